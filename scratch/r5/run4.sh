#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "maxpool" 2>&1 | tail -5
bash scratch/r5/run3.sh
