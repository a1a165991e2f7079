#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "mix" 2>&1 | tail -8
bash scratch/r5/run3.sh
