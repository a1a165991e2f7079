#!/bin/bash
# round 4, GPU call 4: the whole GPU suite on the current tree + the driver's bench command
mkdir -p gpurun_out/r4d
cd $GRAFT_REPO_ROOT
export VLFB_PARITY_DIR=$GRAFT_REPO_ROOT/gpurun_out/r4d/parity
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4d/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4d/pytest_gpu.log
tail -15 gpurun_out/r4d/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4d/bench_default.json 2> gpurun_out/r4d/bench_default.err
cut -c1-400 gpurun_out/r4d/bench_default.json
