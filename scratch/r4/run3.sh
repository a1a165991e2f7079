#!/bin/bash
# round 4, GPU call 3: parity on the benchmarked plan (8 clips vs 1-clip engines), mix in the model tests, full-size parity tables
mkdir -p gpurun_out/r4c
cd $GRAFT_REPO_ROOT
export VLFB_PARITY_DIR=$GRAFT_REPO_ROOT/gpurun_out/r4c/parity
timeout 1500 python -m pytest tests/test_bench_plan_gpu.py -q -s -m gpu > gpurun_out/r4c/plan.log 2>&1
echo "plan rc=$?" >> gpurun_out/r4c/plan.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "mix or full_size" > gpurun_out/r4c/model.log 2>&1
echo "model rc=$?" >> gpurun_out/r4c/model.log
grep -E "^\[|passed|failed|rc=|Error|error|assert" gpurun_out/r4c/plan.log | cut -c1-400 | tail -30
grep -E "passed|failed|rc=" gpurun_out/r4c/model.log | tail
