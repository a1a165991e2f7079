#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "test_forward_backward_matches_oracle and mix" > $O/small_mix.log 2>&1
grep "identical ReLU\|passed\|failed" $O/small_mix.log | cut -c1-300
bash scratch/r5/run3.sh
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_bench_plan_gpu.py tests/test_train_loop_gpu.py tests/test_step_graph_gpu.py -q -m gpu -x > $O/others.log 2>&1
tail -3 $O/others.log
