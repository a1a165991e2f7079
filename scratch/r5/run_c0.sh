#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dgrad_s2_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "test_forward_backward_matches_oracle or fp16_path_matches" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_bench_plan_gpu.py tests/test_train_loop_gpu.py tests/test_step_graph_gpu.py -q -m gpu -x 2>&1 | tail -3
B="--steps 40 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-fp16-line --no-mix-line"
for r in 1 2; do for dt in mix fp16; do for sw in True False; do
  timeout 300 python bench.py --dtype $dt $B --engine SPARSE_SHORTCUT_DGRAD=$sw 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt sparse=$sw', d['value'], 'clips/s', d['config']['final_loss'])"
done; done; done
