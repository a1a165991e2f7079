#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
B="--steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line"
for a in 1 0 1 0; do
  VLFB_BWD_AUX=$a timeout 300 python bench.py $B > gpurun_out/r4l/b_$a.json 2> gpurun_out/r4l/b_$a.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/b_$a.json')); print('BWD_AUX=$a', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r4l/b_$a.err
done
for a in 1 0; do
  VLFB_BWD_AUX=$a timeout 300 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4l/m_$a.json 2> gpurun_out/r4l/m_$a.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/m_$a.json')); print('mix BWD_AUX=$a', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r4l/m_$a.err
done
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "reproducible or tiny or small" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_step_graph_gpu.py tests/test_bench_plan_gpu.py -q -m gpu -k "fp16 or trace or replay" 2>&1 | tail -4
