#!/bin/bash
# round 4, GPU call 7: threading test; same-box table of the mix knobs
mkdir -p gpurun_out/r4g
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_threads_gpu.py -q -m gpu > gpurun_out/r4g/threads.log 2>&1; tail -3 gpurun_out/r4g/threads.log
B="--dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line"
for cfgs in "MIX_W2=True MIX_NL_F32=True MIX_TRUNK2=True" "MIX_W2=True MIX_NL_F32=True MIX_TRUNK2=False" "MIX_W2=True MIX_NL_F32=False MIX_TRUNK2=False" "MIX_W2=False MIX_NL_F32=False MIX_TRUNK2=False" "MIX_W2=False MIX_NL_F32=True MIX_TRUNK2=True" "MIX_W2=True MIX_NL_F32=True MIX_TRUNK2=True"; do
  n=$(echo $cfgs | tr ' =' '__')
  timeout 300 python bench.py $B --engine $cfgs > gpurun_out/r4g/bench_$n.json 2> gpurun_out/r4g/bench_$n.err
  python -c "import json; d=json.load(open('gpurun_out/r4g/bench_$n.json')); print('$cfgs', d['value'], d['ms_per_step'])" || tail -2 gpurun_out/r4g/bench_$n.err
done
timeout 300 python bench.py --dtype fp16 --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > gpurun_out/r4g/bench_fp16.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r4g/bench_fp16.json')); print('fp16', d['value'], d['ms_per_step'], 'traffic', d['roofline']['traffic'])"
timeout 300 python bench.py --dtype split --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > gpurun_out/r4g/bench_split.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r4g/bench_split.json')); print('split', d['value'], d['ms_per_step'])"
