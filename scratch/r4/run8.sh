#!/bin/bash
mkdir -p gpurun_out/r4h
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_step_graph_gpu.py -q -m gpu -k "replayed_trace" > gpurun_out/r4h/trace.log 2>&1; tail -2 gpurun_out/r4h/trace.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "forward_backward_matches_oracle and mix" > gpurun_out/r4h/model.log 2>&1; tail -2 gpurun_out/r4h/model.log
B="--dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line"
for rep in 1 2; do for hs in True False; do
  timeout 300 python bench.py $B --engine HALF_COPIES_ON_SIDE=$hs > gpurun_out/r4h/bench_$hs_$rep.json 2> gpurun_out/r4h/bench_$hs_$rep.err
  python -c "import json; d=json.load(open('gpurun_out/r4h/bench_$hs_$rep.json')); print('HALF_COPIES_ON_SIDE=$hs', d['value'], d['ms_per_step'], d['config']['final_loss'])" || tail -2 gpurun_out/r4h/bench_$hs_$rep.err
done; done
