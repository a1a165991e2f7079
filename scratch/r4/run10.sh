#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
B="--steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line"
for rep in 1 2; do for pr in 0 1; do
  VLFB_NT_PRIO=$pr timeout 300 python bench.py $B > gpurun_out/r4j/b_${pr}_$rep.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r4j/b_${pr}_$rep.json')); print('VLFB_NT_PRIO=$pr', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
