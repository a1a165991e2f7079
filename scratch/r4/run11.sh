#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
B="--steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line"
for fr in 1.0 0.75 0.5 0.625 0.375 0.875 1.0; do
  VLFB_SIDE_CU_FRACTION=$fr timeout 300 python bench.py $B > gpurun_out/r4k/b_$fr.json 2> gpurun_out/r4k/b_$fr.err
  python -c "import json; d=json.load(open('gpurun_out/r4k/b_$fr.json')); print('SIDE_CU_FRACTION=$fr', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r4k/b_$fr.err
done
for fr in 1.0 0.5 0.75; do
  VLFB_SIDE_CU_FRACTION=$fr timeout 300 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4k/m_$fr.json 2> gpurun_out/r4k/m_$fr.err
  python -c "import json; d=json.load(open('gpurun_out/r4k/m_$fr.json')); print('mix SIDE_CU_FRACTION=$fr', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r4k/m_$fr.err
done
