#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
B="--dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line"
for m in all nonoverlap none; do
  VLFB_EXP_POOL_LO=$m timeout 420 python scratch/r5/mix_variants.py ava_r50_lfb_nl 2>&1 | grep "^\[" | cut -c60-330 | sed "s/^/pool_lo=$m /"
done
for r in 1 2; do for m in all nonoverlap none; do
  VLFB_EXP_POOL_LO=$m timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool_lo=$m', d['value'], 'clips/s')"
done; done
