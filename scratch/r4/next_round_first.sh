#!/bin/bash
# First GPU call of the next round: does the fp32 head (Engine.MIX_HEAD_F32, DESIGN.md 7 "Located: it is the HEAD") move the mix
# gradients TOWARDS the oracle?  Prediction from scratch/r4/emu_hybrid.py: identical-decisions median 4.0e-4 -> ~1.6e-4,
# max 9.7e-4 -> ~7.3e-4 (conv1_w) on the full-size AVA clip; small size 3.1e-4 -> ~1.5e-4.
#   gpurun --timeout 900 -- 'bash scratch/r4/next_round_first.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
for on in 0 1; do
  VLFB_MIX_HEAD_F32=$on timeout 200 python scratch/r4/mix_check.py ava_r50_lfb_nl charades_r50_baseline 2>&1 | grep "^\[" | grep "trunk2=True" > gpurun_out/r5a/small_head$on.txt
  VLFB_MIX_HEAD_F32=$on MIX_FULL=1 timeout 400 python scratch/r4/mix_check.py ava_r50_lfb_nl > gpurun_out/r5a/full_head$on.txt 2>&1
done
for on in 0 1; do echo "== head_f32=$on"; cut -c1-260 gpurun_out/r5a/small_head$on.txt; grep "^\[" gpurun_out/r5a/full_head$on.txt | cut -c1-260; done
# speed: the same step with and without the switch
for on in 0 1; do
  VLFB_MIX_HEAD_F32=$on timeout 200 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r5a/bench_head$on.json 2>/dev/null
  python -c "import json; d=json.loads(open('gpurun_out/r5a/bench_head$on.json').read().splitlines()[-1]); print('head_f32=$on', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
done
# then, with the fp32 head: plain fp16 DGRAD weights in res2 (+ res3)  (emulated max 6.6e-4 / 7.5e-4; expect 241-248 clips/s)
for skip in res2 res2,res3; do
  VLFB_MIX_HEAD_F32=1 VLFB_MIX_W2_SKIP=$skip MIX_FULL=1 timeout 400 python scratch/r4/mix_check.py ava_r50_lfb_nl 2>&1 | grep "^\[" | cut -c1-260
  VLFB_MIX_HEAD_F32=1 VLFB_MIX_W2_SKIP=$skip timeout 200 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('skip=$skip', d['value'], 'clips/s')"
done
