#!/bin/bash
# Round 5, GPU call 1: (A) mix parity switches at full size, both presets; (B) their speed; (C) r03 tree vs HEAD A/B (bf16 / fp16)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 420 python scratch/r5/mix_variants.py ava_r50_lfb_nl > $O/full_ava.txt 2>&1
grep "^\[" $O/full_ava.txt | cut -c1-330
timeout 300 python scratch/r5/mix_variants.py charades_r50_baseline > $O/full_charades.txt 2>&1
grep "^\[" $O/full_charades.txt | cut -c1-330
MIX_SMALL=1 timeout 200 python scratch/r5/mix_variants.py ava_r50_lfb_nl charades_r50_baseline > $O/small.txt 2>&1
grep "^\[" $O/small.txt | cut -c1-330
B="--steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line"
for v in "0:" "1:" "1:res2" "1:res2,res3"; do
  h=${v%%:*}; s=${v#*:}
  VLFB_MIX_HEAD_F32=$h VLFB_MIX_W2_SKIP=$s timeout 200 python bench.py --dtype mix $B 2>/dev/null | tail -1 > $O/bench_mix_h${h}_s${s//,/_}.json
  python -c "import json; d=json.loads(open('$O/bench_mix_h${h}_s${s//,/_}.json').read()); print('mix head_f32=$h skip=$s', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
done
# (C) regression bisect: the round-3 tree (a5f9ac8) against HEAD on one box, alternating
for r in 1 2; do
  for t in r03:bf16 head:bf16 head:fp16 r03:fp16; do
    tree=${t%%:*}; dt=${t#*:}
    if [ $tree = r03 ]; then bp=scratch/ab/r03tree/bench.py; ex="--no-cpu-baseline --no-fp32-line --no-split-line"; else bp=bench.py; ex="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line"; fi
    timeout 200 python $bp --dtype $dt --steps 40 --warmup 5 $ex --detail $O/detail_${tree}_${dt}_$r.txt 2>$O/err_${tree}_${dt}_$r.txt | tail -1 > $O/bench_${tree}_${dt}_$r.json
    python -c "import json; d=json.loads(open('$O/bench_${tree}_${dt}_$r.json').read()); print('$tree $dt', d['value'], 'clips/s', d['ms_per_step'], 'ms NT', d['roofline']['achieved'], 'TN', d.get('roofline_wgrad',{}).get('achieved'))" || tail -3 $O/err_${tree}_${dt}_$r.txt
  done
done
