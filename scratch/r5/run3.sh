#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 420 python scratch/r5/mix_variants.py ava_r50_lfb_nl > $O/full_ava.txt 2>&1
grep "^\[" $O/full_ava.txt | cut -c1-330 || tail -20 $O/full_ava.txt
timeout 300 python scratch/r5/mix_variants.py charades_r50_baseline > $O/full_charades.txt 2>&1
grep "^\[" $O/full_charades.txt | cut -c1-330 || tail -20 $O/full_charades.txt
timeout 200 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line 2>$O/bench.err | tail -1 > $O/bench_mix.json
python -c "import json; d=json.loads(open('$O/bench_mix.json').read()); print('mix', d['value'], 'clips/s', d['ms_per_step'], 'ms')" || tail -5 $O/bench.err
