#!/bin/bash
# round 4, GPU call 6: two-term residual-stream gradient in "mix": parity small + full size, step time
mkdir -p gpurun_out/r4f
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r4/mix_check.py > gpurun_out/r4f/mix_check.txt 2>&1
echo "mix_check rc=$?" >> gpurun_out/r4f/mix_check.txt
grep -v amdgpu gpurun_out/r4f/mix_check.txt | tail -6 | cut -c1-330
MIX_FULL=1 timeout 900 python scratch/r4/mix_check.py > gpurun_out/r4f/mix_check_full.txt 2>&1
echo "mix_check_full rc=$?" >> gpurun_out/r4f/mix_check_full.txt
grep -v "^   " gpurun_out/r4f/mix_check_full.txt | grep -v amdgpu | tail -4 | cut -c1-330
for t2 in 1 0; do
VLFB_MIX_TRUNK2=$t2 timeout 300 python bench.py --dtype mix --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4f/bench_mix_t$t2.json 2> gpurun_out/r4f/bench_mix_t$t2.err
python -c "import json; d=json.load(open('gpurun_out/r4f/bench_mix_t$t2.json')); print('TRUNK2=$t2', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r4f/bench_mix_t$t2.err
done
