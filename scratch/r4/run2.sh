#!/bin/bash
# round 4, GPU call 2: "mix" with fp32 non-local gradients -- parity (small, full size), 8-clip step time
mkdir -p gpurun_out/r4b
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r4/mix_check.py > gpurun_out/r4b/mix_check.txt 2>&1
echo "mix_check rc=$?" >> gpurun_out/r4b/mix_check.txt
MIX_FULL=1 timeout 900 python scratch/r4/mix_check.py > gpurun_out/r4b/mix_check_full.txt 2>&1
echo "mix_check_full rc=$?" >> gpurun_out/r4b/mix_check_full.txt
timeout 300 python bench.py --dtype mix --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line --detail gpurun_out/r4b/detail_mix.txt > gpurun_out/r4b/bench_mix.json 2> gpurun_out/r4b/bench_mix.err
timeout 300 python bench.py --dtype mix --single-stream --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line --detail gpurun_out/r4b/detail_mix_single.txt > gpurun_out/r4b/bench_mix_single.json 2> gpurun_out/r4b/bench_mix_single.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4b/prof -- python $GRAFT_REPO_ROOT/bench.py --dtype mix --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line > $GRAFT_REPO_ROOT/gpurun_out/r4b/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4b/prof -name "*kernel_stats*" | head -3
find gpurun_out/r4b/prof -name "*kernel_trace*" -size +20M -delete
grep -v amdgpu gpurun_out/r4b/mix_check.txt | tail -9; grep -v "^   " gpurun_out/r4b/mix_check_full.txt | tail -4; cut -c1-200 gpurun_out/r4b/bench_mix.json
