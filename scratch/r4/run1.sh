#!/bin/bash
# round 4, GPU call 1: the "mix" dtype -- parity at the small test size, then the 8-clip step time (W2 on / off)
mkdir -p gpurun_out/r4a
cd $GRAFT_REPO_ROOT
export PYTHONPATH=
timeout 600 python scratch/r4/mix_check.py > gpurun_out/r4a/mix_check.txt 2>&1
echo "mix_check rc=$?" >> gpurun_out/r4a/mix_check.txt
for w2 in 1 0; do
  VLFB_MIX_W2=$w2 timeout 300 python bench.py --dtype mix --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line --detail gpurun_out/r4a/detail_mix_w2_$w2.txt > gpurun_out/r4a/bench_mix_w2_$w2.json 2> gpurun_out/r4a/bench_mix_w2_$w2.err
done
timeout 300 python bench.py --dtype split --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4a/bench_split.json 2> gpurun_out/r4a/bench_split.err
timeout 300 python bench.py --dtype fp16 --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4a/bench_fp16.json 2> gpurun_out/r4a/bench_fp16.err
tail -5 gpurun_out/r4a/mix_check.txt; cat gpurun_out/r4a/bench_mix_w2_1.json | cut -c1-300; tail -3 gpurun_out/r4a/bench_mix_w2_1.err
