#!/bin/bash
# Round-4 measurement batch (run on the GPU box through gpurun, from the repo root): the whole GPU suite with the parity
# tables, the driver's bench command, bench lines of the BASELINE.json configs, rocprofv3 kernel traces (default fp16 and mix),
# the two PMC passes for HBM traffic, per-launch tables.  Outputs under gpurun_out/final4/.
O=gpurun_out/final4
mkdir -p $O
R=$(pwd)
export VLFB_PARITY_DIR=$R/$O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
B="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --steps 60 --warmup 5"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_c4_driver_cmd.err
python bench.py $B --steps 30 --detail $O/per_launch_two_stream.txt > /dev/null 2>&1
python bench.py $B --steps 30 --single-stream --detail $O/per_launch_single_stream.txt > /dev/null 2>&1
python bench.py $B --dtype bf16 > $O/bench_c4_bf16.json 2>/dev/null
python bench.py $B --dtype mix --steps 30 --detail $O/per_launch_mix_two_stream.txt > $O/bench_c4_mix.json 2>/dev/null
python bench.py $B --dtype mix --steps 20 --single-stream --detail $O/per_launch_mix_single_stream.txt > /dev/null 2>&1
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 8 > $O/bench_c2_8clips.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 2 > $O/bench_c2_2clips.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl > $O/bench_c3_frozen.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl --set MODEL.FREEZE_BACKBONE False > $O/bench_c3_unfrozen.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl --set MODEL.FREEZE_BACKBONE False --dtype mix --steps 20 > $O/bench_c3_unfrozen_mix.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 30 > $O/bench_c5_fp16.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 10 --dtype mix > $O/bench_c5_mix.json 2>$O/bench_c5_mix.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_mix -o stats -- python $R/bench.py --dtype mix --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_mix.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/pmc_write.log 2>&1
cd $R
python scratch/prof_summary.py $O/prof > $O/rocprofv3_kernel_stats.txt 2>&1
python scratch/prof_summary.py $O/prof_mix > $O/rocprofv3_kernel_stats_mix.txt 2>&1
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line" > /dev/null 2>&1
python scratch/timeline2.py $O/prof $O/timeline_step.txt > /dev/null 2>&1
python scratch/timeline2.py $O/prof_mix $O/timeline_step_mix.txt > /dev/null 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof $O/prof_mix
ls $O
head -c 500 $O/bench_c4_driver_cmd.json
