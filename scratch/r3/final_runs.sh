#!/bin/bash
# Round-3 measurement batch (run on the GPU box through gpurun, from the repo root): bench lines for the BASELINE.json
# configs, the split-bf16 parity path, rocprofv3 kernel traces of the bench command (bf16 and split), the two PMC passes
# for HBM traffic, per-launch tables and the full-size parity tables.  Outputs under gpurun_out/final3/.
O=gpurun_out/final3
mkdir -p $O
R=$(pwd)
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 60 --warmup 5"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_c4_driver_cmd.err
python bench.py $B --dtype split --steps 20 --detail $O/per_launch_split_two_stream.txt > $O/bench_c4_split.json 2>/dev/null
python bench.py $B --dtype split --steps 20 --single-stream --detail $O/per_launch_split_single_stream.txt > /dev/null 2>&1
python bench.py $B --steps 30 --detail $O/per_launch_two_stream.txt > /dev/null 2>&1
python bench.py $B --steps 30 --single-stream --detail $O/per_launch_single_stream.txt > /dev/null 2>&1
python bench.py $B --dtype fp16 > $O/bench_c4_fp16.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 8 > $O/bench_c2_8clips.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 2 > $O/bench_c2_2clips.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl > $O/bench_c3_frozen.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl --set MODEL.FREEZE_BACKBONE False > $O/bench_c3_unfrozen.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 30 > $O/bench_c5_bf16.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 30 --dtype fp16 > $O/bench_c5_fp16.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 8 --dtype split > $O/bench_c5_split.json 2>/dev/null
VLFB_SPLIT_MATH=6,3 python bench.py $B --dtype split --steps 20 > $O/bench_c4_split_x6_forward.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_split -o stats -- python $R/bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_split.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/pmc_write.log 2>&1
cd $R
python scratch/prof_summary.py $O/prof > $O/rocprofv3_kernel_stats.txt 2>&1
python scratch/prof_summary.py $O/prof_split > $O/rocprofv3_kernel_stats_split.txt 2>&1
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line" > /dev/null 2>&1
python scratch/timeline2.py $O/prof $O/timeline_step.txt > /dev/null 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof $O/prof_split
VLFB_PARITY_DIR=$R/$O timeout 900 python -m pytest tests/test_model_gpu.py -q -k "full_size_clip" > $O/parity_fullsize.log 2>&1
tail -2 $O/parity_fullsize.log
ls $O
head -c 600 $O/bench_c4_driver_cmd.json
