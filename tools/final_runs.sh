#!/bin/bash
# The measurement batch of a round (run on the GPU box through gpurun, from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/final_runs.sh final6'
# smoke, the driver's bench command, per-launch tables, bench lines of the other BASELINE.json configs, rocprofv3 kernel
# traces (mix and fp16), the two PMC passes for HBM traffic (mix).  Outputs under gpurun_out/<name>/; the summaries are
# copied to profiles/rNN_* by hand.  Also: the full-size parity tables (VLFB_PARITY_DIR), the SQ-counter pass (MFMA-pipe busy,
# LDS conflicts, stall shares per kernel) and the same-box A/B of the two-plane forward (VLFB_MIX_PAIR=0 = the round-5 form).
# A second argument "pmc" runs only the counter passes (into the same directory).
O=gpurun_out/${1:-final}
mkdir -p $O
R=$(pwd)
X="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --no-fp16-line --no-bf16-line"
if [ "$2" != "pmc" ]; then
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_c4_driver_cmd.err
B="$X --steps 40 --warmup 5"
python bench.py $B --steps 30 --detail $O/per_launch_mix_two_stream.txt > $O/bench_c4_mix.json 2>/dev/null
VLFB_MIX_PAIR=0 python bench.py $B --steps 30 > $O/bench_c4_mix_r5_forward.json 2>/dev/null
python bench.py $B --steps 30 > $O/bench_c4_mix_again.json 2>/dev/null
python bench.py $B --steps 20 --single-stream --detail $O/per_launch_mix_single_stream.txt > /dev/null 2>&1
python bench.py $B --dtype fp16 --detail $O/per_launch_fp16_two_stream.txt > $O/bench_c4_fp16.json 2>/dev/null
python bench.py $B --dtype fp16 --steps 30 --single-stream --detail $O/per_launch_fp16_single_stream.txt > /dev/null 2>&1
python bench.py $B --dtype bf16 > $O/bench_c4_bf16.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 8 > $O/bench_c2_8clips.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 2 > $O/bench_c2_2clips.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 8 --dtype fp16 > $O/bench_c2_8clips_fp16.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl > $O/bench_c3_frozen.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl --set MODEL.FREEZE_BACKBONE False > $O/bench_c3_unfrozen.json 2>/dev/null
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 10 > $O/bench_c5_mix.json 2>$O/bench_c5_mix.err
python bench.py $B --workload ava_r101_lfb_nl_3l --frames 64 --steps 20 --dtype fp16 > $O/bench_c5_fp16.json 2>/dev/null
VLFB_PARITY_DIR=$R/$O python -m pytest tests/test_model_gpu.py -q -x -k test_full_size_clip_matches_oracle > $O/parity_run.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof_mix -o stats -- python $R/bench.py --steps 12 --warmup 3 $X > $R/$O/prof_mix.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_fp16 -o stats -- python $R/bench.py --dtype fp16 --steps 20 --warmup 5 $X > $R/$O/prof_fp16.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_mix > $O/rocprofv3_kernel_stats_mix.txt 2>&1
python tools/prof_summary.py $O/prof_fp16 > $O/rocprofv3_kernel_stats_fp16.txt 2>&1
python tools/timeline.py $O/prof_mix $O/timeline_step_mix.txt > /dev/null 2>&1
python tools/timeline.py $O/prof_fp16 $O/timeline_step_fp16.txt > /dev/null 2>&1
rm -rf $O/prof_mix $O/prof_fp16
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/pmc_write.log 2>&1
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU"
rocprofv3 --kernel-trace --pmc $SQ -d $R/$O/pmc_sq -o q -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $R/$O/pmc_sq16 -o q -- python $R/bench.py --dtype fp16 --steps 3 --warmup 1 $X > $R/$O/pmc_sq16.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 $X" > /dev/null 2>$O/pmc_traffic.err
python tools/pmc_sq.py $O/pmc_sq $O/pmc_sq_mix.txt "python bench.py --steps 3 --warmup 1 $X" > /dev/null 2>$O/pmc_sq.err
python tools/pmc_sq.py $O/pmc_sq16 $O/pmc_sq_fp16.txt "python bench.py --dtype fp16 --steps 3 --warmup 1 $X" > /dev/null 2>>$O/pmc_sq.err
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq16
ls $O
head -c 400 $O/bench_c4_driver_cmd.json; echo
cat $O/pmc_hbm_traffic.txt
