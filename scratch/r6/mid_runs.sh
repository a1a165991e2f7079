#!/bin/bash
# intermediate measurement batch: full-size parity tables, single-stream per-launch table, SQ counters (mix)
O=gpurun_out/${1:-mid}
mkdir -p $O
R=$(pwd)
X="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --no-fp16-line --no-bf16-line"
VLFB_PARITY_DIR=$R/$O python -m pytest tests/test_model_gpu.py -q -x -k test_full_size_clip_matches_oracle -s > $O/parity_run.log 2>&1
tail -3 $O/parity_run.log
python bench.py $X --steps 20 --single-stream --detail $O/per_launch_mix_single_stream.txt > $O/bench_single.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/$O/pmc_sq -o q -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/pmc_sq.log 2>&1
cd $R
python tools/pmc_sq.py $O/pmc_sq $O/pmc_sq_mix.txt "python bench.py --steps 3 --warmup 1 $X" > /dev/null 2> $O/pmc_sq.err
rm -rf $O/pmc_sq
ls $O
