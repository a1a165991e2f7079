#!/bin/bash
# rocprofv3 kernel trace of the mix step -> kernel stats + timeline of one step (run on the GPU box from the repo root)
O=gpurun_out/${1:-prof}
mkdir -p $O
R=$(pwd)
X="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --no-fp16-line --no-bf16-line"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof_mix -o stats -- python $R/bench.py --steps 12 --warmup 3 $X ${@:2} > $R/$O/prof_mix.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_mix > $O/rocprofv3_kernel_stats_mix.txt 2>&1
python tools/timeline.py $O/prof_mix $O/timeline_step_mix.txt > /dev/null 2>&1
[ -n "$KEEP_DB" ] || rm -rf $O/prof_mix
