#!/bin/bash
# rocprofv3 kernel trace of the mix step -> per-kernel table (tools/prof_summary.py) in gpurun_out/$1/kernel_stats_mix.txt
R=$GRAFT_REPO_ROOT; O=gpurun_out/${1:-r5h}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$O
rocprofv3 --kernel-trace --stats -d $R/$O/prof_mix -o stats -- python $R/bench.py --dtype mix --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/prof_mix.log 2>&1
python $R/tools/prof_summary.py $R/$O/prof_mix > $R/$O/kernel_stats_mix.txt 2>&1
rm -rf $R/$O/prof_mix
head -50 $R/$O/kernel_stats_mix.txt | cut -c1-170
