#!/bin/bash
# the two PMC passes of scratch/r3/final_runs.sh alone (profiles/hbm_traffic.json carries the sha256 of csrc/)
O=gpurun_out/pmc; mkdir -p $O; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line > $R/$O/pmc_write.log 2>&1
cd $R
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line"
rm -rf $O/pmc_fetch $O/pmc_write
