#!/bin/bash
# the two PMC passes of scratch/final_runs.sh alone (HBM traffic per launch family and per kernel)
O=gpurun_out/final
mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line > $R/$O/pmc_write.log 2>&1
cd $R
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line"
python scratch/pmc_kernel.py $O/pmc_fetch $O/pmc_write > $O/pmc_per_kernel.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
