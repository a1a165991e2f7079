#!/bin/bash
# (rocprofv3 under `timeout`: a counter set the hardware cannot collect aborts the tool and leaves it hanging)
# one PMC pass over the L2 (TCC) of the mix step: hit rate of the operand traffic per kernel family (GPU box, repo root)
O=gpurun_out/${1:-pmc_l2}
mkdir -p $O
R=$(pwd)
X="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --no-fp16-line --no-bf16-line"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $R/$O/a -o a -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/a.log 2>&1
cd $R
python tools/pmc_mem.py $O/pmc_l2_mix.txt "python bench.py --steps 3 --warmup 1 $X" $O/a > /dev/null 2> $O/pmc_l2.err
rm -rf $O/a
tail -3 $O/a.log; tail -5 $O/pmc_l2.err; cat $O/pmc_l2_mix.txt | cut -c1-600
