#!/bin/bash
# two PMC passes over the vector-memory path of the mix step (run on the GPU box from the repo root)
O=gpurun_out/${1:-pmc_mem}
mkdir -p $O
R=$(pwd)
X="--no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --no-fp16-line --no-bf16-line"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $R/$O/a -o a -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/a.log 2>&1
rocprofv3 --kernel-trace --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE -d $R/$O/b -o b -- python $R/bench.py --steps 3 --warmup 1 $X > $R/$O/b.log 2>&1
cd $R
python tools/pmc_mem.py $O/pmc_mem_mix.txt "python bench.py --steps 3 --warmup 1 $X" $O/a $O/b > /dev/null 2> $O/pmc_mem.err
rm -rf $O/a $O/b
tail -3 $O/a.log; tail -3 $O/b.log; cat $O/pmc_mem.err | tail -5; cat $O/pmc_mem_mix.txt | cut -c1-700
