#!/bin/bash
O=gpurun_out/${1:-r3e}; mkdir -p $O
export TMPDIR=/tmp
for c in res5_3x3 res4_3x3 res5_2c res4_2a_t res3_3x3 res2_2c; do
  python scratch/r3/sp_probe.py $c 10 1 >> $O/probe_tile128.txt 2>&1
  python scratch/r3/sp_probe.py $c 10 0 >> $O/probe_auto.txt 2>&1
done
bash scratch/r3/pmc_sp.sh res5_3x3 $O/pmc_res5_3x3_tile128.txt 1
bash scratch/r3/pmc_sp.sh res5_2c $O/pmc_res5_2c_tile128.txt 1
cat $O/probe_tile128.txt; cat $O/probe_auto.txt
