#!/bin/bash
# SQ / TCC counter passes over an arbitrary command: scratch/pmc_cmd.sh "<command>" <out.txt>   (run from the repo root)
# CAUTION: never finished on `scratch/stem_probe 8 32 224` within 280 s at the end of round 2 (the call was killed by the budget
# clamp, nothing came back) -- unverified which pass stalls; try it on a one-kernel command with its own short `timeout` first.
R=$(pwd); C=$1; O=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM -d /tmp/pmcp/a -o a -- $R/$C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU -d /tmp/pmcp/b -o b -- $R/$C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d /tmp/pmcp/c -o c -- $R/$C > /dev/null 2>&1
cd $R
python scratch/pmc_kernel.py /tmp/pmcp/a /tmp/pmcp/b /tmp/pmcp/c > $O 2>&1
