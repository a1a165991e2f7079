#!/bin/bash
# PMC passes over one case of sp_probe.py: scratch/r3/pmc_sp.sh <case> <out.txt> [algo]
R=$(pwd); C=$1; O=$2; A=${3:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM -d /tmp/pmcp/a -o a -- python $R/scratch/r3/sp_probe.py $C 3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pmcp/b -o b -- python $R/scratch/r3/sp_probe.py $C 3 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d /tmp/pmcp/c -o c -- python $R/scratch/r3/sp_probe.py $C 3 $A > /dev/null 2>&1
cd $R
python scratch/pmc_kernel.py /tmp/pmcp/a /tmp/pmcp/b /tmp/pmcp/c > $O 2>&1
