#!/bin/bash
# PMC passes over one case of a probe binary: scratch/pmc_probe.sh <probe> "<case substring>" <out.txt>
R=$(pwd); P=$1; C=$2; O=$3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM -d /tmp/pmcp/a -o a -- $R/$P 3 "$C" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM -d /tmp/pmcp/b -o b -- $R/$P 3 "$C" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d /tmp/pmcp/c -o c -- $R/$P 3 "$C" > /dev/null 2>&1
cd $R
python scratch/pmc_kernel.py /tmp/pmcp/a /tmp/pmcp/b /tmp/pmcp/c > $O 2>&1
