#!/bin/bash
O=gpurun_out/${1:-r3g}; mkdir -p $O
timeout 900 python scratch/r3/diag_split.py ava_r50_lfb_nl > $O/diag_ava.txt 2>&1
grep -v "^  lfb_nl\|amdgpu" $O/diag_ava.txt | head -150
