#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by several per cent): scratch/ab/run.sh <rounds> libA libB ...
L=video-long-term-feature-banks_amd/lib/vlfb/libvlfb_hip.so
cp $L /tmp/keep.so
R=$1; shift
for r in $(seq $R); do for v in "$@"; do cp scratch/ab/$v.so $L; echo -n "$v: "; timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-fp32-line 2>&1 | tail -1 | cut -c70-130; done; done
cp /tmp/keep.so $L
