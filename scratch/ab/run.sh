#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by several per cent): scratch/ab/run.sh <rounds> <dtype> libA libB ...
# (scratch/ab/<lib>.so are alternative builds of libvlfb_hip.so; the original is restored at the end)
cd $GRAFT_REPO_ROOT
L=video-long-term-feature-banks_amd/lib/vlfb/libvlfb_hip.so
cp $L /tmp/keep.so
R=$1; DT=$2; shift; shift
for r in $(seq $R); do for v in "$@"; do cp scratch/ab/$v.so $L
  timeout 300 python bench.py --dtype $DT --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line --no-fp16-line --no-mix-line 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['roofline_families']; print('$v', d['value'], 'clips/s |', ' '.join('%s %.2f ms' % (k, f[k]['ms_per_step']) for k in f), '| loss', d['config']['final_loss'])"
done; done
cp /tmp/keep.so $L
