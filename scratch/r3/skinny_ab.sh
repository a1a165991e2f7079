#!/bin/bash
O=gpurun_out/skinny; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k skinny 2>&1 | tail -6
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 60 --warmup 5"
python bench.py $B --detail $O/per_launch.txt > $O/on.json 2>/dev/null
grep "M=33 " $O/per_launch.txt
python bench.py $B --workload charades_r50_lfb_nl > $O/c3_on.json 2>/dev/null
python - <<PY
import json
for f in ("on","c3_on"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["ms_per_step"])
PY
