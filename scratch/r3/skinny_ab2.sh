#!/bin/bash
O=gpurun_out/skinny; mkdir -p $O
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 60 --warmup 5"
for r in 1 2; do
VLFB_SKINNY=0 python bench.py $B > $O/off$r.json 2>/dev/null
python bench.py $B > $O/on$r.json 2>/dev/null
done
VLFB_SKINNY=0 python bench.py $B --workload charades_r50_lfb_nl > $O/c3_off.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl > $O/c3_on.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.load(open(f)); print(f, d["value"], d["ms_per_step"])
PY
bash scratch/r3/pmc_only.sh | tail -2
