#!/bin/bash
O=gpurun_out/s2ab; mkdir -p $O
timeout 600 python -m pytest tests/test_dgrad_s2_gpu.py -x -q 2>&1 | tail -8
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 20 --warmup 4 --dtype split"
VLFB_SPLIT_S2=0 python bench.py $B > $O/a_off.json 2>$O/a.err
VLFB_SPLIT_S2_1X1=0 python bench.py $B > $O/b_3x3only.json 2>$O/b.err
python bench.py $B > $O/c_all.json 2>$O/c.err
python bench.py $B --detail $O/per_launch.txt --single-stream > /dev/null 2>&1
grep "s122" $O/per_launch.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR",e); continue
    print("%-36s %8.1f ms=%.2f loss=%s" % (f,d["value"],d["ms_per_step"],d["config"]["final_loss"]))
PY
