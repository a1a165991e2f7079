#!/bin/bash
O=gpurun_out/x3ab; mkdir -p $O
timeout 600 python -m pytest tests/test_split_gpu.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "not full_size" 2>&1 | tail -5
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 20 --warmup 4 --dtype split"
python bench.py $B --engine FPROP_PLANES=False > $O/a_nofp.json 2>$O/a.err
python bench.py $B > $O/b_fp.json 2>$O/b.err
python bench.py $B --engine PLANES_SCOPE=all > $O/c_all.json 2>$O/c.err
python bench.py $B --engine PLANES_SCOPE=all PLANES_MAX_NUMEL=110000000 > $O/d_all_big.json 2>$O/d.err
python bench.py $B --detail $O/per_launch_x3.txt --single-stream > /dev/null 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR",e); continue
    print("%-36s %8.1f ms=%.2f loss=%s" % (f,d["value"],d["ms_per_step"],d["config"]["final_loss"]))
PY
tail -3 $O/*.err | grep -v amdgpu | head -20
