#!/bin/bash
O=gpurun_out/lag; mkdir -p $O
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 60 --warmup 5"
for lag in 0 2 4 8 16 32 1000; do
  python bench.py $B --engine WGRAD_LAG=$lag > $O/bf16_lag$lag.json 2>/dev/null
done
python bench.py $B --engine WGRAD_LAG=0 > $O/bf16_lag0b.json 2>/dev/null
for lag in 0 8 1000; do
  python bench.py $B --steps 20 --dtype split --engine WGRAD_LAG=$lag > $O/split_lag$lag.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR",e); continue
    print("%-36s %8.1f ms=%.3f loss=%s" % (f,d["value"],d["ms_per_step"],d["config"]["final_loss"]))
PY
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2>/dev/null
