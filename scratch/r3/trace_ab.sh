#!/bin/bash
# STEP_TRACE A/B: bit-identity tests, then host enqueue time and clips/s with and without the recorded step
O=gpurun_out/trace; mkdir -p $O
timeout 600 python -m pytest tests/test_step_graph_gpu.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
B="--no-cpu-baseline --no-fp32-line --no-split-line --steps 60 --warmup 5"
for rep in 1 2; do
python bench.py $B > $O/bf16_streams_$rep.json 2>/dev/null
python bench.py $B --engine STEP_TRACE=True > $O/bf16_trace_$rep.json 2>/dev/null
done
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 2 > $O/c2_streams.json 2>/dev/null
python bench.py $B --workload charades_r50_baseline --clips-per-gpu 2 --engine STEP_TRACE=True > $O/c2_trace.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl > $O/c3_streams.json 2>/dev/null
python bench.py $B --workload charades_r50_lfb_nl --engine STEP_TRACE=True > $O/c3_trace.json 2>/dev/null
python bench.py $B --dtype split --steps 20 > $O/split_streams.json 2>/dev/null
python bench.py $B --dtype split --steps 20 --engine STEP_TRACE=True > $O/split_trace.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    print("%-40s %8.1f ms=%.2f host=%.2f loss=%s" % (f, d["value"], d["ms_per_step"], d.get("host_enqueue_ms_per_step",-1), d["config"].get("final_loss")))
PY
