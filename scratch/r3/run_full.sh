#!/bin/bash
# whole GPU suite + smoke + the driver's bench command
O=gpurun_out/${1:-r3f}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_gpu.log; tail -2 $O/smoke.log; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value","ms_per_step","model_flops_utilisation","host_enqueue_ms_per_step")}, d.get("split_path"), d.get("fp32_path"), d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
