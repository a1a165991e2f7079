#!/bin/bash
# development A/B: captured step vs stream step, one box
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line "$@" 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
run
run --single-stream
run --single-stream --graph step
run --graph forward
run --graph step
run
