#!/bin/bash
mkdir -p gpurun_out/r4i
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -k "wgrad_with_bias or reproducible or roi_align or roi_head" > gpurun_out/r4i/t.log 2>&1; tail -5 gpurun_out/r4i/t.log | cut -c1-300
for i in 1 2 3; do timeout 300 python bench.py --dtype mix --steps 30 --warmup 4 --no-cpu-baseline --no-fp32-line --no-split-line > gpurun_out/r4i/b$i.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r4i/b$i.json')); print(d['value'], repr(d['config']['final_loss']))"; done
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > gpurun_out/r4i/f.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r4i/f.json')); print('fp16', d['value'], repr(d['config']['final_loss']))"
