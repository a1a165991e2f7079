#!/bin/bash
# final validation of the round-4 tree: the new whole-config tests first (fail fast), then the full GPU suite, then the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final4d
timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "more_shipped or every_shipped" > gpurun_out/final4d/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/final4d/new_tests.log
tail -3 gpurun_out/final4d/new_tests.log
timeout 700 python -m pytest tests -q -m gpu > gpurun_out/final4d/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/final4d/pytest_gpu.log
tail -4 gpurun_out/final4d/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 > gpurun_out/final4d/bench.json 2> gpurun_out/final4d/bench.err
echo "bench rc=$?"
python -c "import json; d=json.loads(open('gpurun_out/final4d/bench.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('mix_path',{}).get('value'), d.get('split_path',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
