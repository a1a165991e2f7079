#!/bin/bash
# round 4, GPU call 5: fused bias gradient -- unit test, model tests that touch bias gradients, same-box A/B
mkdir -p gpurun_out/r4e
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -k "wgrad_with_bias" > gpurun_out/r4e/unit.log 2>&1
echo "unit rc=$?" >> gpurun_out/r4e/unit.log
grep -E "^\[|passed|failed|rc=" gpurun_out/r4e/unit.log | cut -c1-220 | tail -24
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "forward_backward_matches_oracle and (ava_r50_lfb_nl or charades_r50_lfb_nl)" > gpurun_out/r4e/model.log 2>&1
echo "model rc=$?" >> gpurun_out/r4e/model.log
tail -3 gpurun_out/r4e/model.log
for rep in 1 2; do for fb in True False; do
  timeout 300 python bench.py --dtype bf16 --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line --engine FUSE_BIAS_GRAD=$fb > gpurun_out/r4e/bench_fb_${fb}_$rep.json 2> gpurun_out/r4e/bench_fb_${fb}_$rep.err
  python -c "import json; d=json.load(open('gpurun_out/r4e/bench_fb_${fb}_$rep.json')); print('FUSE_BIAS_GRAD=$fb', d['value'], d['ms_per_step'])"
done; done
