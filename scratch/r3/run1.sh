#!/bin/bash
# round 3, GPU call 1: split-bf16 kernels -- unit parity, model parity (small + full size), first bench
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_split_gpu.py -x -q > $O/t_split.log 2>&1; echo "split unit rc=$?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_model_gpu.py -q -s -k "test_forward_backward_matches_oracle and split" > $O/t_model_small.log 2>&1; echo "model small rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-line --detail $O/split_detail.txt > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-line --single-stream --detail $O/split_detail_1s.txt > $O/bench_split_1s.json 2> $O/bench_split_1s.err; echo "bench split 1-stream rc=$?" | tee -a $O/summary.txt
VLFB_PARITY_DIR=$O timeout 900 python -m pytest tests/test_model_gpu.py -q -s -k "test_full_size_clip_matches_oracle" > $O/t_full.log 2>&1; echo "full size rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-line > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?" | tee -a $O/summary.txt
tail -3 $O/t_split.log; tail -3 $O/t_model_small.log; tail -3 $O/t_full.log; cut -c1-400 $O/bench_split.json; cut -c1-300 $O/bench_bf16.json
