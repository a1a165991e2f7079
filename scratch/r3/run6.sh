#!/bin/bash
O=gpurun_out/${1:-r3l}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_split_gpu.py -x -q > $O/t_split.log 2>&1; echo "split unit rc=$?"
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "test_forward_backward_matches_oracle and split" > $O/t_small.log 2>&1; echo "small rc=$?"
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --detail $O/split_detail.txt > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?"
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --single-stream --detail $O/split_detail_1s.txt > $O/bench_split_1s.json 2> $O/bench_split_1s.err; echo "bench split 1s rc=$?"
tail -4 $O/t_split.log; grep "identical\|rror" $O/t_small.log | head; tail -2 $O/t_small.log; cut -c1-240 $O/bench_split.json; echo; cut -c1-240 $O/bench_split_1s.json; tail -3 $O/bench_split.err
