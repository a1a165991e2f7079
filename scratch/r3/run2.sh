#!/bin/bash
# round 3, GPU call: split unit tests + split bench (two-stream and single-stream per-launch tables)
O=gpurun_out/${1:-r3b}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_split_gpu.py -x -q > $O/t_split.log 2>&1; echo "split unit rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-line --detail $O/split_detail.txt > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --dtype split --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-line --single-stream --detail $O/split_detail_1s.txt > $O/bench_split_1s.json 2> $O/bench_split_1s.err; echo "bench split 1-stream rc=$?" | tee -a $O/summary.txt
tail -3 $O/t_split.log; cut -c1-330 $O/bench_split.json; echo; cut -c1-330 $O/bench_split_1s.json
