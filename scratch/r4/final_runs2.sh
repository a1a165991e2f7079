#!/bin/bash
# Round-4 closing batch on the final kernel sources: the whole GPU suite (parity tables + JSON stamped with the csrc hash),
# smoke, the driver's bench command, the two PMC passes for HBM traffic, rocprofv3 kernel stats.  Outputs: gpurun_out/final4b/.
O=gpurun_out/final4b
mkdir -p $O
R=$(pwd)
export VLFB_PARITY_DIR=$R/$O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line > $R/$O/prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_mix -o stats -- python $R/bench.py --dtype mix --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_mix.log 2>&1
cd $R
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.txt $O/hbm_traffic.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-split-line --no-mix-line" > /dev/null 2>&1
python scratch/prof_summary.py $O/prof > $O/rocprofv3_kernel_stats.txt 2>&1
python scratch/prof_summary.py $O/prof_mix > $O/rocprofv3_kernel_stats_mix.txt 2>&1
python scratch/timeline2.py $O/prof $O/timeline_step.txt > /dev/null 2>&1
python scratch/timeline2.py $O/prof_mix $O/timeline_step_mix.txt > /dev/null 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof $O/prof_mix
cp $O/hbm_traffic.json profiles/hbm_traffic.json; cp $O/parity_fullsize_*.json profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_cmd.json 2> $O/bench_c4_driver_cmd.err
ls $O; head -c 400 $O/bench_c4_driver_cmd.json
