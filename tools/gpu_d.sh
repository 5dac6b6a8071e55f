#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/d; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "attention" > $out/att_tests.log 2>&1; echo "rc=$?" >> $out/att_tests.log
timeout 300 python3 tools/time_attention3.py > $out/time3.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_train_step.py -q > $out/mod_tests.log 2>&1; echo "rc=$?" >> $out/mod_tests.log
timeout 600 python3 bench.py --workload full --cpu-rows 0 > $out/bench_full.json 2> $out/bench_full.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o p -- python3 bench.py --workload full --steps 3 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_full.log 2>&1
f=$(find $out/prof_full -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $out/kernel_stats_full.csv; rm -rf $out/prof_full
tail -15 $out/att_tests.log; cat $out/time3.log; tail -12 $out/mod_tests.log; cut -c1-200 $out/bench_full.json
