#!/bin/bash
# round-3 final GPU session: whole GPU suite, then the full-forward bench line and kernel stats
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
timeout 3000 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
tail -4 $out/gpu_tests.log
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
cut -c1-180 $out/bench_full.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o p -- python3 bench.py --workload full --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof_full.log 2>&1
f=$(find $out/prof_full -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > $out/kernel_stats_full.csv
rm -rf $out/prof_full
head -8 $out/kernel_stats_full.csv | cut -c1-120
