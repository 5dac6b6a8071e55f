#!/bin/bash
# round-4 baseline: one-stream kernel trace of the full forward (true per-kernel durations) + the bench lines
export TMPDIR=/tmp
out=gpurun_out/r04a; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full1 -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof_full1.log 2>&1
f=$(find $out/prof_full1 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" | cut -c1-400 > $out/kernel_stats_full_one_stream.csv
t=$(find $out/prof_full1 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/timeline.py "$t" > $out/timeline_full_one_stream.txt 2>&1
rm -rf $out/prof_full1
timeout 600 python3 bench.py --workload full --cpu-rows 0 --exact-steps 0 > $out/bench_full.json 2> $out/bench_full.err
timeout 600 python3 bench.py --workload full --one-stream --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/bench_full_one.json 2> $out/bench_full_one.err
cut -c1-200 $out/bench_full.json $out/bench_full_one.json
head -30 $out/kernel_stats_full_one_stream.csv | cut -c1-160
