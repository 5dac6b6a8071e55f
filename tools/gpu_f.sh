#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03; mkdir -p $out
timeout 600 python3 bench.py --workload full 2>/dev/null | tail -1 > $out/bench_full.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o p -- python3 bench.py --workload full --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_full.log 2>&1
f=$(find $out/prof_full -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $out/kernel_stats_full.csv; rm -rf $out/prof_full
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_full_$c -o p -- python3 bench.py --workload full --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/pmc_full_$c.log 2>&1
  f=$(find $out/pmc_full_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/pmc_full_$c.csv; rm -rf $out/pmc_full_$c
done
python3 tools/pmc_summary.py $out full > $out/pmc_summary_full.txt 2>&1
timeout 600 python3 bench.py 2>/dev/null | tail -1 > $out/bench_cfg3_b.json
cut -c1-300 $out/bench_full.json; head -12 $out/kernel_stats_full.csv | cut -c1-150; cat $out/pmc_summary_full.txt | head -8; cut -c1-200 $out/bench_cfg3_b.json
