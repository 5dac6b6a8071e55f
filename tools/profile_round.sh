#!/bin/bash
# Dev tool (GPU box): refresh the bench lines and rocprofv3 kernel-stat summaries that profiles/ holds.
# usage: bash tools/profile_round.sh   (writes under gpurun_out/round/)
export TMPDIR=/tmp
out=gpurun_out/round; mkdir -p $out
python3 bench.py 2>/dev/null | tail -1 > $out/bench_cfg3.json
python3 bench.py --workload cfg2 2>/dev/null | tail -1 > $out/bench_cfg2.json
python3 bench.py --workload cfg5 2>/dev/null | tail -1 > $out/bench_cfg5.json
python3 bench.py --workload full 2>/dev/null | tail -1 > $out/bench_full.json
for w in cfg3 cfg2 full; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" > $out/kernel_stats_$w.csv
  rm -rf $out/prof_$w
done
ls -la $out
