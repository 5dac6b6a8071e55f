#!/bin/bash
# Dev tool (GPU box): bench lines + rocprofv3 kernel-stat summaries + per-dispatch timelines that profiles/r05_* hold.
# usage: bash tools/profile_round5.sh [tag]   (writes under gpurun_out/r05<tag>/)
export TMPDIR=/tmp
out=gpurun_out/r05$1; mkdir -p $out
python3 bench.py 2>$out/bench_cfg3.err | tail -1 > $out/bench_cfg3.json
python3 bench.py --workload full 2>$out/bench_full.err | tail -1 > $out/bench_full.json
python3 bench.py --workload fullref --steps 20 --warmup 3 2>$out/bench_fullref.err | tail -1 > $out/bench_fullref.json
python3 bench.py --workload refdefault 2>/dev/null | tail -1 > $out/bench_refdefault.json
for w in cfg3 full fullref; do
  extra=""; [ $w != cfg3 ] && extra="--one-stream --no-one-stream-pass --no-half-text-pass"
  steps=2; [ $w = fullref ] && steps=6
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps $steps --warmup 2 --cpu-rows 0 --exact-steps 0 $extra > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -60 "$f" | cut -c1-600 > $out/kernel_stats_$w.csv
  t=$(find $out/prof_$w -name "*kernel_trace.csv" | head -1)
  if [ -n "$t" ] && [ $w != cfg3 ]; then
    python3 tools/launch_census.py "$t" > $out/launch_census_$w.txt 2>&1
    python3 tools/timeline.py "$t" > $out/timeline_${w}_one_stream.txt 2>&1
  fi
  rm -rf $out/prof_$w
done
ls -la $out | head -40
