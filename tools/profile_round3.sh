#!/bin/bash
# Dev tool (GPU box): the bench lines, rocprofv3 kernel-stat summaries and PMC traffic figures that profiles/r03_* hold.
# usage: bash tools/profile_round3.sh   (writes under gpurun_out/r03/)
export TMPDIR=/tmp
out=gpurun_out/r03; mkdir -p $out
python3 bench.py 2>/dev/null | tail -1 > $out/bench_cfg3.json
python3 bench.py --workload cfg2 2>/dev/null | tail -1 > $out/bench_cfg2.json
python3 bench.py --workload cfg5 2>/dev/null | tail -1 > $out/bench_cfg5.json
python3 bench.py --workload full 2>/dev/null | tail -1 > $out/bench_full.json
python3 bench.py --workload cfg4 2>/dev/null | tail -1 > $out/bench_cfg4.json
python3 bench.py --workload codeshard 2>/dev/null | tail -1 > $out/bench_codeshard.json
for w in cfg3 cfg2 cfg5 full cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" > $out/kernel_stats_$w.csv
  rm -rf $out/prof_$w
done
# PMC traffic, its own passes (no --stats, no trace domains besides the kernel trace)
for w in cfg3 cfg2 cfg5 full; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/pmc_${w}_$c.log 2>&1
    f=$(find $out/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $out/pmc_${w}_$c.csv
    rm -rf $out/pmc_${w}_$c
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_calib_$c -o p -- python3 tools/calib_gather.py > $out/pmc_calib_$c.log 2>&1
  f=$(find $out/pmc_calib_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/pmc_calib_$c.csv
  rm -rf $out/pmc_calib_$c
done
python3 tools/pmc_summary.py $out cfg3 cfg2 cfg5 full calib > $out/pmc_summary.txt 2>&1
ls -la $out | head -60; cat $out/pmc_summary.txt | head -80
