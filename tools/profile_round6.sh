#!/bin/bash
# Dev tool (GPU box): the bench lines, rocprofv3 kernel-stat summaries, timelines and PMC passes that profiles/r06_* hold.
# usage: bash tools/profile_round6.sh   (writes under gpurun_out/r06p/)
export TMPDIR=/tmp
out=gpurun_out/r06p; mkdir -p $out
( time python3 bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err ) 2> $out/bench_cfg3.time
python3 bench.py --workload full 2>/dev/null | tail -1 > $out/bench_full.json
python3 bench.py --workload full --rows 256 --steps 20 --warmup 3 2>/dev/null | tail -1 > $out/bench_full_rows256.json
python3 bench.py --workload fullref --steps 50 --warmup 5 2>/dev/null | tail -1 > $out/bench_fullref.json
python3 bench.py --workload fullref --rows 4096 --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/bench_fullref_rows4096.json
python3 bench.py --workload refdefault 2>/dev/null | tail -1 > $out/bench_refdefault.json
python3 bench.py --workload cfg2 2>/dev/null | tail -1 > $out/bench_cfg2.json
python3 bench.py --workload cfg5 2>/dev/null | tail -1 > $out/bench_cfg5.json
python3 bench.py --workload codeshard 2>/dev/null | tail -1 > $out/bench_codeshard.json
python3 bench.py --workload cfg4 --precomputed-encoders --steps 20 --warmup 20 2>/dev/null | tail -1 > $out/bench_cfg4_vq_only.json
python3 bench.py --workload cfg4 --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/bench_cfg4.json
for d in near_codes clustered_codebook heavy_tail; do
  python3 bench.py --data $d --steps 3 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $out/bench_cfg3_data_$d.json
done
python3 bench.py --workload refdefault --data clustered_codebook --steps 3 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > $out/bench_refdefault_data_clustered_codebook.json
for w in cfg3 full fullref; do
  extra=""; [ $w = full ] || [ $w = fullref ] && extra="--one-stream --no-one-stream-pass --no-half-text-pass"
  steps=2; [ $w = fullref ] && steps=6
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps $steps --warmup 2 --cpu-rows 0 --exact-steps 0 --no-extra-workloads $extra > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -60 "$f" | cut -c1-600 > $out/kernel_stats_$w.csv
  t=$(find $out/prof_$w -name "*kernel_trace.csv" | head -1)
  if [ -n "$t" ] && [ $w = full ]; then
    python3 tools/launch_census.py "$t" > $out/launch_census_$w.txt 2>&1
    python3 tools/timeline.py "$t" > $out/timeline_${w}_one_stream.txt 2>&1
  fi
  rm -rf $out/prof_$w
done
bash tools/r05/timeline_fullref.sh > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $out/timeline_fullref.txt
bash tools/r06/pmc_small_search.sh > /dev/null 2>&1; cp gpurun_out/r06/pmc_small_search.txt $out/pmc_small_search_after.txt
# PMC traffic of the headline's kernels, their own passes (no trace domains beside --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_cfg3_$c -o p -- python3 bench.py --workload cfg3 --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 --no-clock-probe --no-extra-workloads > $out/pmc_cfg3_$c.log 2>&1
  f=$(find $out/pmc_cfg3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/pmc_cfg3_$c.csv
  rm -rf $out/pmc_cfg3_$c
done
python3 tools/pmc_summary.py $out cfg3 > $out/pmc_summary.txt 2>&1
rm -f $out/pmc_cfg3_FETCH_SIZE.csv $out/pmc_cfg3_WRITE_SIZE.csv
timeout 1500 python3 -m pytest tests/ -x -q -m gpu 2>&1 | tail -12 > $out/gpu_suite_tail.txt
ls -la $out | head -60
