#!/bin/bash
# Dev tool (GPU box): the bench lines, rocprofv3 kernel-stat summaries and PMC traffic figures that profiles/r04_* hold.
# usage: bash tools/profile_round4.sh   (writes under gpurun_out/r04/)
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
python3 bench.py 2>/dev/null | tail -1 > $out/bench_cfg3.json
python3 bench.py --workload full 2>/dev/null | tail -1 > $out/bench_full.json
python3 bench.py --workload cfg2 2>/dev/null | tail -1 > $out/bench_cfg2.json
python3 bench.py --workload cfg5 2>/dev/null | tail -1 > $out/bench_cfg5.json
python3 bench.py --workload cfg4 2>/dev/null | tail -1 > $out/bench_cfg4.json
python3 bench.py --workload cfg4 --precomputed-encoders 2>/dev/null | tail -1 > $out/bench_cfg4_vq_only.json
python3 bench.py --workload codeshard 2>/dev/null | tail -1 > $out/bench_codeshard.json
python3 bench.py --workload refdefault 2>/dev/null | tail -1 > $out/bench_refdefault.json
python3 bench.py --workload fullref --steps 20 --warmup 3 2>/dev/null | tail -1 > $out/bench_fullref.json
for w in cfg3 full refdefault; do
  extra=""; [ $w = full ] && extra="--no-one-stream-pass --no-half-text-pass"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 $extra > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" | cut -c1-600 > $out/kernel_stats_$w.csv
  rm -rf $out/prof_$w
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_f1 -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass --no-half-text-pass > $out/prof_f1.log 2>&1
f=$(find $out/prof_f1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-600 > $out/kernel_stats_full_one_stream.csv
rm -rf $out/prof_f1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -o p -- python3 bench.py --workload cfg4 --precomputed-encoders --steps 3 --warmup 2 --cpu-rows 0 > $out/prof_c4.log 2>&1
f=$(find $out/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" | cut -c1-600 > $out/kernel_stats_cfg4_vq_only.csv
rm -rf $out/prof_c4
# launches per forward by origin (B = 256 at the reference's default shape; B = 4096 at D = 768)
for w in fullref full; do
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$w -o p -- python3 bench.py --workload $w --one-stream --steps 6 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass --no-half-text-pass > $out/tr_$w.log 2>&1
  t=$(find $out/tr_$w -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/launch_census.py "$t" > $out/launch_census_$w.txt 2>&1
  rm -rf $out/tr_$w
done
# PMC traffic, its own passes (no --stats, no trace domains besides the kernel trace)
for w in cfg3 full refdefault; do
  extra=""; [ $w = full ] && extra="--one-stream --no-one-stream-pass --no-half-text-pass"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 $extra > $out/pmc_${w}_$c.log 2>&1
    f=$(find $out/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $out/pmc_${w}_$c.csv
    rm -rf $out/pmc_${w}_$c
  done
done
# matrix-pipe busy of the full forward's kernels
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcm_$c -o p -- python3 bench.py --workload full --one-stream --no-one-stream-pass --no-half-text-pass --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/pmcm_$c.log 2>&1
  f=$(find $out/pmcm_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/pmcm_full_$c.csv
  rm -rf $out/pmcm_$c
done
python3 tools/pmc_summary.py $out cfg3 full refdefault > $out/pmc_summary.txt 2>&1
ls -la $out | head -60; cat $out/pmc_summary.txt | head -60
