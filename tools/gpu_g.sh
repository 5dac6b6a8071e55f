#!/bin/bash
# round-3 GPU session G: whole GPU suite after the projection change; cfg3 / full bench lines + cfg3 kernel stats
export TMPDIR=/tmp
out=gpurun_out/g; mkdir -p $out
timeout 3000 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
timeout 600 python3 bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
for w in cfg3; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -45 "$f" > $out/kernel_stats_$w.csv
  rm -rf $out/prof_$w
done
tail -8 $out/gpu_tests.log; cut -c1-600 $out/bench_cfg3.json; cut -c1-300 $out/bench_full.json
