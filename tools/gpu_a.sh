#!/bin/bash
# round-3 GPU session A: new tests first, then the whole GPU suite, then bench lines + kernel stats of full and cfg4
export TMPDIR=/tmp
out=gpurun_out/a; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_distributed.py -q -x -s > $out/new_tests.log 2>&1; echo "new tests rc=$?" >> $out/new_tests.log
timeout 2400 python3 -m pytest tests -m gpu -q -s --deselect tests/test_gpu_split_gemm.py --deselect tests/test_gpu_distributed.py > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
timeout 600 python3 bench.py --workload cfg4 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
for w in full cfg4; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -45 "$f" > $out/kernel_stats_$w.csv
  rm -rf $out/prof_$w
done
tail -5 $out/new_tests.log; tail -8 $out/gpu_tests.log; cat $out/bench_full.json | cut -c1-400
