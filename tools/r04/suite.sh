#!/bin/bash
# whole GPU suite + the full-forward bench lines (multi-stream and one-stream)
export TMPDIR=/tmp
out=gpurun_out/r04b; mkdir -p $out
timeout 2400 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
tail -5 $out/gpu_tests.log
timeout 600 python3 bench.py --workload full --cpu-rows 0 --exact-steps 0 > $out/bench_full.json 2> $out/bench_full.err
cut -c1-300 $out/bench_full.json; tail -3 $out/bench_full.err
