#!/bin/bash
# round-3 final GPU session: whole GPU suite, then the headline and full-forward bench lines
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
timeout 3000 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
tail -4 $out/gpu_tests.log
timeout 600 python3 bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
cut -c1-180 $out/bench_cfg3.json; cut -c1-180 $out/bench_full.json
