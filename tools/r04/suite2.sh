#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04m; mkdir -p $out
timeout 2400 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
tail -5 $out/gpu_tests.log
timeout 900 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err; cut -c1-200 $out/bench_full.json
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
