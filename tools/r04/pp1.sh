#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04c; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_filter.py -m gpu -q -x -k "attention or filter or search or rescore or scan" > $out/tests.log 2>&1; tail -5 $out/tests.log
timeout 600 python3 tools/time_attention_variants.py 0 2 > $out/att_variants.log 2>&1; cat $out/att_variants.log
timeout 600 python3 bench.py --workload full --one-stream --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/bench_full_one.json 2> $out/bench_full_one.err; cut -c1-200 $out/bench_full_one.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
rm -rf $out/prof
