#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/j; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_full_size.py tests/test_gpu_kernels.py -q -x > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -3 $out/tests.log
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
cut -c1-200 $out/bench_full.json
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $out/timeline.txt 2>&1
rm -rf $out/prof
tail -2 $out/timeline.txt
