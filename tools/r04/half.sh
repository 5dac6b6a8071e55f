#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04i; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py tests/test_gpu_robustness.py -m gpu -q -x > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout 900 python3 bench.py --workload full --cpu-rows 0 --exact-steps 0 > $out/bench_full.json 2> $out/bench_full.err; tail -2 $out/bench_full.err
python3 -c "
import json; d=json.load(open('$out/bench_full.json')); print(d['value'], d['one_stream'], d.get('half_precision_text')); print(d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
