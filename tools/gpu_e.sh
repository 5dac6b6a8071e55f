#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/e; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_filter.py tests/test_gpu_modules.py -q -x > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
timeout 300 python3 tools/time_split_gemm.py > $out/gemm.log 2>&1
timeout 600 python3 bench.py --workload full --cpu-rows 0 > $out/bench_full.json 2> $out/bench_full.err
timeout 600 python3 bench.py --workload cfg4 --cpu-rows 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
tail -4 $out/tests.log; head -12 $out/gemm.log; cut -c1-200 $out/bench_full.json; python3 -c "
import json; d=json.load(open('$out/bench_cfg4.json')); r=d['roofline']; print('cfg4', d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['kernel_share_of_step'])"
