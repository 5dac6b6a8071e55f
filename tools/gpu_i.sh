#!/bin/bash
# round-3 GPU session I: the 4-image split GEMM in the forward -- tests that touch it, then bench lines
export TMPDIR=/tmp
out=gpurun_out/i; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_modules.py tests/test_gpu_full_size.py tests/test_gpu_filter.py tests/test_gpu_kernels.py -q -x > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -4 $out/tests.log
timeout 600 python3 bench.py --workload full > $out/bench_full.json 2> $out/bench_full.err
timeout 600 python3 bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err
cut -c1-330 $out/bench_full.json; cut -c1-330 $out/bench_cfg3.json
