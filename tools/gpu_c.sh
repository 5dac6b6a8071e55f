#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/c; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_train_step.py -q -x -k "kmeans or ema_codebook or fp16 or refuses or packed_cross" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
timeout 300 python3 tools/time_split_gemm.py > $out/gemm.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg4 -o p -- python3 bench.py --workload cfg4 --text-layers 0 --steps 3 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof_cfg4.log 2>&1
f=$(find $out/prof_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" > $out/kernel_stats_cfg4_noenc.csv; rm -rf $out/prof_cfg4
timeout 300 python3 bench.py --workload cfg4 --text-layers 0 --cpu-rows 0 > $out/bench_cfg4_noenc.json 2>$out/bench_cfg4_noenc.err
tail -4 $out/tests.log; cat $out/gemm.log
