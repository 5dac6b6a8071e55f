#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04e; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_filter.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py tests/test_gpu_robustness.py tests/test_gpu_modules.py tests/test_gpu_split_gemm.py -m gpu -q -x > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-400 > $out/kernel_stats_full_one_stream.csv
rm -rf $out/prof
python3 - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r04e/kernel_stats_full_one_stream.csv')))
for r in rows[1:14]:
    try: print(f"{r[0][:55]:55s} calls {int(r[1]):4d}  {float(r[2])/5/1e6:7.3f} ms/step avg {float(r[3])/1e3:8.1f} us")
    except Exception: pass
PY
timeout 600 python3 bench.py --workload full --cpu-rows 0 --exact-steps 1 > $out/bench_full.json 2> $out/bench_full.err; cut -c1-200 $out/bench_full.json; python3 -c "
import json; d=json.load(open('$out/bench_full.json')); print(d['one_stream'], d['exact_fp32_path']['outputs_bit_identical_to_default_path'])"
