#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04j; mkdir -p $out
timeout 2400 python3 -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_modules.py tests/test_gpu_train_step.py tests/test_gpu_train_kernels.py tests/test_gpu_full_size.py -m gpu -q -x > $out/tests.log 2>&1; tail -6 $out/tests.log | cut -c1-300
timeout 900 python3 bench.py --workload cfg4 --precomputed-encoders --cpu-rows 0 > $out/bench_cfg4_vq_only.json 2> $out/bench_cfg4_vq_only.err; cut -c1-200 $out/bench_cfg4_vq_only.json; tail -2 $out/bench_cfg4_vq_only.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload cfg4 --precomputed-encoders --steps 3 --warmup 2 --cpu-rows 0 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" | cut -c1-300 > $out/kernel_stats_cfg4_vq_only.csv
rm -rf $out/prof
python3 - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r04j/kernel_stats_cfg4_vq_only.csv')))
tot=0
for r in rows[1:]:
    try: tot+=float(r[2])
    except Exception: pass
print("sum of listed kernel time per step: %.2f ms" % (tot/5/1e6))
for r in rows[1:26]:
    try: print(f"  {r[0][:70]:70s} calls {int(r[1]):5d}  {float(r[2])/5/1e6:7.3f} ms/step")
    except Exception: pass
print("Cijk kernels:", [r[0][:40] for r in rows[1:] if r and r[0].startswith('Cijk')][:5])
PY
timeout 900 python3 bench.py --workload cfg4 --cpu-rows 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; cut -c1-200 $out/bench_cfg4.json
