#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04g; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "pinned_to_the_reference" -s > $out/tests.log 2>&1; grep -E "census|passed|failed" $out/tests.log
timeout 900 python3 bench.py --workload refdefault > $out/bench_refdefault.json 2> $out/bench_refdefault.err; cut -c1-260 $out/bench_refdefault.json; tail -2 $out/bench_refdefault.err
timeout 900 python3 bench.py --workload fullref --steps 20 --warmup 3 > $out/bench_fullref.json 2> $out/bench_fullref.err; cut -c1-260 $out/bench_fullref.json; tail -2 $out/bench_fullref.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload refdefault --steps 2 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-400 > $out/kernel_stats_refdefault.csv
rm -rf $out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof2 -o p -- python3 bench.py --workload fullref --steps 20 --warmup 3 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof2.log 2>&1
f=$(find $out/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-400 > $out/kernel_stats_fullref.csv
rm -rf $out/prof2
python3 - <<'PY'
import csv
for w,steps in (("refdefault",3),("fullref",23)):
    rows=list(csv.reader(open(f'gpurun_out/r04g/kernel_stats_{w}.csv')))
    print(w)
    for r in rows[1:16]:
        try: print(f"  {r[0][:60]:60s} calls {int(r[1]):4d}  {float(r[2])/steps/1e6:7.3f} ms/step avg {float(r[3])/1e3:8.1f} us")
        except Exception: pass
PY
