#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_gpu_dropout_stats.py tests/test_gpu_small_width.py -x -q 2>&1 | tail -15 > gpurun_out/r06/tests_a.txt
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -k "f20 or f21 or f22 or f23" 2>&1 | tail -15 > gpurun_out/r06/tests_b.txt
cat gpurun_out/r06/tests_a.txt gpurun_out/r06/tests_b.txt
( time python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err ) 2> gpurun_out/r06/bench_default.time
tail -3 gpurun_out/r06/bench_default.err; cat gpurun_out/r06/bench_default.time
for d in near_codes clustered_codebook heavy_tail; do
  python bench.py --data $d --steps 3 --warmup 1 --cpu-rows 0 > gpurun_out/r06/bench_cfg3_$d.json 2> gpurun_out/r06/bench_cfg3_$d.err
  tail -2 gpurun_out/r06/bench_cfg3_$d.err
done
bash tools/r06/chunk_kernel_time.sh 2>&1 | tail -6
python tools/r06/ab_wv_epilogue.py > gpurun_out/r06/ab_wv_epilogue.txt 2>&1; cat gpurun_out/r06/ab_wv_epilogue.txt | tail -8
python - <<'PY'
import json
for f in ["bench_default","bench_cfg3_near_codes","bench_cfg3_clustered_codebook","bench_cfg3_heavy_tail"]:
    try:
        d=json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"],2), d["roofline"]["frac"], (d.get("fallback_rows") or {}).get("rows_handed_to_the_exact_kernel_per_step"),
              [round(s["candidates_per_row"],1) for s in (d.get("fallback_rows") or {}).get("searches",[])])
        if "extra" in d and "workloads" in d["extra"]:
            for k,v in d["extra"]["workloads"].items():
                if isinstance(v,dict): print("   ",k, v.get("value"), v.get("ms_per_step"), v.get("dominant_kernel"), v.get("frac"), v.get("clock_ghz"), v.get("error"), v.get("hip_graph_replay"))
    except Exception as e:
        print(f,"ERR",e)
PY
