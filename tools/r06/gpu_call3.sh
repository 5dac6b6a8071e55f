#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_small_width.py -x -q 2>&1 | tail -12 > gpurun_out/r06/tests_small.txt; cat gpurun_out/r06/tests_small.txt
python bench.py --workload fullref --steps 50 --warmup 5 --cpu-rows 0 > gpurun_out/r06/bench_fullref.json 2> gpurun_out/r06/bench_fullref.err; tail -2 gpurun_out/r06/bench_fullref.err
python bench.py --workload fullref --rows 4096 --steps 10 --warmup 2 --cpu-rows 0 > gpurun_out/r06/bench_fullref_rows4096.json 2>> gpurun_out/r06/bench_fullref.err
python bench.py --data clustered_codebook --steps 3 --warmup 1 --cpu-rows 0 > gpurun_out/r06/bench_cfg3_clustered_codebook_cap112.json 2> gpurun_out/r06/bench_cfg3_clustered2.err; tail -2 gpurun_out/r06/bench_cfg3_clustered2.err
python bench.py --steps 5 --warmup 1 --cpu-rows 0 --no-extra-workloads > gpurun_out/r06/bench_cfg3_cap112.json 2>/dev/null
python - <<'PY'
import json
for f in ["bench_fullref","bench_fullref_rows4096","bench_cfg3_clustered_codebook_cap112","bench_cfg3_cap112"]:
    try:
        d=json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"],3), d["roofline"].get("kernel"), d["roofline"]["frac"], (d.get("hip_graph_replay") or {}).get("value"),
              (d.get("fallback_rows") or {}).get("rows_handed_to_the_exact_kernel_per_step"), d["roofline"].get("other_kernels",{}).keys())
    except Exception as e:
        print(f,"ERR",e)
PY
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/tests_full.txt; cat gpurun_out/r06/tests_full.txt
