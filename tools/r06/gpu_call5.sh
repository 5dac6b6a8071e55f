#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_small_width.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8
python bench.py --workload fullref --steps 50 --warmup 5 --cpu-rows 0 > gpurun_out/r06/bench_fullref_2pass.json 2> gpurun_out/r06/bench_fullref_2pass.err; tail -2 gpurun_out/r06/bench_fullref_2pass.err
python bench.py --workload fullref --rows 4096 --steps 10 --warmup 2 --cpu-rows 0 > gpurun_out/r06/bench_fullref_rows4096_2pass.json 2>/dev/null
python - <<'PY'
import json
for f in ["bench_fullref_2pass","bench_fullref_rows4096_2pass"]:
    try:
        d=json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"],4), d["roofline"].get("kernel"), d["roofline"]["frac"], (d.get("hip_graph_replay") or {}))
    except Exception as e:
        print(f,"ERR",e)
PY
bash tools/r05/timeline_fullref.sh > /dev/null 2>&1; cat gpurun_out/tl/timeline.txt
