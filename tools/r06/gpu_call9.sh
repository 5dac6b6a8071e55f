#!/bin/bash
mkdir -p gpurun_out/r06
python tools/r06/crossover_small.py 2>&1 | grep "^D=" > gpurun_out/r06/crossover_after.txt; grep -c "AUTO" gpurun_out/r06/crossover_after.txt; awk '{ if (($0 ~ /filter faster/ && $0 !~ /takes filter/) || ($0 ~ /f32 faster/ && $0 !~ /takes f32/)) print "MISPICK: " $0 }' gpurun_out/r06/crossover_after.txt
for r in 256 512 1024; do
  python bench.py --workload full --rows $r --steps 20 --warmup 3 --cpu-rows 0 --exact-steps 0 --no-half-text-pass --no-one-stream-pass --no-clock-probe 2>/dev/null | tail -1 > gpurun_out/r06/full_rows_${r}_newrule.json
done
python - <<'PY'
import json
for r in (256,512,1024):
    d=json.loads(open(f"gpurun_out/r06/full_rows_{r}_newrule.json").read().strip().splitlines()[-1])
    print(r, round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["kernel"])
PY
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_modules.py tests/test_gpu_small_width.py tests/test_gpu_no_host_read.py -x -q 2>&1 | tail -4
