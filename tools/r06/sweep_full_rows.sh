#!/bin/bash
# verdict r05 item 1a: is the full forward faster per code when a call's qf / context fit the 256 MB Infinity Cache?  rows per call swept.
mkdir -p gpurun_out/r06
for r in 256 512 1024 2048 4096 8192; do
  python bench.py --workload full --rows $r --steps 10 --warmup 3 --cpu-rows 0 --exact-steps 0 --no-half-text-pass --no-one-stream-pass --no-clock-probe > gpurun_out/r06/full_rows_$r.json 2> gpurun_out/r06/full_rows_$r.err
  python bench.py --workload full --rows $r --one-stream --steps 10 --warmup 3 --cpu-rows 0 --exact-steps 0 --no-half-text-pass --no-one-stream-pass --no-clock-probe > gpurun_out/r06/full_rows_${r}_one_stream.json 2>> gpurun_out/r06/full_rows_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/full_rows_*.json'), key=lambda s:(len(s),s)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
