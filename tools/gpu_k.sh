#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
  for g in 4 2 6 32; do
    echo -n "grid $g x CUs: "; MEDTOK_DEV_IMG_GRID=$g timeout 600 python3 bench.py --workload full --cpu-rows 0 --exact-steps 0 --steps 10 --no-one-stream-pass 2>/dev/null | cut -c44-75
  done
done
