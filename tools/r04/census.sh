#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04l; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do python3 tools/r04/ab_fullref.py; python3 tools/r04/ab_fullref.py SPLIT_MIN_ROWS=1; python3 tools/r04/ab_fullref.py SPLIT_MIN_ROWS=1 COMBINE_MAX_EXTRA_FLOPS=-1; done 2>/dev/null
for w in fullref full; do
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$w -o p -- python3 bench.py --workload $w --one-stream --steps 6 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass --no-half-text-pass > $out/tr_$w.log 2>&1
  t=$(find $out/tr_$w -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/launch_census.py "$t" | tee $out/launch_census_$w.txt
  rm -rf $out/tr_$w
done
