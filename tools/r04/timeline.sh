#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04h; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/timeline.py "$t" > $out/timeline_full_one_stream.txt 2>&1
rm -rf $out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/timeline.py "$t" > $out/timeline_full_multi.txt 2>&1
rm -rf $out/prof
tail -3 $out/timeline_full_one_stream.txt; tail -3 $out/timeline_full_multi.txt
