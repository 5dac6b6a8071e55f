#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04d; mkdir -p $out
for i in 1 2; do for v in shipped att0 one_stream att0_one_stream; do python3 tools/ab_forward.py $v 2>/dev/null; done; done | tee $out/ab.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --one-stream --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-60,200-330
[ -n "$f" ] && head -40 "$f" | cut -c1-400 > $out/kernel_stats_full_one_stream.csv
rm -rf $out/prof
