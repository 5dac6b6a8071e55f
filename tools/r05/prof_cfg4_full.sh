#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats of the whole cfg 4 step (stand-in encoders included).   usage: bash tools/r05/prof_cfg4_full.sh
export TMPDIR=/tmp
out=gpurun_out/c4full; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload cfg4 --steps 3 --warmup 2 --cpu-rows 0 --no-clock-probe > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-260 > $out/kernel_stats_cfg4.csv
rm -rf $out/prof
cat $out/kernel_stats_cfg4.csv | awk -F'","' '{printf "%10.1f us/step x%6.1f  %s\n", $3/5000, $2/5, substr($1,2,90)}' | head -32
