#!/bin/bash
# Dev tool (GPU box): per-dispatch timeline of one full-forward step -- which stream runs what, when (tools/timeline.py)
export TMPDIR=/tmp
out=gpurun_out/j; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --workload full --steps 3 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-one-stream-pass > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $out/timeline.txt 2>&1
rm -rf $out/prof
tail -2 $out/timeline.txt
