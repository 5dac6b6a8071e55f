#!/bin/bash
# Dev tool (GPU box): per-dispatch timeline of the D = 768 forward at the reference's batch (bench.py --workload full --rows 256, one stream)
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/tl256 -o p -- python3 bench.py --workload full --rows 256 --one-stream --steps 4 --warmup 2 --cpu-rows 0 --exact-steps 0 --no-half-text-pass --no-one-stream-pass --no-clock-probe > $out/tl256.log 2>&1
t=$(find $out/tl256 -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$t" > $out/timeline_full_rows256.txt 2>&1
rm -rf $out/tl256
cat $out/timeline_full_rows256.txt
