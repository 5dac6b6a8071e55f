#!/bin/bash
# verdict r05 item 1a, second half: per-code KERNEL time of the full forward (one stream, rocprofv3 --kernel-trace --stats) at several
# codes per call.  A call of <= ~1024 codes keeps its qf / context (<= 256 MB each) inside the 256 MB Infinity Cache; if cache-resident
# chunks paid, the attention / GEMM kernels would take LESS time per code there than at 4096 codes per call (host launch cost excluded:
# these are device durations).  usage: bash tools/r06/chunk_kernel_time.sh   (writes gpurun_out/r06/chunk_kernel_time.txt)
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
: > $out/chunk_kernel_time.txt
for r in 512 1024 2048 4096; do
  steps=$((16384 / r)); [ $steps -gt 16 ] && steps=16
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_chunk_$r -o p -- python3 bench.py --workload full --rows $r --one-stream --steps $steps --warmup 2 --cpu-rows 0 --exact-steps 0 --no-half-text-pass --no-one-stream-pass --no-clock-probe > $out/prof_chunk_$r.log 2>&1
  f=$(find $out/prof_chunk_$r -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $r $steps >> $out/chunk_kernel_time.txt <<'PY'
import csv, sys
f, rows, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
calls = steps + 2                      # warm-up steps run the same kernels
tot = {}
for r in csv.DictReader(open(f)):
    n = r["Name"]
    key = ("attention_pp" if "attention_pp" in n else "attention_other" if "attention" in n else "split_gemm" if "split_gemm" in n else
           "filter_f16" if "filter_f16" in n else "rescore" if "rescore" in n else "layernorm" if "layernorm" in n else
           "split_half/images" if ("split_half" in n or "half_image" in n) else "rownorm" if "rownorm" in n else "other")
    tot[key] = tot.get(key, 0.0) + float(r["TotalDurationNs"])
codes = rows * calls
print(f"rows/call {rows}: kernel ns per code -- " + ", ".join(f"{k} {v / codes:.0f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])) + f" | all {sum(tot.values()) / codes:.0f}")
PY
  rm -rf $out/prof_chunk_$r
done
cat $out/chunk_kernel_time.txt
