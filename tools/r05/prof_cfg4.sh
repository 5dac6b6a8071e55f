#!/bin/bash
# Dev tool (GPU box): the VQ-side training step (cfg 4 with precomputed encoder outputs) -- bench line, rocprofv3 kernel stats, launches per step:
# profiles/r05_bench_cfg4_vq_only.json, r05_kernel_stats_cfg4_vq_only.csv, r05_launches_per_step_cfg4_vq_only.txt.   usage: bash tools/r05/prof_cfg4.sh
export TMPDIR=/tmp
out=gpurun_out/c4; mkdir -p $out
python3 bench.py --workload cfg4 --precomputed-encoders 2>/dev/null | tail -1 > $out/bench_cfg4_vq_only.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload cfg4 --precomputed-encoders --steps 3 --warmup 2 --cpu-rows 0 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" | cut -c1-400 > $out/kernel_stats_cfg4_vq_only.csv
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $out/trace_summary.txt <<'PY'
import csv,sys,collections,re
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
# last step = between the last two multi_tensor_apply groups; simpler: count per name over all, divide by 5 steps
c=collections.Counter(); d=collections.Counter()
for r in rows:
    n=re.sub(r'^void ','',r["Kernel_Name"])[:60]; c[n]+=1; d[n]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
print("total launches", len(rows), "per step ~", len(rows)/5.0)
for n,v in d.most_common(40): print(f"{v/5e3:9.1f} us/step x{c[n]/5.0:6.1f}  {n}")
PY
rm -rf $out/prof
python3 -c "
import json; d=json.load(open('$out/bench_cfg4_vq_only.json')); print(d['value'], d['ms_per_step'])"
head -45 $out/trace_summary.txt
