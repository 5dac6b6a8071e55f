#!/bin/bash
# Dev tool (GPU box): the VQ-side training step (cfg 4 with precomputed encoder outputs) -- bench line, rocprofv3 kernel stats, launches per
# step and the per-launch timeline of one step; then the whole cfg 4 step with the stand-in encoders.
# -> profiles/r06_bench_cfg4_vq_only.json, r06_kernel_stats_cfg4_vq_only.csv, r06_launches_per_step_cfg4_vq_only.txt, r06_timeline_cfg4_vq_only.txt,
#    r06_bench_cfg4.json (copied from gpurun_out/c4 by hand).    usage: bash tools/r06/prof_cfg4.sh
export TMPDIR=/tmp
out=gpurun_out/c4; mkdir -p $out
python3 bench.py --workload cfg4 --precomputed-encoders --steps 20 --warmup 20 2>/dev/null | tail -1 > $out/bench_cfg4_vq_only.json    # (first process on a fresh box: a long warm-up)
python3 bench.py --workload cfg4 --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/bench_cfg4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload cfg4 --precomputed-encoders --steps 3 --warmup 2 --cpu-rows 0 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" | cut -c1-400 > $out/kernel_stats_cfg4_vq_only.csv
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" $out <<'PY'
import csv,sys,collections,re
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
out=sys.argv[2]
nm=lambda r: re.sub(r'^void ','',r["Kernel_Name"])
marks=[i for i,r in enumerate(rows) if "usage_multi_finish" in r["Kernel_Name"]]
step=rows[marks[-2]:marks[-1]]
t0=int(step[0]["Start_Timestamp"]); prev=t0; busy=0
with open(out+"/timeline_cfg4_vq_only.txt","w") as f:
    f.write("# one training step, launch by launch: start offset (us), duration, idle gap in front of it, kernel (rocprofv3 --kernel-trace; cut at usage_multi_finish_kernel = end of a forward)\n")
    for r in step:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        f.write(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  gap {(s-prev)/1e3:7.1f}  {nm(r)[:120]}\n")
        busy+=e-s; prev=max(prev,e)
wall=int(step[-1]["End_Timestamp"])-t0
c=collections.Counter(); d=collections.Counter()
for r in step:
    n=nm(r)[:100]; c[n]+=1; d[n]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
lib=sum(v for n,v in c.items() if not n.startswith("at::") and not n.startswith("__amd_rocclr"))
with open(out+"/launches_per_step_cfg4_vq_only.txt","w") as f:
    f.write(f"one step under rocprofv3: {len(step)} launches ({lib} of the library's own kernels, {len(step)-lib} torch / runtime), wall {wall/1e6:.3f} ms, kernel-busy {busy/1e6:.3f} ms\n")
    for n,v in d.most_common(70): f.write(f"{v/1e3:9.1f} us x{c[n]:4d}  {n}\n")
print(open(out+"/launches_per_step_cfg4_vq_only.txt").read()[:1500])
PY
rm -rf $out/prof
python3 -c "
import json
for n in ('bench_cfg4_vq_only','bench_cfg4'):
    d=json.load(open('$out/'+n+'.json')); print(n, d['value'], d['ms_per_step'], d['config'].get('encoders_ms_per_step'))"
