#!/bin/bash
# Dev tool (GPU box): the VQ-side training step (cfg 4, precomputed encoder outputs) as a TIMELINE of its last step -- every launch with its
# start offset, duration and the idle gap before it -- plus busy time / wall of the step and the per-kernel sums.
# usage: bash tools/r06/prof_cfg4_timeline.sh   -> gpurun_out/c4t/{timeline.txt,summary.txt}
export TMPDIR=/tmp
out=gpurun_out/c4t; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --workload cfg4 --precomputed-encoders --steps 3 --warmup 2 --cpu-rows 0 --no-extra-workloads > $out/prof.log 2>&1
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" $out <<'PY'
import csv,sys,collections,re
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
out=sys.argv[2]
nm=lambda r: re.sub(r'^void ','',r["Kernel_Name"])
# steps end with the optimizer's last multi_tensor_apply group: cut at usage_multi_finish_kernel (one per forward)
marks=[i for i,r in enumerate(rows) if "usage_multi_finish" in r["Kernel_Name"]]
lo,hi=marks[-2],marks[-1]
step=rows[lo:hi]
t0=int(step[0]["Start_Timestamp"]); prev=t0; busy=0
with open(out+"/timeline.txt","w") as f:
    for r in step:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        f.write(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  gap {(s-prev)/1e3:7.1f}  {nm(r)[:110]}\n")
        busy+=e-s; prev=max(prev,e)
wall=int(step[-1]["End_Timestamp"])-t0
c=collections.Counter(); d=collections.Counter()
for r in step:
    n=nm(r)[:90]; c[n]+=1; d[n]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
with open(out+"/summary.txt","w") as f:
    f.write(f"one step (between two usage_multi_finish launches): {len(step)} launches, wall {wall/1e6:.3f} ms, kernel-busy {busy/1e6:.3f} ms\n")
    for n,v in d.most_common(60): f.write(f"{v/1e3:9.1f} us x{c[n]:4d}  {n}\n")
print(open(out+"/summary.txt").read())
PY
rm -rf $out/prof
