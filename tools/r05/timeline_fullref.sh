export TMPDIR=/tmp
out=gpurun_out/tl; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --workload fullref --steps 6 --warmup 2 --cpu-rows 0 --exact-steps 0 --one-stream --no-one-stream-pass > $out/prof.log 2>&1
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $out/timeline.txt <<'PY'
import csv,sys,re
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
marks=[i for i,n in enumerate(names) if "cross_attention64_kernel" in n]
a,b=marks[-3],marks[-2]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-t0
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} us  {re.sub(r'^void ','',r['Kernel_Name'])[:90]}")
print("dispatches per forward", b-a)
PY
cat $out/timeline.txt
rm -rf $out/prof
