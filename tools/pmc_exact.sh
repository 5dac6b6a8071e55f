#!/bin/bash
# Dev tool (GPU box): clock and matrix-pipe counters of the exact fp32 search kernel (rocprofv3 --pmc, own pass).  usage: bash tools/pmc_exact.sh [n k d topk]
export TMPDIR=/tmp
out=gpurun_out/pmc_exact; mkdir -p $out
n=${1:-600000}; k=${2:-16384}; d=${3:-768}; t=${4:-5}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/run -o p -- python3 tools/one_search.py $n $k $d $t 1 > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_exact/run/**/*counter_collection.csv", recursive=True)[0]
t = glob.glob("gpurun_out/pmc_exact/run/**/*kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    dur[r["Kernel_Name"].split("<")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("<")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_exact/summary.csv", "w") as o:
    o.write("kernel,launches,ms_each,GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,clock_GHz,mfma_busy_frac\n")
    for k, v in agg.items():
        if "search_f32" not in k: continue
        g, m = v["GRBM_GUI_ACTIVE"], v["SQ_VALU_MFMA_BUSY_CYCLES"]
        ms = dur[k]
        for i in range(len(g)):
            o.write(f"{k},{i},{ms[i]:.3f},{g[i]:.4g},{v['SQ_BUSY_CU_CYCLES'][i]:.4g},{m[i]:.4g},{g[i]/ms[i]/1e6:.3f},{m[i]/(g[i]*128):.3f}\n")
print(open("gpurun_out/pmc_exact/summary.csv").read())
PY
