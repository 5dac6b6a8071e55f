#!/bin/bash
# Dev tool (GPU box): instruction-issue counters of the D = 64 filter kernel (refdefault), one counter group per pass, kernel-trace only.
# usage: bash tools/r05/pmc_valu_refdefault.sh   -> gpurun_out/valu/summary.txt
export TMPDIR=/tmp
out=gpurun_out/valu; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F16" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o p -- python3 bench.py --workload refdefault --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 --no-clock-probe > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/p$i.csv
  rm -rf $out/p$i
done
python3 - <<'PY' > $out/summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("gpurun_out/valu/p*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if not any(s in k for s in ("filter_rows64", "rescore", "rownorm", "filter_f16")): continue
    print("==", k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[k][c]:16.1f} per launch  ({cnt[k][c]} launches)")
PY
cat $out/summary.txt
