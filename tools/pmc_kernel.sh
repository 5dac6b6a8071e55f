#!/bin/bash
# Dev tool (GPU box): SQ counters of one kernel from a small driver script.  usage: bash tools/pmc_kernel.sh <script.py> <tag> <kernel-name-substring>
export TMPDIR=/tmp
out=gpurun_out/pmc; mkdir -p $out
script=$1; tag=$2; kname=$3
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_$i -o p -- python3 $script > $out/${tag}_$i.log 2>&1
  f=$(find $out/${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_$i.csv
  rm -rf $out/${tag}_$i
done
python3 - $tag "$kname" <<'PY'
import csv, sys, collections, glob
tag, kname = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); n = collections.Counter()
for f in sorted(glob.glob(f"gpurun_out/pmc/{tag}_[0-9].csv")):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open(f"gpurun_out/pmc/{tag}_summary.txt", "w") as o:
    for k in sorted(agg):
        o.write(f"{k:32s} {agg[k] / max(n[k], 1):14.5g}  (per launch, {n[k]} launches)\n")
print(open(f"gpurun_out/pmc/{tag}_summary.txt").read())
PY
