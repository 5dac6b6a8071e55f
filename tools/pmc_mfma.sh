#!/bin/bash
# Dev tool (GPU box): matrix-pipe utilisation counters per kernel (rocprofv3 --pmc, its own pass).  usage: bash tools/pmc_mfma.sh <workload>
export TMPDIR=/tmp
out=gpurun_out/pmc; mkdir -p $out
w=${1:-cfg3}
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${w}_$tag -o p -- python3 bench.py --workload $w --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/${w}_$tag.log 2>&1
  f=$(find $out/${w}_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${w}_$tag.csv
  rm -rf $out/${w}_$tag
done
python3 - $w <<'PY'
import csv, sys, collections, glob
w = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc/{w}_SQ_*.csv"):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
        for n in ("filter_f16_kernel", "to_half_kernel"):
            if n in k: k = n
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[(k, r["Counter_Name"])] += 1
    for (k, c), n in seen.items(): cnt[k] = max(cnt[k], n)
with open(f"gpurun_out/pmc/{w}_mfma_summary.csv", "w") as o:
    names = sorted({c for v in agg.values() for c in v})
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if k.startswith(("at::", "Cijk", "__amd", "rocprim")): continue
        o.write(f"{k[:50]},{cnt[k]}," + ",".join(f"{v.get(c, 0):.4g}" for c in names) + "\n")
print(open(f"gpurun_out/pmc/{w}_mfma_summary.csv").read())
PY
