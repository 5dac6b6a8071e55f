#!/bin/bash
# Dev tool (GPU box): issue / wait counters of the small-batch search kernel of the B = 256 forward (fullref), one counter group per
# rocprofv3 pass, kernel-trace only.   usage: bash tools/r06/pmc_small_search.sh   -> gpurun_out/r06/pmc_small_search.txt
export TMPDIR=/tmp
out=gpurun_out/r06pmc; mkdir -p $out gpurun_out/r06
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o p -- python3 bench.py --workload fullref --steps 6 --warmup 2 --cpu-rows 0 --exact-steps 0 --one-stream --no-one-stream-pass > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/p$i.csv
  rm -rf $out/p$i
done
python3 - <<'PY' > gpurun_out/r06/pmc_small_search.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("gpurun_out/r06pmc/p*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if not any(s in k for s in ("search_f32_multi", "merge_assign_multi", "cross_attention64_kernel")): continue
    print("==", k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[k][c]:16.1f} per launch  ({cnt[k][c]} launches)")
PY
tail -3 $out/p1.log; ls $out | head; cat gpurun_out/r06/pmc_small_search.txt
