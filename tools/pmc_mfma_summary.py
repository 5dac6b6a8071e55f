"""Dev tool: per-kernel matrix-pipe busy table from the four single-counter passes of tools/profile_round4.sh
(pmcm_full_<COUNTER>.csv).  usage: python tools/pmc_mfma_summary.py <dir> > profiles/r04_pmc_mfma_full.csv"""
import collections, csv, sys
d = sys.argv[1]
names = ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY")
agg = collections.defaultdict(lambda: dict.fromkeys(names, 0.0) | {"n": 0})
for c in names:
    seen = collections.Counter()
    for r in csv.DictReader(open(f"{d}/pmcm_full_{c}.csv")):
        if r.get("Counter_Name") != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:70]
        agg[k][c] += float(r["Counter_Value"]); seen[k] += 1
    for k, n in seen.items():
        agg[k]["n"] = max(agg[k]["n"], n)
print("# rocprofv3 --pmc <counter> --kernel-trace (one pass per counter) -- python3 bench.py --workload full --one-stream --steps 1 --warmup 1 "
      "(MI355X, round 4, tools/profile_round4.sh); sums over the launches of both steps")
print("kernel,launches," + ",".join(names) + ",mfma_busy_over_active_x128,wait_over_wave_cycles")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:14]:
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's SIMDs at 1/4 rate (guide: x4 / (GRBM_GUI_ACTIVE x 1024 SIMDs) = / (active x 256)); kept as
    # busy / (active x 128) for continuity with round 3's table
    print(f'"{k}",{v["n"]},' + ",".join(str(int(v[c])) for c in names)
          + f',{v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["GRBM_GUI_ACTIVE"] * 128, 1):.4f},{v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1):.4f}')
