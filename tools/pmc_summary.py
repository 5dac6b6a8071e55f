"""Dev tool: per-kernel fabric traffic from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round3.sh (KB per dispatch in the
csv).  Prints raw counter bytes per launch; corrections (x2 on FETCH_SIZE for wide streaming reads, the gather calibration) are
applied by the reader, not here.  usage: python tools/pmc_summary.py <dir> <workload> ..."""
import collections, csv, json, sys
d, names = sys.argv[1], sys.argv[2:]
res = {}
for w in names:
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = list(csv.DictReader(open(f"{d}/pmc_{w}_{c}.csv")))
        except FileNotFoundError:
            continue
        seen = collections.Counter()
        for r in rows:
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:70]
            agg[k][c] += float(r["Counter_Value"]) * 1024.0
            seen[k] += 1
        for k, n in seen.items():
            agg[k]["n"] = max(agg[k]["n"], n)
    res[w] = {k: {"launches": v["n"], "fetch_bytes_raw": v["FETCH_SIZE"] / max(v["n"], 1), "write_bytes_raw": v["WRITE_SIZE"] / max(v["n"], 1)}
              for k, v in agg.items() if v["n"]}
    print(f"== {w}")
    for k, v in sorted(res[w].items(), key=lambda kv: -(kv[1]["fetch_bytes_raw"] + kv[1]["write_bytes_raw"]) * kv[1]["launches"])[:14]:
        print(f"  {k:70s} x{v['launches']:3d}  FETCH_SIZE {v['fetch_bytes_raw']/1e9:8.3f} GB  WRITE_SIZE {v['write_bytes_raw']/1e9:8.3f} GB  per launch (raw)")
json.dump(res, open(f"{d}/pmc_raw.json", "w"), indent=1)
