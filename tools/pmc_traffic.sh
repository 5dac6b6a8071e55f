#!/bin/bash
# Dev tool (GPU box): fabric-side traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE are
# KB per dispatch; FETCH_SIZE is doubled per MI355X_MICROARCH.md, section HBM).  usage: bash tools/pmc_traffic.sh <workload> ...
export TMPDIR=/tmp
out=gpurun_out/pmc; mkdir -p $out
for w in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/${w}_$c.log 2>&1
    f=$(find $out/${w}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $out/${w}_$c.csv
    rm -rf $out/${w}_$c
  done
done
python3 - "$@" <<'PY'
import csv, json, sys, collections
out = {}
for w in sys.argv[1:]:
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = list(csv.DictReader(open(f"gpurun_out/pmc/{w}_{c}.csv")))
        except FileNotFoundError:
            continue
        seen = collections.Counter()
        for r in rows:
            if r.get("Counter_Name") != c: continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
            agg[k][c] += float(r["Counter_Value"]) * 1024.0
            seen[k] += 1
        for k, n in seen.items(): agg[k]["n"] = max(agg[k]["n"], n)
    out[w] = {k: {"launches": v["n"], "fetch_bytes_x2": 2 * v["FETCH_SIZE"] / max(v["n"], 1), "write_bytes": v["WRITE_SIZE"] / max(v["n"], 1),
                  "total_bytes": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / max(v["n"], 1)} for k, v in agg.items() if v["n"]}
rows = {"cfg3": 600000, "cfg2": 100000, "full": 4096, "cfg5": 600000}
flat = {"_meta": {"taken": "round 2 (tools/pmc_traffic.sh)", "rows": {w: rows.get(w) for w in out}}}
for w, d in out.items(): flat[w] = {k: v["total_bytes"] for k, v in d.items()}
json.dump(flat, open("gpurun_out/pmc/pmc_traffic.json", "w"), indent=1)
json.dump(out, open("gpurun_out/pmc/summary.json", "w"), indent=1)
for w, d in out.items():
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["total_bytes"] * kv[1]["launches"])[:12]:
        print(w, k[:50], v)
PY
