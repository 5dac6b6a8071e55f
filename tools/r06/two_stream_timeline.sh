#!/bin/bash
# Dev tool (GPU box): per-dispatch timeline (rocprofv3 --kernel-trace) and shader clock of a cfg3 step with its four searches on ONE
# stream and on TWO (verdict r05 item 2: what the overlap hides and what it costs).  -> gpurun_out/r06/two_stream_timeline.txt
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
: > $out/two_stream_timeline.txt
for m in one two; do
  rocprofv3 --kernel-trace --output-format csv -d $out/ts_$m -o p -- python3 tools/r06/two_stream_step.py $m 2 > $out/ts_$m.log 2>&1
  t=$(find $out/ts_$m -name "*kernel_trace.csv" | head -1)
  echo "== cfg3 step, searches on $m stream(s); clock: $(grep streams $out/ts_$m.log | cut -c1-160)" >> $out/two_stream_timeline.txt
  python3 - "$t" >> $out/two_stream_timeline.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "clock_probe" not in r["Kernel_Name"]]
# the last step: from the last but one codebook normalisation (rownorm over 49152 rows is the first big launch of a step) on
marks = [i for i, r in enumerate(rows) if "wsq_max_regions" in r["Kernel_Name"]]
sel = rows[marks[-1] - 1:] if marks else rows[-60:]
t0 = int(sel[0]["Start_Timestamp"])
qs = {}
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 50.0:
        continue                                    # (only the launches that matter: >= 50 us)
    q = r.get("Queue_Id", "?"); qs.setdefault(q, len(qs))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    n = re.sub(r"^void ", "", r["Kernel_Name"])[:44]
    print(f"  {s / 1e3:8.2f} ms  +{d / 1e3:7.2f} ms  q{qs[q]}  {n}")
print(f"  step span {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e6:.2f} ms")
PY
  rm -rf $out/ts_$m
done
cat $out/two_stream_timeline.txt
