"""Dev tool: per-dispatch timeline of one bench step from a rocprofv3 --kernel-trace CSV (start offset, duration, queue)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else -1
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:70]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the first kernel of pooled(): find repeating marker = split_half_kernel with the largest grid? simpler: take the last 1/3 of dispatches
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "segment_mean" in n]
if len(marks) >= 2:
    a, b = marks[-2] + 1, marks[-1] + 1
    # a step spans from just after the previous segment_mean .. ; shift to include the rest of the step after segment_mean
    sel = rows[a:b]
else:
    sel = rows[-200:]
t0 = int(sel[0]["Start_Timestamp"])
qs = {}
for r in sel:
    q = r.get("Queue_Id", "?"); qs.setdefault(q, len(qs))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{qs[q]}  {short(r['Kernel_Name'])}")
print("span us", (int(sel[-1]["End_Timestamp"]) - t0) / 1e3, "dispatches", len(sel))
