"""Dev tool: print the kernel sequence (start offset, duration, name) of the last N launches of a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-int(sys.argv[2]):]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  %8.1f us  %s" % ((a - t0) / 1e3, (b - a) / 1e3, r["Kernel_Name"][:80]))
