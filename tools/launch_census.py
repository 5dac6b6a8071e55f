"""Dev tool: launches per forward by origin, from a rocprofv3 --kernel-trace CSV of bench.py --workload full / fullref -- a forward
starts at pack_mask_len_kernel (the first launch of CrossAttention.pooled; the two modality-specific searches in front of it are
attributed to the following forward: same count).  usage: python tools/launch_census.py kernel_trace.csv"""
import csv, collections, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "pack_mask_len_kernel" in n]
if len(marks) < 3:
    sys.exit("fewer than three forwards in the trace")
a, b = marks[-3], marks[-1]                     # two whole forwards
cls = collections.Counter()
detail = collections.Counter()
for n in names[a:b]:
    if "at::native" in n or "rocprim" in n or "at::cuda" in n:
        k = "torch (at::native / rocprim)"
    elif "rocclr" in n:
        k = "runtime copy / fill (hipMemcpyAsync, hipMemsetAsync)"
    elif n.startswith("Cijk"):
        k = "hipBLASLt GEMM (Cijk_*)"
    else:
        k = "this library"
    cls[k] += 1
    if k != "this library":
        detail[n[:110]] += 1
print(f"launches per forward (mean of 2): {(b - a) / 2:.1f}")
for k, v in cls.most_common():
    print(f"  {v / 2:6.1f}  {k}")
for k, v in detail.most_common(12):
    print(f"      {v / 2:5.1f}  {k}")
