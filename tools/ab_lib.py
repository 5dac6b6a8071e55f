"""Dev tool: the full forward (or cfg3: argv[2]) on the in-tree library or on another build of it (argv[1] = path of a .so, or
'tree'), one library per process; alternate from a shell loop on one box."""
import sys, time
sys.path.insert(0, ".")
import torch
from pathlib import Path
import medtok_amd._lib as L
if sys.argv[1] != "tree":
    L._SO = Path(sys.argv[1]).resolve()
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
what = sys.argv[2] if len(sys.argv) > 2 else "full"
w = bench.Full(4096, dev, 0, ops.PATH_AUTO) if what == "full" else bench.Cfg3(600000, dev, 0, ops.PATH_AUTO)
n, steps = (4096, 10) if what == "full" else (600000, 4)
def run():
    for _ in range(2): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): w.step()
    torch.cuda.synchronize(); return n * steps / (time.perf_counter() - t0)
print(f"{sys.argv[1][-12:]:12s} {what}", "  ".join(f"{run()/1e3:8.1f}k" for _ in range(3)), flush=True)
ops.profile_begin()
for _ in range(steps): w.step()
torch.cuda.synchronize()
prof = ops.profile_end()
print("   kernel ms per launch:", {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in prof.items() if v["launches"]}, flush=True)
if what != "full":
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    from medtok_amd.inference import quantize_pooled
    x = w.pooled_text
    w.vq._search(x, "shared", False); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(5): w.vq._search(x, "shared", False)
    ev[1].record(); torch.cuda.synchronize()
    print(f"   one shared search (rownorm + filter + rescore): {ev[0].elapsed_time(ev[1]) / 5:.3f} ms", flush=True)
