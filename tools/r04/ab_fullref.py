"""Dev tool: the B = 256 forward at the reference's default shape (bench.FullRefDefault) under one setting per process."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from medtok_amd import ops
import medtok_amd.vector_quantization_soft_one_new as vq
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    setattr(vq, k, type(getattr(vq, k))(float(v)) if not isinstance(getattr(vq, k), bool) else bool(int(v)))
dev = torch.device("cuda:0")
w = bench.FullRefDefault(256, dev, 0, ops.PATH_AUTO)
def run(steps=50):
    for _ in range(5): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): w.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
print(" ".join(sys.argv[1:]) or "shipped", "  ".join(f"{run():.3f} ms" for _ in range(3)), flush=True)
