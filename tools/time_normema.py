"""Dev tool: NormEMAVectorQuantizer train step (no autograd) with the one-call head on / off, alternated in one process."""
import sys, statistics
sys.path.insert(0, ".")
import torch
from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
dev = torch.device("cuda:0")
n, k, d = 100000, 8192, 768
q = NormEMAVectorQuantizer(k, d, 0.25).to(dev).train()
z = torch.randn(n, d, 1, 1, device=dev)
res = {True: [], False: []}
with torch.no_grad():
    for r in range(6):
        for fused in (True, False):
            q.fused_head = fused
            q(z); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): q(z)
            e1.record(); torch.cuda.synchronize()
            if r: res[fused].append(e0.elapsed_time(e1) / 5)
for f in (True, False): print(f"fused_head={f}: median {statistics.median(res[f]):.3f} ms  (min {min(res[f]):.3f})", flush=True)
