"""Dev tool: exact fp32 search at small batch vs the cap on code splits."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
for n, K, D in ((256, 49152, 768), (256, 16384, 768), (1024, 49152, 768), (64, 49152, 768), (256, 21000, 64), (256, 7000, 64), (1024, 21000, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    ref = None
    for cap in (32, 64, 128, 256, 512):
        ops.debug_plan_override(search_max_splits=cap)
        for _ in range(3): i, d = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): i, d = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        if ref is None: ref = (i.clone(), d.clone())
        print(f"n={n} K={K} D={D} max splits {cap}: {dt*1e6:.0f} us ({2*n*K*D/dt/1e12:.1f} TF) same bits {torch.equal(i, ref[0]) and torch.equal(d, ref[1])}", flush=True)
