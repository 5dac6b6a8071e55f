"""Dev tool: fp16-filter path at small / medium batch vs the number of code splits (ops.plan_path: per-call plan hook), against the exact path."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = 768
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for n, K in ((1024, 49152), (2048, 49152), (4096, 49152), (1024, 8192), (2048, 8192), (4096, 16384)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    line = f"n={n} K={K}: exact {t(lambda: ops.topk_search(xh, xs, wh, ws, 5, 1)):.0f} us | filter auto {t(lambda: ops.topk_search(xh, xs, wh, ws, 5, 2)):.0f}"
    for sp in (16, 32, 64, 128):
        line += f" | S={sp}: {t(lambda: ops.topk_search(xh, xs, wh, ws, 5, ops.plan_path(2, filter_splits=sp))):.0f}"
    print(line, flush=True)
