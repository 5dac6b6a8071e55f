"""Dev tool (GPU box): the exact fp32 search at a few hundred rows against the cap on its code splits (plan bit search_max_splits;
the default cap is 64 from three row tiles up, min(256, CUs) / row tiles below).  ms per topk_search call, k = 5."""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")


def t(n, k, d, path, iters=30):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    for _ in range(3): ops.topk_search(xh, xs, wh, ws, 5, path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.topk_search(xh, xs, wh, ws, 5, path)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for d, k in ((768, 49152), (768, 16384), (64, 21000), (256, 49152)):
    for n in (256, 384, 512, 768, 1024, 2048):
        line = f"D={d} K={k} N={n}: default {t(n, k, d, ops.PATH_F32_MFMA):.3f}"
        for cap in (32, 64, 96, 128, 192, 255):
            line += f" | cap {cap}: {t(n, k, d, ops.plan_path(ops.PATH_F32_MFMA, search_max_splits=cap)):.3f}"
        line += f" | filter {t(n, k, d, ops.PATH_F16_FILTER):.3f}"
        print(line, flush=True)
