"""Dev tool: where does the fp16-filter path start to beat the exact fp32 path?"""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
def t(n, k, d, topk, path, iters=20):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    for _ in range(3): ops.topk_search(xh, xs, wh, ws, topk, path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.topk_search(xh, xs, wh, ws, topk, path)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for d, k in ((64, 21000), (768, 8192), (768, 49152), (64, 2048)):
    for n in (256, 512, 1024, 2048, 4096, 8192, 16384):
        a, b = t(n, k, d, 5, 1), t(n, k, d, 5, 2)
        print(f"D={d} K={k} N={n}: f32 {a:.3f} ms  filter {b:.3f} ms  -> {'filter' if b < a else 'f32'}", flush=True)
