"""Dev tool: time the search kernel alone at a few shapes (GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops

def run(n, k, d, topk, path=0, iters=3):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(n, d, generator=g).to(dev); W = torch.randn(k, d, generator=g).to(dev)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    ops.topk_search(xh, xs, wh, ws, topk, path); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.topk_search(xh, xs, wh, ws, topk, path)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"n={n} K={k} d={d} topk={topk} path={path}: {ms:.3f} ms  {2*n*k*d/ms/1e9:.1f} TFLOP/s  {n/ms*1e3:.0f} rows/s", flush=True)

if __name__ == "__main__":
    run(100000, 8192, 768, 1)
    run(100000, 8192, 768, 5)
    run(600000, 16384, 768, 5, iters=1)
    run(65536, 16384, 768, 5)
    run(4096, 8192, 768, 5)
    run(256, 21000, 64, 5, iters=20)
