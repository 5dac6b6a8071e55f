"""Dev tool: filter_f16_kernel time vs code splits / XCD-aware block order (plan forced through ops.debug_plan_override)."""
import os, sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
N, K, D = int(sys.argv[1]) if len(sys.argv) > 1 else 600000, int(sys.argv[2]) if len(sys.argv) > 2 else 49152, 768
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
ref = None
for splits, xcd in ((0, 0), (2, 1), (4, 0), (4, 1), (8, 0), (8, 1), (16, 1)):
    ops.debug_plan_override(filter_splits=splits or -1, filter_xcd=xcd)
    for _ in range(2): idx, dist = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    ops.profile_begin()
    import time; t0 = time.perf_counter()
    for _ in range(3): idx, dist = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    p = ops.profile_end()["filter_f16_kernel"]
    if ref is None: ref = (idx.clone(), dist.clone())
    same = torch.equal(idx, ref[0]) and torch.equal(dist, ref[1])
    print(f"splits={splits or 'auto'} xcd={xcd}: filter kernel {p['ms']/p['launches']:.2f} ms ({p['flops']/p['ms']/1e9:.0f} TF), whole search {dt*1e3:.2f} ms, same bits: {same}", flush=True)
