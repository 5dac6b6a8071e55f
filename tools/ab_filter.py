"""Dev tool: filter kernel time at the cfg-3 shapes, same bits as the exact path.
MEDTOK_TOOL_LIB=<path> binds an alternative build of the same ABI."""
import sys, time
sys.path.insert(0, ".")
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):            # dev A/B: an alternative build of the same ABI
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = 768
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(600000, 16384), (600000, 49152)] if len(sys.argv) < 2 else [(int(sys.argv[1]), int(sys.argv[2]))]
for N, K in shapes:
    x = torch.randn(N, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    ir, dr = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
    for rnd in range(2):
        ops.SEARCH_STATS = {}
        for _ in range(2): idx, dist = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
        fb = ops.SEARCH_STATS.get("fallback_rows"); ops.SEARCH_STATS = None
        torch.cuda.synchronize(); ops.profile_begin(); t0 = time.perf_counter()
        for _ in range(3): idx, dist = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        p = ops.profile_end()["filter_f16_kernel"]
        print(f"N={N} K={K}: filter kernel {p['ms']/p['launches']:.2f} ms ({p['flops']/p['ms']/1e9:.0f} TF), whole search {dt*1e3:.2f} ms, "
              f"fallback rows {fb}, same bits as exact: {torch.equal(idx, ir) and torch.equal(dist, dr)}", flush=True)
    del x, W, xh, wh
