"""Dev tool: candidate-list capacity (MEDTOK_FILTER_CAP builds via MEDTOK_TOOL_LIB): rows handed to the exact kernel and search time,
random and clustered inputs."""
import sys, time
sys.path.insert(0, ".")
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = 768
g = torch.Generator(device=dev).manual_seed(0)
for N, K, kind in ((600000, 16384, "random"), (600000, 49152, "random"), (300000, 16384, "clustered"), (300000, 16384, "near-codes")):
    W = torch.randn(K, D, device=dev, generator=g)
    if kind == "clustered":      # codes in 64 tight clusters: many near-ties per row
        W = torch.randn(64, D, device=dev, generator=g)[torch.randint(0, 64, (K,), device=dev, generator=g)] + 0.05 * W
    x = torch.randn(N, D, device=dev, generator=g)
    if kind == "near-codes":
        x = W[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.02 * x
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    ops.SEARCH_STATS = {}
    ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    st = dict(ops.SEARCH_STATS); ops.SEARCH_STATS = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{kind:11s} N={N} K={K}: fallback rows {st.get('fallback_rows')} of {N}, search {dt*1e3:.2f} ms", flush=True)
    del x, W, xh, wh
