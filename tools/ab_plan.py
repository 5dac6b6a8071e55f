"""Dev tool: filter-kernel time of one build under several launch plans (ops.debug_plan_override):
    python tools/ab_plan.py K lib.so  -> splits 1/2/4/8, XCD-aware order on/off"""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import _lib, ops
K, lib = int(sys.argv[1]), sys.argv[2]
if lib != "default": _lib.use_library(lib)
dev = torch.device("cuda:0"); N, D = 600000, 768
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
for splits, xcd in ((2, 1), (1, 0), (2, 0), (4, 1), (8, 1), (4, 0)):
    ops.debug_plan_override(filter_splits=splits, filter_xcd=xcd)
    ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize(); ops.profile_begin()
    for _ in range(3): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    p = ops.profile_end()["filter_f16_kernel"]
    print(f"K={K} {lib} splits={splits} xcd={xcd}: {p['ms']/p['launches']:.2f} ms", flush=True)
ops.debug_plan_override()
