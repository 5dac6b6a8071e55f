"""Dev tool: filter kernel time only (results are garbage under the ablation macros) for the build in MEDTOK_TOOL_LIB."""
import sys, time
sys.path.insert(0, ".")
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):            # dev A/B: an alternative build of the same ABI
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
N, K, D = 600000, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 768
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
for rnd in range(2):
    for _ in range(2): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize(); ops.profile_begin()
    for _ in range(3): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    p = ops.profile_end()["filter_f16_kernel"]
    print(f"filter kernel {p['ms']/p['launches']:.2f} ms ({p['flops']/p['ms']/1e9:.0f} TF)", flush=True)
