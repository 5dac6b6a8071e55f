"""Dev tool (GPU box): the D <= 64 filter kernel (filter_rows64_kernel) against the general one on the same search -- ids and
distances must be the same bits; times alternated in one process.  usage: python tools/r04/ab_rows64.py [N] [K] [D] [k]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from medtok_amd import _lib, ops
import os
if os.environ.get('DBGLIB'):           # (a variant build of the library, for A/B runs of kernel changes)
    _lib.use_library(os.environ['DBGLIB'])

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 21_000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g)
W = torch.randn(K, D, device=dev, generator=g)
xh, xs = ops.rownorm(x)
wh, ws = ops.rownorm(W)
paths = {"general": _lib.plan_path(ops.PATH_F16_FILTER, filter_rows64=False), "rows64": _lib.plan_path(ops.PATH_F16_FILTER, filter_rows64=True),
         "rows64wide": _lib.plan_path(ops.PATH_F16_FILTER, filter_rows64="wide"), "exact": ops.PATH_F32_MFMA}
out = {}
for name, p in paths.items():
    out[name] = ops.topk_search(xh, xs, wh, ws, k, p)
torch.cuda.synchronize()
for name in ("rows64", "rows64wide", "exact"):
    print(f"{name} vs general: ids equal {torch.equal(out[name][0], out['general'][0])}  distances equal {torch.equal(out[name][1], out['general'][1])}")
for rnd in range(3):
    for name in ("general", "rows64wide", "rows64"):
        ops.profile_begin()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ops.topk_search(xh, xs, wh, ws, k, paths[name])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        pr = ops.profile_end()["filter_f16_kernel"]
        fms = pr["ms"] / max(pr["launches"], 1)
        print(f"round {rnd} {name:10s} {dt * 1e3:8.3f} ms per search   filter kernel {fms:7.3f} ms = {pr['flops'] / max(pr['launches'], 1) / fms / 1e9:7.1f} TFLOP/s "
              f"= {pr['flops'] / max(pr['launches'], 1) / fms / 1e9 / 2500:.3f} of the f16 peak")
