"""Dev tool: whole one-call forward (rownorm + filter + re-score with fused assignment) and its filter kernel for several BUILDS,
alternated in one process:  python tools/ab_rescore.py K rounds lib1.so lib2.so ...   ('default' = in-tree build).
forward - filter ~ re-score + rownorm (0.7 ms) + small kernels.  Timing only under ablation macros."""
import sys, statistics
sys.path.insert(0, ".")
import torch
from medtok_amd import _lib, ops
dev = torch.device("cuda:0")
K, rounds, libs = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
N, D = 600000, 768
default = str(_lib.library_path())
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
def bind(path):
    _lib._lib = None; _lib._SO = type(_lib._SO)(default if path == "default" else path); _lib.load()
bind(libs[0])
wh, ws = ops.rownorm(W)
res = {l: [] for l in libs}
for r in range(rounds + 1):
    for l in libs:
        bind(l)
        ops.soft_vq_forward(x, wh, ws, 5, ops.PATH_F16_FILTER, want_sqerr=False)
        torch.cuda.synchronize(); ops.profile_begin()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.soft_vq_forward(x, wh, ws, 5, ops.PATH_F16_FILTER, want_sqerr=False)
        e1.record(); torch.cuda.synchronize()
        p = ops.profile_end()["filter_f16_kernel"]
        if r: res[l].append((e0.elapsed_time(e1) / 3, p["ms"] / p["launches"]))
for l in libs:
    f = statistics.median(v[0] for v in res[l]); k = statistics.median(v[1] for v in res[l])
    print(f"K={K} {l}: forward {f:.2f} ms, filter kernel {k:.2f} ms, rest {f - k:.2f} ms", flush=True)
