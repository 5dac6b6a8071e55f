"""Dev tool: filter-kernel time of several BUILDS of the library, alternated inside one process (same data, same thermal state):
    python tools/ab_libs.py K rounds lib1.so lib2.so ...        ('default' = the in-tree build)
Reports per build the median over rounds of the mean of 3 launches.  Timing only: ablation builds return garbage."""
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
xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
res = {l: [] for l in libs}
for r in range(rounds + 1):
    for l in libs:
        bind(l)
        ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
        torch.cuda.synchronize(); ops.profile_begin()
        for _ in range(3): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
        torch.cuda.synchronize()
        p = ops.profile_end()["filter_f16_kernel"]
        if r: res[l].append(p["ms"] / p["launches"])
for l in libs:
    v = res[l]
    print(f"K={K} {l}: median {statistics.median(v):.2f} ms  (min {min(v):.2f}, max {max(v):.2f})  {2.0*N*K*D/statistics.median(v)/1e9:.0f} TF", flush=True)
