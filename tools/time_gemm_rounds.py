"""Dev tool: out_proj-shaped split product (K = N = 768) at several row counts; + the HBM write rate of a plain fill."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = 768
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
w = torch.randn(D, D, device=dev) / D ** 0.5
ws = ops.split_half(w, dp=D, scale=2048.0)
bias = torch.randn(D, device=dev)
for M in (16384, 21000 * 3, 65536, 84000, 131072, 262144):
    x = torch.randn(M, D, device=dev); xs = ops.split_half(x)
    a = t(lambda: ops.split_gemm(xs, ws, n_g=D, k_g=D, bias=bias, unscale=1 / 2048.0))
    b = t(lambda: ops.split_gemm(xs, ws, n_g=D, k_g=D, bias=bias, unscale=1 / 2048.0, want_f32=False, want_split=True))
    tiles = (M + 255) // 256 * 3
    print(f"M={M:7d} tiles={tiles:5d} rounds={tiles/256:5.2f}  fp32 out {a:7.1f} us ({2*M*D*D/a/1e6:6.1f} TF)   split out {b:7.1f} us ({2*M*D*D/b/1e6:6.1f} TF)")
buf = torch.empty(84000 * 768, device=dev)
f = t(lambda: buf.fill_(1.0))
print(f"fill of {buf.numel()*4/1e6:.0f} MB: {f:.1f} us = {buf.numel()*4/f/1e6:.2f} TB/s")
src = torch.randn_like(buf)
c = t(lambda: buf.copy_(src))
print(f"copy of {buf.numel()*4/1e6:.0f} MB: {c:.1f} us = {2*buf.numel()*4/c/1e6:.2f} TB/s (read + write)")
