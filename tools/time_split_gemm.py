"""Dev tool: the four dense products of a cross-attention layer at the `full` workload's shapes, each timed alone
(split-fp16 GEMM vs the library's fp32 GEMM), D = 768, 4 heads."""
import sys, time
sys.path.insert(0, ".")
import torch
from pathlib import Path
import medtok_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "tree":
    L._SO = Path(sys.argv[1]).resolve()          # another build of the library (A/B: one library per process)
from medtok_amd import ops
dev = torch.device("cuda:0")
D, H, hd = 768, 4, 192

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

def w(rows, cols):
    m = torch.randn(rows, cols, device=dev) / cols ** 0.5
    return ops.split_half(m, dp=cols, scale=2048.0), m

for M in ((84000,) if len(sys.argv) > 2 else (84000, 4096, 1024)):
    x = torch.randn(M, D, device=dev); xs = ops.split_half(x)
    (wq, wq32), (wk, _), (wv, _), (wo, wo32) = w(H * hd, D), w(H * D, hd), w(H * hd, D), w(D, H * hd)
    q = ops.split_gemm(xs, wq, n_g=H * hd, k_g=D, want_f32=False, want_split=True)[1]
    ctx = torch.randn(M, H * D, device=dev); cs = ops.split_half(ctx)
    att = ops.split_gemm(cs, wv, n_g=hd, k_g=D, groups=H, a_group_cols=D, b_group_rows=hd, want_f32=False, want_split=True)[1]
    fl = 2.0 * M * D * D
    r = {
        "G1 x.Wq^T (split out)": t(lambda: ops.split_gemm(xs, wq, n_g=H * hd, k_g=D, want_f32=False, want_split=True)),
        "G1 fp32 out": t(lambda: ops.split_gemm(xs, wq, n_g=H * hd, k_g=D)),
        "G2 fold per head (fp32 out [M, 4D])": t(lambda: ops.split_gemm(q, wk, n_g=D, k_g=hd, groups=H, a_group_cols=hd, b_group_rows=D)),
        "G3 Wv per head (split out)": t(lambda: ops.split_gemm(cs, wv, n_g=hd, k_g=D, groups=H, a_group_cols=D, b_group_rows=hd, want_f32=False, want_split=True)),
        "G4 out_proj (fp32 out)": t(lambda: ops.split_gemm(att, wo, n_g=D, k_g=H * hd)),
        "split_half x [M, D]": t(lambda: ops.split_half(x)),
        "split_half ctx [M, 4D]": t(lambda: ops.split_half(ctx)),
        "library fp32 x @ Wq^T": t(lambda: torch.mm(x, wq32.t())),
    }
    print(f"M = {M}: fp32-equivalent flops per product {fl/1e9:.1f} G")
    for k, v in r.items():
        print(f"   {k:40s} {v*1e3:8.1f} us   {fl / v / 1e9:7.1f} TF fp32-eq" if "split_half" not in k else f"   {k:40s} {v*1e3:8.1f} us")
