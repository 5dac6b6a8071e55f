"""Dev tool: bulk search at the reference's default width (e_dim = 64, n_e = 21000 / regions of 7000), both paths."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for n, K in ((600000, 7000), (600000, 21000), (100000, 21000)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    wh, ws = ops.rownorm(W)
    outs = {}
    for name, path in (("auto", ops.PATH_AUTO), ("exact", ops.PATH_F32_MFMA), ("filter", ops.PATH_F16_FILTER)):
        for _ in range(2): r = ops.soft_vq_forward(x, wh, ws, 5, path, want_sqerr=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = ops.soft_vq_forward(x, wh, ws, 5, path, want_sqerr=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        outs[name] = r
        print(f"n={n} K={K} D={D} {name}: {dt*1e3:.2f} ms  {2.0*n*K*D/dt/1e12:.1f} TF  {n/dt/1e6:.2f} M rows/s", flush=True)
    print("  same bits:", torch.equal(outs["exact"]["idx"], outs["filter"]["idx"]) and torch.equal(outs["exact"]["dist"], outs["filter"]["dist"]), flush=True)
