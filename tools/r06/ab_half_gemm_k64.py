"""Dev tool (GPU box): round 6's two changes to the dense products against the forms before them (medtok_debug_set_half_gemm_k32: bit 0 =
32-deep stages of the one-pass product, bit 1 = tiles to the XCDs by row tile whatever the row-tile count): bit equality of a few
products, their times, and the cfg 4 VQ-side training step, alternated in one process.
python tools/r06/ab_half_gemm_k64.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def k32(on): lib.medtok_debug_set_half_gemm_k32(int(on))
g = torch.Generator(device=dev).manual_seed(0)
for (m, n, k, groups) in ((131072, 768, 768, 1), (256, 768, 768, 1), (3072, 768, 832, 7), (5656, 3072, 768, 1), (5656, 768, 3072, 1), (768, 768, 4736, 28), (3072, 768, 832, 7), (300, 64, 64, 1), (1000, 256, 192, 4)):
    for dt in (torch.bfloat16, torch.float16):
        if groups == 1:
            a = torch.randn(m, k, device=dev, generator=g).to(dt); b = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(dt)
            call = lambda: ops.half_gemm(a, b, n_g=n, k_g=k)
        else:       # the weight-gradient form: A [m, groups * k] grouped along its columns, B [groups * n, k]
            a = torch.randn(m, groups * k, device=dev, generator=g).to(dt); b = (torch.randn(groups * n, k, device=dev, generator=g) * 0.05).to(dt)
            call = lambda: ops.half_gemm(a, b, n_g=n, k_g=k, groups=groups, a_group_cols=k, b_group_rows=n)
        res, us = {}, {}
        for mode in (3, 1, 0):
            k32(mode)
            for _ in range(3): c = call()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): c = call()
            torch.cuda.synchronize(); us[mode] = (time.perf_counter() - t0) / 20 * 1e6
            res[mode] = c.clone()
        print(f"m={m:6d} n={n:4d} k={k:4d} groups={groups:2d} {str(dt)[6:]:8s}: before {us[3]:7.1f} us, dense tile order {us[1]:7.1f} us, + 64-deep stages {us[0]:7.1f} us, "
              f"equal bits: {torch.equal(res[3], res[0]) and torch.equal(res[1], res[0])}", flush=True)
wl = bench.Cfg4(256, dev, seed=0, path=ops.PATH_AUTO, precomputed=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
for rnd in range(3):
    for mode, what in ((3, "both as before round 6"), (1, "dense tile order, 32-deep stages"), (2, "by row tile, 64-deep stages"), (0, "both (HEAD)")):
        k32(mode)
        wl.step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): wl.step()
        torch.cuda.synchronize()
        print(f"round {rnd}: cfg4 VQ-side step, {what:34s}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
k32(0)
