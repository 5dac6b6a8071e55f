"""Dev tool: randomized shapes for the filter kernel of rows of <= 64 elements (filter_rows64_kernel: learning tiles, scanned tiles,
revisited tiles; code splits; ragged last tiles; every k-list length) against the exact fp32 path and the general filter kernel:
ids and distances must be the same bits.    python tools/r04/fuzz_rows64.py [cases] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from medtok_amd import ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, t0 = 0, time.time()
for c in range(cases):
    n = int(rng.choice([1, 127, 128, 129, 513, 4097, 20000, 70001, 200000][: 8 if c % 8 else 9]))
    tiles = int(rng.choice([1, 2, 3, 7, 8, 9, 12, 16, 17, 33, 40, 83]))
    K = max(1, 256 * tiles + int(rng.choice([-255, -100, -1, 0, 0, 0])))
    D = int(rng.choice([36, 40, 60, 64, 64, 64]))
    k = int(rng.choice([1, 2, 5, 5, 8]))
    if k > K: k = 1
    kind = int(rng.integers(0, 5))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    if kind == 1: W[K // 2:] = W[: K - K // 2].clone()                                   # duplicated codes (ties)
    if kind == 2: x = x * 0.01 + W[torch.randint(0, K, (n,), device=dev, generator=g)]   # rows close to codes
    if kind == 3 and K >= 10:                                                            # near-copies of neighbours
        m = min(W[::5].shape[0], W[1::5].shape[0])
        W[::5][:m] = W[1::5][:m] + 1e-3 * torch.randn(m, D, device=dev, generator=g)
    if kind == 4: x[::7] = 0                                                             # zero rows
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
    for env in (dict(filter_rows64=True), dict(filter_rows64=True, filter_splits=int(rng.choice([1, 2, 4, 8]))), dict(filter_rows64=False)):
        i1, d1 = ops.topk_search(xh, xs, wh, ws, k, ops.plan_path(ops.PATH_F16_FILTER, **env))
        if not (torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))):
            bad += 1
            print(f"MISMATCH case {c}: n={n} K={K} D={D} k={k} kind={kind} env={env} rows differing {(i0 != i1).any(1).sum().item()}", flush=True)
print(f"{cases} cases x 3 plans, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
