"""Dev tool: randomized shapes / distributions, forced fp16-filter path vs the exact fp32 path: ids and distances must be
bit-identical.  usage: python tools/fuzz_search.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from medtok_amd import ops, _lib
if os.environ.get('DBGLIB'):           # (a variant build of the library)
    _lib.use_library(os.environ['DBGLIB'])
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rng.choice([1, 7, 255, 256, 257, 1000, 4097, 20000, 70001, 300000][: 9 if c % 10 else 10]))
    K = int(rng.choice([1, 5, 31, 256, 257, 1000, 4096, 8191, 16384, 20001]))
    D = int(rng.choice([4, 32, 60, 64, 100, 128, 260, 768, 1024]))
    k = int(rng.choice([1, 2, 5, 8]))
    if k > K: k = 1
    kind = rng.integers(0, 5)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    if kind == 1: W[K // 2:] = W[: K - K // 2].clone()                       # duplicated codes (ties)
    if kind == 2: x = x * 0.01 + W[torch.randint(0, K, (n,), device=dev, generator=g)]   # rows close to codes
    if kind == 3: W = W * torch.rand(K, 1, device=dev, generator=g) * 3                  # un-normalised codes
    if kind == 4: x[:: 7] = 0                                         # zero rows
    norm = kind != 3
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W, normalize=norm) if not norm else ops.rownorm(W)
    if not norm: wh = W.contiguous()
    for env in ({}, dict(filter_splits=int(rng.choice([1, 2, 4, 8])), filter_xcd=True)):
        i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
        i1, d1 = ops.topk_search(xh, xs, wh, ws, k, ops.plan_path(ops.PATH_F16_FILTER, **env))
        ok = torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
        if not ok:
            bad += 1
            print(f"MISMATCH case {c}: n={n} K={K} D={D} k={k} kind={kind} env={env} rows differing {(i0 != i1).any(1).sum().item()}", flush=True)
print(f"{cases} cases x 2 plans, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
