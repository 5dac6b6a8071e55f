"""Dev tool (GPU box): timing-dependent faults only show in repetition at full occupancy -- large filter searches (both D <= 64 kernels, the
general one) repeated REPS times each against ONE exact result.   python tools/r05/repeat_large_searches.py [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = total = 0
for seed, (n, K, D, k) in enumerate([(600000, 21000, 64, 5), (600000, 7000, 64, 5), (300000, 21000, 60, 8), (600000, 21000, 32, 1), (200000, 20001, 768, 5),
                                     (4097, 20001, 32, 5), (8192, 49152, 768, 5)]):
    g = torch.Generator(device=dev).manual_seed(seed + 50)
    x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
    for rep in range(reps):
        for kind in (True, "wide") if D <= 64 else (None,):
            path = ops.plan_path(ops.PATH_F16_FILTER, filter_rows64=kind) if kind is not None else ops.PATH_F16_FILTER
            i1, d1 = ops.topk_search(xh, xs, wh, ws, k, path)
            total += 1
            if not (torch.equal(i0, i1) and torch.equal(d0, d1)):
                bad += 1; print("MISMATCH", n, K, D, k, kind, rep, int((i0 != i1).any(1).sum()), flush=True)
    print("shape", n, K, D, k, "done", flush=True)
print("searches:", total, "mismatches:", bad)
sys.exit(1 if bad else 0)
