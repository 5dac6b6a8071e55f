"""Dev tool (GPU box): the filter's measured score error against the bound's budget terms (the quantities
tests/test_gpu_filter.py::test_filter_score_error_bound asserts on), printed.   python tools/r05/measure_filter_error.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT)]
import numpy as np, torch
from medtok_amd import ops
from oracle import oracle
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rng = np.random.default_rng(0)
for d in (64, 256, 768, 1000):
    x = rng.standard_normal((512, d), dtype=np.float32); W = rng.standard_normal((1024, d), dtype=np.float32)
    x[:8] = np.abs(x[:8]); W[:8] = np.abs(W[:8]); x[8] = 1.0; W[8] = 1.0
    xh, xs = oracle.rownorm(x); wh, ws = oracle.rownorm(W)
    s_apx = ops.debug_filter_scores(T(xh), T(xs), T(wh), T(ws)).cpu().numpy()
    xr = (xh * 256).astype(np.float16).astype(np.float64) / 256
    wr = (wh * 256).astype(np.float16).astype(np.float64) / 256
    err = np.abs(s_apx - xr @ wr.T)
    model = d * 2.0 ** -24 * (np.abs(xr) @ np.abs(wr).T) + (d + 64) * 2.0 ** -25 * ws[None, :]       # one ulp (round to nearest) per addition
    print(f"d={d:5d}  max accumulation error / one-ulp-per-addition model = {(err / model).max():.3f}   (budget = 4 x the model)")
