"""Dev tool (GPU box): what the library's event bracketing costs a cfg 4 training step (VQ side alone): none / the dense products only / every
kind, alternated in one process.   python tools/r06/ab_cfg4_profile_cost.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=ops.PATH_AUTO, precomputed=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
def timed(steps=20):
    t0 = time.perf_counter()
    for _ in range(steps): wl.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rnd in range(3):
    for mode in ("none", "split_gemm only", "all kinds"):
        if mode == "split_gemm only": ops.profile_begin(kinds=["split_gemm_kernel"])
        elif mode == "all kinds": ops.profile_begin()
        ms = timed()
        if mode != "none": ops.profile_end()
        print(f"round {rnd}: events around {mode:16s}: {ms:.2f} ms/step", flush=True)
