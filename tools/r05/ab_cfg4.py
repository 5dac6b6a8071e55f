"""Dev tool (GPU box): the full cfg 4 train step (stand-in encoders + VQ side) with this round's training switches on and off, alternated
in one process on one box.   python tools/r05/ab_cfg4.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
from medtok_amd import ops
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=ops.PATH_AUTO)
def timed(steps=5):
    wl.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): wl.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rnd in range(3):
    for half, merge in ((True, True), (False, False), (True, False), (False, True)):
        vqmod.AUTOCAST_HALF_PRODUCTS, vqmod.MERGE_SIDES_IN_TRAINING = half, merge
        print(f"round {rnd}: half products {half!s:5} merged sides {merge!s:5}: {timed():.2f} ms/step", flush=True)
