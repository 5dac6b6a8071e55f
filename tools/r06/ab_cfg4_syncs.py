"""Dev tool (GPU box): what the two host reads of a training step cost at cfg 4 (VQ side alone, precomputed encoder outputs): the usage
counts at the end of VectorQuantizer.forward (show_usage) and the node-count read of CrossAttention.pooled.  Alternated in one
process.   python tools/r06/ab_cfg4_syncs.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=ops.PATH_AUTO, precomputed=True)
def timed(steps=10):
    wl.step(); wl.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): wl.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rnd in range(3):
    for usage in (True, False):
        wl.model.quantize.show_usage = usage
        print(f"round {rnd}: show_usage {usage!s:5}: {timed():.2f} ms/step", flush=True)
