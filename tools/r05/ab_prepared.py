"""Dev tool (GPU box): the full forward (bench.py --workload full) with and without the prepared codebook, alternated in one process.
usage: python tools/r05/ab_prepared.py [rows]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = bench.Full(rows, dev, seed=0, path=0)
for _ in range(3):
    wl.step()
for rnd in range(3):
    for on in (False, True):
        vqmod.PREPARED_CODEBOOK = on
        wl.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            wl.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"round {rnd} prepared={on!s:5s} {dt * 1e3:7.3f} ms per forward = {rows / dt / 1e3:7.1f} k codes/s", flush=True)
