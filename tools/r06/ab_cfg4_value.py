"""Dev tool (GPU box): the VQ-side training step (cfg 4, precomputed encoders) with a module constant of vector_quantization_soft_one_new set
to each of several values, alternated in one process.   python tools/r06/ab_cfg4_value.py NAME v1 v2 [v3 ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
name, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
for rnd in range(3):
    for v in values:
        setattr(vqmod, name, v)
        wl.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): wl.step()
        torch.cuda.synchronize()
        print(f"round {rnd} {name}={v:6d} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per step", flush=True)
