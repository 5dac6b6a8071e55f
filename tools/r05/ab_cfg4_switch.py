"""Dev tool (GPU box): the VQ-side training step (cfg 4, precomputed encoders) with a module switch of vector_quantization_soft_one_new
on / off, alternated in one process.   python tools/r05/ab_cfg4_switch.py SWITCH_NAME [rounds]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
name = sys.argv[1]; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
for _ in range(4): wl.step()
torch.cuda.synchronize()
for rnd in range(rounds):
    for on in (False, True):
        setattr(vqmod, name, on)
        wl.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): wl.step()
        torch.cuda.synchronize()
        print(f"round {rnd} {name}={on!s:5s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per step", flush=True)
