import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch, bench
from medtok_amd import loss as L
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
for rnd in range(3):
    for on in (False, True):
        L.FUSED_LOSS_ASSEMBLY = on
        wl.step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): wl.step()
        torch.cuda.synchronize()
        print(f"round {rnd} FUSED_LOSS_ASSEMBLY={on!s:5s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per step", flush=True)
