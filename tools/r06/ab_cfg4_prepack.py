import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch, bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
ca = wl.model.quantize.cross_attn
mx = int(torch.bincount(wl.inputs.batch).max())
for _ in range(6): wl.step()
torch.cuda.synchronize()
def t(n=10):
    wl.step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(4):
    for name, pre, bound in (("plain", False, None), ("prepack", True, None), ("bound", False, mx)):
        vqmod.PREPACK_CODES = pre; ca.max_nodes_bound = bound
        print(f"round {rnd} {name:8s} {t():7.3f} ms", flush=True)
ca.max_nodes_bound = None
