import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch, bench
import medtok_amd.vector_quantization_soft_one_new as vqmod
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
ca = wl.model.quantize.cross_attn
mx = int(torch.bincount(wl.inputs.batch).max())
print("max nodes", mx)
for _ in range(5): wl.step()
torch.cuda.synchronize()
for rnd in range(3):
    for on in (False, True):
        vqmod._EXPERIMENT_TRAIN_BOUND = on
        ca.max_nodes_bound = mx if on else None
        for usage in (True, False):
            wl.model.quantize.show_usage = usage
            wl.step(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): wl.step()
            torch.cuda.synchronize()
            print(f"round {rnd} bound={on!s:5s} show_usage={usage!s:5s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per step", flush=True)
