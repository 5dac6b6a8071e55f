"""Dev tool (GPU box): per-step wall times of the cfg 4 VQ-side training step (synchronised after every step), with and without the
Python garbage collector -- where the occasional 17-24 ms step of the A/B loops comes from.   python tools/r06/cfg4_step_distribution.py"""
import gc, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
wl = bench.Cfg4(256, dev, seed=0, path=ops.PATH_AUTO, precomputed=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
for mode in ("gc on", "gc off", "gc on", "gc off"):
    (gc.enable if mode == "gc on" else gc.disable)()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts2 = sorted(ts)
    print(f"{mode}: median {ts2[30]:.2f} ms, p90 {ts2[54]:.2f}, max {ts2[-1]:.2f}; steps over 13 ms: {[round(t, 1) for t in ts if t > 13]}; "
          f"allocator: {torch.cuda.memory_stats(dev)['num_alloc_retries']} retries, {torch.cuda.memory_reserved(dev) >> 20} MiB reserved", flush=True)
gc.enable()
