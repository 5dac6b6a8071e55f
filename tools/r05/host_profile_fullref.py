"""Dev tool (GPU box): where the HOST time of the eager B = 256 forward at the reference's shape goes (cProfile over 300 forwards, the GPU
queue drained every 20 so that back-pressure does not show up as host time).   python tools/r05/host_profile_fullref.py [sort]"""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
dev = torch.device("cuda:0")
wl = bench.FullRefDefault(256, dev, seed=0, path=0)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300):
    wl.step()
torch.cuda.synchronize()
print(f"eager forward: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
pr = cProfile.Profile()
for blk in range(15):
    pr.enable()
    for i in range(20):
        wl.step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(45)
