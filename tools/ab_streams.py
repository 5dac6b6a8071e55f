"""Dev tool: does re-creating the side streams change the forward's speed?  (HIP maps streams onto a few hardware queues.)"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from medtok_amd import ops
import medtok_amd.vector_quantization_soft_one_new as vq
dev = torch.device("cuda:0")
w = bench.Full(4096, dev, 0, ops.PATH_AUTO)
def run(steps=10):
    for _ in range(2): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): w.step()
    torch.cuda.synchronize(); return 4096 * steps / (time.perf_counter() - t0)
print("same streams, 6 runs:      ", "  ".join(f"{run()/1e3:6.1f}k" for _ in range(6)), flush=True)
out = []
for i in range(8):
    vq._side_streams.clear()
    out.append(run())
print("streams re-created each run:", "  ".join(f"{x/1e3:6.1f}k" for x in out), flush=True)
print("same streams again:         ", "  ".join(f"{run()/1e3:6.1f}k" for _ in range(3)), flush=True)
vq.STREAM_PRIORITY = (0, 0, 0)
out = []
for i in range(6):
    vq._side_streams.clear()
    out.append(run())
print("re-created, equal priority: ", "  ".join(f"{x/1e3:6.1f}k" for x in out), flush=True)
