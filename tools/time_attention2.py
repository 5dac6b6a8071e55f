"""Dev tool: per-chunk latency / throughput regimes of ops.shared_kv_attention (uniform shapes)."""
import sys, time
sys.path.insert(0, ".")
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 768, 4
EXACT = bool(int(_os.environ.get("ATT_EXACT", "0")))
def run(B, rows, T, reps=20):
    q = torch.randn(B * rows, D, device=dev) * 0.05
    kv = torch.randn(B * T, D, device=dev)
    code = torch.arange(B, device=dev)
    a = (q, code * rows, torch.full((B,), rows, device=dev), kv, code * T, torch.full((B,), T, device=dev), rows, 192 ** -0.5, EXACT)
    for _ in range(3): ops.shared_kv_attention(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.shared_kv_attention(*a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fl = B * rows * T * D * 4.0
    tiles = B * ((rows + 31) // 32); chunks = (T + 31) // 32
    print(f"B={B:5d} rows={rows:4d} T={T:4d}: {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF  | blocks {tiles:6d} x {chunks:3d} chunks -> {dt*1e6/chunks/max(1, tiles/256):.2f} us per chunk-round", flush=True)
shapes = [(256, 32, 512), (256, 32, 32), (256, 32, 64), (256, 32, 128), (512, 32, 512), (2048, 32, 512), (256, 160, 512), (2048, 96, 256)]
if len(sys.argv) > 2: shapes = [(2048, 96, 256), (2048, 32, 512)]
for sh in shapes: run(*sh)
