"""Dev tool: per-unit time of the 64-row DMA attention kernel on uniform shapes (unit = 64 query rows x 16 keys of one block)."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
def run(B, rows, T, shared_keys=False, reps=20, variant=0):
    q = torch.randn(B * rows, D, device=dev) * 0.05
    kv = torch.randn((1 if shared_keys else B) * T, D, device=dev)
    img = ops.split_half(kv)
    code = torch.arange(B, device=dev)
    ks = code * 0 if shared_keys else code * T
    a = (q, code * rows, torch.full((B,), rows, device=dev), img, ks, torch.full((B,), T, device=dev), rows, 192 ** -0.5, False, variant)
    for _ in range(3): ops.shared_kv_attention_split(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.shared_kv_attention_split(*a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fl = B * rows * T * D * 4.0
    tiles = B * ((rows + 63) // 64); chunks = (T + 15) // 16
    print(f"v{variant} B={B:5d} rows={rows:4d} T={T:4d} shared_keys={int(shared_keys)}: {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF  | {tiles:6d} blocks x {chunks:3d} chunks -> {dt*1e6/chunks/max(1, tiles/256):.2f} us per unit-round", flush=True)
for sh in [(256, 64, 512, True), (256, 64, 512, False), (2048, 64, 512, True), (2048, 64, 512, False), (2048, 128, 256, False), (2048, 128, 256, True), (4096, 80, 256, False)]:
    run(*sh, variant=0); run(*sh, variant=1)
