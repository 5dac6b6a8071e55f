"""Dev tool (GPU box): the dense tile order of the dense products (round 6) on and off (medtok_debug_set_half_gemm_k32 bit 1) for the inference
forwards at the reference's batch -- fullref (e_dim = 64) and full --rows 256 (D = 768) -- alternated in one process.
python tools/r06/ab_dense_order_inference.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
for name, wl in (("fullref", bench.FullRefDefault(256, dev, 0, ops.PATH_AUTO)), ("full --rows 256", bench.Full(256, dev, 0, ops.PATH_AUTO)),
                 ("full (4096 rows)", bench.Full(4096, dev, 0, ops.PATH_AUTO))):
    for _ in range(5): wl.step()
    torch.cuda.synchronize()
    for rnd in range(3):
        for mode in (2, 0):
            lib.medtok_debug_set_half_gemm_k32(mode)
            for _ in range(3): wl.step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 50 if wl.rows <= 256 else 10
            for _ in range(n): wl.step()
            torch.cuda.synchronize()
            print(f"{name:18s} round {rnd}: {'tiles by row tile (before)' if mode else 'dense order (HEAD)        '}: {(time.perf_counter() - t0) / n * 1e3:8.4f} ms per forward", flush=True)
lib.medtok_debug_set_half_gemm_k32(0)
