"""Dev tool: where filter_rows64_kernel's cycles go -- per-wave s_memtime sums of its loop segments (TIMED instantiation through
medtok_debug_filter_probe) on one search at the reference's shape.    python tools/r04/filter_probe64.py [N] [K] [out.json]"""
import ctypes, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from medtok_amd import _lib, ops
import os
if os.environ.get('DBGLIB'):
    _lib.use_library(os.environ['DBGLIB'])
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 21000
D = 64
g = torch.Generator(device=dev).manual_seed(0)
xh, xs = ops.rownorm(torch.randn(N, D, device=dev, generator=g))
wh, ws = ops.rownorm(torch.randn(K, D, device=dev, generator=g))
lib = _lib.load()
nb = lib.medtok_search_workspace_bytes(N, K, D, 5, _lib.plan_path(ops.PATH_F16_FILTER, filter_rows64="wide"))   # (the timed kernel's plan)
wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
probe = torch.zeros(16384 * 32, dtype=torch.int64, device=dev)
nblk = ctypes.c_int64(0)
def run():
    rc = lib.medtok_debug_filter_probe(xh.data_ptr(), xs.data_ptr(), N, wh.data_ptr(), ws.data_ptr(), K, D, 5, wsb.data_ptr(), nb, probe.data_ptr(), probe.numel() * 8,
                                       ctypes.byref(nblk), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.medtok_last_error()
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
raw = probe[: nblk.value * 32].view(nblk.value, 4, 8).cpu()
hit_cycles = (raw[:, :, 6] >> 32).double()
raw[:, :, 6] &= 0xFFFFFFFF
d = raw.double()
tiles = d[:, :, 5].sum()
seg = ["wait for own copies (s_waitcnt vmcnt(0): DMA and candidate stores)", "tile barrier", "DMA issue (8 + 1 instructions)", "32 MFMAs + 16 operand reads", "epilogue (scan, hits, merge, restart)"]
tot = d[:, :, :5].sum()
out = {"workload": f"one search, {N} rows x K={K}, D={D}, k=5", "timed_launch_incl_prep_ms": dt * 1e3, "blocks": int(nblk.value),
       "tiles_per_wave_mean": float(d[:, :, 5].mean()), "cycles_per_tile_and_wave": float(tot / tiles), "hit_sequences_per_tile_and_wave": float(d[:, :, 6].sum() / tiles), "segments": {}}
print(f"{nblk.value} blocks, {float(d[:, :, 5].mean()):.0f} code tiles per wave; timed launch with operand prep {dt * 1e3:.2f} ms")
print(f"cycles per code tile and wave: {float(tot / tiles):.0f}  (32 MFMAs of 32 cycles = 1024); hit sequences per tile and wave: {float(d[:, :, 6].sum() / tiles):.2f} of 32 quad tests")
print(f"  of the epilogue, inside the hit blocks (quad hit sequences of an accumulator tile, with the scalar branches around them): {float(hit_cycles.sum() / tiles):.0f} cycles/tile")
out["hit_block_cycles_per_tile"] = float(hit_cycles.sum() / tiles)
for i, nm in enumerate(seg):
    c = float(d[:, :, i].sum() / tiles)
    out["segments"][nm] = {"cycles_per_tile": c, "share": c / float(tot / tiles)}
    print(f"  {nm:72s} {c:7.0f} cycles/tile  {100 * c / float(tot / tiles):5.1f} %")
life = d[:, :, :5].sum(2).mean(1)
print(f"block lifetime: mean {float(life.mean()):.0f} cycles; sum over blocks / 512 block slots = {float(life.sum()) / 512 / 1e6:.2f} Mcycles")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
