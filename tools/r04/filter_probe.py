"""Dev tool (round 4): where the fp16 filter kernel's cycles go -- per-wave s_memtime sums of its stage-loop segments on one search
of cfg 3 (600 000 rows x K codes, D = 768), written by the TIMED instantiation through medtok_debug_filter_probe.
    python tools/r04/filter_probe.py [K] [out.json]"""
import ctypes, json, sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import _lib, ops
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N, D = 600000, 768
g = torch.Generator(device=dev).manual_seed(0)
xh, xs = ops.rownorm(torch.randn(N, D, device=dev, generator=g))
wh, ws = ops.rownorm(torch.randn(K, D, device=dev, generator=g))
lib = _lib.load()
nb = lib.medtok_search_workspace_bytes(N, K, D, 5, ops.PATH_F16_FILTER)
wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
probe = torch.zeros(8192 * 64, dtype=torch.int64, device=dev)
nblk = ctypes.c_int64(0)
def run():
    rc = lib.medtok_debug_filter_probe(xh.data_ptr(), xs.data_ptr(), N, wh.data_ptr(), ws.data_ptr(), K, D, 5, wsb.data_ptr(), nb, probe.data_ptr(), probe.numel() * 8,
                                       ctypes.byref(nblk), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.medtok_last_error()
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
# the untimed product kernel for reference
for _ in range(2): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
torch.cuda.synchronize(); t0 = time.perf_counter(); ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER); torch.cuda.synchronize(); dts = time.perf_counter() - t0
d = probe[: nblk.value * 64].view(nblk.value, 8, 8).cpu().double()
live = d[:, 0, 5] > 0
d = d[live]
seg = ["mfma_group_1 (+ operand reads)", "wait for own copies (s_waitcnt vmcnt)", "stage barrier (s_barrier)", "mfma_group_2 (+ DMA issue, operand reads)", "tile epilogue (scan, hits, merge, restart)"]
stages = d[:, :, 5].sum()
tot = d[:, :, :5].sum()
out = {"workload": f"one search, {N} rows x K={K}, D={D}, k=5 (main launch of the plan)", "timed_launch_incl_prep_ms": dt * 1e3, "whole_search_untimed_ms": dts * 1e3,
       "blocks": int(live.sum()), "stages_per_wave_mean": float(d[:, :, 5].mean()), "cycles_per_stage_and_wave": float(tot / stages), "segments": {}}
print(f"K={K}: {int(live.sum())} blocks, {float(d[:, :, 5].mean()):.0f} stages per wave; timed launch (with operand prep) {dt*1e3:.2f} ms; whole untimed search {dts*1e3:.2f} ms")
print(f"cycles per stage and wave: {float(tot / stages):.0f}   (16 + 16 MFMAs of 32 cycles = 1024 if the wave had the matrix pipe to itself; two waves share a SIMD)")
for i, nm in enumerate(seg):
    c = float(d[:, :, i].sum() / stages)
    early, late = float(d[:, :4, i].sum() / d[:, :4, 5].sum()), float(d[:, 4:, i].sum() / d[:, 4:, 5].sum())
    out["segments"][nm] = {"cycles_per_stage": c, "share": c / float(tot / stages), "waves_0_3": early, "waves_4_7": late}
    print(f"  {nm:48s} {c:7.0f} cycles/stage  {100 * c / float(tot / stages):5.1f} %   (waves 0-3: {early:6.0f}, waves 4-7: {late:6.0f})")
life = d[:, :, :5].sum(2).mean(1)
print(f"block lifetime (sum of segments): mean {float(life.mean()):.0f} cycles; sum over blocks / 256 CUs = {float(life.sum()) / 256 / 1e6:.2f} Mcycles")
out["block_cycles_sum_over_256_cus_M"] = float(life.sum()) / 256 / 1e6
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
