"""Dev tool (GPU box): exact fp32 path vs fp16 shortlist at the small-batch shapes of a B = 256 forward at BASELINE's width
(bench.py --workload full --rows 256: shared search 512 rows x 49152 codes, specific searches 256 x 16384; D = 768, k = 5),
and one size up.  Prints ms per search for both paths and what AUTO resolves to."""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")


def t(n, k, d, topk, path, iters=30):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    try:
        for _ in range(3): ops.topk_search(xh, xs, wh, ws, topk, path)
    except Exception as exc:
        return float("nan")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.topk_search(xh, xs, wh, ws, topk, path)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for d, k in ((768, 49152), (768, 16384), (768, 8192), (256, 49152)):
    for n in (128, 256, 512, 1024, 2048, 4096):
        a, b = t(n, k, d, 5, ops.PATH_F32_MFMA), t(n, k, d, 5, ops.PATH_F16_FILTER)
        auto = "filter" if ops.takes_filter_path(n, k, d, 5) else "f32"
        print(f"D={d} K={k} N={n}: f32 {a:.3f} ms  filter {b:.3f} ms  -> {'filter' if b < a else 'f32'} faster; AUTO takes {auto}", flush=True)
