"""Dev tool (GPU box): verdict r05 item 1c, measured instead of estimated -- the full forward (bench.Full, 4096 codes, one stream and
streams) with (a) the shipped library, (b) the shipped library called layer by layer (no fused C call: the reference point for c),
(c) the `wv_epi` variant (tools/r06/build_variants.py: the attention kernel carries the minimum extra work of a fused W_v,h epilogue)
with the W_v GEMM launch dropped from the layer chain (what the fusion would save).  (c) returns wrong values: only its time counts.
One fresh process per setting, alternated ROUNDS times.     python tools/r06/ab_wv_epilogue.py"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from medtok_amd import _lib, ops
    mode = sys.argv[2]
    if mode == "wv_epi":
        _lib.use_library(str(ROOT / "devlib" / "wv_epi" / "libmedtok_vq.so"))
    import bench
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    if sys.argv[3] == "one":
        vqmod.SIDE_STREAM_MIN_CODES = 0
    if mode in ("layerwise", "wv_epi"):
        vqmod.FUSED_LAYER_CALL = False
    if mode == "wv_epi":
        real = ops.split_gemm
        def fake(a, b, n_g, k_g, groups=1, a_group_cols=0, b_group_rows=0, **kw):
            # the per-head W_v product (groups = heads, a_group_cols = Dw, k_g = Dw, n_g = hp): dropped -- its output would come out
            # of the attention kernel's epilogue; stand-in images of the right shape from the context's first columns
            if groups > 1 and a_group_cols == k_g and kw.get("want_split"):
                hi, lo = a
                w = groups * n_g
                return None, (hi[:, :w].contiguous(), lo[:, :w].contiguous())
            return real(a, b, n_g, k_g, groups=groups, a_group_cols=a_group_cols, b_group_rows=b_group_rows, **kw)
        ops.split_gemm = fake
    dev = torch.device("cuda:0")
    wl = bench.Full(4096, dev, seed=0, path=0)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    print(json.dumps({"ms": (time.perf_counter() - t0) / n * 1e3}))
    sys.exit(0)
ROUNDS = 3
res = {}
for r in range(ROUNDS):
    for streams in ("one", "multi"):
        for mode in ("shipped", "layerwise", "wv_epi"):
            out = subprocess.run([sys.executable, __file__, "--child", mode, streams], capture_output=True, text=True, timeout=900)
            try:
                res.setdefault((mode, streams), []).append(json.loads(out.stdout.strip().splitlines()[-1])["ms"])
            except Exception:
                res.setdefault((mode, streams), []).append(None)
                print(mode, streams, "FAILED", out.stderr[-800:])
for (mode, streams), v in res.items():
    print(f"full forward, 4096 codes, {streams:5s} stream(s), {mode:10s} ms per step:", " ".join("%.3f" % x if x else "fail" for x in v))
