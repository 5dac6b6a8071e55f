"""Dev tool (GPU box): one whole fused search (rownorm + filter + re-score with the soft assignment; 600 000 rows, D = 768) under variant
builds of the library, one fresh process per build, alternated.   python tools/r05/ab_search_libs.py K name [name ...]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from medtok_amd import _lib, ops
    if os.environ.get("DBGLIB"): _lib.use_library(os.environ["DBGLIB"])
    K = int(sys.argv[2]); dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(600000, 768, device=dev, generator=g); W = torch.randn(K, 768, device=dev, generator=g)
    wh, ws = ops.rownorm(W)
    out = torch.empty_like(x)
    for _ in range(2): r = ops.soft_vq_forward(x, wh, ws, 5, want_sqerr=False, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): r = ops.soft_vq_forward(x, wh, ws, 5, want_sqerr=False, out=out)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"ms": e0.elapsed_time(e1) / 4, "chk": int(r["idx"].sum().item())}))
    sys.exit(0)
K = sys.argv[1]; names = sys.argv[2:]
res = {n: [] for n in names}
for r in range(3):
    for n in names:
        env = dict(os.environ)
        if n != "shipped": env["DBGLIB"] = str(ROOT / "devlib" / n / "libmedtok_vq.so")
        out = subprocess.run([sys.executable, __file__, "--child", K], env=env, capture_output=True, text=True, timeout=600)
        try: res[n].append(json.loads(out.stdout.strip().splitlines()[-1]))
        except Exception: res[n].append(None); print(n, "FAILED", out.stderr[-300:])
for n in names: print(f"K={K} {n:8s} fused search ms:", " ".join("%.3f" % v["ms"] if v else "fail" for v in res[n]), " checksum", {v["chk"] for v in res[n] if v})
