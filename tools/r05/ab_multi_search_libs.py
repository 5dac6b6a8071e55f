"""Dev tool (GPU box): the batched search call of the B = 256 forward at the reference's shape (3 searches: 512 rows x 21 000 codes, twice
256 x 7 000; e_dim 64, k = 5) under variant builds, one process per build.   python tools/r05/ab_multi_search_libs.py name [name ...]"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from medtok_amd import _lib, ops
    if os.environ.get("DBGLIB"): _lib.use_library(os.environ["DBGLIB"])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    what, wsq = ops.rownorm(torch.randn(21000, 64, device=dev, generator=g))
    xs = [torch.randn(512, 64, device=dev, generator=g), torch.randn(256, 64, device=dev, generator=g), torch.randn(256, 64, device=dev, generator=g)]
    regs = [(0, 21000), (0, 7000), (14000, 21000)]
    searches = [dict(x=x, what=what[lo:hi], wsq=wsq[lo:hi].contiguous()) for x, (lo, hi) in zip(xs, regs)]
    for _ in range(20): ops.soft_vq_forward_multi(searches, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.soft_vq_forward_multi(searches, 5)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"us": e0.elapsed_time(e1) / 200 * 1e3}))
    sys.exit(0)
names = sys.argv[1:]; res = {n: [] for n in names}
for r in range(3):
    for n in names:
        env = dict(os.environ)
        if n != "shipped": env["DBGLIB"] = str(ROOT / "devlib" / n / "libmedtok_vq.so")
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=600)
        try: res[n].append(json.loads(out.stdout.strip().splitlines()[-1])["us"])
        except Exception: res[n].append(None); print(n, "FAILED", out.stderr[-400:])
for n in names: print(f"{n:10s} us per call (3 launches):", " ".join("%.1f" % v if v else "fail" for v in res[n]))
