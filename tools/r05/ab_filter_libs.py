"""Dev tool (GPU box): the fp16 filter kernel's time under variant builds of the library (tools/r05/build_mutants.py), one fresh process
per build (a library is bound once per process), alternated ROUNDS times.   python tools/r05/ab_filter_libs.py K[:D] name [name ...]   (D = 768 unless given)
('shipped' = the product library).  Prints the mean filter-kernel milliseconds per 600 000-row search of every build and round."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from medtok_amd import _lib, ops
    if os.environ.get("DBGLIB"): _lib.use_library(os.environ["DBGLIB"])
    K, _, D = sys.argv[2].partition(":"); K = int(K); D = int(D or 768); dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(600000, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    for _ in range(2): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    ops.profile_begin()
    for _ in range(4): ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    p = ops.profile_end()["filter_f16_kernel"]
    print(json.dumps({"ms": p["ms"] / max(p["launches"], 1), "launches": p["launches"]}))
    sys.exit(0)
K = sys.argv[1]; names = sys.argv[2:]; ROUNDS = 3
res = {n: [] for n in names}
for r in range(ROUNDS):
    for n in names:
        env = dict(os.environ)
        if n != "shipped": env["DBGLIB"] = str(ROOT / "devlib" / n / "libmedtok_vq.so")
        out = subprocess.run([sys.executable, __file__, "--child", K], env=env, capture_output=True, text=True, timeout=600)
        try: res[n].append(json.loads(out.stdout.strip().splitlines()[-1])["ms"])
        except Exception: res[n].append(None); print(n, "FAILED", out.stderr[-400:])
for n in names: print(f"K={K} {n:10s} filter kernel ms per search:", " ".join("%.3f" % v if v else "fail" for v in res[n]))
