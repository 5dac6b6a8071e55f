"""Dev tool (GPU box): bench workloads under variant builds of the library (tools/r05/build_mutants.py), one fresh process per build,
alternated ROUNDS times.   python tools/r05/ab_forward_libs.py {full|cfg3|cfg4vq|fullref} name [name ...]      ('shipped' = the product library)"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from medtok_amd import _lib
    if os.environ.get("DBGLIB"): _lib.use_library(os.environ["DBGLIB"])
    import bench
    dev = torch.device("cuda:0")
    what = sys.argv[2]
    if what == "cfg4vq":
        wl = bench.Cfg4(256, dev, seed=0, path=0, precomputed=True)
    elif what == "cfg3":
        wl = bench.Cfg3(600000, dev, seed=0, path=0)
    elif what == "fullref":
        wl = bench.FullRefDefault(256, dev, seed=0, path=0)
    else:
        wl = bench.Full(4096, dev, seed=0, path=0)
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    n = 100 if what == "fullref" else 5 if what == "cfg3" else 10
    t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize()
    print(json.dumps({"ms": (time.perf_counter() - t0) / n * 1e3}))
    sys.exit(0)
what = sys.argv[1]; names = sys.argv[2:]; ROUNDS = 3
res = {n: [] for n in names}
for r in range(ROUNDS):
    for n in names:
        env = dict(os.environ)
        if n != "shipped": env["DBGLIB"] = str(ROOT / "devlib" / n / "libmedtok_vq.so")
        out = subprocess.run([sys.executable, __file__, "--child", what], env=env, capture_output=True, text=True, timeout=900)
        try: res[n].append(json.loads(out.stdout.strip().splitlines()[-1])["ms"])
        except Exception: res[n].append(None); print(n, "FAILED", out.stderr[-600:])
for n in names: print(f"{what} {n:10s} ms per step:", " ".join("%.3f" % v if v else "fail" for v in res[n]))
