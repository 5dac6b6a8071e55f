"""cfg3 step (bench.Cfg3) with the four searches on one stream vs on two (inference.TWO_STREAM_MIN_ROWS), alternated in one process.
usage: python tools/r06/ab_two_stream.py [rows] [rounds] [steps]"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from medtok_amd import inference, ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
wl = bench.Cfg3(rows, dev, 0, ops.PATH_AUTO)


def run(two):
    inference.TWO_STREAM_MIN_ROWS = 1 if two else 0
    out = wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = wl.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


res = {"rows": rows, "steps": steps, "one_stream_ms": [], "two_stream_ms": []}
ref = None
for r in range(rounds):
    a, oa = run(False)
    b, ob = run(True)
    res["one_stream_ms"].append(a)
    res["two_stream_ms"].append(b)
    res["bit_identical"] = all(torch.equal(u, v) for u, v in zip(oa, ob))
with ops.ClockProbe(dev, max_seconds=30.0) as probe:
    inference.TWO_STREAM_MIN_ROWS = 1
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
res["clock_two_stream"] = probe.result()
with ops.ClockProbe(dev, max_seconds=30.0) as probe:
    inference.TWO_STREAM_MIN_ROWS = 0
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
res["clock_one_stream"] = probe.result()
print(json.dumps(res))
