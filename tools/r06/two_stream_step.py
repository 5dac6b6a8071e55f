"""Dev tool (GPU box): a few cfg3 steps with the four searches on two HIP streams (inference.TWO_STREAM_MIN_ROWS = 1) or on one -- the
program tools/r06/two_stream_timeline.sh runs under rocprofv3; prints the shader clock of the steps (ops.ClockProbe).
usage: python tools/r06/two_stream_step.py {one|two} [steps]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from medtok_amd import inference, ops  # noqa: E402

two = len(sys.argv) > 1 and sys.argv[1] == "two"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
inference.TWO_STREAM_MIN_ROWS = 1 if two else 0
wl = bench.Cfg3(600000, dev, 0, ops.PATH_AUTO)
wl.step()
torch.cuda.synchronize()
with ops.ClockProbe(dev, max_seconds=20.0) as probe:
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
print(json.dumps({"streams": 2 if two else 1, "clock": probe.result()}))
