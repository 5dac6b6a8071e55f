"""Dev tool: the full forward under ONE launch-order / stream setting per process (argv[1]); run the settings alternately from a
shell loop on one box.  (Settings must not be switched inside a process: HIP maps streams onto a few hardware queues, and
re-created streams land on other queues -- 315 k vs 410 k codes/s for the same code, tools/ab_streams.py.)"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from medtok_amd import ops
import medtok_amd.vector_quantization_soft_one_new as vq
variants = {
    "shipped": {},
    "text_front": dict(TEXT_CHAIN_AFTER_LAYER=-1),
    "text_behind_1": dict(TEXT_CHAIN_AFTER_LAYER=1),
    "prio_equal": dict(STREAM_PRIORITY=(0, 0, 0)),
    "images_high": dict(STREAM_PRIORITY=(-1, 0, -1)),
    "searches_high": dict(STREAM_PRIORITY=(-1, -1, 0)),
    "no_lpt": dict(LPT_ORDER=False),
    "one_stream": dict(SIDE_STREAM_MIN_CODES=0),
    "att0": dict(ATTENTION_VARIANT=0),
    "att0_one_stream": dict(ATTENTION_VARIANT=0, SIDE_STREAM_MIN_CODES=0),
}
name = sys.argv[1]
for k, v in variants[name].items(): setattr(vq, k, v)
dev = torch.device("cuda:0")
w = bench.Full(4096, dev, 0, ops.PATH_AUTO)
def run(steps=10):
    for _ in range(2): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): w.step()
    torch.cuda.synchronize(); return 4096 * steps / (time.perf_counter() - t0)
print(f"{name:16s}", "  ".join(f"{run()/1e3:6.1f}k" for _ in range(3)), flush=True)
