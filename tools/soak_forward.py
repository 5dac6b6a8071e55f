"""Dev tool: the multi-stream inference forward run repeatedly at BASELINE sizes; every output of every run must be bit-identical to
the first run and to the single-stream forward (a missing stream dependency shows up as a difference, eventually)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from medtok_amd import ops
import medtok_amd.vector_quantization_soft_one_new as vqmod
dev = torch.device("cuda:0")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w = bench.Full(4096, dev, 0, ops.PATH_AUTO)
torch.manual_seed(0)
vq = vqmod.VectorQuantizer(w.N_E, w.D, 0.25, 0.0, True, True, [w.D, w.D], k=w.TOPK).to(dev).eval()      # (with the usage window)
def forward():
    vq._norm_cache = None
    vq.codebook_used.zero_()
    with torch.no_grad():
        out = vq(w.h, w.text, w.nodes, w.mask, w.batch)
    torch.cuda.synchronize()
    return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
def same(a, b):
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, torch.Tensor):
            if not torch.equal(x, y): bad.append(k)
        elif isinstance(x, tuple):
            for i, (p, q) in enumerate(zip(x, y)):
                if isinstance(p, torch.Tensor) and not torch.equal(p, q): bad.append(f"{k}[{i}]")
        elif x != y:
            bad.append(k)
    return bad
keep = vqmod.SIDE_STREAM_MIN_CODES
vqmod.SIDE_STREAM_MIN_CODES = 0
single = forward()
vqmod.SIDE_STREAM_MIN_CODES = keep
first = forward()
print("multi-stream vs single-stream:", same(single, first) or "identical")
fails = 0
for i in range(runs):
    # vary the timing: different amounts of unrelated work in flight when the forward starts
    junk = [torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev) for _ in range(i % 4)]
    bad = same(first, forward())
    if bad:
        fails += 1; print("run", i, "differs in", bad)
print(f"{runs} runs, {fails} with differences")
sys.exit(1 if fails else 0)
