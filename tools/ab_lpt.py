"""Dev tool: pooled() with / without longest-first block order."""
import sys, time, statistics
sys.path.insert(0, ".")
import torch
from medtok_amd import vector_quantization_soft_one_new as M
from oracle import synth
dev = torch.device("cuda:0")
for B, L, Mx, D in ((4096, 512, 40, 768), (16384, 512, 40, 64)):
    torch.manual_seed(0)
    v = M.VectorQuantizer(3000, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
    text, mask, nodes, batch = (t.to(dev) for t in synth.ragged_batch("tf", B, L, Mx, D, 0))
    res = {True: [], False: []}
    outs = {}
    with torch.no_grad():
        for r in range(5):
            for flag in (True, False):
                M.LPT_ORDER = flag
                outs[flag] = v.cross_attn.pooled(text, mask, nodes, batch); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): v.cross_attn.pooled(text, mask, nodes, batch)
                torch.cuda.synchronize()
                if r: res[flag].append((time.perf_counter() - t0) / 5 * 1e3)
    same = all(torch.equal(a, b) for a, b in zip(outs[True], outs[False]))
    print(f"B={B} D={D}: LPT {statistics.median(res[True]):.3f} ms, plain {statistics.median(res[False]):.3f} ms, same bits {same}", flush=True)
