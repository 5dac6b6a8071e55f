"""Dev tool: cross_attn.pooled() of two versions of the module file (medtok_amd/_vq_prev.py = the previous one), alternated."""
import sys, time, statistics
sys.path.insert(0, ".")
import torch
from medtok_amd import vector_quantization_soft_one_new as new
from medtok_amd import _vq_prev as old
from oracle import synth
dev = torch.device("cuda:0")
for B, L, M, D in ((256, 512, 40, 64), (1024, 512, 40, 64), (256, 64, 20, 768)):
    torch.manual_seed(0)
    vn = new.VectorQuantizer(3000, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
    vo = old.VectorQuantizer(3000, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
    vo.load_state_dict(vn.state_dict())
    text, mask, nodes, batch = (t.to(dev) for t in synth.ragged_batch("tf", B, L, M, D, 0))
    res = {"new": [], "old": []}
    with torch.no_grad():
        for r in range(7):
            for name, v in (("new", vn), ("old", vo)):
                v.cross_attn.pooled(text, mask, nodes, batch); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20): v.cross_attn.pooled(text, mask, nodes, batch)
                torch.cuda.synchronize()
                if r: res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"B={B} L={L} D={D}: new {statistics.median(res['new']):.3f} ms, old {statistics.median(res['old']):.3f} ms", flush=True)
