"""Dev tool: does quantize_pooled capture into a HIP graph (torch.cuda.CUDAGraph), and what does replay save?"""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from medtok_amd.inference import quantize_pooled
dev = torch.device("cuda:0")
for B, D, n_e in ((256, 64, 21000), (1024, 64, 21000), (256, 768, 49152), (1024, 768, 49152)):
    torch.manual_seed(0)
    v = VectorQuantizer(n_e, D, 0.25, 0.0, True, False, [D, D]).to(dev).eval()
    h = torch.randn(B, 2 * D, device=dev); pt = torch.randn(B, D, device=dev); pg = torch.randn(B, D, device=dev)
    with torch.no_grad():
        for _ in range(3): ref = quantize_pooled(v, h, pt, pg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): quantize_pooled(v, h, pt, pg)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): quantize_pooled(v, h, pt, pg)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = quantize_pooled(v, h, pt, pg)
        g.replay(); torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 50
    print(f"B={B} D={D} n_e={n_e}: eager {eager*1e3:.3f} ms, graph replay {rep*1e3:.3f} ms, same outputs: {same}", flush=True)
