"""Dev tool (GPU box): host-side cost of the pieces of the B = 256 eval forward (enqueue time without synchronisation, 300 calls each)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
wl = bench.FullRefDefault(256, dev, 0, ops.PATH_AUTO)
vq = wl.vq
norm = vq._normalised_codebook()
what, wsq = norm
def host(fn, n=300):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
with torch.no_grad():
    both = vq.cross_attn.pooled_small(wl.text, wl.mask, wl.nodes, wl.batch)
    proj = vq.project_both(wl.h)
    e, k, B = 64, 5, 256
    emb = torch.empty(B, 2 * e, device=dev)
    def searches():
        s = [dict(x=both.view(2 * B, e), what=what, wsq=wsq, out=emb.view(2 * B, e))]
        for i, t in enumerate(("text", "graph")):
            lo, hi = vq._region(t)
            s.append(dict(x=proj[:, i * e:(i + 1) * e], what=what[lo:hi], wsq=wsq[lo:hi].contiguous()))
        return ops.soft_vq_forward_multi(s, k)
    res = searches()
    ids = [res[0]["idx"].view(B, 2 * k), res[1]["idx"], res[2]["idx"]]
    for name, fn in (("pooled_small", lambda: vq.cross_attn.pooled_small(wl.text, wl.mask, wl.nodes, wl.batch)),
                     ("project_both", lambda: vq.project_both(wl.h)),
                     ("soft_vq_forward_multi (3 searches)", searches),
                     ("usage_update_multi_", lambda: ops.usage_update_multi_(vq.codebook_used, ids, vq.n_e)),
                     ("usage + .cpu()", lambda: ops.usage_update_multi_(vq.codebook_used, ids, vq.n_e).cpu()),
                     ("whole forward (show_usage=True)", lambda: vq(wl.h, wl.text, wl.nodes, wl.mask, wl.batch))):
        h, w = host(fn)
        print(f"{name:40s} host {h:7.1f} us   wall {w:7.1f} us per call", flush=True)
    # ---- inside project_both
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    z = wl.h
    img = ops.split_half(z, dp=128)
    lt, lg = vq.proj_text, vq.proj_graph
    key = tuple((t.data_ptr(), t._version) for t in (lt.weight, lt.bias, lg.weight, lg.bias)) + (lt.weight.device,)
    w_split, unscale, bias = vq._medtok_proj_both_cache[1]
    for name, fn in (("  split_half(z)", lambda: ops.split_half(z, dp=128)),
                     ("  split_gemm(grouped)", lambda: ops.split_gemm(img, w_split, n_g=64, k_g=64, groups=2, a_group_cols=64, b_group_rows=64, bias=bias, unscale=unscale)),
                     ("  key tuple", lambda: tuple((t.data_ptr(), t._version) for t in (lt.weight, lt.bias, lg.weight, lg.bias)) + (lt.weight.device,)),
                     ("  _cached lookup", lambda: vqmod._cached(vq, "_medtok_proj_both_cache", key, None, lt.weight.device)),
                     ("  is_current_stream_capturing", lambda: torch.cuda.is_current_stream_capturing()),
                     ("  torch.empty x10", lambda: [torch.empty((256, 64), device=dev) for _ in range(10)])):
        h, w = host(fn)
        print(f"{name:40s} host {h:7.1f} us   wall {w:7.1f} us per call", flush=True)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): vq.project_both(wl.h)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
