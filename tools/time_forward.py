"""Dev tool: full VectorQuantizer.forward at the reference's default shape (B=256/GPU, L=512, D=64, n_e=21000)."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from oracle import synth
dev = torch.device("cuda:0")
import sys as _s
SHAPES = ((256, 512, 40, 64, 21000), (1024, 512, 40, 64, 21000), (256, 64, 20, 768, 24576)) if len(_s.argv) < 2 else ((4096, 512, 40, 64, 21000), (16384, 512, 40, 64, 21000))
for B, L, M, D, n_e in SHAPES:
    torch.manual_seed(0)
    v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
    text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
    z = torch.randn(B, 2 * D)
    args = [t.to(dev) for t in (z, text, nodes, mask, batch)]
    with torch.no_grad():
        for _ in range(3): v(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): v(*args)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        # pieces
        t1 = time.perf_counter()
        for _ in range(20): v.cross_attn.pooled(args[1], args[3], args[2], args[4])
        torch.cuda.synchronize(); dc = (time.perf_counter() - t1) / 20
        for _ in range(3): v.cross_attn.pooled(args[1], args[3], args[2], args[4], fold=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20): v.cross_attn.pooled(args[1], args[3], args[2], args[4], fold=True)
        torch.cuda.synchronize(); df = (time.perf_counter() - t1) / 20
        for _ in range(3): v.cross_attn.pooled(args[1], args[3], args[2], args[4], fold=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20): v.cross_attn.pooled(args[1], args[3], args[2], args[4], fold=False)
        torch.cuda.synchronize(); dn = (time.perf_counter() - t1) / 20
    print(f"B={B} L={L} D={D} n_e={n_e}: forward {dt*1e3:.2f} ms ({B/dt:.0f} codes/s), cross-attention part {dc*1e3:.2f} ms (forced fold/packed {df*1e3:.2f}, projected keys {dn*1e3:.2f})", flush=True)
