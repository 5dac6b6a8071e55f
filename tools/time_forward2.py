import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from medtok_amd import ops
from oracle import synth
dev = torch.device("cuda:0")
B, L, M, D, n_e = 256, 512, 40, 64, 21000
torch.manual_seed(0)
v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
z = torch.randn(B, 2 * D)
args = [t.to(dev) for t in (z, text, nodes, mask, batch)]
def T(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("forward", T(lambda: v(*args)))
    print("pooled", T(lambda: v.cross_attn.pooled(args[1], args[3], args[2], args[4])))
    x = torch.randn(B, D, device=dev)
    print("specific_embedding", T(lambda: v.specific_embedding(x, "text")))
    print("_search only", T(lambda: v._search(x, "text", False)))
    idx = torch.randint(0, 7000, (B, 5), device=dev)
    print("usage_update_", T(lambda: ops.usage_update_(v.codebook_used, idx, n_e)))
    print("usage + item", T(lambda: v.codebook_usage(idx, "shared")))
    what, wsq = v._normalised_codebook()
    print("soft_vq_forward", T(lambda: ops.soft_vq_forward(x, what[:7000], wsq[:7000].contiguous(), 5)))
    xh, xs = ops.rownorm(x)
    print("topk_search", T(lambda: ops.topk_search(xh, xs, what[:7000], wsq[:7000].contiguous(), 5)))
    v.show_usage = False
    print("forward(no usage)", T(lambda: v(*args)))
