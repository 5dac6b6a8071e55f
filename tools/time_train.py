"""Dev tool: VectorQuantizer train step (forward with z_aug + total_loss + backward) at training shapes, split into pieces."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from medtok_amd import loss as L
from oracle import synth
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, Lt, M, D, n_e in ((256, 512, 40, 64, 21000), (256, 512, 40, 768, 49152)):
    torch.manual_seed(0)
    v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).train()
    for m in v.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    text, mask, nodes, batch = synth.ragged_batch("tt", B, Lt, M, D, 0)
    z, za = torch.randn(B, 2 * D), torch.randn(B, 2 * D)
    z, text, nodes, mask, batch, za = [t.to(dev) for t in (z, text, nodes, mask, batch, za)]
    z.requires_grad_(True); za.requires_grad_(True)

    def step():
        v.zero_grad(set_to_none=True)
        r = v(z, text, nodes, mask, batch, za)
        loss, _ = L.total_loss(r)
        loss.backward()

    def fwd_only():
        with torch.no_grad():
            v(z, text, nodes, mask, batch, za)

    def spec():
        v.zero_grad(set_to_none=True)
        zq, (vq, cm, _, _), _ = v.specific_embedding(z[:, :D], "text")
        (vq + cm + zq.square().mean()).backward()

    def losses():
        a, b = z[:, :D], za[:, :D]
        s = L.shared_loss(a, b, a, b); p = L.specific_loss(a, b, a, b, a, b)
        sum(s + p).backward()

    def xattn():
        pt, pg = v.cross_attn.pooled(text, mask, nodes, batch)
        (pt.sum() + pg.sum()).backward()

    print(f"B={B} L={Lt} D={D} n_e={n_e}: train step {timeit(step):.2f} ms | no-grad forward {timeit(fwd_only):.2f} | "
          f"one specific search fwd+bwd {timeit(spec):.2f} | loss.py fwd+bwd {timeit(losses):.2f} | cross-attn fwd+bwd {timeit(xattn):.2f}", flush=True)
