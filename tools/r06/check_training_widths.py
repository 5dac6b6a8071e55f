"""Dev tool (GPU box): CrossAttention.pooled in TRAINING (fp32 and bf16 autocast) at every kernel width against the plain-torch comparator
(pooled_reference: nn.MultiheadAttention on padded batches), outputs and gradients.   python tools/r06/check_training_widths.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from medtok_amd.vector_quantization_soft_one_new import CrossAttention
dev = torch.device("cuda:0")
for d in (64, 128, 256, 384, 512, 640, 768, 96):
    torch.manual_seed(d)
    ca = CrossAttention(d, 4, dropout=0.0).to(dev).train()
    bsz, L = 6, 40
    counts = torch.tensor([3, 0, 9, 1, 70, 2])
    batch = torch.repeat_interleave(torch.arange(bsz), counts).to(dev)
    valid = torch.tensor([40, 5, 1, 17, 33, 9])
    mask = (torch.arange(L)[None, :] < valid[:, None]).long().to(dev)
    text0, nodes0 = torch.randn(bsz, L, d, device=dev), torch.randn(int(counts.sum()), d, device=dev)
    pa, pb = torch.randn(bsz, d, device=dev), torch.randn(bsz, d, device=dev)
    def run(fn, autocast=None):
        ca.zero_grad(set_to_none=True)
        t, n = text0.clone().requires_grad_(), nodes0.clone().requires_grad_()
        import contextlib
        with (torch.autocast("cuda", dtype=autocast) if autocast else contextlib.nullcontext()):
            pt, pg = fn(t, mask, n, batch)
        keep = counts.to(dev) > 0                  # (a code without nodes: the kernels give zero context, the padded comparator a softmax over nothing)
        ((pt.float() * pa)[keep].sum() + (pg.float() * pb)[keep].sum()).backward()
        return [pt.float()[keep].detach(), pg.float()[keep].detach(), t.grad[keep].clone(), n.grad.clone(), ca.model[0].multihead_attn.in_proj_weight.grad.clone()]
    ref = run(ca.pooled_reference)
    for ac in (None, torch.bfloat16):
        got = run(ca.pooled, ac)
        errs = [float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9) for a, b in zip(got, ref)]
        print(f"D = {d:4d} autocast {str(ac):15s}: rel err of (pooled text, pooled graph, d text, d nodes, d in_proj) = {[f'{e:.1e}' for e in errs]}", flush=True)
