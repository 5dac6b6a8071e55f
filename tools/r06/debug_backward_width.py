import sys, numpy as np, torch
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from medtok_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for d in (640, 384, 512):
    ql, kl = 40, 50
    q = (torch.randn(ql, d, device=dev) * 0.3); kv = torch.randn(kl, d, device=dev); d_out = torch.randn(ql, d, device=dev)
    z = torch.zeros(1, dtype=torch.int64, device=dev)
    args = (q, z, z + ql, kv, z, z + kl)
    out, lse = ops.shared_kv_attention_train(*args, ql, 0.2, 0.0, 7)
    # torch reference
    qr, kr = q.clone().double().requires_grad_(), kv.clone().double().requires_grad_()
    p = torch.softmax(qr @ kr.t() * 0.2, -1); o = p @ kr
    o.backward(d_out.double())
    print(d, "fwd err", float((out.double() - o).abs().max()))
    for half in (None, torch.bfloat16, torch.float16):
        dq, dkv = ops.shared_kv_attention_backward(*args, ql, kl, 0.2, 0.0, 7, out, lse, d_out, half=half)
        e = (dkv.double() - kr.grad).abs()
        print(d, half, "dq err", float((dq.double() - qr.grad).abs().max()) / float(qr.grad.abs().max()), "dkv err", float(e.max()) / float(kr.grad.abs().max()),
              "worst row/col", int(e.max(1).values.argmax()), int(e.max(0).values.argmax()), "nan", bool(torch.isnan(dkv).any()),
              "col err profile", [round(float(e[:, c:c+32].max()), 3) for c in range(0, d, 32)][:20])
