"""Dev tool: the attention backward (dQ + dKV kernels, profiled separately by the library) on uniform shapes."""
import sys
sys.path.insert(0, ".")
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D = 768
for B, rows, T in ((1024, 96, 256), (1024, 32, 512)):
    q = torch.randn(B * rows, D, device=dev) * 0.05
    kv = torch.randn(B * T, D, device=dev)
    code = torch.arange(B, device=dev)
    qs, ql, ks, kl = code * rows, torch.full((B,), rows, device=dev), code * T, torch.full((B,), T, device=dev)
    out, lse = ops.shared_kv_attention_train(q, qs, ql, kv, ks, kl, rows, 192 ** -0.5, 0.1, 7)
    do = torch.randn_like(out)
    for _ in range(2): ops.shared_kv_attention_backward(q, qs, ql, kv, ks, kl, rows, T, 192 ** -0.5, 0.1, 7, out, lse, do)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10): ops.shared_kv_attention_backward(q, qs, ql, kv, ks, kl, rows, T, 192 ** -0.5, 0.1, 7, out, lse, do)
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 10
    fl = B * rows * T * D * 2.0 * 5      # five products of 2 D flops per (row, key)
    print(f"B={B} rows={rows} T={T}: backward {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TF", flush=True)
