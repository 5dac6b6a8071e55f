"""Dev tool: time ops.shared_kv_attention at the cfg-4 shapes (B=256 codes, <=512 tokens, <=40 nodes, D=768, 4 heads)."""
import sys, time
sys.path.insert(0, ".")
import torch
import os as _os
if _os.environ.get("MEDTOK_TOOL_LIB"):
    from medtok_amd import _lib as _l; _l.use_library(_os.environ["MEDTOK_TOOL_LIB"])
from medtok_amd import ops
from oracle import synth
dev = torch.device("cuda:0")
B, L, M, D, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 512, 40, (int(sys.argv[2]) if len(sys.argv) > 2 else 768), 4
text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
counts = torch.bincount(batch, minlength=B); starts = torch.cumsum(counts, 0) - counts
valid = mask.sum(1)
code = torch.arange(B)
def T(x): return x.to(dev)
qg = torch.randn(nodes.shape[0] * H, D, device=dev) * 0.05
qt = torch.randn(B * H, D, device=dev) * 0.05
textf = T(text.reshape(B * L, D)); nd = T(nodes)
args_g = (qg, T(starts * H), T(counts * H), textf, T(code * L), T(valid), int(counts.max()) * H, 192 ** -0.5)
args_t = (qt, T(code * H), T(torch.full((B,), H)), nd, T(starts), T(counts), H, 192 ** -0.5)
flops_g = float((counts * H * valid).sum()) * D * 4
flops_t = float((H * counts).sum()) * D * 4
for name, a, fl in (("graph side", args_g, flops_g), ("text side", args_t, flops_t)):
    for _ in range(3): ops.shared_kv_attention(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.shared_kv_attention(*a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e6:.0f} us, {fl/dt/1e12:.1f} TFLOP/s useful ({fl/1e9:.1f} GF)", flush=True)
