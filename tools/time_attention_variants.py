"""Dev tool: the graph side's attention launch of the `full` workload (bench.Full's ragged shapes: 4096 codes, 1..40 nodes x 4
heads, 1..512 tokens, D = 768, keys as (hi, lo) images, LPT order) on each variant of ops.shared_kv_attention_split, alternated
in one process; checks that the variants return the same bits.   python tools/time_attention_variants.py [variants...] [--codes N]"""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops, _lib
import os
if os.environ.get('DBGLIB'):
    _lib.use_library(os.environ['DBGLIB'])
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
variants = [int(a) for a in args] or [0, 2, 258]          # (258 = 2 | 0x100: fp32 keys, split inside the kernel)
B = int(sys.argv[sys.argv.index("--codes") + 1]) if "--codes" in sys.argv else 4096
D, H, L = 768, 4, 512
g = torch.Generator(device=dev).manual_seed(77)
tok = torch.randint(1, L + 1, (B,), device=dev, generator=g)
n_nodes = torch.randint(1, 41, (B,), device=dev, generator=g)
text = torch.randn(B * L, D, device=dev, generator=g)
q = torch.randn(int(n_nodes.sum()) * H, D, device=dev, generator=g) * 0.05
images = ops.split_half(text, seg_len=tok, seg_rows=L)
starts = torch.cumsum(n_nodes, 0) - n_nodes
order = torch.argsort(tok, descending=True)
q_start, q_len, k_start, k_len = (starts * H)[order], (n_nodes * H)[order], (torch.arange(B, device=dev) * L)[order], tok[order]
pairs = float((n_nodes * H * tok).sum())
nbytes = 4.0 * D * (float(tok.sum()) + 2.0 * float((n_nodes * H).sum()))
def run(v, split_out=False):
    keys = text if v & 0x100 else images
    return ops.shared_kv_attention_split(q, q_start, q_len, keys, k_start, k_len, 160, 192 ** -0.5, split_out=split_out, variant=v & 0xFF)
outs = {v: run(v) for v in variants}
torch.cuda.synchronize()
ref = outs[variants[0]]
for v in variants[1:]:
    same = torch.equal(outs[v], ref)
    print(f"variant {v} vs {variants[0]}: bit-identical {same}; max abs diff {float((outs[v] - ref).abs().max()):.3e} (scale {float(ref.abs().max()):.3e})")
for rnd in range(3):
    for v in variants:
        for _ in range(2): run(v, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run(v, True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"round {rnd} variant {v}: {dt*1e3:7.3f} ms  {4.0*D*pairs/dt/1e12:6.1f} TFLOP/s fp32-equiv  {nbytes/dt/1e12:5.2f} TB/s algorithmic", flush=True)
