"""Dev tool: the three attention cores on the `full` workload's ragged shapes (graph side: <= 40 nodes x 4 heads against <= 512 tokens)."""
import sys, time
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D, H, B, L = 768, 4, 4096, 512
g = torch.Generator(device=dev).manual_seed(0)
tok = torch.randint(1, L + 1, (B,), device=dev, generator=g)
nn_ = torch.randint(1, 41, (B,), device=dev, generator=g)
q_len = nn_ * H
q_start = torch.cumsum(q_len, 0) - q_len
q = torch.randn(int(q_len.sum()), D, device=dev, generator=g) * 0.05
text = torch.randn(B * L, D, device=dev, generator=g)
kv_start = torch.arange(B, device=dev) * L
order = torch.argsort(tok, descending=True)
a = (q, q_start[order].contiguous(), q_len[order].contiguous())
k = (kv_start[order].contiguous(), tok[order].contiguous())
pairs = float((q_len * tok).sum())
def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
r32 = t(lambda: ops.shared_kv_attention(*a, text, *k, 160, 192 ** -0.5, True))
r16 = t(lambda: ops.shared_kv_attention(*a, text, *k, 160, 192 ** -0.5, False))
sp = t(lambda: ops.split_half(text, seg_len=tok, seg_rows=L))
img = ops.split_half(text, seg_len=tok, seg_rows=L)
rd = t(lambda: ops.shared_kv_attention_split(*a, img, *k, 160, 192 ** -0.5, variant=1))
rd0 = t(lambda: ops.shared_kv_attention_split(*a, img, *k, 160, 192 ** -0.5, variant=0))
rds = t(lambda: ops.shared_kv_attention_split(*a, img, *k, 160, 192 ** -0.5, split_out=True, variant=0))
print(f"32-row DMA, two blocks per CU: {rd0:.3f} ms ({4.0 * D * pairs / rd0 / 1e9:.0f} TF); with (hi, lo) output images {rds:.3f} ms")
fl = 4.0 * D * pairs
print(f"pairs {pairs:.3g}: fp32 kernel {r32:.3f} ms ({fl/r32/1e9:.0f} TF) | f16x3 32-row {r16:.3f} ms ({fl/r16/1e9:.0f} TF) | 64-row DMA {rd:.3f} ms ({fl/rd/1e9:.0f} TF) + masked split of the text {sp:.3f} ms")
o1 = ops.shared_kv_attention(*a, text, *k, 160, 192 ** -0.5, False); o2 = ops.shared_kv_attention_split(*a, img, *k, 160, 192 ** -0.5)
print("max |diff| between the two f16x3 kernels:", float((o1 - o2).abs().max()), "of", float(o1.abs().max()))
