"""Dev tool: one launch pair of the 64-row DMA attention kernel on a uniform, L2-resident shape (for rocprofv3 --pmc passes)."""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
D, B, rows, T = 768, 2048, 64, 512
q = torch.randn(B * rows, D, device=dev) * 0.05
kv = torch.randn(T, D, device=dev)
img = ops.split_half(kv)
code = torch.arange(B, device=dev)
a = (q, code * rows, torch.full((B,), rows, device=dev), img, code * 0, torch.full((B,), T, device=dev), rows, 192 ** -0.5)
for _ in range(3): ops.shared_kv_attention_split(*a)
torch.cuda.synchronize()
