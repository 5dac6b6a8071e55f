"""Dev tool: torch-profiler breakdown of CrossAttention.pooled (no grad) at B=256, L=512, D=768."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from oracle import synth
dev = torch.device("cuda:0")
B, L, M, D = 256, 512, 40, 768
v = VectorQuantizer(3072, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
print("valid tokens", int(mask.sum()), "of", B * L, "| nodes", nodes.shape[0], "max per code", int(torch.bincount(batch).max()))
args = [t.to(dev) for t in (text, mask, nodes, batch)]
with torch.no_grad():
    for _ in range(3): v.cross_attn.pooled(*args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5): v.cross_attn.pooled(*args)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
