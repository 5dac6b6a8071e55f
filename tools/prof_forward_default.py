"""Dev tool: kernel breakdown of VectorQuantizer.forward (eval) at the reference's default shape."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from oracle import synth
dev = torch.device("cuda:0")
B, L, M, D, n_e = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 512, 40, 64, 21000
v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
z = torch.randn(B, 2 * D)
args = [t.to(dev) for t in (z, text, nodes, mask, batch)]
with torch.no_grad():
    for _ in range(3): v(*args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5): v(*args)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
