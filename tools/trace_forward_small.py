"""Dev tool: one VectorQuantizer.forward at the reference's default shape, for rocprofv3 --kernel-trace."""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from oracle import synth
dev = torch.device("cuda:0")
B, L, M, D, n_e = 256, 512, 40, 64, 21000
torch.manual_seed(0)
v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
text, mask, nodes, batch = synth.ragged_batch("tf", B, L, M, D, 0)
args = [t.to(dev) for t in (torch.randn(B, 2 * D), text, nodes, mask, batch)]
with torch.no_grad():
    for _ in range(4): v(*args)
    torch.cuda.synchronize()
print("done")
