"""Dev tool: VectorQuantizer train step under bf16 autocast (cfg-4 shape: B=256, L=512, D=768, n_e=49152), with a kernel breakdown."""
import sys, time
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
from medtok_amd import loss as L
from oracle import synth
dev = torch.device("cuda:0")
B, Lt, M, D, n_e = 256, 512, 40, 768, 49152
torch.manual_seed(0)
v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).train()
text, mask, nodes, batch = synth.ragged_batch("tt", B, Lt, M, D, 0)
z, za = torch.randn(B, 2 * D), torch.randn(B, 2 * D)
z, text, nodes, mask, batch, za = [t.to(dev) for t in (z, text, nodes, mask, batch, za)]
z.requires_grad_(True); za.requires_grad_(True)
opt = torch.optim.Adam(v.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        r = v(z, text.bfloat16(), nodes, mask, batch, za)
        loss, _ = L.total_loss(r)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"bf16-autocast train step: {(time.perf_counter()-t0)/10*1e3:.2f} ms", flush=True)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
