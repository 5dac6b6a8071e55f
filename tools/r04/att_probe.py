"""Dev tool (round 4): where the ping-pong attention kernel's time goes on the `full` workload's graph-side launch -- ablation
variants (no MFMA / no key copies / no softmax) and per-wave cycle counts of the phases (s_memtime)."""
import sys, time, ctypes
sys.path.insert(0, ".")
import torch
from medtok_amd import ops, _lib
dev = torch.device("cuda:0")
B, D, H, L = 4096, 768, 4, 512
g = torch.Generator(device=dev).manual_seed(77)
tok = torch.randint(1, L + 1, (B,), device=dev, generator=g)
n_nodes = torch.randint(1, 41, (B,), device=dev, generator=g)
text = torch.randn(B * L, D, device=dev, generator=g)
q = torch.randn(int(n_nodes.sum()) * H, D, device=dev, generator=g) * 0.05
images = ops.split_half(text, seg_len=tok, seg_rows=L)
starts = torch.cumsum(n_nodes, 0) - n_nodes
order = torch.argsort(tok, descending=True)
q_start, q_len, k_start, k_len = (starts * H)[order], (n_nodes * H)[order], (torch.arange(B, device=dev) * L)[order], tok[order]
k_same = torch.zeros_like(k_start)           # every code reads the SAME key rows: L2-resident keys
def run(v, ks=k_start):
    return ops.shared_kv_attention_split(q, q_start, q_len, images, ks, k_len, 160, 192 ** -0.5, split_out=True, variant=v)
def t(v, ks=k_start, reps=10):
    for _ in range(2): run(v, ks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): run(v, ks)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
# (round 4 also timed ablation builds -- no MFMA / no key copies / no softmax: profiles/r04_attention_pp_probe.txt; those template bits are gone)
names = {0: "dma <4,6,1,1> (round 3)", 1: "dma <8,3,2,2>", 2: "pp (shipped)"}
for rnd in range(2):
    for v, nm in names.items():
        print(f"round {rnd} {nm:45s} {t(v):7.3f} ms   keys L2-resident: {t(v, k_same):7.3f} ms", flush=True)
# timestamps
lib = _lib.load()
pairs = 3
nblk = pairs * ((B + 7) // 8 * 8)
dbg = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
lib.medtok_debug_set_attention_probe(ctypes.c_void_p(dbg.data_ptr()))
run(2 + 128); torch.cuda.synchronize()
dbg.zero_(); run(2 + 128); torch.cuda.synchronize()
lib.medtok_debug_set_attention_probe(ctypes.c_void_p(0))
d = dbg.view(nblk, 8, 8).cpu().double()
live = d[:, 0, 5] > 0
d = d[live]
print("blocks that ran:", int(live.sum()), " mean chunks/block:", float(d[:, 0, 5].mean()))
for grp in (0, 1):
    w = d[:, 4 * grp:4 * grp + 4, :]
    act = w[:, 0, 6] > 0
    wa = w[act]
    ch = wa[:, :, 5].sum()
    print(f"group {grp}: active in {int(act.sum())} blocks; cycles per chunk and wave: S {float(wa[:, :, 0].sum() / ch):7.0f}  X {float(wa[:, :, 1].sum() / ch):7.0f}  "
          f"V {float(wa[:, :, 2].sum() / ch):7.0f}  wait+barrier {float(wa[:, :, 3].sum() / ch):7.0f}  total {float(wa[:, :, 4].sum() / ch):7.0f}")
    if (~act).any():
        wi = w[~act]
        print(f"         idle in {int((~act).sum())} blocks: wait+barrier per chunk {float(wi[:, :, 3].sum() / wi[:, :, 5].sum()):7.0f}  total {float(wi[:, :, 4].sum() / wi[:, :, 5].sum()):7.0f}")
tot = d[:, 0, 4]
print("block lifetime cycles: mean %.0f  per chunk %.0f;  sum over blocks / 256 CUs = %.3f Mcycles" % (float(tot.mean()), float(tot.sum() / d[:, 0, 5].sum()), float(tot.sum()) / 256 / 1e6))
