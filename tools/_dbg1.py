import sys, numpy as np, torch
sys.path.insert(0, ".")
from medtok_amd import ops
dev = torch.device("cuda:0")
for (n,k,d,topk) in [(257,1025,36,3)]:
    rng = np.random.default_rng(n + k + d)
    x = torch.from_numpy(rng.standard_normal((n, d), dtype=np.float32)).to(dev); W = torch.from_numpy(rng.standard_normal((k, d), dtype=np.float32)).to(dev)
    xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
    torch.cuda.synchronize(); print("norm ok", flush=True)
    i1, d1 = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA); torch.cuda.synchronize(); print("exact ok", flush=True)
    i2, d2 = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F16_FILTER); torch.cuda.synchronize(); print("filter ok", torch.equal(i1,i2), torch.equal(d1,d2), flush=True)
