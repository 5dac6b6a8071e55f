import sys; sys.path.insert(0, ".")
import torch, numpy as np
from medtok_amd import _lib, ops
n, k, d, topk = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
xh, xs = ops.rownorm(x); wh, ws_ = ops.rownorm(W)
lib = _lib.load()
nb = lib.medtok_search_workspace_bytes(n, k, d, topk, 2)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
idx = torch.empty((n, topk), dtype=torch.int64, device=dev); dist = torch.empty((n, topk), device=dev)
rc = lib.medtok_topk_search_f32(xh.data_ptr(), xs.data_ptr(), n, wh.data_ptr(), ws_.data_ptr(), k, d, topk, idx.data_ptr(), dist.data_ptr(), ws.data_ptr(), nb, 2, 0)
torch.cuda.synchronize()
al = lambda v: (v + 255) // 256 * 256
n_pad = (n + 255) // 256 * 256; k_pad = (k + 255) // 256 * 256; dp = (d + 63) // 64 * 64
off = al(n_pad * dp * 2) + al(k_pad * dp * 2)
en_max = ws[off:off + 4].view(torch.float32).item(); off += 256
off += al(k_pad * 4)
fb = ws[off:off + 4].view(torch.int32).item(); off += 256
off += al(n * 4)
row_tiles = n_pad // 256
want = 1 if row_tiles >= 4096 else (2 if row_tiles >= 1024 else (1024 + row_tiles - 1) // row_tiles)
code_tiles = k_pad // 256; want = min(want, code_tiles, 16); tps = (code_tiles + want - 1) // want; splits = (code_tiles + tps - 1) // tps
own = splits * 4
cnt = ws[off:off + n * own * 4].view(torch.int32).view(n, own)
print("rc", rc, "en_max", en_max, "fallback rows", fb, "splits", splits, "own", own)
print("cand count per owner: mean %.1f max %d  p99 %d" % (cnt.float().mean().item(), cnt.max().item(), int(torch.quantile(cnt.float().flatten()[:2000000], 0.99))))
print("per row total: mean %.1f max %d" % (cnt.sum(1).float().mean().item(), cnt.sum(1).max().item()))
print("rows with any owner > 64:", int((cnt > 64).any(1).sum()))
