"""Dev tool: how many rows of the full workload's shared searches the fp16 filter hands to the exact kernel."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from medtok_amd import ops
dev = torch.device("cuda:0")
w = bench.Full(4096, dev, 0, ops.PATH_AUTO)
vq = w.vq
with torch.no_grad():
    pt, pg = vq.cross_attn.pooled(w.text, w.mask, w.nodes, w.batch)
    what, wsq = vq._normalised_codebook()
    for name, x in (("pooled text", pt), ("pooled graph", pg), ("random", torch.randn_like(pt))):
        xhat, xsq = ops.rownorm(x.float().contiguous(), normalize=True)
        ops.SEARCH_STATS = {}
        idx, dist = ops.topk_search(xhat, xsq, what, wsq.contiguous(), 5)
        print(name, ops.SEARCH_STATS, "row norms", float(x.norm(dim=1).min()), float(x.norm(dim=1).max()),
              "cos between rows (mean |offdiag|)", float((xhat[:256] @ xhat[:256].t()).fill_diagonal_(0).abs().mean()),
              "top1-top5 gap mean", float((dist[:, 4] - dist[:, 0]).mean()), "d0 mean", float(dist[:, 0].mean()))
