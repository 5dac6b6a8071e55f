"""CPU PyTorch port of the reference's op SEQUENCE for the VQ path.

TEST INFRASTRUCTURE ONLY (see oracle/medtok_oracle.c header).  Where the C oracle fixes a
GPU-reproducible arithmetic order, this port keeps what the reference actually executes on a
CPU -- a materialised N x K distance matrix from one BLAS GEMM, torch.topk / argmin, a one-hot
matrix and a second GEMM for the EMA sums -- so that bench.py's cpu_baseline times the work
the reference's CPU PyTorch path performs (north_star: "next to the reference's CPU PyTorch
path timed on the host cores").  Pinned against the same golden vectors as the C oracle
(tests/test_oracle_golden.py::test_torch_port_*).

Reference lines restated: vector_quantization_soft_one_new.py:120-125,194-214 and
norm_ema_quantizer.py:166-218.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def distance_matrix(x, y):
    # (|x|^2 + |y|^2) - 2 x y^T, evaluated in that order (:121-123)
    return (x * x).sum(dim=1, keepdim=True) + (y * y).sum(dim=1) - 2 * (x @ y.t())


@torch.no_grad()
def soft_search(x_proj, W_region, k=5):
    """normalise both sides, dense distances, k smallest, softmax(-d), weighted code mix, STE value."""
    xn = F.normalize(x_proj, p=2, dim=-1)
    wn = F.normalize(W_region, p=2, dim=-1)
    d = distance_matrix(xn, wn)
    vals, idx = torch.topk(d, k=k, largest=False)
    w = torch.softmax(-vals, dim=1)
    zq = (w.unsqueeze(-1) * wn[idx]).sum(dim=1)
    sq = ((zq - x_proj) ** 2).mean()
    return dict(idx=idx, dist=vals, w=w, zq=x_proj + (zq - x_proj), xhat=xn, mse=sq)


@torch.no_grad()
def full_tokenize(h_text_proj, h_graph_proj, pooled_text, pooled_graph, W, k=5):
    """The four searches of VectorQuantizer.forward (cross-attention already applied):
    text / graph over their codebook thirds, shared text / graph over all of it."""
    region = W.shape[0] // 3
    a = soft_search(h_text_proj, W[:region], k)
    b = soft_search(h_graph_proj, W[-region:], k)
    c = soft_search(pooled_text, W, k)
    d = soft_search(pooled_graph, W, k)
    emb = torch.cat([a["zq"], b["zq"], c["zq"], d["zq"]], dim=-1)
    tokens = torch.stack([a["idx"], b["idx"], c["idx"], d["idx"]], dim=1)
    weights = torch.stack([a["w"], b["w"], c["w"], d["w"]], dim=1)
    return emb, tokens, weights


@torch.no_grad()
def norm_ema_forward(z, E, cluster_size, beta, decay, training):
    """argmin search + EMA update on z [N, D]; E and cluster_size are updated in place."""
    zn = F.normalize(z, p=2, dim=-1)
    d = distance_matrix(zn, E)
    idx = torch.argmin(d, dim=1)
    zq = E[idx].clone()
    onehot = F.one_hot(idx, E.shape[0]).to(zn.dtype)
    bins = onehot.sum(0)
    cluster_size.mul_(decay).add_(bins, alpha=1 - decay)
    if training:
        empty = bins == 0
        denom = bins.masked_fill(empty, 1.0)
        sums = zn.t() @ onehot
        new = F.normalize((sums / denom.unsqueeze(0)).t(), p=2, dim=-1)
        new = torch.where(empty[:, None], E, new)
        E.mul_(decay).add_(new, alpha=1 - decay)
        E.copy_(F.normalize(E, p=2, dim=-1))
    loss = beta * F.mse_loss(zq, zn)
    return zn + (zq - zn), loss, idx
