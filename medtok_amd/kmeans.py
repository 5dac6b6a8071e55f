"""k-means codebook initialisation (reference norm_ema_quantizer.py:14-57,85-93).

Cosine k-means on l2-normalised samples: assignment is the same nearest-code
search kernel (argmax of the dot product == argmin of the distance for unit
vectors), the per-cluster sums are the EMA statistics kernel.  Parity with the
reference is statistical only: its initial means come from torch.randperm.
"""
from __future__ import annotations

import torch

from . import ops


def sample_vectors(samples, num):
    n = samples.shape[0]
    if n >= num:
        indices = torch.randperm(n, device=samples.device)[:num]
    else:
        indices = torch.randint(0, n, (num,), device=samples.device)
    return samples[indices]


def kmeans(samples, num_clusters, num_iters=10, use_cosine_sim=False):
    if not use_cosine_sim:
        raise NotImplementedError("the reference only ever calls kmeans(..., use_cosine_sim=True) (:90)")
    samples = samples.detach().float().contiguous()
    means = sample_vectors(samples, num_clusters).contiguous()
    _, ssq = ops.rownorm(samples, normalize=False, want_xhat=False)
    bins = None
    for _ in range(num_iters):
        _, msq = ops.rownorm(means, normalize=False, want_xhat=False)
        idx, _ = ops.topk_search(samples, ssq, means, msq, 1)
        bins, sums = ops.ema_stats(samples, idx.view(-1), num_clusters)
        zero = bins == 0
        new_means = sums / bins.masked_fill(zero, 1.0).unsqueeze(-1)
        new_means, _ = ops.rownorm(new_means.contiguous())
        means = torch.where(zero.unsqueeze(-1), means, new_means).contiguous()
    return means, bins
