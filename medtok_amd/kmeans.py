"""k-means codebook initialisation (reference norm_ema_quantizer.py:14-57,85-93).

Cosine k-means on l2-normalised samples: assignment is the same nearest-code
search kernel (argmax of the dot product == argmin of the distance for unit
vectors), the per-cluster sums are the EMA statistics kernel.  The reference's
only randomness is the choice of the initial means (torch.randperm); with the
same `init_means` the iteration is deterministic and is checked against the
reference's own run (fixture F12, tests/test_gpu_modules.py): bucket assignments
identical wherever the fp64 best-vs-second gap exceeds round-off, final means
within 1e-5.
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


def kmeans(samples, num_clusters, num_iters=10, use_cosine_sim=False, init_means=None, trace=None):
    """(means [K, D], bins [K]) like the reference's kmeans (:24-57).  `init_means` replaces the random choice of the
    initial means; `trace` (a list) receives (buckets, means_used) of every iteration (tests)."""
    if not use_cosine_sim:
        raise NotImplementedError("the reference only ever calls kmeans(..., use_cosine_sim=True) (:90)")
    samples = samples.detach().float().contiguous()
    means = (sample_vectors(samples, num_clusters) if init_means is None else init_means.detach().float()).contiguous()
    _, ssq = ops.rownorm(samples, normalize=False, want_xhat=False)
    bins = None
    for _ in range(num_iters):
        _, msq = ops.rownorm(means, normalize=False, want_xhat=False)
        idx, _ = ops.topk_search(samples, ssq, means, msq, 1)
        if trace is not None:
            trace.append((idx.view(-1).clone(), means.clone()))
        bins, sums = ops.ema_stats(samples, idx.view(-1), num_clusters)
        zero = bins == 0
        new_means = sums / bins.masked_fill(zero, 1.0).unsqueeze(-1)
        new_means, _ = ops.rownorm(new_means.contiguous())
        means = torch.where(zero.unsqueeze(-1), means, new_means).contiguous()
    return means, bins
