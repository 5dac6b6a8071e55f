"""k-means codebook initialisation (reference norm_ema_quantizer.py:14-57,85-93).

Both branches of the reference run on the same two kernels: the assignment is the nearest-code search (Euclidean branch :36-39:
argmax of -|s - m|^2 is the search's argmin of |s|^2 + |m|^2 - 2 s.m as it stands; cosine branch :34 on l2-normalised samples
and means: argmax of the dot product; the search is handed ZERO squared norms on both sides, so its distance is exactly
-2 s.m whatever the vectors' lengths -- EmbeddingEMA.init_embed_split feeds column halves of unit rows, whose sampled initial
means are not unit vectors -- on the exact fp32 kernel: the fp16 shortlist's error bound needs the true norms), the per-cluster
sums are the EMA statistics kernel; the cosine branch re-normalises the means (:50-51).  The reference's
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
    samples = samples.detach().float().contiguous()
    means = (sample_vectors(samples, num_clusters) if init_means is None else init_means.detach().float()).contiguous()
    if use_cosine_sim:
        ssq = torch.zeros(samples.shape[0], dtype=torch.float32, device=samples.device)
        zero_msq = torch.zeros(num_clusters, dtype=torch.float32, device=samples.device)
    else:
        _, ssq = ops.rownorm(samples, normalize=False, want_xhat=False)
    bins = None
    for _ in range(num_iters):
        if use_cosine_sim:          # :34 `samples @ means.t()` -> max: (0 + 0) - 2 s.m -> min, ties to the lowest index
            idx, _ = ops.topk_search(samples, ssq, means, zero_msq, 1, ops.PATH_F32_MFMA)
        else:
            _, msq = ops.rownorm(means, normalize=False, want_xhat=False)
            idx, _ = ops.topk_search(samples, ssq, means, msq, 1)
        if trace is not None:
            trace.append((idx.view(-1).clone(), means.clone()))
        bins, sums = ops.ema_stats(samples, idx.view(-1), num_clusters)
        zero = bins == 0
        new_means = sums / bins.masked_fill(zero, 1.0).unsqueeze(-1)
        if use_cosine_sim:
            new_means, _ = ops.rownorm(new_means.contiguous())
        means = torch.where(zero.unsqueeze(-1), means, new_means).contiguous()
    return means, bins
