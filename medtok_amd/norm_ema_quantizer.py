"""Norm-EMA vector quantiser of MedTok on MI355X.

Drop-in for the live classes of the reference's MedTok/norm_ema_quantizer.py:
`l2norm`, `ema_inplace`, `norm_ema_inplace`, `EmbeddingEMA`,
`NormEMAVectorQuantizer` -- same constructor arguments, return values and
state_dict keys (`cluster_size`, `embedding.weight`, `embedding.cluster_size`,
`embedding.embed_avg`, `embedding.initted`).

The forward pass (reference :166-218) runs as gfx950 kernels:
normalise -> argmin search (no N x K matrix) -> gather -> histogram + row-ordered
segmented sum (no one-hot, no second GEMM) -> all-reduce of [bins | embed_sum]
-> fused EMA codebook update.  The classes the reference never instantiates
(CVectorQuantiser, FeaturePool, lookup-only VectorQuantizer) are out of scope.
"""
from __future__ import annotations

import torch
import torch.distributed as distributed
import torch.nn as nn

from . import distributed as mdist
from . import ops


class _L2NormFunction(torch.autograd.Function):
    """F.normalize(t, p=2, dim=-1) with the forward done by the rownorm kernel."""

    @staticmethod
    def forward(ctx, t):
        flat = t.detach().reshape(-1, t.shape[-1]).float()
        that, sq = ops.rownorm(flat)          # sq = |that|^2 per row, in the kernel's summation order: what the search consumes
        ctx.save_for_backward(t, that)
        ctx.mark_non_differentiable(sq)
        return that.view(t.shape), sq

    @staticmethod
    def backward(ctx, g, _g_sq=None):
        t, that = ctx.saved_tensors
        that = that.view(t.shape)
        nrm = t.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        return (g - that * (that * g).sum(-1, keepdim=True)) / nrm


def l2norm(t):
    """x / max(||x||, 1e-12) along the last dim (reference :8-9)."""
    return _L2NormFunction.apply(t)[0]


def l2norm_with_sq(t):
    """(l2norm(t), the squared norms of its rows [numel / D]) from ONE pass of the rownorm kernel."""
    return _L2NormFunction.apply(t)


def ema_inplace(moving_avg, new, decay):
    """moving_avg <- decay * moving_avg + (1 - decay) * new (reference :11-12).
    Plain torch: the only call sites left outside the fused kernels are [K]-sized."""
    moving_avg.data.mul_(decay).add_(new, alpha=(1 - decay))


def norm_ema_inplace(moving_avg, new, decay):
    """EMA step followed by re-normalisation (reference :136-138)."""
    moving_avg.data.mul_(decay).add_(new, alpha=(1 - decay))
    moving_avg.data.copy_(l2norm(moving_avg.data))


class EmbeddingEMA(nn.Module):
    """Codebook state holder (reference :60-134)."""

    def __init__(self, num_tokens, codebook_dim, decay=0.99, eps=1e-5, kmeans_init=True, codebook_init_path=""):
        super().__init__()
        self.num_tokens = num_tokens
        self.codebook_dim = codebook_dim
        self.decay = decay
        self.eps = eps
        if codebook_init_path == "":
            if not kmeans_init:
                weight = torch.nn.functional.normalize(torch.randn(num_tokens, codebook_dim), p=2, dim=-1)
            else:
                weight = torch.zeros(num_tokens, codebook_dim)
            self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        else:
            weight = torch.load(codebook_init_path, map_location="cpu").clone()
            self.register_buffer("initted", torch.Tensor([True]))
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.cluster_size = nn.Parameter(torch.zeros(num_tokens), requires_grad=False)
        self.embed_avg = nn.Parameter(weight.clone(), requires_grad=False)
        self.update = True

    @torch.jit.ignore
    def init_embed_(self, data):
        if self.initted:
            return
        from .kmeans import kmeans
        embed, cluster_size = kmeans(data, self.num_tokens, 10, use_cosine_sim=True)
        self.weight.data.copy_(embed)
        self.cluster_size.data.copy_(cluster_size)
        self.initted.data.copy_(torch.Tensor([True]))

    @torch.jit.ignore
    def init_embed_split(self, data, split):
        """Two independent k-means runs over the column halves of `data`, means concatenated (reference :96-107; no caller there)."""
        if self.initted:
            return
        from .kmeans import kmeans
        embed1, cluster_size1 = kmeans(data[:, :split[0]].contiguous(), self.num_tokens, 10, use_cosine_sim=True)
        embed2, cluster_size2 = kmeans(data[:, split[0]:].contiguous(), self.num_tokens, 10, use_cosine_sim=True)
        self.weight.data.copy_(torch.cat([embed1, embed2], dim=-1))
        self.cluster_size.data.copy_((cluster_size1 + cluster_size2) / 2.)
        self.initted.data.copy_(torch.Tensor([True]))

    @torch.jit.ignore
    def init_embed_with_ind(self, data, inds):
        """(reference :109-116: `inds` is accepted and unused there too)"""
        self.init_embed_(data)

    def forward(self, embed_id):
        return torch.nn.functional.embedding(embed_id, self.weight)

    def cluster_size_ema_update(self, new_cluster_size):
        self.cluster_size.data.mul_(self.decay).add_(new_cluster_size, alpha=1 - self.decay)

    def embed_avg_ema_update(self, new_embed_avg):
        self.embed_avg.data.mul_(self.decay).add_(new_embed_avg, alpha=1 - self.decay)

    def weight_update(self, num_tokens):
        n = self.cluster_size.sum()
        smoothed = (self.cluster_size + self.eps) / (n + num_tokens * self.eps) * n
        self.weight.data.copy_(self.embed_avg / smoothed.unsqueeze(1))


class NormEMAVectorQuantizer(nn.Module):
    def __init__(self, n_embed, embedding_dim, beta, decay=0.99, eps=1e-5, statistic_code_usage=True,
                 kmeans_init=False, codebook_init_path=""):
        super().__init__()
        if embedding_dim % 4:
            raise ValueError("embedding_dim must be a multiple of 4 (float4 kernels)")
        self.codebook_dim = embedding_dim
        self.num_tokens = n_embed
        self.beta = beta
        self.decay = decay
        self.search_path = ops.PATH_AUTO
        self.embedding = EmbeddingEMA(self.num_tokens, self.codebook_dim, decay, eps, kmeans_init, codebook_init_path)
        self.statistic_code_usage = statistic_code_usage
        if statistic_code_usage:
            self.register_buffer("cluster_size", torch.zeros(n_embed))
        # like the reference (:155-159) the choice is made once, at construction time
        self.fused_head = True           # l2norm + search in one library call where no autograd graph is needed (forward)
        if distributed.is_available() and distributed.is_initialized():
            self.all_reduce_fn = mdist.all_reduce_sum        # torch.distributed.all_reduce (RCCL on GPUs), SUM, in place
        else:
            self.all_reduce_fn = nn.Identity()

    def reset_cluster_size(self, device):
        if self.statistic_code_usage:
            self.register_buffer("cluster_size", torch.zeros(self.num_tokens))
            self.cluster_size = self.cluster_size.to(device)

    def forward(self, z):
        b, c, h, w = z.shape
        # 'b c h w -> b h w c' (reference :169); for the usual [N, D, 1, 1] input this is a view
        z = z.permute(0, 2, 3, 1)
        need_grad = torch.is_grad_enabled() and z.requires_grad
        # (`if self.initted` in init_embed_ is a host read in the reference too: done once here)
        initted = bool(self.embedding.initted)
        if self.fused_head and not need_grad and initted and z.is_cuda and z.dtype == torch.float32:
            # l2norm and the nearest-code search in ONE library call: on the fp16-shortlist path the normalising pass also writes
            # the fp16 image the shortlist streams (same bits as the two calls of the other branch)
            E = self.embedding.weight.data                  # stored normalised, NOT re-normalised (:175-177)
            _, esq = ops.rownorm(E, normalize=False, want_xhat=False)
            zd, zsq, idx2, _ = ops.normalized_search(z.reshape(-1, self.codebook_dim), E, esq, 1, self.search_path)
            z_flat = zd
            n = zd.shape[0]
        else:
            z, zsq = l2norm_with_sq(z)                      # the rows' |z|^2 come out of the same pass (no second read of z)
            z_flat = z.reshape(-1, self.codebook_dim)
            zd = z_flat.detach()
            n = zd.shape[0]
            if not initted:
                self.embedding.init_embed_(zd)
            E = self.embedding.weight.data
            _, esq = ops.rownorm(E, normalize=False, want_xhat=False)
            idx2, _ = ops.topk_search(zd, zsq, E, esq, 1, self.search_path)
        encoding_indices = idx2.view(-1)
        # gather BEFORE the EMA update (:181)
        if need_grad:
            _, zq, _ = ops.soft_assign(zd, E, encoding_indices, None, hard=True, want_w=False, want_sqerr=False, raw=True)
        else:   # straight-through value and squared error in the same pass
            _, zq_ste, row_sqerr = ops.soft_assign(zd, E, encoding_indices, None, hard=True, want_w=False)

        k, d = self.num_tokens, self.codebook_dim
        if not self.training:
            bins = ops.code_histogram(encoding_indices, k)
            self.all_reduce_fn(bins)
            ops.ema_cluster_size_(self.cluster_size, bins, self.decay)
        if self.training and self.embedding.update:
            bins, embed_sum, stats = ops.ema_stats(zd, encoding_indices, k, fused=True)
            # one collective for both statistics (the reference issues two, :195,:203)
            self.all_reduce_fn(stats)
            ops.ema_apply_(E, self.cluster_size, bins, embed_sum, self.decay)

        if need_grad:
            loss = self.beta * torch.mean((zq - z_flat) ** 2)          # F.mse_loss(z_q.detach(), z)
            z_q = z_flat + (zq - z_flat).detach()
        else:
            loss = self.beta * ops.sum_scale(row_sqerr, (1.0 / (n * d)) if n else float("nan"))      # mean of nothing: nan, like F.mse_loss
            z_q = zq_ste
        z_q = z_q.view(b, h, w, c).permute(0, 3, 1, 2)
        return z_q, loss, encoding_indices
