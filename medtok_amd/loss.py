"""Cross-modal losses of MedTok (drop-in for the reference's MedTok/loss.py:40-110).

Same function names, arguments and return structures (4-tuples of 0-dim fp32
tensors, differentiable).  All of it -- InfoNCE, the alignment term and the
orthogonality term, forward and backward -- runs on the library's gfx950 kernels
through autograd.Functions.  The GAN losses of loss.py:5-37 have no caller in the
reference and are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


class _InfoNCEFunction(torch.autograd.Function):
    """InfoNCE on the gfx950 kernels (medtok_info_nce_forward_f32 / _backward_f32): normalise, B x B logits,
    log-softmax against the diagonal and the mean in three launches; the backward is one launch and reads the
    softmax rows the forward kept.  fp32 throughout (the reference's autocast also runs normalize and
    cross_entropy in fp32; only its B x B matmul drops to bf16)."""

    @staticmethod
    def forward(ctx, q, k, temperature):
        qf, kf = q.detach().float().contiguous(), k.detach().float().contiguous()
        loss, prob, ws = ops.info_nce_forward(qf, kf, temperature)
        ctx.save_for_backward(qf, kf, prob, ws)
        ctx.temperature = temperature
        ctx.in_dtypes = (q.dtype, k.dtype)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        qf, kf, prob, ws = ctx.saved_tensors
        gq, gk = ops.info_nce_backward(qf, kf, prob, ws, g_loss.float().contiguous(), ctx.temperature)
        return gq.to(ctx.in_dtypes[0]), gk.to(ctx.in_dtypes[1]), None


def info_nce_loss(q, k, temperature=0.07):
    """InfoNCE with in-batch negatives (reference :40-56).

    The reference concatenates [positive | off-diagonal negatives] and takes
    cross-entropy against column 0; that is the cross-entropy of the full
    similarity matrix against its diagonal, which the HIP kernels evaluate
    without the masked copy and the concatenation.  Device tensors only: like
    the rest of the package there is no CPU path (MedTokLibraryError)."""
    if q.shape[-1] % 4:                                  # kernels stride float4; zero columns change nothing
        pad = 4 - q.shape[-1] % 4
        q, k = F.pad(q, (0, pad)), F.pad(k, (0, pad))
    return _InfoNCEFunction.apply(q, k, float(temperature))


def _pad4(*ts):
    """kernels stride float4; zero columns change neither dot products nor norms.  A width that is not a multiple of 4 costs one
    padded COPY of each operand per call (F.pad); widths that are (every configuration of the reference: e_dim 64 / 768) cost nothing."""
    d = ts[0].shape[-1]
    if d % 4 == 0:
        return ts
    return tuple(F.pad(t, (0, 4 - d % 4)) for t in ts)


class _AlignmentFunction(torch.autograd.Function):
    """mean_b <mu1[b], mu2[b]>: row dots (medtok_row_dot_f32) + the fixed-order fp64 mean; the gradients are the other
    operand scaled by g / B on the device (medtok_scale_by_device_scalar_f32)."""

    @staticmethod
    def forward(ctx, mu1, mu2):
        a, b = mu1.detach().float().contiguous(), mu2.detach().float().contiguous()
        ctx.save_for_backward(a, b)
        ctx.in_dtypes = (mu1.dtype, mu2.dtype)
        return ops.sum_scale(ops.row_dot(a, b), (1.0 / a.shape[0]) if a.shape[0] else float("nan"))     # mean of nothing: nan, like torch.mean

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.float().contiguous()
        c = (1.0 / a.shape[0]) if a.shape[0] else float("nan")
        return (ops.scale_by_device_scalar(b, g, None, c).to(ctx.in_dtypes[0]),
                ops.scale_by_device_scalar(a, g, None, c).to(ctx.in_dtypes[1]))


class _OrthogonalFunction(torch.autograd.Function):
    """|| z^T z* ||_F on the exact fp32 MFMA GEMM (medtok_small_gemm_f32) and medtok_frobenius_f32; backward
    G = g M / ||M||, dz = z* G^T, dz* = z G (two more small GEMMs)."""

    @staticmethod
    def forward(ctx, z, z_star):
        a, b = z.detach().float().contiguous(), z_star.detach().float().contiguous()
        m = ops.small_gemm(a, b, trans_a=True)                       # [d1, d2]
        nrm = ops.frobenius(m)
        ctx.save_for_backward(a, b, m, nrm)
        ctx.in_dtypes = (z.dtype, z_star.dtype)
        return nrm

    @staticmethod
    def backward(ctx, g):
        a, b, m, nrm = ctx.saved_tensors
        gm = ops.scale_by_device_scalar(m, g.float().contiguous(), nrm)         # dL/dM
        dz = ops.small_gemm(b, gm, trans_b=True)                     # [B, d1] = z* G^T
        dzs = ops.small_gemm(a, gm)                                  # [B, d2] = z G
        return dz.to(ctx.in_dtypes[0]), dzs.to(ctx.in_dtypes[1])


def alignment_loss(mu1, mu2):
    """Mean row-wise dot product (reference :59-64)."""
    mu1, mu2 = _pad4(mu1, mu2)
    return _AlignmentFunction.apply(mu1, mu2)


def orthogonal_loss(z, z_star):
    """Frobenius norm of z^T z* (reference :66-83)."""
    if z_star.shape[-1] % 4:                              # (M's row length is z*'s width; z's width may be anything)
        z_star = F.pad(z_star, (0, 4 - z_star.shape[-1] % 4))
    return _OrthogonalFunction.apply(z, z_star)


class _NormalizeFunction(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) on the rownorm kernel, backward on medtok_normalize_backward_f32."""

    @staticmethod
    def forward(ctx, x):
        xf = x.detach().float().contiguous()
        xhat, _ = ops.rownorm(xf)
        ctx.save_for_backward(xf, xhat)
        ctx.in_dtype = x.dtype
        return xhat

    @staticmethod
    def backward(ctx, g):
        xf, xhat = ctx.saved_tensors
        return ops.normalize_backward(g.float().contiguous(), xhat, xf).to(ctx.in_dtype)


def shared_loss(z1, z2, x1, x2, beta=0.1):
    """(nce(z1,z2), align(x1^,x2^), nce(z2,z1), align(x2^,x1^)) (reference :86-96);
    `beta` is applied by the caller (train_MedTok.py:221-224), as in the reference."""
    x1, x2 = _pad4(x1, x2)
    x1n = _NormalizeFunction.apply(x1)
    x2n = _NormalizeFunction.apply(x2)
    return info_nce_loss(z1, z2), alignment_loss(x1n, x2n), info_nce_loss(z2, z1), alignment_loss(x2n, x1n)


def specific_loss(z1, z1_aug, z2, z2_aug, z1_c, z2_c, lamb=0.1):
    """(nce([z1|z2_c],[z1_aug|z2_c]), orth(z1,z1_c), nce([z2|z1_c],[z2_aug|z1_c]), orth(z2,z2_c))
    (reference :98-110); `lamb` is applied by the caller (train_MedTok.py:233-235)."""
    a = info_nce_loss(torch.cat([z1, z2_c], dim=-1), torch.cat([z1_aug, z2_c], dim=-1))
    b = info_nce_loss(torch.cat([z2, z1_c], dim=-1), torch.cat([z2_aug, z1_c], dim=-1))
    return a, orthogonal_loss(z1, z1_c), b, orthogonal_loss(z2, z2_c)


def total_loss(quantized_result, shared_loss_beta=0.1, specific_loss_lamb=0.1):
    """Loss assembly of train_MedTok.py:213-238 from a VectorQuantizer.forward() dict.
    Returns (loss, parts) with parts holding every scalar the reference logs."""
    r = quantized_result
    codebook = (r["shared_embed_loss"][0] + r["shared_embed_loss"][1]
                + r["text_specific_loss"][0] + r["text_specific_loss"][1]
                + r["graph_specific_loss"][0] + r["graph_specific_loss"][1])
    s11, s12, s21, s22 = shared_loss(r["shared_text_embedding"], r["shared_graph_embedding"],
                                     r["text_feature"], r["graph_feature"])
    shared_all = (s11 - shared_loss_beta * s12) + (s21 - shared_loss_beta * s22)
    p11, p12, p21, p22 = specific_loss(z1=r["specific_embedding_text"], z1_aug=r["specific_embedding_text_aug"],
                                       z2=r["specific_embedding_graph"], z2_aug=r["specific_embedding_graph_aug"],
                                       z1_c=r["shared_text_embedding"], z2_c=r["shared_graph_embedding"])
    specific_all = (p11 + specific_loss_lamb * p12) + (p21 + specific_loss_lamb * p22)
    loss = codebook + shared_all + specific_all
    parts = dict(codebook_loss=codebook, shared_loss=(s11, s12, s21, s22), specific_loss=(p11, p12, p21, p22),
                 shared_loss_all=shared_all, specific_loss_all=specific_all)
    return loss, parts
