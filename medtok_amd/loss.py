"""Cross-modal losses of MedTok (drop-in for the reference's MedTok/loss.py:40-110).

Same function names, arguments and return structures (4-tuples of 0-dim fp32
tensors, differentiable).  InfoNCE -- four calls per training step -- runs on the
library's gfx950 kernels (forward and backward); the alignment and orthogonality
terms are one reduction / one D x D product each and stay torch code.  The GAN
losses of loss.py:5-37 have no caller in the reference and are out of scope.
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


def alignment_loss(mu1, mu2):
    """Mean row-wise dot product (reference :59-64)."""
    return (mu1 * mu2).sum(dim=1).mean()


def orthogonal_loss(z, z_star):
    """Frobenius norm of z^T z* (reference :66-83)."""
    return torch.linalg.matrix_norm(z.t() @ z_star, ord="fro")


def shared_loss(z1, z2, x1, x2, beta=0.1):
    """(nce(z1,z2), align(x1^,x2^), nce(z2,z1), align(x2^,x1^)) (reference :86-96);
    `beta` is applied by the caller (train_MedTok.py:221-224), as in the reference."""
    x1n = F.normalize(x1, p=2, dim=-1)
    x2n = F.normalize(x2, p=2, dim=-1)
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
