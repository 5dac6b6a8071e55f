"""numpy front-end of the CPU oracle (oracle/medtok_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under medtok_amd/ imports this module.

Each function mirrors one C-ABI entry point of the product library
(include/medtok_vq.h) and cites the reference lines it restates; the
arithmetic order is documented in medtok_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libmedtok_oracle.so"
_lib = None

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (seconds)."""
    src = _HERE / "medtok_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "libmedtok_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        _lib = C.CDLL(str(_SO))
        _lib.oracle_usage_update.restype = C.c_int64
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def rownorm(x, normalize=True):
    """F.normalize(x, p=2, dim=-1) and |xhat|^2 (norm_ema_quantizer.py:8-9)."""
    x, xp = _f(x)
    n, d = x.shape
    xhat = np.empty_like(x)
    sqn = np.empty(n, np.float32)
    rc = lib().oracle_rownorm_f32(xp, C.c_int64(n), d, int(bool(normalize)),
                                  xhat.ctypes.data_as(_f32p), sqn.ctypes.data_as(_f32p))
    assert rc == 0
    return xhat, sqn


def topk_search(xhat, xsq, what, wsq, topk):
    """topk smallest get_distance() entries per row
    (vector_quantization_soft_one_new.py:120-125,157,203; k=1 is
    norm_ema_quantizer.py:175-179)."""
    xhat, xp = _f(xhat); xsq, xsp = _f(xsq); what, wp = _f(what); wsq, wsp = _f(wsq)
    n, d = xhat.shape
    k = what.shape[0]
    idx = np.empty((n, topk), np.int64)
    dist = np.empty((n, topk), np.float32)
    rc = lib().oracle_topk_search_f32(xp, xsp, C.c_int64(n), wp, wsp, C.c_int64(k), d, topk,
                                      idx.ctypes.data_as(_i64p), dist.ctypes.data_as(_f32p))
    assert rc == 0, rc
    return idx, dist


def distance(xhat, xsq, what, wsq):
    xhat, xp = _f(xhat); xsq, xsp = _f(xsq); what, wp = _f(what); wsq, wsp = _f(wsq)
    n, d = xhat.shape
    k = what.shape[0]
    out = np.empty((n, k), np.float32)
    rc = lib().oracle_distance_f32(xp, xsp, C.c_int64(n), wp, wsp, C.c_int64(k), d,
                                   out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def scores(xhat, what):
    """Canonical chain dot products [n, K] (small cases)."""
    xhat, xp = _f(xhat); what, wp = _f(what)
    n, d = xhat.shape
    k = what.shape[0]
    out = np.empty((n, k), np.float32)
    rc = lib().oracle_scores_f32(xp, C.c_int64(n), wp, C.c_int64(k), d, out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def soft_assign(xref, what, idx, dist, hard=False, raw=False):
    """softmax(-d) weights, weighted code mix, STE value, row squared error
    (vector_quantization_soft_one_new.py:158-182,204-214; hard=True is
    norm_ema_quantizer.py:181,212-214)."""
    xref, xp = _f(xref); what, wp = _f(what); idx, ip = _i(idx); dist, dp = _f(dist)
    n, d = xref.shape
    topk = 1 if idx.ndim == 1 else idx.shape[1]
    w = np.empty((n, topk), np.float32)
    zq = np.empty((n, d), np.float32)
    se = np.empty(n, np.float32)
    rc = lib().oracle_soft_assign_f32(xp, wp, ip, dp, C.c_int64(n), d, topk, int(bool(hard)) | (2 if raw else 0),
                                      w.ctypes.data_as(_f32p), zq.ctypes.data_as(_f32p),
                                      se.ctypes.data_as(_f32p))
    assert rc == 0
    return w, zq, se


def _fopt(a):
    if a is None:
        return None, None
    return _f(a)


def soft_vq_backward(x, xhat, what, idx, w, g_zq=None, g_xhat=None, g_out=None, g_vq=0.0, g_commit=0.0,
                     vq_scale=0.0, commit_scale=0.0):
    """(gx [n,d], g_code [n*k,d]): the gradient autograd sends through the k selected columns of the graph
    built at vector_quantization_soft_one_new.py:157-182,203-214 (tolerance checker, double accumulation)."""
    x, xp = _f(x); xhat, hp = _f(xhat); what, wp = _f(what); idx, ip = _i(idx); w, wtp = _f(w)
    g_zq, gzp = _fopt(g_zq); g_xhat, ghp = _fopt(g_xhat); g_out, gop = _fopt(g_out)
    n, d = x.shape
    topk = idx.shape[1]
    gx = np.empty((n, d), np.float32)
    gc = np.empty((n * topk, d), np.float32)
    rc = lib().oracle_soft_vq_backward_f32(xp, hp, wp, ip, wtp, C.c_int64(n), d, topk, gzp, ghp, gop,
                                           C.c_float(g_vq), C.c_float(g_commit), C.c_float(vq_scale), C.c_float(commit_scale),
                                           gx.ctypes.data_as(_f32p), gc.ctypes.data_as(_f32p))
    assert rc == 0
    return gx, gc


def normalize_backward(g, vhat, v):
    g, gp = _f(g); vhat, hp = _f(vhat); v, vp = _f(v)
    out = np.empty_like(v)
    rc = lib().oracle_normalize_backward_f32(gp, hp, vp, C.c_int64(v.shape[0]), v.shape[1], out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def info_nce(q, k, temperature=0.07, g_loss=1.0):
    """(loss, gq, gk) of loss.py:40-56 and its gradient."""
    q, qp = _f(q); k, kp = _f(k)
    b, d = q.shape
    loss = np.empty(1, np.float32)
    gq, gk = np.empty_like(q), np.empty_like(k)
    rc = lib().oracle_info_nce_f32(qp, kp, C.c_int64(b), d, C.c_float(temperature), C.c_float(g_loss),
                                   loss.ctypes.data_as(_f32p), gq.ctypes.data_as(_f32p), gk.ctypes.data_as(_f32p))
    assert rc == 0
    return float(loss[0]), gq, gk


def shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale):
    """Ragged attention core with raw rows as keys and values (folded nn.MultiheadAttention, :30,45)."""
    q, qp = _f(q); kv, kp = _f(kv)
    qs, qsp = _i(q_start); ql, qlp = _i(q_len); ks, ksp = _i(kv_start); kl, klp = _i(kv_len)
    out = np.full_like(q, np.nan)
    rc = lib().oracle_shared_kv_attention_f32(qp, qsp, qlp, kp, ksp, klp, C.c_int64(len(qs)), q.shape[1], C.c_float(scale),
                                              out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, scale, dropout_p=0.0, seed=0, d_out=None):
    """Training-mode attention core: (out, lse) or, with d_out, (out, lse, dq, dkv) -- dropout by the stateless hash mask
    (vector_quantization_soft_one_new.py:21,30,45 after folding the projections; autograd of the same)."""
    q, qp = _f(q); kv, kp = _f(kv)
    qs, qsp = _i(q_start); ql, qlp = _i(q_len); ks, ksp = _i(kv_start); kl, klp = _i(kv_len)
    out = np.zeros_like(q); lse = np.full(q.shape[0], -np.inf, np.float32)
    dq = np.zeros_like(q) if d_out is not None else None
    dkv = np.zeros_like(kv) if d_out is not None else None
    dop = _f(d_out)[1] if d_out is not None else None
    rc = lib().oracle_shared_kv_attention_train_f32(qp, qsp, qlp, kp, ksp, klp, C.c_int64(len(qs)), q.shape[1], C.c_float(scale),
                                                    C.c_float(dropout_p), C.c_uint32(seed), out.ctypes.data_as(_f32p), lse.ctypes.data_as(_f32p),
                                                    dop, None if dq is None else dq.ctypes.data_as(_f32p),
                                                    None if dkv is None else dkv.ctypes.data_as(_f32p), C.c_int64(kv.shape[0]))
    assert rc == 0
    return (out, lse) if d_out is None else (out, lse, dq, dkv)


def row_dot(a, b):
    """<a[r], b[r]> per row (the summand of alignment_loss, loss.py:63)."""
    a, ap = _f(a); b, bp = _f(b)
    out = np.empty(a.shape[0], np.float32)
    assert lib().oracle_row_dot_f32(ap, bp, C.c_int64(a.shape[0]), a.shape[1], out.ctypes.data_as(_f32p)) == 0
    return out


def small_gemm(A, B, trans_a=False, trans_b=False):
    """op(A) @ op(B), one fp32 fmaf chain over k per entry (torch.mm of loss.py:79 and its backward)."""
    A, Ap = _f(A); B, Bp = _f(B)
    (m, k), (sam, sak) = ((A.shape[1], A.shape[0]), (1, A.shape[1])) if trans_a else ((A.shape[0], A.shape[1]), (A.shape[1], 1))
    (k2, n), (sbk, sbn) = ((B.shape[1], B.shape[0]), (1, B.shape[1])) if trans_b else ((B.shape[0], B.shape[1]), (B.shape[1], 1))
    assert k == k2
    out = np.empty((m, n), np.float32)
    assert lib().oracle_small_gemm_f32(Ap, C.c_int64(sam), C.c_int64(sak), Bp, C.c_int64(sbk), C.c_int64(sbn), m, n, k,
                                       out.ctypes.data_as(_f32p)) == 0
    return out


def frobenius(x):
    """||x||_F (torch.norm(p='fro'), loss.py:82)."""
    x, xp = _f(x)
    out = np.empty(1, np.float32)
    assert lib().oracle_frobenius_f32(xp, C.c_int64(x.shape[0]), x.shape[1], out.ctypes.data_as(_f32p)) == 0
    return out[0]


def residual_layernorm(a, b, gamma, beta, eps):
    """LayerNorm(a + b) * gamma + beta per row (the tail of CrossAttentionLayer, vector_quantization_soft_one_new.py:47-50)."""
    a, ap = _f(a); b, bp = _f(b); gamma, gp = _f(gamma); beta, bep = _f(beta)
    y = np.empty_like(a)
    assert lib().oracle_residual_layernorm_f32(ap, bp, gp, bep, C.c_int64(a.shape[0]), a.shape[1], C.c_float(eps), y.ctypes.data_as(_f32p)) == 0
    return y


def segment_mean(x, seg_start, seg_len):
    """Mean of rows [seg_start[b], +seg_len[b]) of x per segment (`.mean(dim=0)` over a code's nodes, :140-141)."""
    x, xp = _f(x)
    ss = np.ascontiguousarray(seg_start, np.int64); sl = np.ascontiguousarray(seg_len, np.int64)
    out = np.empty((len(ss), x.shape[1]), np.float32)
    i64p = C.POINTER(C.c_int64)
    assert lib().oracle_segment_mean_f32(xp, ss.ctypes.data_as(i64p), sl.ctypes.data_as(i64p), C.c_int64(len(ss)), x.shape[1],
                                         out.ctypes.data_as(_f32p)) == 0
    return out


def ema_stats(zhat, idx, k_codes):
    """bins and embed_sum ([K,D]) of norm_ema_quantizer.py:194,202."""
    zhat, zp = _f(zhat); idx, ip = _i(idx)
    n, d = zhat.shape
    bins = np.empty(k_codes, np.float32)
    es = np.empty((k_codes, d), np.float32)
    rc = lib().oracle_ema_stats_f32(zp, ip, C.c_int64(n), d, C.c_int64(k_codes),
                                    bins.ctypes.data_as(_f32p), es.ctypes.data_as(_f32p))
    assert rc == 0
    return bins, es


def ema_apply(E, cluster_size, bins, embed_sum, decay):
    """In-place codebook / cluster_size update of norm_ema_quantizer.py:197-210."""
    assert E.dtype == np.float32 and E.flags.c_contiguous
    assert cluster_size.dtype == np.float32 and cluster_size.flags.c_contiguous
    bins, bp = _f(bins); embed_sum, ep = _f(embed_sum)
    k, d = E.shape
    rc = lib().oracle_ema_apply_f32(E.ctypes.data_as(_f32p), cluster_size.ctypes.data_as(_f32p),
                                    bp, ep, C.c_int64(k), d, C.c_float(decay),
                                    C.c_float(1 - decay))
    assert rc == 0


def ema_cluster_size(cluster_size, bins, decay):
    assert cluster_size.dtype == np.float32 and cluster_size.flags.c_contiguous
    bins, bp = _f(bins)
    rc = lib().oracle_ema_cluster_size_f32(cluster_size.ctypes.data_as(_f32p), bp,
                                           C.c_int64(cluster_size.shape[0]), C.c_float(decay),
                                           C.c_float(1 - decay))
    assert rc == 0


def usage_update(window, ids, n_codes):
    """codebook_usage() window slide + distinct count
    (vector_quantization_soft_one_new.py:219-236). Returns used/n_codes."""
    assert window.dtype == np.float32 and window.flags.c_contiguous
    ids, ip = _i(np.asarray(ids).reshape(-1))
    cnt = lib().oracle_usage_update(window.ctypes.data_as(_f32p), C.c_int64(window.shape[0]),
                                    ip, C.c_int64(ids.shape[0]), C.c_int64(n_codes))
    assert cnt >= 0
    return cnt / n_codes


# ---- composite restatements of the reference's module-level functions ------

def specific_search(x, W_region, topk, x_is_projected=True):
    """VectorQuantizer.specific_embedding without the Linear and the losses
    (vector_quantization_soft_one_new.py:194-205,214)."""
    xhat, xsq = rownorm(x, True)
    what, wsq = rownorm(W_region, True)
    idx, dist = topk_search(xhat, xsq, what, wsq, topk)
    w, zq, se = soft_assign(x, what, idx, dist)
    return dict(idx=idx, dist=dist, w=w, zq=zq, xhat=xhat, row_sqerr=se)


def norm_ema_forward(z, E, cluster_size, beta, decay, training):
    """NormEMAVectorQuantizer.forward on z [N,D] (norm_ema_quantizer.py:166-218).
    Mutates E / cluster_size like the reference does."""
    zhat, zsq = rownorm(z, True)
    _, esq = rownorm(E, False)
    idx, dist = topk_search(zhat, zsq, E, esq, 1)
    _, zq, se = soft_assign(zhat, E, idx[:, 0], dist, hard=True)
    bins, es = ema_stats(zhat, idx[:, 0], E.shape[0])
    if training:
        ema_apply(E, cluster_size, bins, es, decay)
    else:
        ema_cluster_size(cluster_size, bins, decay)
    loss = np.float32(beta) * np.float32(se.astype(np.float64).sum() / zhat.size)
    return zq, loss, idx[:, 0]
