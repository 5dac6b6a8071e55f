"""Tensor-level wrappers over the C ABI (include/medtok_vq.h).

torch is used for device memory and the current HIP stream only; every
computation below runs in the hand-written gfx950 kernels.  CPU tensors are
rejected: there is no fallback path.
"""
from __future__ import annotations

import contextlib

import torch

from . import _lib
from ._lib import PATH_AUTO, PATH_F16_FILTER, PATH_F32_MFMA, MAX_TOPK, plan_path  # noqa: F401


# Profiling hook (bench.py): when set to a list, every search CALL (all of its kernels) is bracketed
# by HIP events on its launch stream and (start, stop, algorithmic_flops) is appended.  None = no overhead.
SEARCH_TIMER = None


DKV_SOURCES_MAX = _lib.DKV_SOURCES_MAX
PROFILE_KINDS = ("filter_f16_kernel", "search_f32_kernel", "shared_kv_attention_kernel", "shared_kv_attention_backward_kernels", "split_gemm_kernel")


def profile_begin(kinds=None) -> None:
    """Start the library's own per-kernel timing (HIP events around each matrix-pipe launch).  kinds: names out of PROFILE_KINDS to
    bracket (None: all) -- a training step has ~180 library launches, and the event pairs of all of them cost it ~0.5 ms."""
    if kinds is None:
        _lib.check(_lib.load().medtok_profile_begin(), "medtok_profile_begin")
        return
    mask = 0
    for k in kinds:
        mask |= 1 << PROFILE_KINDS.index(k)
    _lib.check(_lib.load().medtok_profile_begin_kinds(mask), "medtok_profile_begin_kinds")


def profile_end() -> dict:
    """Stop it; returns {kernel: dict(ms, flops, launches)} (synchronises on the recorded events)."""
    import ctypes as C
    names = PROFILE_KINDS
    n = len(names)
    ms, fl, ln = (C.c_double * n)(), (C.c_double * n)(), (C.c_int * n)()
    _lib.check(_lib.load().medtok_profile_end(ms, fl, ln), "medtok_profile_end")
    return {names[i]: dict(ms=ms[i], flops=fl[i], launches=ln[i]) for i in range(n)}


MedTokLibraryError = _lib.MedTokLibraryError


class ClockProbe:
    """Shader clock of a timed region (bench.py): `with ClockProbe(device, max_seconds) as p: ...timed work on OTHER streams...`;
    afterwards p.result() = dict(ghz_mean, ghz_min, ghz_max, per_xcd).  One idle wavefront per XCD on a stream of its own; it ends
    when the context exits (a pinned host flag) or after max_seconds, whichever is first."""

    def __init__(self, device, max_seconds: float = 30.0):
        self.device = torch.device(device)
        self.max_ticks = int(max_seconds * 1e8)
        self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.out = torch.zeros((8, 4), dtype=torch.int64, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)

    def __enter__(self):
        with _on(self.device):
            _lib.check(_lib.load().medtok_debug_clock_probe(self.flag.data_ptr(), self.max_ticks, self.out.data_ptr(), self.stream.cuda_stream),
                       "medtok_debug_clock_probe")
        return self

    def __exit__(self, *exc):
        self.flag[0] = 1
        self.stream.synchronize()
        return False

    def result(self) -> dict:
        o = self.out.cpu().tolist()
        ghz = [c / (r * 10.0) for c, r, _, _ in o if r > 0]          # cycles per 10 ns tick -> GHz
        if not ghz:
            return {}
        return dict(ghz_mean=sum(ghz) / len(ghz), ghz_min=min(ghz), ghz_max=max(ghz), seconds=max(r for _, r, _, _ in o) / 1e8,
                    per_xcd={int(x): c / (r * 10.0) for c, r, x, _ in o if r > 0},
                    how="s_memtime cycles / s_memrealtime (100 MHz) ticks of one idle wavefront per XCD while the probe ran")


ATTENTION_MAX_WIDTH = 1024                 # widest cross-attention: inference only above ATTENTION_MAX_TRAIN_WIDTH
ATTENTION_MAX_TRAIN_WIDTH = 768


def attention_width(d: int) -> int:
    """Width the ragged attention kernels run a D-wide problem at: D itself for 64 and multiples of 128 up to 768, and 1024 (BERT-large
    features; inference only: the training kernels' 32-key chunk of fp32 rows does not fit the LDS there), else the next such width
    (the caller appends zero columns).  Wider than 1024 is refused."""
    if d <= 64:
        return 64
    w = (d + 127) // 128 * 128
    if w > 768:
        w = 1024
    if d > ATTENTION_MAX_WIDTH:
        raise _lib.MedTokLibraryError(f"cross-attention width D = {d} is not supported by the gfx950 attention kernels: they take D <= 1024 "
                                      f"(64, a multiple of 128 up to 768, or 1024 natively; anything else zero-padded to the next such width)")
    return w


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_SAME_DEVICE = contextlib.nullcontext()


def _stream(t: torch.Tensor) -> int:
    """the raw handle of torch's current stream on t's device (the C accessor where this torch has it: every wrapper below asks once per
    launch, and the Python-level Stream object costs more than some of the launches)"""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _stream_of(dev: torch.device) -> int:
    """raw handle of torch's current stream on `dev`"""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


def _on(dev: torch.device):
    """`with torch.cuda.device(dev)` only where it changes something: a tensor on another device than the current one"""
    return _SAME_DEVICE if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MedTokLibraryError(
            f"{name}: expected a tensor on an MI355X (cuda/HIP) device, got "
            f"{getattr(t, 'device', type(t))}; medtok_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)


def pad_dim(d: int) -> int:
    """Kernels need D % 4 == 0; zero columns change neither norms nor dots."""
    return (d + 3) // 4 * 4


def rownorm(x: torch.Tensor, normalize: bool = True, want_xhat: bool = True):
    """(xhat, |xhat|^2) of F.normalize(x, dim=-1). x: [n, d] fp32."""
    x = _dev(x, "x")
    n, d = x.shape
    lib = _lib.load()
    sqn = torch.empty(n, dtype=torch.float32, device=x.device)
    if normalize:
        xhat = torch.empty_like(x)
    else:
        xhat = x if want_xhat else None
    with _on(x.device):
        _lib.check(lib.medtok_rownorm_f32(x.data_ptr(), n, d, int(normalize), _ptr(xhat), sqn.data_ptr(),
                                          _stream(x)), "medtok_rownorm_f32")
    return xhat, sqn


# Test hook: when set to a dict, topk_search records how many rows the fp16 filter handed to the exact kernel ("fallback_rows";
# one host sync per search -- never set in product code).
SEARCH_STATS = None


def topk_search(xhat, xsq, what, wsq, topk: int, path: int = PATH_AUTO):
    """idx [n, topk] int64, dist [n, topk] fp32: the topk nearest codes per row."""
    xhat, xsq = _dev(xhat, "xhat"), _dev(xsq, "xsq")
    what, wsq = _dev(what, "what"), _dev(wsq, "wsq")
    n, d = xhat.shape
    k = what.shape[0]
    if what.shape[1] != d or xsq.shape[0] != n or wsq.shape[0] != k:
        raise ValueError("topk_search: shape mismatch")
    lib = _lib.load()
    idx = torch.empty((n, topk), dtype=torch.int64, device=xhat.device)
    dist = torch.empty((n, topk), dtype=torch.float32, device=xhat.device)
    nb = lib.medtok_search_workspace_bytes(n, k, d, topk, path)
    ws = _ws(nb, xhat)
    timer = SEARCH_TIMER
    with _on(xhat.device):
        if timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.medtok_topk_search_f32(xhat.data_ptr(), xsq.data_ptr(), n, what.data_ptr(), wsq.data_ptr(),
                                              k, d, topk, idx.data_ptr(), dist.data_ptr(), ws.data_ptr(),
                                              ws.numel(), path, _stream(xhat)), "medtok_topk_search_f32")
        if timer is not None:
            e1.record()
            timer.append((e0, e1, 2.0 * n * k * d))
        if SEARCH_STATS is not None:
            off = lib.medtok_debug_filter_fallback_count_offset(n, k, d, topk, path)
            if off != 2 ** 64 - 1:
                SEARCH_STATS["fallback_rows"] = int(ws[off: off + 4].view(torch.int32).item())
                SEARCH_STATS["rows"] = n
    return idx, dist


def merge_topk_lists(dist_parts, idx_parts):
    """Exact top-k of the union of per-shard lists: dist_parts [P, n, k] fp32, idx_parts [P, n, k] int64 (global ids)."""
    dist_parts, idx_parts = _dev(dist_parts, "dist_parts"), _dev(idx_parts, "idx_parts", torch.int64)
    parts, n, k = dist_parts.shape
    lib = _lib.load()
    idx = torch.empty((n, k), dtype=torch.int64, device=dist_parts.device)
    dist = torch.empty((n, k), dtype=torch.float32, device=dist_parts.device)
    with _on(dist_parts.device):
        _lib.check(lib.medtok_merge_topk_lists_f32(dist_parts.data_ptr(), idx_parts.data_ptr(), n, parts, k, idx.data_ptr(),
                                                   dist.data_ptr(), _stream(dist_parts)), "medtok_merge_topk_lists_f32")
    return idx, dist


def debug_filter_scores(xhat, xsq, what, wsq):
    """Test hook: approximate scores s~ [n, K] of the fp16 filter."""
    xhat, xsq, what, wsq = _dev(xhat, "xhat"), _dev(xsq, "xsq"), _dev(what, "what"), _dev(wsq, "wsq")
    n, d = xhat.shape
    k = what.shape[0]
    lib = _lib.load()
    out = torch.empty((n, k), dtype=torch.float32, device=xhat.device)
    ws = _ws(lib.medtok_debug_filter_scores_workspace_bytes(n, k, d), xhat)
    with _on(xhat.device):
        _lib.check(lib.medtok_debug_filter_scores_f32(xhat.data_ptr(), xsq.data_ptr(), n, what.data_ptr(), wsq.data_ptr(), k, d,
                                                      out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(xhat)),
                   "medtok_debug_filter_scores_f32")
    return out


def _zq_out(out, n, d, like):
    """Output buffer for zq: a fresh [n, d] tensor, or the caller's column block of a wider row-major tensor."""
    if out is None:
        return torch.empty((n, d), dtype=torch.float32, device=like.device), d
    if not (out.is_cuda and out.dtype == torch.float32 and out.shape == (n, d) and out.stride(1) == 1 and out.stride(0) % 4 == 0
            and out.data_ptr() % 16 == 0):
        raise ValueError("out must be an fp32 [n, d] device view with unit column stride and 16-byte aligned rows")
    return out, out.stride(0)


def soft_assign(xref, what, idx, dist, hard: bool = False, want_w: bool = True, want_sqerr: bool = True,
                raw: bool = False, out=None):
    """(w [n,k], zq [n,d], row_sqerr [n]); zq is the straight-through value unless raw.
    `out` may be a column block of a wider tensor (e.g. emb[:, d:2*d])."""
    xref, what = _dev(xref, "xref"), _dev(what, "what")
    idx = _dev(idx, "idx", torch.int64)
    n, d = xref.shape
    topk = 1 if idx.dim() == 1 else idx.shape[1]
    dist = None if hard and dist is None else _dev(dist, "dist")
    lib = _lib.load()
    w = torch.empty((n, topk), dtype=torch.float32, device=xref.device) if want_w else None
    zq, zstride = _zq_out(out, n, d, xref)
    se = torch.empty(n, dtype=torch.float32, device=xref.device) if want_sqerr else None
    with _on(xref.device):
        _lib.check(lib.medtok_soft_assign_f32(xref.data_ptr(), what.data_ptr(), idx.data_ptr(), _ptr(dist), n, d, topk,
                                              int(hard) | (2 if raw else 0), _ptr(w), zq.data_ptr(), zstride, _ptr(se), _stream(xref)),
                   "medtok_soft_assign_f32")
    return w, zq, se


def soft_vq_backward(x, xhat, what, idx, w, g_zq=None, g_xhat=None, g_out=None, g_vq=None, g_commit=None,
                     vq_scale: float = 0.0, commit_scale: float = 0.0, want_gx: bool = True, want_g_code: bool = True):
    """Sparse backward of the soft top-k assignment: (gx [n, d], g_code [n*k, d]).

    g_zq / g_xhat / g_out: [n, d] upstream gradients w.r.t. the code mix, the normalised rows and the
    straight-through output (None = zero); g_vq / g_commit: 0-dim DEVICE tensors, the upstream gradients of
    mean((zq - x)^2) seen from zq and from x, scaled by vq_scale / commit_scale inside the kernel.
    g_code[r*k + j] is the gradient w.r.t. the normalised code what[idx[r, j]]."""
    x, xhat, what, w = _dev(x, "x"), _dev(xhat, "xhat"), _dev(what, "what"), _dev(w, "w")
    idx = _dev(idx, "idx", torch.int64)
    n, d = x.shape
    topk = idx.shape[1]
    opt = [None if t is None else _dev(t, nm) for t, nm in ((g_zq, "g_zq"), (g_xhat, "g_xhat"), (g_out, "g_out"),
                                                            (g_vq, "g_vq"), (g_commit, "g_commit"))]
    gx = torch.empty((n, d), dtype=torch.float32, device=x.device) if want_gx else None
    g_code = torch.empty((n * topk, d), dtype=torch.float32, device=x.device) if want_g_code else None
    lib = _lib.load()
    with _on(x.device):
        _lib.check(lib.medtok_soft_vq_backward_f32(x.data_ptr(), xhat.data_ptr(), what.data_ptr(), idx.data_ptr(), w.data_ptr(),
                                                   n, d, topk, *[_ptr(t) for t in opt], float(vq_scale), float(commit_scale),
                                                   _ptr(gx), _ptr(g_code), _stream(x)), "medtok_soft_vq_backward_f32")
    return gx, g_code


def normalize_backward(g, vhat, v, live=None):
    """Backward of F.normalize(v, dim=-1): (g - vhat (vhat . g)) / max(|v|, 1e-12), row-wise.  live ([n] fp32): rows with a 0 there are
    known to hold an all-zero g and are written as zeros without being read (the same bits)."""
    g, vhat, v = _dev(g, "g"), _dev(vhat, "vhat"), _dev(v, "v")
    n, d = v.shape
    out = torch.empty_like(v)
    lib = _lib.load()
    with _on(v.device):
        if live is not None:
            live = _dev(live, "live")
            if live.numel() != n:
                raise ValueError("normalize_backward: live must have one entry per row")
            _lib.check(lib.medtok_normalize_backward_sparse_f32(g.data_ptr(), vhat.data_ptr(), v.data_ptr(), live.data_ptr(), n, d, out.data_ptr(), _stream(v)),
                       "medtok_normalize_backward_sparse_f32")
        else:
            _lib.check(lib.medtok_normalize_backward_f32(g.data_ptr(), vhat.data_ptr(), v.data_ptr(), n, d, out.data_ptr(), _stream(v)),
                       "medtok_normalize_backward_f32")
    return out


def info_nce_forward(q, k, temperature: float):
    """(loss 0-dim, prob [b, b], ws): InfoNCE of loss.py:40-56; prob and ws feed info_nce_backward."""
    q, k = _dev(q, "q"), _dev(k, "k")
    b, d = q.shape
    if k.shape != q.shape:
        raise ValueError("info_nce: q and k must have the same shape")
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    prob = torch.empty((b, b), dtype=torch.float32, device=q.device)
    ws = _ws(lib.medtok_info_nce_workspace_bytes(b, d), q)
    with _on(q.device):
        _lib.check(lib.medtok_info_nce_forward_f32(q.data_ptr(), k.data_ptr(), b, d, float(temperature), loss.data_ptr(),
                                                   prob.data_ptr(), ws.data_ptr(), ws.numel(), _stream(q)), "medtok_info_nce_forward_f32")
    return loss, prob, ws


def info_nce_backward(q, k, prob, ws, g_loss, temperature: float):
    """(gq, gk) [b, d] from the forward's prob / ws and the upstream 0-dim device gradient."""
    q, k, prob, g_loss = _dev(q, "q"), _dev(k, "k"), _dev(prob, "prob"), _dev(g_loss, "g_loss")
    b, d = q.shape
    gq, gk = torch.empty_like(q), torch.empty_like(k)
    lib = _lib.load()
    with _on(q.device):
        _lib.check(lib.medtok_info_nce_backward_f32(q.data_ptr(), k.data_ptr(), prob.data_ptr(), g_loss.data_ptr(), b, d,
                                                    float(temperature), gq.data_ptr(), gk.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    _stream(q)), "medtok_info_nce_backward_f32")
    return gq, gk


def row_dot(a, b):
    """out[r] = <a[r], b[r]> (fp32, the rownorm summation order)."""
    a, b = _dev(a, "a"), _dev(b, "b")
    n, d = a.shape
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    with _on(a.device):
        _lib.check(_lib.load().medtok_row_dot_f32(a.data_ptr(), b.data_ptr(), n, d, out.data_ptr(), _stream(a)), "medtok_row_dot_f32")
    return out


def small_gemm(A, B, trans_a: bool = False, trans_b: bool = False):
    """op(A) @ op(B) in exact fp32 (one fmaf chain per entry, k ascending).  A, B: contiguous fp32 matrices; the transposes
    are strides, nothing is copied."""
    A, B = _dev(A, "A"), _dev(B, "B")
    (m, k), (sam, sak) = ((A.shape[1], A.shape[0]), (1, A.shape[1])) if trans_a else ((A.shape[0], A.shape[1]), (A.shape[1], 1))
    (k2, n), (sbk, sbn) = ((B.shape[1], B.shape[0]), (1, B.shape[1])) if trans_b else ((B.shape[0], B.shape[1]), (B.shape[1], 1))
    if k != k2:
        raise ValueError(f"small_gemm: inner dimensions differ ({k} vs {k2})")
    C = torch.empty((m, n), dtype=torch.float32, device=A.device)
    with _on(A.device):
        _lib.check(_lib.load().medtok_small_gemm_f32(A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn, m, n, k, C.data_ptr(), _stream(A)),
                   "medtok_small_gemm_f32")
    return C


def frobenius(x):
    """0-dim fp32 tensor ||x||_F of a contiguous [rows, d] matrix (d % 4 == 0)."""
    x = _dev(x, "x")
    rows, d = x.shape
    lib = _lib.load()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = _ws(lib.medtok_frobenius_workspace_bytes(rows), x)
    with _on(x.device):
        _lib.check(lib.medtok_frobenius_f32(x.data_ptr(), rows, d, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x)), "medtok_frobenius_f32")
    return out


def split_half(x, dp: int = None, scale: float = 1.0, seg_len=None, seg_rows: int = 0):
    """(hi, lo) fp16 images [n, dp] of the fp32 matrix x [n, d] (last dim contiguous; a row-strided view is fine): x * scale =
    hi + lo to ~2^-22 relative.  dp (default: d rounded up to 8) pads with zero columns.  seg_len (int64 [n / seg_rows] on the
    device): only the first seg_len[b] rows of every segment of seg_rows rows are converted (the rest stay uninitialised)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.MedTokLibraryError("split_half: expected an fp32 [n, d] matrix with contiguous rows on an MI355X device")
    n, d = x.shape
    dp = (d + 7) // 8 * 8 if dp is None else int(dp)
    hi = torch.empty((n, dp), dtype=torch.float16, device=x.device)
    lo = torch.empty((n, dp), dtype=torch.float16, device=x.device)
    with _on(x.device):
        _lib.check(_lib.load().medtok_split_half_f32(x.data_ptr(), n, d, x.stride(0) if n > 1 else d, dp, float(scale), hi.data_ptr(), lo.data_ptr(),
                                                     _ptr(None if seg_len is None else _dev(seg_len, "seg_len", torch.int64)), int(seg_rows), _stream(x)),
                   "medtok_split_half_f32")
    return hi, lo


def split_gemm(a, b, n_g: int, k_g: int, groups: int = 1, a_group_cols: int = 0, b_group_rows: int = 0, bias=None, unscale: float = 1.0,
               want_f32: bool = True, want_split: bool = False):
    """unscale * (A . B^T) + bias on three fp16 MFMA passes (medtok_split_gemm_f16).  a = (hi, lo) [m, lda], b = (hi, lo)
    [b_rows, ldb] fp16 images; see include/medtok_vq.h for the grouped form.  Returns (c fp32 [m, groups * n_g] or None,
    (c_hi, c_lo) or None)."""
    a_hi, a_lo = a
    b_hi, b_lo = b
    for t in (a_hi, a_lo, b_hi, b_lo):
        if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.dim() == 2):
            raise _lib.MedTokLibraryError("split_gemm: operands must be contiguous fp16 matrices on an MI355X device")
    m, lda = a_hi.shape
    b_rows, ldb = b_hi.shape
    n_out = groups * n_g
    bias = None if bias is None else _dev(bias, "bias")
    c = torch.empty((m, n_out), dtype=torch.float32, device=a_hi.device) if want_f32 else None
    ch = torch.empty((m, n_out), dtype=torch.float16, device=a_hi.device) if want_split else None
    cl = torch.empty((m, n_out), dtype=torch.float16, device=a_hi.device) if want_split else None
    with _on(a_hi.device):
        _lib.check(_lib.load().medtok_split_gemm_f16(a_hi.data_ptr(), a_lo.data_ptr(), m, lda, int(a_group_cols), b_hi.data_ptr(), b_lo.data_ptr(),
                                                     b_rows, ldb, int(b_group_rows), int(n_g), int(k_g), int(groups), _ptr(bias), float(unscale),
                                                     _ptr(c), n_out, _ptr(ch), _ptr(cl), n_out, _stream(a_hi)), "medtok_split_gemm_f16")
    return c, ((ch, cl) if want_split else None)


def half_gemm(a, b, n_g: int, k_g: int, groups: int = 1, a_group_cols: int = 0, b_group_rows: int = 0, bias=None, unscale: float = 1.0):
    """c [m, groups * n_g] fp32 = unscale * (A . B^T) + bias in ONE half-precision pass with fp32 accumulation (medtok_half_gemm_f32):
    a [m, lda], b [b_rows, ldb] both torch.float16 or both torch.bfloat16, contiguous; grouped as split_gemm."""
    if a.dtype != b.dtype or a.dtype not in (torch.float16, torch.bfloat16):
        raise TypeError("half_gemm: operands must both be float16 or both bfloat16")
    for t in (a, b):
        if not (t.is_cuda and t.is_contiguous() and t.dim() == 2):
            raise _lib.MedTokLibraryError("half_gemm: operands must be contiguous 2-d matrices on an MI355X device")
    m, lda = a.shape
    b_rows, ldb = b.shape
    bias = None if bias is None else _dev(bias, "bias")
    c = torch.empty((m, groups * n_g), dtype=torch.float32, device=a.device)
    with _on(a.device):
        _lib.check(_lib.load().medtok_half_gemm_f32(a.data_ptr(), m, lda, int(a_group_cols), b.data_ptr(), b_rows, ldb, int(b_group_rows), int(n_g), int(k_g),
                                                    int(groups), _ptr(bias), float(unscale), c.data_ptr(), groups * n_g, int(a.dtype == torch.bfloat16),
                                                    _stream(a)), "medtok_half_gemm_f32")
    return c


def half_image(x, dp: int, dtype, transpose: bool = False, group_cols: int = 0, col_sums: bool = False):
    """the fp16 / bf16 image of the fp32 matrix x [n, d] (unit column stride): [n, dp] with zero columns past d, or (transpose) of x^T,
    [d, dp] with dp >= n, stacked as dp / group_cols groups of [d, group_cols] (medtok_half_image_f32)"""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.MedTokLibraryError("half_image: expected an fp32 [n, d] matrix with contiguous rows on an MI355X device")
    n, d = x.shape
    if transpose:
        gc = int(group_cols) or int(dp)
        shape = (int(dp) // gc * d, gc)
    else:
        shape = (n, int(dp))
    out = torch.empty(shape, dtype=dtype, device=x.device)
    if col_sums:            # (transposed image + the column sums of x from the same pass: medtok_half_image_t_sums_f32)
        if not transpose:
            raise ValueError("half_image: col_sums comes with the transposed image")
        partials = torch.empty(((int(dp) + 63) // 64, d), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _lib.check(_lib.load().medtok_half_image_t_sums_f32(x.data_ptr(), n, d, x.stride(0) if n > 1 else d, int(dp), int(group_cols),
                                                                int(dtype == torch.bfloat16), out.data_ptr(), partials.data_ptr(), _stream(x)),
                       "medtok_half_image_t_sums_f32")
        return out, partials.sum(0)
    with _on(x.device):
        _lib.check(_lib.load().medtok_half_image_f32(x.data_ptr(), n, d, x.stride(0) if n > 1 else d, int(dp), int(bool(transpose)), int(group_cols),
                                                     int(dtype == torch.bfloat16), out.data_ptr(), _stream(x)), "medtok_half_image_f32")
    return out


def half_image_pair(x, dp: int, np_: int, dtype, group_cols: int = 0, col_sums: bool = False):
    """(half_image(x, dp), half_image(x, np_, transpose=True, group_cols)) from ONE pass over x (medtok_half_image_pair_f32); rows that the
    rows of x do not reach in the transposed image's padding stay unwritten only where half_image(transpose=True) leaves them so too"""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.MedTokLibraryError("half_image_pair: expected an fp32 [n, d] matrix with contiguous rows on an MI355X device")
    n, d = x.shape
    if n == 0 or int(dp) > (d + 63) // 64 * 64:
        pair = half_image(x, dp, dtype), half_image(x, np_, dtype, transpose=True, group_cols=group_cols)
        return (*pair, x.sum(0)) if col_sums else pair
    gc = int(group_cols) or int(np_)
    plain = torch.empty((n, int(dp)), dtype=dtype, device=x.device)
    t = torch.empty((int(np_) // gc * d, gc), dtype=dtype, device=x.device)
    if col_sums:            # (+ the column sums of x from the same pass: per 64-row tile in the kernel, over the tiles here)
        partials = torch.empty(((int(np_) + 63) // 64, d), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _lib.check(_lib.load().medtok_half_image_pair_sums_f32(x.data_ptr(), n, d, x.stride(0) if n > 1 else d, int(dp), int(np_), int(group_cols),
                                                                   int(dtype == torch.bfloat16), plain.data_ptr(), t.data_ptr(), partials.data_ptr(),
                                                                   _stream(x)), "medtok_half_image_pair_sums_f32")
        return plain, t, partials.sum(0)
    with _on(x.device):
        _lib.check(_lib.load().medtok_half_image_pair_f32(x.data_ptr(), n, d, x.stride(0) if n > 1 else d, int(dp), int(np_), int(group_cols),
                                                          int(dtype == torch.bfloat16), plain.data_ptr(), t.data_ptr(), _stream(x)), "medtok_half_image_pair_f32")
    return plain, t


def absmax(x):
    """0-dim device fp32 tensor max |x| (no host read): feeds the power-of-two prescale of training-mode split operands."""
    x = _dev(x, "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.load().medtok_absmax_f32(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x)), "medtok_absmax_f32")
    return out


def split_half_scaled(x, dp: int, amax=None, transpose: bool = False, group_cols: int = 0):
    """(hi, lo) fp16 images of the fp32 matrix x [n, d], prescaled by the power of two that `amax` (0-dim device tensor, from absmax;
    None: 1) selects.  transpose=False: [n, dp], dp >= d; transpose=True: the images of x^T, [d, dp], dp >= n -- with group_cols (a
    multiple of 64 dividing dp) stacked as [dp / group_cols * d, group_cols]: group g holds columns [g group_cols, + group_cols)."""
    x = _dev(x, "x")
    n, d = x.shape
    if transpose:
        gc = int(group_cols) or int(dp)
        shape = (int(dp) // gc * d, gc)
    else:
        shape = (n, int(dp))
    hi = torch.empty(shape, dtype=torch.float16, device=x.device)
    lo = torch.empty(shape, dtype=torch.float16, device=x.device)
    with _on(x.device):
        _lib.check(_lib.load().medtok_split_half_scaled_f32(x.data_ptr(), n, d, d, int(dp), _ptr(amax), int(bool(transpose)), int(group_cols),
                                                            hi.data_ptr(), lo.data_ptr(), _stream(x)), "medtok_split_half_scaled_f32")
    return hi, lo


def split_gemm_scaled(a, b, n_g: int, k_g: int, bias=None, unscale: float = 1.0, amax_a=None, amax_b=None, groups: int = 1, a_group_cols: int = 0,
                      b_group_rows: int = 0):
    """c [m, groups * n_g] fp32 = unscale / (pow2(amax_a) pow2(amax_b)) * A . B^T + bias from (hi, lo) images made by split_half_scaled
    (grouped as medtok_split_gemm_f16)."""
    (a_hi, a_lo), (b_hi, b_lo) = a, b
    m, lda = a_hi.shape
    b_rows, ldb = b_hi.shape
    c = torch.empty((m, groups * n_g), dtype=torch.float32, device=a_hi.device)
    bias = None if bias is None else _dev(bias, "bias")
    with _on(a_hi.device):
        _lib.check(_lib.load().medtok_split_gemm_scaled_f16(a_hi.data_ptr(), a_lo.data_ptr(), m, lda, int(a_group_cols), b_hi.data_ptr(), b_lo.data_ptr(),
                                                            b_rows, ldb, int(b_group_rows), int(n_g), int(k_g), int(groups), _ptr(bias), float(unscale),
                                                            _ptr(amax_a), _ptr(amax_b), c.data_ptr(), groups * n_g, _stream(a_hi)),
                   "medtok_split_gemm_scaled_f16")
    return c


def residual_layernorm(a, b, gamma, beta, eps: float, split_dp: int = 0):
    """LayerNorm(a + b) * gamma + beta per row of the contiguous fp32 matrices a, b [n, d] (the tail of CrossAttentionLayer,
    vector_quantization_soft_one_new.py:47-50); d % 4 == 0, d <= 4096.  split_dp > 0: returns (y, (y_hi, y_lo)) with the (hi, lo)
    fp16 images [n, split_dp] of y that split_half(y, dp=split_dp) would make, written by the same kernel."""
    a, b, gamma, beta = _dev(a, "a"), _dev(b, "b"), _dev(gamma, "gamma"), _dev(beta, "beta")
    if a.shape != b.shape or a.dim() != 2 or gamma.shape != (a.shape[1],) or beta.shape != (a.shape[1],):
        raise ValueError(f"residual_layernorm: shapes a={tuple(a.shape)} b={tuple(b.shape)} gamma={tuple(gamma.shape)} beta={tuple(beta.shape)}")
    n, d = a.shape
    y = torch.empty_like(a)
    hi = lo = None
    if split_dp:
        hi = torch.empty((n, split_dp), dtype=torch.float16, device=a.device)
        lo = torch.empty((n, split_dp), dtype=torch.float16, device=a.device)
    with _on(a.device):
        _lib.check(_lib.load().medtok_residual_layernorm_split_f32(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, d, float(eps),
                                                                   y.data_ptr(), _ptr(hi), _ptr(lo), int(split_dp), _stream(a)),
                   "medtok_residual_layernorm_split_f32")
    return (y, (hi, lo)) if split_dp else y


def pack_codes(mask, batch, heads: int, lpt: bool, count_bound: int = 0, status=None):
    """The prologue of the batched cross-attention in three small launches (include/medtok_vq.h: medtok_pack_codes_checked): mask [B, L]
    (bool / int32 / int64), batch [n_nodes] int64 -> dict(valid_len, counts, starts, t_start, t_len, g_start, g_len, tok_start,
    g_kv_len: int64 [B] device tensors; stats: int64 [4] = largest count, id range, unsorted flag -- not yet read back).
    status (int32 device tensor, optional): what a caller that never reads stats back needs flagged -- bit 0 unsorted, bit 1 id out of
    range, bit 2 a code with more than count_bound nodes -- OR-ed into status[0] on the device."""
    if not (isinstance(mask, torch.Tensor) and mask.is_cuda and mask.dim() == 2):
        raise _lib.MedTokLibraryError("pack_codes: expected a [B, L] mask on an MI355X device")
    if mask.dtype not in (torch.bool, torch.uint8, torch.int32, torch.int64):
        mask = mask != 0
    mask = mask.contiguous()
    batch = _dev(batch.reshape(-1), "batch", torch.int64)
    bsz, seq_len = mask.shape
    dev = mask.device
    out = torch.empty((9, bsz), dtype=torch.int64, device=dev)
    stats = torch.empty(4, dtype=torch.int64, device=dev)
    lib = _lib.load()
    ws = _ws(lib.medtok_pack_codes_workspace_bytes(bsz), mask)
    with _on(dev):
        if status is not None and not (status.is_cuda and status.dtype == torch.int32 and status.device == dev):
            raise _lib.MedTokLibraryError("pack_codes: status must be an int32 tensor on the mask's device")
        _lib.check(lib.medtok_pack_codes_checked(mask.data_ptr(), mask.element_size(), bsz, seq_len, batch.data_ptr(), batch.numel(), int(heads),
                                                 int(bool(lpt)), *[out[i].data_ptr() for i in range(9)], stats.data_ptr(), int(count_bound),
                                                 _ptr(status), ws.data_ptr(), ws.numel(), _stream(mask)),
                   "medtok_pack_codes")
    names = ("valid_len", "counts", "starts", "t_start", "t_len", "g_start", "g_len", "tok_start", "g_kv_len")
    r = {k: out[i] for i, k in enumerate(names)}
    r["stats"] = stats
    return r


def cross_attention_layer(rows, rows_images, w, q_start, q_len, max_q_len: int, kv, kv_split, kv_start, kv_len, scale: float, variant: int,
                          gamma, beta, eps: float, want_images: bool):
    """One CrossAttentionLayer at inference in one C call (include/medtok_vq.h: medtok_cross_attention_layer_f32).  rows [R, d] fp32;
    rows_images: their (hi, lo) images [R, dw] or None; w: the dict of CrossAttention._split_weights; kv fp32 [Rk, dw] or kv_split =
    (hi, lo | None) images, or kv_split = the fp32 rows themselves (split inside the kernel: shared_kv_attention_split).  Returns y [R, d] (and its images [R, dw] when want_images)."""
    rows = _dev(rows, "rows")
    n_rows, d = rows.shape
    heads, hp, dw = w["heads"], w["hp"], w["dw"]
    dev = rows.device
    y = torch.empty_like(rows)
    yh = torch.empty((n_rows, dw), dtype=torch.float16, device=dev) if want_images else None
    yl = torch.empty((n_rows, dw), dtype=torch.float16, device=dev) if want_images else None
    lib = _lib.load()
    ws = _ws(lib.medtok_cross_attention_layer_workspace_bytes(n_rows, d, dw, heads, hp), rows)
    (wq, wq_u), (wk, wk_u), (wv, wv_u), (wo, wo_u) = w["wq"], w["wk"], w["wv"], w["wo"]
    xh, xl = rows_images if rows_images is not None else (None, None)
    if torch.is_tensor(kv_split):               # fp32 key rows, turned into their images inside the attention kernel
        kh, kl, variant = _dev(kv_split, "kv"), None, int(variant) | ATTENTION_F32_KEYS
    else:
        kh, kl = kv_split if kv_split is not None else (None, None)
    with _on(dev):
        _lib.check(lib.medtok_cross_attention_layer_f32(
            rows.data_ptr(), _ptr(xh), _ptr(xl), n_rows, d, dw, heads, hp,
            wq[0].data_ptr(), wq[1].data_ptr(), float(wq_u), w["bq"].data_ptr(), wk[0].data_ptr(), wk[1].data_ptr(), float(wk_u),
            wv[0].data_ptr(), wv[1].data_ptr(), float(wv_u), w["bv"].data_ptr(), wo[0].data_ptr(), wo[1].data_ptr(), float(wo_u), w["bo"].data_ptr(),
            q_start.data_ptr(), q_len.data_ptr(), q_start.numel(), int(max_q_len), _ptr(kv if kv_split is None else None), _ptr(kh), _ptr(kl),
            kv_start.data_ptr(), kv_len.data_ptr(), float(scale), int(variant), gamma.data_ptr(), beta.data_ptr(), float(eps),
            y.data_ptr(), _ptr(yh), _ptr(yl), ws.data_ptr(), ws.numel(), _stream(rows)), "medtok_cross_attention_layer_f32")
    return (y, (yh, yl)) if want_images else y


SMALL_ATTENTION_LAYER_FLOATS = 4 * 64 * 64 + 6 * 64


def cross_attention_small(text, mask, nodes, batch, weights, layers: int, scale: float, ln_eps: float, pooled, status, exact_f32: bool = False):
    """CrossAttention.pooled at e_dim = 64 with 4 heads in two launches and no host read (include/medtok_vq.h:
    medtok_cross_attention_small_f32).  text [B, L, 64] fp32, mask [B, L] (bool / int32 / int64), nodes [N, 64] fp32 with a sorted
    batch vector [N] int64, weights [layers, SMALL_ATTENTION_LAYER_FLOATS] (CrossAttention._small_weights), pooled [B, 2, 64] fp32
    (written: [:, 0] the attended CLS rows, [:, 1] the node means), status int32 [4] (zeroed once by its owner).
    exact_f32 (tests only): the attention core on the fp32 matrix pipe instead of the product's split-fp16 core."""
    text, nodes, weights = _dev(text, "text"), _dev(nodes, "nodes"), _dev(weights, "weights")
    batch = _dev(batch.reshape(-1), "batch", torch.int64)
    if not (isinstance(mask, torch.Tensor) and mask.is_cuda and mask.dim() == 2):
        raise _lib.MedTokLibraryError("cross_attention_small: expected a [B, L] mask on an MI355X device")
    if mask.dtype not in (torch.bool, torch.uint8, torch.int32, torch.int64):
        mask = mask != 0
    mask = mask.contiguous()
    bsz, seq_len, d = text.shape
    n_nodes = nodes.shape[0]
    if pooled.shape != (bsz, 2, d) or pooled.dtype != torch.float32 or not pooled.is_contiguous():
        raise ValueError("cross_attention_small: pooled must be a contiguous fp32 [B, 2, d] tensor")
    y_nodes = torch.empty((max(n_nodes, 1), d), dtype=torch.float32, device=text.device)
    with _on(text.device):
        lib = _lib.load()
        _lib.check((lib.medtok_debug_cross_attention_small_exact_f32 if exact_f32 else lib.medtok_cross_attention_small_f32)(
            text.data_ptr(), mask.data_ptr(), mask.element_size(), bsz, seq_len, nodes.data_ptr() if n_nodes else 0, batch.data_ptr() if n_nodes else 0,
            n_nodes, d, 4, int(layers), weights.data_ptr(), float(scale), float(ln_eps), y_nodes.data_ptr(), pooled.data_ptr(), 2 * d, d,
            status.data_ptr(), _stream(text)), "medtok_cross_attention_small_f32")
    return pooled


def segment_mean(x, seg_start, seg_len):
    """out[b] = mean of rows [seg_start[b], seg_start[b] + seg_len[b]) of the contiguous fp32 matrix x [rows, d] (rows added in
    order; an empty segment gives zeros) -- the `.mean(dim=0)` over a code's graph nodes (:140-141).  seg_* int64 device vectors."""
    x = _dev(x, "x")
    seg_start, seg_len = _dev(seg_start, "seg_start", torch.int64), _dev(seg_len, "seg_len", torch.int64)
    n_seg, d = seg_start.numel(), x.shape[1]
    if seg_len.numel() != n_seg:
        raise ValueError("segment_mean: seg_start and seg_len differ in length")
    out = torch.empty((n_seg, d), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.load().medtok_segment_mean_f32(x.data_ptr(), seg_start.data_ptr(), seg_len.data_ptr(), n_seg, d, out.data_ptr(), _stream(x)),
                   "medtok_segment_mean_f32")
    return out


def scale_by_device_scalar(x, num, den=None, c: float = 1.0):
    """x * (c * num / den) with num / den 0-dim device tensors (no host sync)."""
    x, num = _dev(x, "x"), _dev(num, "num")
    den = None if den is None else _dev(den, "den")
    out = torch.empty_like(x)
    with _on(x.device):
        _lib.check(_lib.load().medtok_scale_by_device_scalar_f32(x.data_ptr(), x.numel(), num.data_ptr(), _ptr(den), float(c), out.data_ptr(),
                                                                 _stream(x)), "medtok_scale_by_device_scalar_f32")
    return out


def _attention_outputs(q, split_out):
    """fp32 result, or its (hi, lo) fp16 images (rows that belong to no code stay uninitialised either way)"""
    if not split_out:
        return torch.empty_like(q), None, None
    return None, torch.empty(q.shape, dtype=torch.float16, device=q.device), torch.empty(q.shape, dtype=torch.float16, device=q.device)


def shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, max_q_len: int, scale: float, exact_f32: bool = False, split_out: bool = False):
    """out[r] = softmax_j(scale * <q[r], kv[j]>) . kv over each code's own (ragged) query and key rows.
    q [Rq, d], kv [Rk, d] fp32; *_start / *_len int64 [n_codes] on the device; d = 64 or a multiple of 128 up to 768.
    Default: both products as three fp16 MFMAs over (hi, lo) pairs (fp32-accurate, 16/3 the fp32 pipe's rate); exact_f32: the
    fp32-MFMA kernel the training forward uses.  split_out: return the (hi, lo) fp16 images of the result instead (written by the
    kernel for the dense product that follows)."""
    q, kv = _dev(q, "q"), _dev(kv, "kv")
    qs, ql = _dev(q_start, "q_start", torch.int64), _dev(q_len, "q_len", torch.int64)
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    out, oh, ol = _attention_outputs(q, split_out)
    lib = _lib.load()
    with _on(q.device):
        _lib.check(lib.medtok_shared_kv_attention_f32(q.data_ptr(), qs.data_ptr(), ql.data_ptr(), kv.data_ptr(), ks.data_ptr(),
                                                      kl.data_ptr(), qs.numel(), int(max_q_len), q.shape[1], float(scale),
                                                      _ptr(out), _ptr(oh), _ptr(ol), int(bool(exact_f32)), _stream(q)), "medtok_shared_kv_attention_f32")
    return (oh, ol) if split_out else out


ATTENTION_SPLIT_WIDTHS = (128, 256, 384, 512, 768, 1024)
ATTENTION_HALF_KEY_WIDTHS = (256, 512, 768)        # widths at which fp16 keys are taken as they stand (variant 2, no lo image)


ATTENTION_F32_KEYS = 0x100                 # MEDTOK_ATTENTION_F32_KEYS (include/medtok_vq.h)


def shared_kv_attention_split(q, q_start, q_len, kv_split, kv_start, kv_len, max_q_len: int, scale: float, split_out: bool = False, variant: int = 0):
    """shared_kv_attention for wide batches: the keys as the (hi, lo) fp16 images of split_half (made once per forward), 64 query
    rows per block, keys copied into LDS by DMA.  d in ATTENTION_SPLIT_WIDTHS."""
    q = _dev(q, "q")
    if torch.is_tensor(kv_split):               # fp32 key rows, turned into their images inside the kernel (variant 2, d = 256 / 512 / 768)
        kh, kl_ = _dev(kv_split, "kv"), None
        if kh.dim() != 2 or kh.shape[1] != q.shape[1]:
            raise _lib.MedTokLibraryError("shared_kv_attention_split: fp32 keys must be [Rk, d]")
        variant = int(variant) | ATTENTION_F32_KEYS
    else:
        kh, kl_ = kv_split                      # kl_ = None: fp16 keys as they stand (no lo image; variant 2, d = 256 / 512 / 768)
        for t in (kh, kl_):
            if t is not None and not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == q.shape[1]):
                raise _lib.MedTokLibraryError("shared_kv_attention_split: the key images must be contiguous fp16 [Rk, d] device tensors")
    qs, ql = _dev(q_start, "q_start", torch.int64), _dev(q_len, "q_len", torch.int64)
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    out, oh, ol = _attention_outputs(q, split_out)
    with _on(q.device):
        _lib.check(_lib.load().medtok_shared_kv_attention_split_f32(q.data_ptr(), qs.data_ptr(), ql.data_ptr(), kh.data_ptr(), _ptr(kl_),
                                                                    ks.data_ptr(), kl.data_ptr(), qs.numel(), int(max_q_len), q.shape[1], float(scale),
                                                                    _ptr(out), _ptr(oh), _ptr(ol), int(variant), _stream(q)), "medtok_shared_kv_attention_split_f32")
    return (oh, ol) if split_out else out


ATTENTION_TRAIN_SPLIT_WIDTHS = (256, 512, 768)


def _into(buf, name, shape, like):
    """a caller's output buffer (a contiguous fp32 tensor -- e.g. a row range of a larger one -- that a kernel writes into)"""
    t = _dev(buf, name)
    if t is not buf:
        raise ValueError(f"{name} must be contiguous (the kernel writes into it)")
    if tuple(t.shape) != tuple(shape) or t.device != like.device:
        raise ValueError(f"{name} {tuple(t.shape)} on {t.device} does not match {tuple(shape)} on {like.device}")
    return t


def shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, max_q_len: int, scale: float, dropout_p: float = 0.0, seed: int = 0,
                              split: bool = False, out=None, lse=None):
    """(out, lse): the attention core with dropout on the probabilities (stateless hash mask) and the log-sum-exp per query row.
    split=True (widths ATTENTION_TRAIN_SPLIT_WIDTHS): on the three-pass fp16 products instead of the exact fp32 matrix pipe -- the
    same mask bits, outputs within ~1e-6 relative, a third of the time at D = 768 (the autocast trainer's form).
    out / lse: the caller's buffers ([rows, d] zeros / [rows] -inf where rows may belong to no code), e.g. row ranges of larger ones."""
    q, kv = _dev(q, "q"), _dev(kv, "kv")
    qs, ql = _dev(q_start, "q_start", torch.int64), _dev(q_len, "q_len", torch.int64)
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    out = torch.zeros_like(q) if out is None else _into(out, "out", q.shape, q)              # rows that belong to no code stay zero
    lse = (torch.full((q.shape[0],), float("-inf"), dtype=torch.float32, device=q.device) if lse is None else _into(lse, "lse", (q.shape[0],), q))
    lib = _lib.load()
    fn, name = ((lib.medtok_shared_kv_attention_train_split_f32, "medtok_shared_kv_attention_train_split_f32") if split else
                (lib.medtok_shared_kv_attention_train_f32, "medtok_shared_kv_attention_train_f32"))
    with _on(q.device):
        _lib.check(fn(q.data_ptr(), qs.data_ptr(), ql.data_ptr(), kv.data_ptr(), ks.data_ptr(), kl.data_ptr(), qs.numel(), int(max_q_len), q.shape[1],
                      float(scale), float(dropout_p), int(seed) & 0xFFFFFFFF, out.data_ptr(), lse.data_ptr(), _stream(q)), name)
    return out, lse


def shared_kv_attention_backward(q, q_start, q_len, kv, kv_start, kv_len, max_q_len: int, max_kv_len: int, scale: float, dropout_p: float,
                                 seed: int, out, lse, d_out, half=None, dkv_into=None, accumulate=False, dq_into=None):
    """(dq, dkv) of shared_kv_attention_train for the upstream gradient d_out.  half = torch.float16 / torch.bfloat16: the four matrix
    products in one half-precision pass with fp32 accumulation (autocast callers); None: exact fp32 MFMA.
    dkv_into: a contiguous fp32 [kv_rows, d] buffer that receives dkv (returned as dkv); with accumulate=True the key gradient is
    ADDED to what it holds (the layers of CrossAttention share their keys: one buffer, no add pass).  dq_into: the buffer for dq."""
    q, kv, out, lse, d_out = _dev(q, "q"), _dev(kv, "kv"), _dev(out, "out"), _dev(lse, "lse"), _dev(d_out, "d_out")
    qs, ql = _dev(q_start, "q_start", torch.int64), _dev(q_len, "q_len", torch.int64)
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    dq = torch.empty_like(q) if dq_into is None else _into(dq_into, "dq_into", q.shape, q)
    if dkv_into is None:
        if accumulate:
            raise ValueError("shared_kv_attention_backward: accumulate=True needs the buffer to add to (dkv_into)")
        dkv = torch.empty_like(kv)
    else:
        dkv = _dev(dkv_into, "dkv_into")
        if dkv is not dkv_into:
            raise ValueError("shared_kv_attention_backward: dkv_into must be contiguous (the kernel writes into it)")
        if dkv.shape != kv.shape or dkv.device != kv.device:
            raise ValueError(f"shared_kv_attention_backward: dkv_into {tuple(dkv.shape)} on {dkv.device} does not match kv {tuple(kv.shape)} on {kv.device}")
    lib = _lib.load()
    ws = _ws(lib.medtok_shared_kv_attention_backward_workspace_bytes(q.shape[0]), q)
    args = (q.data_ptr(), qs.data_ptr(), ql.data_ptr(), kv.data_ptr(), ks.data_ptr(), kl.data_ptr(), qs.numel(), int(max_q_len), int(max_kv_len),
            q.shape[0], kv.shape[0], q.shape[1], float(scale), float(dropout_p), int(seed) & 0xFFFFFFFF, out.data_ptr(), lse.data_ptr(),
            d_out.data_ptr(), dq.data_ptr(), dkv.data_ptr(), ws.data_ptr(), ws.numel())
    with _on(q.device):
        if accumulate:
            mode = 1 if half == torch.float16 else (2 if half == torch.bfloat16 else 0)
            _lib.check(lib.medtok_shared_kv_attention_backward_acc_f32(*args, mode, 1, _stream(q)), "medtok_shared_kv_attention_backward_acc_f32")
        elif half in (torch.float16, torch.bfloat16):
            _lib.check(lib.medtok_shared_kv_attention_backward_half_f32(*args, int(half == torch.bfloat16), _stream(q)), "medtok_shared_kv_attention_backward_half_f32")
        else:
            _lib.check(lib.medtok_shared_kv_attention_backward_f32(*args, _stream(q)), "medtok_shared_kv_attention_backward_f32")
    return dq, dkv


def shared_kv_attention_backward_dq(q, q_start, q_len, kv, kv_start, kv_len, max_q_len: int, max_kv_len: int, scale: float, dropout_p: float,
                                    seed: int, out, lse, d_out, half=None, dq_into=None):
    """(dq, delta): the query gradient alone (medtok_shared_kv_attention_backward_acc_f32, accumulate_dkv = 2) and delta = <d_out, out> per
    query row -- what shared_kv_attention_dkv_multi takes, with the call's other tensors, as one of its sources."""
    q, kv, out, lse, d_out = _dev(q, "q"), _dev(kv, "kv"), _dev(out, "out"), _dev(lse, "lse"), _dev(d_out, "d_out")
    qs, ql = _dev(q_start, "q_start", torch.int64), _dev(q_len, "q_len", torch.int64)
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    dq = torch.empty_like(q) if dq_into is None else _into(dq_into, "dq_into", q.shape, q)
    delta = torch.empty((max(q.shape[0], 1),), dtype=torch.float32, device=q.device)
    mode = 1 if half == torch.float16 else (2 if half == torch.bfloat16 else 0)
    with _on(q.device):
        _lib.check(_lib.load().medtok_shared_kv_attention_backward_acc_f32(
            q.data_ptr(), qs.data_ptr(), ql.data_ptr(), kv.data_ptr(), ks.data_ptr(), kl.data_ptr(), qs.numel(), int(max_q_len), int(max_kv_len),
            q.shape[0], kv.shape[0], q.shape[1], float(scale), float(dropout_p), int(seed) & 0xFFFFFFFF, out.data_ptr(), lse.data_ptr(),
            d_out.data_ptr(), dq.data_ptr(), 0, delta.data_ptr(), delta.numel() * 4, mode, 2, _stream(q)), "medtok_shared_kv_attention_backward_acc_f32")
    return dq, delta


def shared_kv_attention_dkv_multi(sources, kv, kv_start, kv_len, max_kv_len: int, half=None):
    """dkv [kv_rows, d]: the key gradient of several attention calls over the same keys in ONE launch (medtok_shared_kv_attention_dkv_multi_f32).
    sources: dicts with q, d_out, lse, delta (of shared_kv_attention_backward_dq), q_start, q_len, scale, dropout_p, seed."""
    kv = _dev(kv, "kv")
    ks, kl = _dev(kv_start, "kv_start", torch.int64), _dev(kv_len, "kv_len", torch.int64)
    count = len(sources)
    if not 1 <= count <= _lib.DKV_SOURCES_MAX:
        raise ValueError(f"shared_kv_attention_dkv_multi: 1..{_lib.DKV_SOURCES_MAX} sources per call")
    descs = (_lib.DkvSource * count)()
    keep = []
    for i, src in enumerate(sources):
        t = [_dev(src[k], k) for k in ("q", "d_out", "lse", "delta")] + [_dev(src[k], k, torch.int64) for k in ("q_start", "q_len")]
        if t[0].shape[1] != kv.shape[1] or t[1].shape != t[0].shape or t[2].numel() < t[0].shape[0] or t[3].numel() < t[0].shape[0] or t[4].numel() != ks.numel():
            raise ValueError("shared_kv_attention_dkv_multi: a source's tensors do not fit together")
        keep.append(t)
        descs[i] = _lib.DkvSource(*[x.data_ptr() for x in t], float(src["scale"]), float(src["dropout_p"]), int(src["seed"]) & 0xFFFFFFFF, 0)
    dkv = torch.empty_like(kv)
    mode = 1 if half == torch.float16 else (2 if half == torch.bfloat16 else 0)
    with _on(kv.device):
        _lib.check(_lib.load().medtok_shared_kv_attention_dkv_multi_f32(descs, count, kv.data_ptr(), ks.data_ptr(), kl.data_ptr(), ks.numel(), int(max_kv_len),
                                                                        kv.shape[0], kv.shape[1], dkv.data_ptr(), mode, _stream(kv)),
                   "medtok_shared_kv_attention_dkv_multi_f32")
    return dkv


def sum_scale(vals: torch.Tensor, scale: float) -> torch.Tensor:
    """0-dim fp32 tensor = scale * sum(vals) (fp64 accumulation, fixed order)."""
    vals = _dev(vals, "vals")
    out = torch.empty((), dtype=torch.float32, device=vals.device)
    lib = _lib.load()
    with _on(vals.device):
        _lib.check(lib.medtok_sum_scale_f32(vals.data_ptr(), vals.numel(), float(scale), out.data_ptr(),
                                            _stream(vals)), "medtok_sum_scale_f32")
    return out


def ema_stats(zhat, idx, k_codes: int, fused: bool = False):
    """(bins [K] fp32, embed_sum [K, D] fp32).  With fused=True both are views of ONE buffer
    [embed_sum | bins], returned as a third value, so a single all-reduce covers both statistics
    (embed_sum first keeps it 16-byte aligned)."""
    zhat = _dev(zhat, "zhat")
    idx = _dev(idx, "idx", torch.int64)
    n, d = zhat.shape
    lib = _lib.load()
    stats = torch.empty(k_codes * (d + 1), dtype=torch.float32, device=zhat.device)
    es = stats[: k_codes * d].view(k_codes, d)
    bins = stats[k_codes * d:]
    ws = _ws(lib.medtok_ema_stats_workspace_bytes(n, k_codes), zhat)
    with _on(zhat.device):
        _lib.check(lib.medtok_ema_stats_f32(zhat.data_ptr(), idx.data_ptr(), n, d, k_codes, bins.data_ptr(),
                                            es.data_ptr(), ws.data_ptr(), ws.numel(), _stream(zhat)),
                   "medtok_ema_stats_f32")
    return (bins, es, stats) if fused else (bins, es)


def code_histogram(idx, k_codes: int) -> torch.Tensor:
    """bins [K] fp32: number of rows assigned to each code (exact)."""
    idx = _dev(idx.reshape(-1), "idx", torch.int64)
    lib = _lib.load()
    bins = torch.empty(k_codes, dtype=torch.float32, device=idx.device)
    ws = _ws(lib.medtok_code_histogram_workspace_bytes(k_codes), idx)
    with _on(idx.device):
        _lib.check(lib.medtok_code_histogram_f32(idx.data_ptr(), idx.numel(), k_codes, bins.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), _stream(idx)), "medtok_code_histogram_f32")
    return bins


def ema_apply_(E, cluster_size, bins, embed_sum, decay: float) -> None:
    """In-place codebook + cluster_size update."""
    for name, t in (("E", E), ("cluster_size", cluster_size)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.MedTokLibraryError(f"ema_apply_: {name} must be a contiguous fp32 device tensor")
    bins, embed_sum = _dev(bins, "bins"), _dev(embed_sum, "embed_sum")
    k, d = E.shape
    lib = _lib.load()
    with _on(E.device):
        _lib.check(lib.medtok_ema_apply_f32(E.data_ptr(), cluster_size.data_ptr(), bins.data_ptr(), embed_sum.data_ptr(),
                                            k, d, float(decay), float(1 - decay), _stream(E)), "medtok_ema_apply_f32")


def ema_cluster_size_(cluster_size, bins, decay: float) -> None:
    if not (cluster_size.is_cuda and cluster_size.dtype == torch.float32 and cluster_size.is_contiguous()):
        raise _lib.MedTokLibraryError("ema_cluster_size_: cluster_size must be a contiguous fp32 device tensor")
    bins = _dev(bins, "bins")
    lib = _lib.load()
    with _on(bins.device):
        _lib.check(lib.medtok_ema_cluster_size_f32(cluster_size.data_ptr(), bins.data_ptr(), cluster_size.numel(),
                                                   float(decay), float(1 - decay), _stream(bins)),
                   "medtok_ema_cluster_size_f32")


def usage_update_(window: torch.Tensor, ids: torch.Tensor, n_codes: int) -> torch.Tensor:
    """Slide `window` (fp32, in place) by ids.numel(), append ids, return the
    distinct-value count as a device int32 scalar (no host sync)."""
    if not (window.is_cuda and window.dtype == torch.float32 and window.is_contiguous()):
        raise _lib.MedTokLibraryError("usage_update_: window must be a contiguous fp32 device tensor")
    ids = _dev(ids.reshape(-1), "ids", torch.int64)
    lib = _lib.load()
    count = torch.empty((), dtype=torch.int32, device=window.device)
    ws = _ws(lib.medtok_usage_workspace_bytes(window.numel(), n_codes), window)
    with _on(window.device):
        _lib.check(lib.medtok_usage_update(window.data_ptr(), window.numel(), ids.data_ptr(), ids.numel(), n_codes,
                                           count.data_ptr(), ws.data_ptr(), ws.numel(), _stream(window)),
                   "medtok_usage_update")
    return count


def multi_search_eligible(n: int, k_codes: int, d: int, topk: int) -> bool:
    """may this search go into a soft_vq_forward_multi call (small: the exact fp32-MFMA path, at most 4096 rows)?"""
    return bool(_lib.load().medtok_soft_vq_multi_eligible(int(n), int(k_codes), int(d), int(topk)))


def soft_vq_forward_multi(searches, topk: int, want_sqerr: bool = False):
    """Several small soft top-k searches in one C call and three launches (include/medtok_vq.h: medtok_soft_vq_forward_multi_f32).
    searches: list of dict(x [n, d] fp32, what [K, d], wsq [K], out = optional [n, d] view for zq); every entry must be
    multi_search_eligible.  Returns a list of dict(xhat, idx, dist, w, zq, row_sqerr) -- the bits of soft_vq_forward per entry
    (row_sqerr: None unless want_sqerr, the training forward)."""
    import ctypes as C
    lib = _lib.load()
    count = len(searches)
    if not 1 <= count <= _lib.MULTI_SEARCH_MAX:
        raise ValueError(f"soft_vq_forward_multi: 1..{_lib.MULTI_SEARCH_MAX} searches per call")
    descs = (_lib.SearchDesc * count)()
    outs, keep = [], []
    d = None
    for i, q in enumerate(searches):
        x, what, wsq = q["x"], _dev(q["what"], "what"), _dev(q["wsq"], "wsq")
        # (x may be a column block of a wider row-major matrix: its row stride travels in the descriptor)
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.stride(0) >= x.shape[1] and x.data_ptr() % 16 == 0):
            x = _dev(x, "x")
        n, di = x.shape
        d = di if d is None else d
        if di != d or what.shape[1] != d:
            raise ValueError("soft_vq_forward_multi: all searches share one width")
        dev = x.device
        xhat = torch.empty((n, d), dtype=torch.float32, device=dev)
        idx = torch.empty((n, topk), dtype=torch.int64, device=dev)
        dist = torch.empty((n, topk), dtype=torch.float32, device=dev)
        w = torch.empty((n, topk), dtype=torch.float32, device=dev)
        zq, zstride = _zq_out(q.get("out"), n, d, x)
        sqerr = torch.empty((n,), dtype=torch.float32, device=dev) if want_sqerr else None
        keep.append((x, what, wsq))
        descs[i] = _lib.SearchDesc(x.data_ptr(), n, what.data_ptr(), wsq.data_ptr(), what.shape[0], xhat.data_ptr(), idx.data_ptr(), dist.data_ptr(),
                                   w.data_ptr(), zq.data_ptr(), zstride, x.stride(0) if n > 1 else d, _ptr(sqerr))
        outs.append(dict(xhat=xhat, idx=idx, dist=dist, w=w, zq=zq, row_sqerr=sqerr))
    ws = _ws(lib.medtok_soft_vq_forward_multi_workspace_bytes(descs, count, d, topk), keep[0][0])
    with _on(keep[0][0].device):
        _lib.check(lib.medtok_soft_vq_forward_multi_f32(descs, count, d, topk, ws.data_ptr(), ws.numel(), _stream(keep[0][0])),
                   "medtok_soft_vq_forward_multi_f32")
    return outs


def usage_update_multi_(window: torch.Tensor, ids_list, n_codes: int, extra_word=None) -> torch.Tensor:
    """usage_update_ for several id sets appended in order, in two launches: int32 [len(ids_list)] device tensor of the distinct-value
    counts after each update (no host sync).  extra_word (int32 device tensor): its first element is copied behind the counts (the
    result has one more entry), so that the caller's one host read brings it along; while it is non-zero the window is NOT written
    (include/medtok_vq.h: the caller repeats the forward on repaired inputs)."""
    import ctypes as C
    if not (window.is_cuda and window.dtype == torch.float32 and window.is_contiguous()):
        raise _lib.MedTokLibraryError("usage_update_multi_: window must be a contiguous fp32 device tensor")
    count = len(ids_list)
    if not 1 <= count <= _lib.USAGE_MULTI_MAX:
        raise ValueError(f"usage_update_multi_: 1..{_lib.USAGE_MULTI_MAX} updates per call")
    ids = [_dev(t.reshape(-1), "ids", torch.int64) for t in ids_list]
    ptrs = (C.c_void_p * count)(*[t.data_ptr() for t in ids])
    ms = (C.c_int64 * count)(*[t.numel() for t in ids])
    lib = _lib.load()
    if extra_word is not None and not (extra_word.is_cuda and extra_word.dtype == torch.int32 and extra_word.device == window.device):
        raise _lib.MedTokLibraryError("usage_update_multi_: extra_word must be an int32 tensor on the window's device")
    counts = torch.empty(count + (extra_word is not None), dtype=torch.int32, device=window.device)
    ws = _ws(lib.medtok_usage_multi_workspace_bytes(window.numel(), n_codes, count), window)
    with _on(window.device):
        _lib.check(lib.medtok_usage_update_multi_word(window.data_ptr(), window.numel(), ptrs, ms, count, n_codes, counts.data_ptr(), _ptr(extra_word),
                                                      ws.data_ptr(), ws.numel(), _stream(window)), "medtok_usage_update_multi")
    return counts


def normalized_search(z, what, wsq, topk: int = 1, path: int = PATH_AUTO):
    """(zhat, |zhat|^2, idx [n, topk], dist [n, topk]): F.normalize(z) and its nearest codes in one C call -- the head of
    NormEMAVectorQuantizer.forward (norm_ema_quantizer.py:169-179).  Same bits as rownorm() + topk_search()."""
    z, what, wsq = _dev(z, "z"), _dev(what, "what"), _dev(wsq, "wsq")
    n, d = z.shape
    k = what.shape[0]
    lib = _lib.load()
    dev = z.device
    zhat = torch.empty_like(z)
    zsq = torch.empty(n, dtype=torch.float32, device=dev)
    idx = torch.empty((n, topk), dtype=torch.int64, device=dev)
    dist = torch.empty((n, topk), dtype=torch.float32, device=dev)
    ws = _ws(lib.medtok_normalized_search_workspace_bytes(n, k, d, topk, path), z)
    with _on(dev):
        _lib.check(lib.medtok_normalized_search_f32(z.data_ptr(), n, d, what.data_ptr(), wsq.data_ptr(), k, topk, path, zhat.data_ptr(),
                                                    zsq.data_ptr(), idx.data_ptr(), dist.data_ptr(), ws.data_ptr(), ws.numel(), _stream(z)),
                   "medtok_normalized_search_f32")
    return zhat, zsq, idx, dist


def takes_filter_path(n: int, k_codes: int, d: int, topk: int, path: int = PATH_AUTO) -> bool:
    """does a search of this shape run the fp16 shortlist (and so profit from a prepared codebook)?"""
    return _lib.load().medtok_search_resolved_path(int(n), int(k_codes), int(d), int(topk), int(path)) == PATH_F16_FILTER


def prepare_codebook(weight: torch.Tensor, regions, normalised=None):
    """normalize(weight) plus everything the fp16-filter searches derive from it, once per weight version (include/medtok_vq.h:
    medtok_rownorm_image_f32 + medtok_codebook_prepare_f32; two launches): (what [K, d], wsq [K], {region name: prepared}) for
    regions = {name: (lo, hi)}.  A `prepared` entry goes to soft_vq_forward(..., prepared=...) together with what[lo:hi] / wsq[lo:hi].
    normalised = (what, wsq) of an earlier rownorm(weight): only the image and the regions' values are made (weight is not read)."""
    import ctypes as C
    w = _dev(weight if normalised is None else normalised[0], "weight")
    n, d = w.shape
    if not 1 <= len(regions) <= _lib.PREP_MAX_REGIONS:
        raise ValueError(f"prepare_codebook: 1..{_lib.PREP_MAX_REGIONS} regions")
    lib = _lib.load()
    dp = lib.medtok_filter_image_width(d)
    dev = w.device
    rows = n + 256
    what = torch.empty_like(w) if normalised is None else None
    wsq = torch.empty(n, dtype=torch.float32, device=dev) if normalised is None else None
    image = torch.empty((rows, dp), dtype=torch.float16, device=dev)
    descs = (_lib.RegionDesc * len(regions))()
    prepared = {}
    for i, (name, (lo, hi)) in enumerate(regions.items()):
        k = hi - lo
        if not 0 <= lo < hi <= n:
            raise ValueError(f"prepare_codebook: region {name} = [{lo}, {hi}) outside the codebook")
        wsqp = torch.empty((k + 255) // 256 * 256, dtype=torch.float32, device=dev)
        en_max = torch.empty(1, dtype=torch.float32, device=dev)
        descs[i] = _lib.RegionDesc(lo, k, wsqp.data_ptr(), en_max.data_ptr())
        prepared[name] = dict(image=image[lo:], wsqp=wsqp, en_max=en_max, k=k, d=d)
    with _on(dev):
        if normalised is None:
            _lib.check(lib.medtok_rownorm_image_f32(w.data_ptr(), n, d, what.data_ptr(), wsq.data_ptr(), image.data_ptr(), rows, dp, _stream(w)),
                       "medtok_rownorm_image_f32")
        else:
            what, wsq = w, _dev(normalised[1], "wsq")
            _lib.check(lib.medtok_codebook_image_f32(what.data_ptr(), n, d, image.data_ptr(), rows, dp, _stream(w)), "medtok_codebook_image_f32")
        _lib.check(lib.medtok_codebook_prepare_f32(wsq.data_ptr(), descs, len(regions), _stream(w)), "medtok_codebook_prepare_f32")
    return what, wsq, prepared


# Measurement hook (bench.py, tests; never set in product code): while this is a list, every soft_vq_forward that takes the fp16
# shortlist appends (its workspace, n, K, d, topk, path) -- the workspace stays alive with the entry, nothing is read back;
# filter_stats(entry) reduces it on the device afterwards.
FILTER_STATS = None


def filter_stats(entry) -> dict:
    """What the shortlist pass of the finished search behind `entry` (an element of FILTER_STATS) left behind: candidates per row,
    share of full candidate lists, rows handed to the exact kernel (include/medtok_vq.h: medtok_debug_filter_stats).  One host read."""
    ws, n, k, d, topk, path = entry
    out = torch.zeros(4, dtype=torch.int64, device=ws.device)
    with _on(ws.device):
        _lib.check(_lib.load().medtok_debug_filter_stats(ws.data_ptr(), ws.numel(), 1, n, k, d, topk, path, out.data_ptr(), _stream(ws)),
                   "medtok_debug_filter_stats")
    cand, full, fb, lists = out.cpu().tolist()
    return dict(rows=n, codes=k, candidates_per_row=cand / n, full_lists_share=full / max(lists, 1), lists_per_row=lists / n, fallback_rows=fb)


def _note_filter_stats(ws, n, k, d, topk, path):
    if FILTER_STATS is not None and n > 0 and takes_filter_path(n, k, d, topk, path):
        FILTER_STATS.append((ws, n, k, d, topk, path))


def soft_vq_forward(x, what, wsq, topk: int, path: int = PATH_AUTO, want_sqerr: bool = True, out=None, prepared=None):
    """rownorm -> search -> soft assign in one C call.
    Returns dict(xhat, idx, dist, w, zq, row_sqerr); `out` as in soft_assign.  prepared (inference, want_sqerr = False): the region's
    entry of prepare_codebook -- the same bits without the search's own passes over the codebook."""
    x, what, wsq = _dev(x, "x"), _dev(what, "what"), _dev(wsq, "wsq")
    n, d = x.shape
    k = what.shape[0]
    if prepared is not None and not want_sqerr and SEARCH_TIMER is None and n > 0:
        if prepared["k"] != k or prepared["d"] != d or prepared["image"].device != x.device:
            raise ValueError("soft_vq_forward: `prepared` belongs to another region / width / device")
        lib = _lib.load()
        dev = x.device
        xhat = torch.empty_like(x)
        idx = torch.empty((n, topk), dtype=torch.int64, device=dev)
        dist = torch.empty((n, topk), dtype=torch.float32, device=dev)
        w = torch.empty((n, topk), dtype=torch.float32, device=dev)
        zq, zstride = _zq_out(out, n, d, x)
        ws = _ws(lib.medtok_soft_vq_workspace_bytes(n, k, d, topk, path), x)
        with _on(dev):
            _lib.check(lib.medtok_soft_vq_forward_prepared_f32(x.data_ptr(), n, d, what.data_ptr(), wsq.data_ptr(), k, topk, path,
                                                               prepared["image"].data_ptr(), prepared["wsqp"].data_ptr(), prepared["en_max"].data_ptr(),
                                                               xhat.data_ptr(), idx.data_ptr(), dist.data_ptr(), w.data_ptr(),
                                                               zq.data_ptr(), zstride, ws.data_ptr(), ws.numel(), _stream(x)),
                       "medtok_soft_vq_forward_prepared_f32")
        _note_filter_stats(ws, n, k, d, topk, path)
        return dict(xhat=xhat, idx=idx, dist=dist, w=w, zq=zq, row_sqerr=None)
    if SEARCH_TIMER is not None:         # same kernels, launched piecewise so the search can be bracketed
        xhat, xsq = rownorm(x)
        idx, dist = topk_search(xhat, xsq, what, wsq, topk, path)
        w, zq, se = soft_assign(x, what, idx, dist, want_sqerr=want_sqerr, out=out)
        return dict(xhat=xhat, idx=idx, dist=dist, w=w, zq=zq, row_sqerr=se)
    lib = _lib.load()
    dev = x.device
    xhat = torch.empty_like(x)
    idx = torch.empty((n, topk), dtype=torch.int64, device=dev)
    dist = torch.empty((n, topk), dtype=torch.float32, device=dev)
    w = torch.empty((n, topk), dtype=torch.float32, device=dev)
    zq, zstride = _zq_out(out, n, d, x)
    se = torch.empty(n, dtype=torch.float32, device=dev) if want_sqerr else None
    ws = _ws(lib.medtok_soft_vq_workspace_bytes(n, k, d, topk, path), x)
    with _on(dev):
        _lib.check(lib.medtok_soft_vq_forward_f32(x.data_ptr(), n, d, what.data_ptr(), wsq.data_ptr(), k, topk, path,
                                                  xhat.data_ptr(), idx.data_ptr(), dist.data_ptr(), w.data_ptr(),
                                                  zq.data_ptr(), zstride, _ptr(se), ws.data_ptr(), ws.numel(), _stream(x)),
                   "medtok_soft_vq_forward_f32")
    _note_filter_stats(ws, n, k, d, topk, path)
    return dict(xhat=xhat, idx=idx, dist=dist, w=w, zq=zq, row_sqerr=se)
