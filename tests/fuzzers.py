"""Randomised checks of the HIP library against its own exact path / the C oracle / fp64 -- test infrastructure.

Every function takes (cases, seed, budget_s): it stops after `cases` cases or `budget_s` seconds, whichever comes first, and returns
(cases_run, mismatches: list[str]).  tests/test_gpu_fuzz.py runs them time-boxed with fixed seeds inside `-m gpu`; tools/fuzz_search.py
is their command line (long runs, variant builds of the library).

Why they are in the suite: round 4's three wrong-result bugs (asm readers of MFMA results inside their wait states; the start-value
race of the general filter kernel at D <= 32; a cross-stream buffer overwrite) were all found by these loops, none by a fixed-shape
test."""
from __future__ import annotations

import math
import random
import time

import numpy as np
import torch


def _timer(budget_s):
    t0 = time.time()
    return (lambda: budget_s is not None and time.time() - t0 > budget_s), (lambda: time.time() - t0)


def fuzz_search(cases=100, seed=0, budget_s=None, max_rows=300000, log=None):
    """Random shapes / distributions / plan bits: the fp16-filter path (general kernel, every D from 4 to 1028) against the exact
    fp32 path -- ids and distances must be the same bits."""
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    rows_all = [1, 7, 255, 256, 257, 1000, 4097, 20000, 70001, 300000]
    # every third case is a shape that exposed a past bug: rows of <= 32 elements against many code tiles (round 4's start-value race
    # of the general kernel showed in ~1 % of such searches: they are what repetition is for)
    hot = [(4097, 20001, 32, 1), (20000, 20001, 4, 1), (4097, 20001, 32, 5), (4097, 8191, 16, 1), (20000, 16384, 16, 5)]
    for c in range(cases):
        if over():
            break
        n = int(rng.choice([r for r in rows_all[: 9 if c % 10 else 10] if r <= max_rows]))
        K = int(rng.choice([1, 5, 31, 256, 257, 1000, 4096, 8191, 16384, 20001]))
        D = int(rng.choice([4, 16, 32, 60, 64, 100, 128, 260, 768, 1028]))
        k = int(rng.choice([1, 2, 5, 8]))
        if c % 3 == 2:
            n, K, D, k = hot[(c // 3) % len(hot)]
        if k > K:
            k = 1
        kind = int(rng.integers(0, 5))
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(n, D, device=dev, generator=g)
        W = torch.randn(K, D, device=dev, generator=g)
        if kind == 1:
            W[K // 2:] = W[: K - K // 2].clone()                                        # duplicated codes (ties)
        if kind == 2:
            x = x * 0.01 + W[torch.randint(0, K, (n,), device=dev, generator=g)]        # rows close to codes
        if kind == 3:
            W = W * torch.rand(K, 1, device=dev, generator=g) * 3                       # un-normalised codes
        if kind == 4:
            x[::7] = 0                                                                  # zero rows
        xh, xs = ops.rownorm(x)
        if kind == 3:
            wh, ws = W.contiguous(), ops.rownorm(W, normalize=False)[1]
        else:
            wh, ws = ops.rownorm(W)
        i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
        plans = ({}, dict(filter_splits=int(rng.choice([1, 2, 4, 8])), filter_xcd=bool(rng.integers(0, 2))),
                 dict(filter_tail=bool(rng.integers(0, 2)), filter_rows64=False, filter_splits=int(rng.choice([0, 1, 2, 3]))),
                 dict(search_max_splits=int(rng.choice([1, 3, 16]))))
        for env in plans:
            path = ops.PATH_F32_MFMA if "search_max_splits" in env else ops.PATH_F16_FILTER
            i1, d1 = ops.topk_search(xh, xs, wh, ws, k, ops.plan_path(path, **env))
            if not (torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))):
                bad.append(f"search case {c}: n={n} K={K} D={D} k={k} kind={kind} env={env} rows differing {(i0 != i1).any(1).sum().item()}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_wide_k(cases=100, seed=0, budget_s=None, log=None):
    """More than 8 codes per row (two exact passes of lists of 8 + a join, round 6) against the C oracle's single list of k, bit for bit:
    random shapes, duplicated codes (ties that straddle the pass boundary), rows close to codes, zero rows, forced code splits."""
    from medtok_amd import ops
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, _ = _timer(budget_s)
    bad, c = [], 0
    for c in range(cases):
        if over():
            break
        n = int(rng.choice([1, 33, 257, 1000, 3000]))
        K = int(rng.choice([16, 100, 1000, 5000, 20001]))
        D = int(rng.choice([4, 36, 64, 128, 260]))
        k = int(rng.integers(9, 17))
        if n * K * D > 3.0e9:
            n = max(1, int(3.0e9 / (K * D)))
        kind = int(rng.integers(0, 5))
        x = rng.standard_normal((n, D), dtype=np.float32)
        W = rng.standard_normal((K, D), dtype=np.float32)
        if kind == 1:
            W[K // 3:] = np.resize(W[: max(K // 3, 1)], (K - K // 3, D))             # every code about three times: exact ties
        if kind == 2:
            x = (x * 0.01 + W[rng.integers(0, K, n)]).astype(np.float32)              # rows close to codes
        if kind == 4:
            x[::7] = 0                                                                # zero rows
        xh, xs = O.rownorm(x)
        wh, ws = O.rownorm(W)
        ri, rd = O.topk_search(xh, xs, wh, ws, k)
        T = lambda a: torch.from_numpy(a).to(dev)
        for env in ({}, dict(search_max_splits=int(rng.choice([1, 3, 16])))):
            gi, gd = ops.topk_search(T(xh), T(xs), T(wh), T(ws), k, ops.plan_path(ops.PATH_AUTO if not env else ops.PATH_F32_MFMA, **env))
            if not (np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy().view(np.int32), rd.view(np.int32))):
                bad.append(f"wide_k case {c}: n={n} K={K} D={D} k={k} kind={kind} env={env} rows differing {(gi.cpu().numpy() != ri).any(1).sum()}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_rows64(cases=100, seed=0, budget_s=None, max_rows=200000, log=None):
    """The filter kernel for rows of <= 64 elements (learning tiles, scanned tiles, revisited tiles; code splits; ragged last tiles;
    every k-list length) against the exact fp32 path and the general filter kernel: same bits."""
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    rows_all = [1, 127, 128, 129, 513, 4097, 20000, 70001, 200000]
    for c in range(cases):
        if over():
            break
        n = int(rng.choice([r for r in rows_all[: 8 if c % 8 else 9] if r <= max_rows]))
        tiles = int(rng.choice([1, 2, 3, 7, 8, 9, 12, 16, 17, 33, 40, 83]))
        K = max(1, 256 * tiles + int(rng.choice([-255, -100, -1, 0, 0, 0])))
        D = int(rng.choice([4, 16, 32, 36, 40, 60, 64, 64, 64]))
        k = int(rng.choice([1, 2, 5, 5, 8]))
        if k > K:
            k = 1
        kind = int(rng.integers(0, 5))
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(n, D, device=dev, generator=g)
        W = torch.randn(K, D, device=dev, generator=g)
        if kind == 1:
            W[K // 2:] = W[: K - K // 2].clone()
        if kind == 2:
            x = x * 0.01 + W[torch.randint(0, K, (n,), device=dev, generator=g)]
        if kind == 3 and K >= 10:                                                       # near-copies of neighbours
            m = min(W[::5].shape[0], W[1::5].shape[0])
            W[::5][:m] = W[1::5][:m] + 1e-3 * torch.randn(m, D, device=dev, generator=g)
        if kind == 4:
            x[::7] = 0
        xh, xs = ops.rownorm(x)
        wh, ws = ops.rownorm(W)
        i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
        for env in (dict(filter_rows64=True), dict(filter_rows64=True, filter_splits=int(rng.choice([1, 2, 4, 8]))), dict(filter_rows64=False),
                    dict(filter_rows64="wide", filter_splits=int(rng.choice([0, 1, 2, 4])))):      # both wave-tile forms of the narrow-row kernel
            i1, d1 = ops.topk_search(xh, xs, wh, ws, k, ops.plan_path(ops.PATH_F16_FILTER, **env))
            if not (torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))):
                bad.append(f"rows64 case {c}: n={n} K={K} D={D} k={k} kind={kind} env={env} rows differing {(i0 != i1).any(1).sum().item()}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_attention(cases=60, seed=0, budget_s=None, log=None):
    """Random ragged shapes through the attention core -- the exact fp32 kernel, the split-fp16 kernel, every variant of the wide-batch
    kernels on (hi, lo) images, variant 2 fed fp32 rows (keys split inside the kernel) and fp16 rows as they stand (no lo image) --
    against the C oracle (1e-5 of the output scale); the in-kernel split must equal the image pass bit for bit.  Every other case also runs
    the training kernels (forward with dropout, fp32 and one-pass backward, dQ-only + multi-source dKV) against the oracle."""
    from medtok_amd import ops
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    T = lambda a: torch.from_numpy(a).to(dev)
    for c in range(cases):
        if over():
            break
        d = int(rng.choice([64, 128, 256, 384, 512, 640, 768]))
        n_codes = int(rng.integers(1, 9))
        q_len = rng.integers(0, 200, n_codes).astype(np.int64)
        kv_len = rng.integers(0, 300, n_codes).astype(np.int64)
        if c % 7 == 0:
            kv_len[0] = 0
        if q_len.sum() == 0:
            q_len[0] = 5
        q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
        q = (rng.standard_normal((int(q_len.sum()), d)) * rng.choice([0.05, 0.3, 1.0])).astype(np.float32)
        kv = rng.standard_normal((max(int(kv_len.sum()), 1), d)).astype(np.float32)
        scale = float(rng.choice([0.07, 0.125, 0.25]))
        ref = O.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale)
        err = 0.0
        for exact in (False, True):
            out = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), scale, exact).cpu().numpy()
            err = max(err, np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-30))
        if d in ops.ATTENTION_SPLIT_WIDTHS and kv_len.sum() > 0:
            img = ops.split_half(T(kv))
            a = (T(q), T(q_start), T(q_len))
            b = (T(kv_start), T(kv_len), int(q_len.max()), scale)
            touched = ~np.isnan(ref).all(1)
            td = torch.from_numpy(touched).to(dev)
            sc = max(np.abs(ref[touched]).max(), 1e-30)
            for v in ((0, 1, 2) if d == 768 else ((0, 2) if d in ops.ATTENTION_HALF_KEY_WIDTHS else (0,))):
                out = ops.shared_kv_attention_split(*a, img, *b, variant=v)
                err = max(err, np.abs(out.cpu().numpy() - ref)[touched].max() / sc)
                if v == 2:
                    own = ops.shared_kv_attention_split(*a, T(kv), *b, variant=2)          # KF32: keys split inside the kernel
                    if not torch.equal(own[td], out[td]):
                        err = float("inf")
                    # KLO = false: fp16 keys as they stand -- compare with the oracle on the keys widened back to fp32
                    kv16 = T(kv).half()
                    ref16 = O.shared_kv_attention(q, q_start, q_len, kv16.float().cpu().numpy(), kv_start, kv_len, scale)
                    half = ops.shared_kv_attention_split(*a, (kv16, None), *b, variant=2)
                    err = max(err, np.abs(half.cpu().numpy() - ref16)[touched].max() / max(np.abs(ref16[touched]).max(), 1e-30))
        if not (err <= 1e-5):
            bad.append(f"attention case {c}: d={d} codes={n_codes} q_len={q_len.tolist()} kv_len={kv_len.tolist()} rel err {err:.3g}")
            if log:
                log(bad[-1])
        if c % 2 == 0 and kv_len.sum() > 0:
            # the TRAINING kernels on the same shapes (round 6: the fp32 dKV kernel was wrong at D = 640 and no test ran that width):
            # forward with dropout + backward against the oracle (1e-5), the split forward where it exists, the one-pass backward
            # against the fp32 one (its own tolerance), the key gradient in one multi-source launch against the plain launch
            p_drop, seed = float(rng.choice([0.0, 0.1, 0.3])), int(rng.integers(1 << 30))
            d_out = rng.standard_normal(q.shape).astype(np.float32)
            args = (T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len))
            mq, mk = int(q_len.max()), int(kv_len.max())
            out_o, lse_o, dq_o, dkv_o = O.shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, scale, p_drop, seed, d_out)
            out, lse = ops.shared_kv_attention_train(*args, mq, scale, p_drop, seed)
            dq, dkv = ops.shared_kv_attention_backward(*args, mq, mk, scale, p_drop, seed, out, lse, T(d_out))
            # (relative to the gradient's largest entry, or -- a code with ONE key has P = 1 and a key-side gradient that is zero in exact
            # arithmetic -- to a tenth of its natural scale |dO| |kv|^2 scale sqrt(d): dP - delta then is pure rounding)
            nat = 0.1 * scale * float(np.abs(kv).max()) ** 2 * float(np.abs(d_out).max()) * d ** 0.5
            rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), nat))
            errs = dict(out=rel(out.cpu().numpy(), out_o), dq=rel(dq.cpu().numpy(), dq_o), dkv=rel(dkv.cpu().numpy(), dkv_o))
            if d in ops.ATTENTION_TRAIN_SPLIT_WIDTHS:
                out_s, _ = ops.shared_kv_attention_train(*args, mq, scale, p_drop, seed, split=True)
                errs["out_split"] = rel(out_s.cpu().numpy(), out_o)
            half = torch.bfloat16 if c % 4 == 0 else torch.float16
            dq_h, dkv_h = ops.shared_kv_attention_backward(*args, mq, mk, scale, p_drop, seed, out, lse, T(d_out), half=half)
            tol_h = 6e-2 if half == torch.bfloat16 else 1e-2        # (operands rounded to 8 / 11 bits; sharply peaked rows reach 5e-3 in fp16)
            dq1, delta = ops.shared_kv_attention_backward_dq(*args, mq, mk, scale, p_drop, seed, out, lse, T(d_out))
            multi = ops.shared_kv_attention_dkv_multi([dict(q=T(q), d_out=T(d_out), lse=lse, delta=delta, q_start=args[1], q_len=args[2], scale=scale,
                                                            dropout_p=p_drop, seed=seed)], T(kv), args[4], args[5], mk)
            wrong = [k for k, v in errs.items() if not v <= 1e-5]
            e_h = (rel(dq_h.cpu().numpy(), dq_o), rel(dkv_h.cpu().numpy(), dkv_o))
            if not (e_h[0] <= tol_h and e_h[1] <= tol_h):
                wrong.append(f"half backward ({half}): dq {e_h[0]:.3g} dkv {e_h[1]:.3g} (tolerance {tol_h})")
            if not (torch.equal(dq1, dq) and torch.equal(multi, dkv)):
                wrong.append("dq-only / multi-source dKV differ from the plain launch")
            if wrong:
                bad.append(f"attention (training) case {c}: d={d} codes={n_codes} q_len={q_len.tolist()} kv_len={kv_len.tolist()} p={p_drop}: {wrong} {errs}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_split_gemm(cases=200, seed=0, budget_s=None, log=None):
    """Random shapes through medtok_split_gemm_f16 (plain and grouped, both tile heights, every output combination) against fp64; every
    third shape also through the one-pass half-precision product (medtok_half_gemm_f32)."""
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    for c in range(cases):
        if over():
            break
        groups = rng.choice([1, 1, 1, 2, 4, 3])
        k_g = 32 * rng.randint(1, 24)
        n_g = 4 * rng.randint(1, 200) if rng.random() < 0.7 else rng.choice([64, 128, 192, 256, 384, 768])
        m = rng.choice([1, 7, 255, 256, 257, 1000, 4096, 5000, rng.randint(1, 70000)])
        a_cols = k_g if groups == 1 else k_g + 8 * rng.randint(0, 3)          # column stride between the groups' slices of A
        b_rows = n_g if groups == 1 else n_g + 4 * rng.randint(0, 5)
        lda = (groups - 1) * a_cols + k_g + 8 * rng.randint(0, 2)
        g = torch.Generator(device=dev).manual_seed(seed * 100003 + c)
        a = torch.randn(m, lda, device=dev, generator=g)
        w = torch.randn((groups - 1) * b_rows + n_g, k_g, device=dev, generator=g) / k_g ** 0.5
        bias = torch.randn(groups * n_g, device=dev, generator=g) if rng.random() < 0.6 else None
        amax = float(w.abs().max())
        scale = 2.0 ** (11 - math.floor(math.log2(amax)))
        ws = ops.split_half(w.contiguous(), dp=k_g, scale=scale)
        want_f32, want_split = rng.choice([(True, False), (False, True), (True, True)])
        cf, cs = ops.split_gemm(ops.split_half(a), ws, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=a_cols, b_group_rows=b_rows, bias=bias,
                                unscale=1.0 / scale, want_f32=want_f32, want_split=want_split)
        ref = torch.cat([a[:, h * a_cols: h * a_cols + k_g].double() @ w[h * b_rows: h * b_rows + n_g].double().t() for h in range(groups)], 1)
        if bias is not None:
            ref = ref + bias.double()
        sc = float(ref.abs().max()) + 1e-30
        errs = []
        if cf is not None:
            errs.append(float((cf.double() - ref).abs().max()) / sc)
        if cs is not None:
            errs.append(float((cs[0].double() + cs[1].double() - ref).abs().max()) / sc)
        if not (max(errs) <= 4e-6):
            bad.append(f"split_gemm case {c}: " + str(dict(m=m, n_g=n_g, k_g=k_g, groups=groups, a_cols=a_cols, b_rows=b_rows, lda=lda,
                                                            bias=bias is not None, f32=want_f32, split=want_split)) + f" rel err {max(errs):.3g}")
            if log:
                log(bad[-1])
        if c % 3 == 0:
            # the one-pass half-precision product of the same shape (medtok_half_gemm_f32: 64-deep stages where k_g % 64 == 0, the dense
            # tile order where the row tiles are few) on the operands rounded to 16 bits, against fp64 on the rounded operands
            dt = torch.bfloat16 if c % 2 else torch.float16
            a16, w16 = a.to(dt).contiguous(), w.to(dt).contiguous()
            ch = ops.half_gemm(a16, w16, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=a_cols, b_group_rows=b_rows, bias=bias)
            ref16 = torch.cat([a16[:, h * a_cols: h * a_cols + k_g].double() @ w16[h * b_rows: h * b_rows + n_g].double().t() for h in range(groups)], 1)
            if bias is not None:
                ref16 = ref16 + bias.double()
            err = float((ch.double() - ref16).abs().max()) / (float(ref16.abs().max()) + 1e-30)
            if not err <= 4e-6:
                bad.append(f"half_gemm case {c}: " + str(dict(m=m, n_g=n_g, k_g=k_g, groups=groups, a_cols=a_cols, b_rows=b_rows, lda=lda, dt=str(dt))) + f" rel err {err:.3g}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_small_width(cases=100, seed=0, budget_s=None, log=None):
    """Random ragged batches through the two-launch cross-attention at e_dim = 64 (tiles that span many codes, codes without nodes /
    without valid tokens, node counts around the tile size, every mask dtype, one to three layers) against the layer-by-layer
    product path: 1e-5 of the output scale; and the device-side status word stays clear for sorted batch vectors."""
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    for c in range(cases):
        if over():
            break
        bsz = int(rng.choice([1, 2, 7, 33, 64, 200]))
        seq_len = int(rng.choice([1, 5, 31, 32, 33, 100, 512]))
        max_nodes = int(rng.choice([1, 3, 8, 9, 17, 40, 90]))
        layers = int(rng.choice([1, 2, 2, 3]))
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        torch.manual_seed(int(rng.integers(1 << 30)))
        ca = CrossAttention(64, 4, dropout=0.1, layers=layers).to(dev).eval()
        with torch.no_grad():
            for layer in ca.model:
                layer.multihead_attn.in_proj_bias.normal_(0, 0.3)
                layer.multihead_attn.out_proj.bias.normal_(0, 0.3)
                layer.layer_norm.weight.normal_(1.0, 0.3)
                layer.layer_norm.bias.normal_(0, 0.3)
        text = torch.randn(bsz, seq_len, 64, device=dev, generator=g) * float(rng.choice([0.1, 1.0, 3.0]))
        tok = torch.randint(0 if c % 3 == 0 else 1, seq_len + 1, (bsz,), device=dev, generator=g)
        n_nodes = torch.randint(0 if c % 2 == 0 else 1, max_nodes + 1, (bsz,), device=dev, generator=g)
        mask = torch.arange(seq_len, device=dev)[None, :] < tok[:, None]
        mask = mask if c % 4 == 0 else mask.to([torch.int64, torch.int32, torch.uint8][c % 3])
        batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
        nodes = torch.randn(int(n_nodes.sum()), 64, device=dev, generator=g)
        with torch.no_grad():
            both = ca.pooled_small(text, mask, nodes, batch)
            try:
                ca.check_small_status()
                flagged = False
            except ValueError:
                flagged = True
            pt, pg = ca.pooled(text, mask, nodes, batch)
        err = max(float((both[:, 0] - pt).abs().max()) / max(float(pt.abs().max()), 1e-30),
                  float((both[:, 1] - pg).abs().max()) / max(float(pg.abs().max()), 1e-30))
        if flagged or not (err <= 1e-5):
            bad.append(f"small_width case {c}: B={bsz} L={seq_len} max_nodes={max_nodes} layers={layers} rel err {err:.3g} flagged={flagged}")
            if log:
                log(bad[-1])
    return c + 1, bad


def fuzz_multi_search(cases=100, seed=0, budget_s=None, log=None):
    """Random batched search calls (1 to 6 searches, ragged row and code counts, region slices, k in {1, 2, 5, 8}, widths with and without
    a k tail, strided x, strided zq output) against the single calls: every output bit for bit."""
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    for c in range(cases):
        if over():
            break
        d = int(rng.choice([4, 36, 64, 64, 128, 200, 768]))
        topk = int(rng.choice([1, 2, 5, 5, 8]))
        n_e = int(rng.choice([300, 1000, 3000, 21000])) if d <= 128 else int(rng.choice([300, 1000, 4000]))
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        what, wsq = ops.rownorm(torch.randn(n_e, d, device=dev, generator=g))
        count = int(rng.integers(1, 7))
        searches = []
        for i in range(count):
            n = int(rng.choice([1, 2, 31, 128, 129, 256, 500, 1000]))
            lo = int(rng.integers(0, n_e - max(topk, 8)))
            hi = int(rng.integers(lo + max(topk, 8), n_e + 1))
            if not ops.multi_search_eligible(n, hi - lo, d, topk):
                continue
            wide = torch.randn(n, 2 * d, device=dev, generator=g)
            x = wide[:, d:] if (i % 2 and d % 4 == 0 and (d * 4) % 16 == 0) else torch.randn(n, d, device=dev, generator=g)
            out = torch.empty(n, 3 * d, device=dev)[:, d:2 * d] if i % 3 == 0 else None
            searches.append(dict(x=x, what=what[lo:hi], wsq=wsq[lo:hi].contiguous(), out=out))
        if not searches:
            continue
        train = bool(c % 2)                         # every other case as the training forward calls it: + per-row squared errors
        res = ops.soft_vq_forward_multi(searches, topk, want_sqerr=train)
        for i, (q, r) in enumerate(zip(searches, res)):
            one = ops.soft_vq_forward(q["x"].contiguous(), q["what"], q["wsq"], topk, want_sqerr=train)
            diff = [key for key in ("xhat", "idx", "dist", "w", "zq") + (("row_sqerr",) if train else ()) if not torch.equal(one[key], r[key])]
            if diff:
                bad.append(f"multi_search case {c} search {i}: n={q['x'].shape[0]} K={q['what'].shape[0]} d={d} k={topk} differs in {diff}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def fuzz_prepared(cases=100, seed=0, budget_s=None, log=None):
    """Random codebooks prepared once (ops.prepare_codebook: random overlapping regions, widths the image pads, from raw rows and from
    rows normalised earlier) and random one-call forwards on them -- both filter kernels, the few-rows and the block re-score, zero
    rows, near-copies of codes -- against the same forwards without the prepared arguments: every output bit for bit."""
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    over, elapsed = _timer(budget_s)
    bad, c = [], 0
    for c in range(cases):
        if over():
            break
        d = int(rng.choice([4, 36, 64, 64, 100, 128, 768]))
        topk = int(rng.choice([1, 2, 5, 5, 8]))
        n_e = int(rng.choice([600, 3000, 7001, 21000])) if d <= 128 else int(rng.choice([600, 3000, 6144]))
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        W = torch.randn(n_e, d, device=dev, generator=g)
        if c % 4 == 1:
            W[n_e // 2:] = W[: n_e - n_e // 2] + 1e-3 * torch.randn(n_e - n_e // 2, d, device=dev, generator=g)
        if c % 5 == 2:
            W[3] = 0
        regions = {}
        for name in ("a", "b", "c")[: int(rng.integers(1, 4))]:
            lo = int(rng.integers(0, n_e - max(topk, 8)))
            regions[name] = (lo, int(rng.integers(lo + max(topk, 8), n_e + 1)))
        regions["all"] = (0, n_e)
        if c % 2:
            what, wsq = ops.rownorm(W)
            _, _, prepared = ops.prepare_codebook(None, regions, normalised=(what, wsq))
        else:
            what, wsq, prepared = ops.prepare_codebook(W, regions)
        for name, (lo, hi) in regions.items():
            n = int(rng.choice([1, 130, 4097, 20000, 70001]))
            x = torch.randn(n, d, device=dev, generator=g)
            if c % 3 == 0:
                x[::5] = 0
            out = torch.empty(n, 2 * d, device=dev)[:, :d] if (c % 2 and d % 4 == 0) else None
            kw = dict(want_sqerr=False)
            ref = ops.soft_vq_forward(x, what[lo:hi], wsq[lo:hi].contiguous(), topk, ops.PATH_F16_FILTER, **kw)
            got = ops.soft_vq_forward(x, what[lo:hi], wsq[lo:hi].contiguous(), topk, ops.PATH_F16_FILTER, out=out, prepared=prepared[name], **kw)
            diff = [key for key in ("xhat", "idx", "dist", "w", "zq") if not torch.equal(ref[key], got[key])]
            if diff:
                bad.append(f"prepared case {c} region {name}=[{lo},{hi}): n={n} K={hi - lo} d={d} k={topk} differs in {diff}")
                if log:
                    log(bad[-1])
    return c + 1, bad


def _same_outputs(a, b):
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, torch.Tensor):
            if not torch.equal(x, y):
                bad.append(k)
        elif isinstance(x, tuple):
            for i, (p, q) in enumerate(zip(x, y)):
                if isinstance(p, torch.Tensor) and not torch.equal(p, q):
                    bad.append(f"{k}[{i}]")
        elif x != y:
            bad.append(k)
    return bad


def soak_forward(runs=30, seed=0, budget_s=None, codes=4096, log=None):
    """The multi-stream inference forward at BASELINE sizes, repeatedly, under varying amounts of unrelated work in flight: every
    output of every run must be bit-identical to the first run and to the single-stream forward (a missing stream dependency shows
    up as a difference, eventually)."""
    import bench
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd import ops
    dev = torch.device("cuda:0")
    over, elapsed = _timer(budget_s)
    w = bench.Full(codes, dev, seed, ops.PATH_AUTO)
    torch.manual_seed(seed)
    vq = vqmod.VectorQuantizer(w.N_E, w.D, 0.25, 0.0, True, True, [w.D, w.D], k=w.TOPK).to(dev).eval()      # (with the usage window)

    def forward():
        vq._norm_cache = None
        vq.codebook_used.zero_()
        with torch.no_grad():
            out = vq(w.h, w.text, w.nodes, w.mask, w.batch)
        torch.cuda.synchronize()
        return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    keep = vqmod.SIDE_STREAM_MIN_CODES
    vqmod.SIDE_STREAM_MIN_CODES = 0
    try:
        single = forward()
    finally:
        vqmod.SIDE_STREAM_MIN_CODES = keep
    first = forward()
    bad = [f"soak: multi-stream differs from single-stream in {d}" for d in [_same_outputs(single, first)] if d]
    i = 0
    for i in range(runs):
        if over():
            break
        junk = [torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev) for _ in range(i % 4)]   # noqa: F841
        d = _same_outputs(first, forward())
        if d:
            bad.append(f"soak run {i} differs in {d}")
            if log:
                log(bad[-1])
    return i + 1, bad
