"""GPU: the fp16-filter search path must return the SAME bits as the exact fp32 path (and the oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def both_paths(xh, xs, wh, ws, topk):
    from medtok_amd import ops
    i1, d1 = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA)
    i2, d2 = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F16_FILTER)
    torch.cuda.synchronize()
    return i1, d1, i2, d2


def test_filter_score_error_bound(oracle, dev):
    """|s~ - s| must stay far inside the bound filter_eps() assumes (filter_f16.h): gamma*sqrt(xsq*wsq) for the fp16 rounding and the
    accumulation, plus (D + 64) 2^-23 |e|^2 for accumulators that start at -2^15 |e|^2 -- on random and on worst-case-sign data.
    The scores are the ones the search itself tests: debug_filter_scores runs the production accumulation."""
    from medtok_amd import ops
    rng = np.random.default_rng(0)
    for d in (64, 768, 1000):
        x = rng.standard_normal((96, d), dtype=np.float32)
        W = rng.standard_normal((200, d), dtype=np.float32)
        x[:8] = np.abs(x[:8]); W[:8] = np.abs(W[:8])                 # same-sign rows: sum|x e| = |x||e|-ish
        x[8] = 1.0; W[8] = 1.0                                       # exactly aligned constant vectors
        x[9, :] = 0; x[9, 0] = 1.0; W[9, :] = 0; W[9, 0] = 1.0       # one-hot
        x[10] *= rng.random(d, dtype=np.float32) < 0.05              # sparse row with tiny elements elsewhere
        xh, xs = oracle.rownorm(x); wh, ws = oracle.rownorm(W)
        s_ref = oracle.scores(xh, wh)
        s_apx = ops.debug_filter_scores(_t(xh, dev), _t(xs, dev), _t(wh, dev), _t(ws, dev)).cpu().numpy()
        gamma = 2.0 ** -10 + 2.0 ** -20 + d * 2.0 ** -22
        start = (d + 64) * 2.0 ** -23 * ws[None, :]
        bound = gamma * np.sqrt(xs[:, None] * ws[None, :]) + start
        ratio = np.abs(s_apx - s_ref) / bound
        assert ratio.max() < 0.6, (d, ratio.max())                   # rounding part alone can reach ~0.5 when aligned
        # the accumulation budget (D * 2^-22) on its own: compare against fp64 dot of the ROUNDED operands
        xr = (xh.astype(np.float16 if False else np.float32) * 256).astype(np.float16).astype(np.float64) / 256
        wr = (wh * 256).astype(np.float16).astype(np.float64) / 256
        acc_err = np.abs(s_apx - xr @ wr.T) / (d * 2.0 ** -22 * (np.abs(xr) @ np.abs(wr).T) + start)
        assert acc_err.max() < 0.25, (d, acc_err.max())     # (measured < 0.01: tools/r05/measure_filter_error.py)


@pytest.mark.parametrize("n,k,d,topk", [
    (300, 1100, 64, 5), (1000, 4096, 768, 5), (513, 3000, 100, 1), (2000, 2048, 128, 8),
    (257, 1025, 36, 3), (70000, 2048, 64, 5), (4096, 8192, 768, 5),
])
def test_filter_equals_exact_random(oracle, dev, n, k, d, topk):
    rng = np.random.default_rng(n + k + d)
    x = rng.standard_normal((n, d), dtype=np.float32); W = rng.standard_normal((k, d), dtype=np.float32)
    from medtok_amd import ops
    xh, xs = ops.rownorm(_t(x, dev)); wh, ws = ops.rownorm(_t(W, dev))
    i1, d1, i2, d2 = both_paths(xh, xs, wh, ws, topk)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    if n * k * d <= 4096 * 8192 * 64:
        io, do = oracle.topk_search(xh.cpu().numpy(), xs.cpu().numpy(), wh.cpu().numpy(), ws.cpu().numpy(), topk)
        assert np.array_equal(i2.cpu().numpy(), io) and np.array_equal(d2.cpu().numpy(), do)


def test_filter_adversarial_codebooks(oracle, dev):
    """Near-duplicate clusters (shortlists overflow -> exact fallback), exact duplicates (tie rule),
    and rows equal to codes; all must match the exact path bit for bit."""
    from medtok_amd import ops
    rng = np.random.default_rng(7)
    d, k, n = 128, 4096, 1500
    centers = rng.standard_normal((8, d), dtype=np.float32)
    W = centers[rng.integers(0, 8, k)] + 1e-3 * rng.standard_normal((k, d), dtype=np.float32)   # 8 tight clusters of ~512 codes
    W[2000:2100] = W[100:200]                                                                    # exact duplicates
    x = centers[rng.integers(0, 8, n)] + 1e-3 * rng.standard_normal((n, d), dtype=np.float32)
    x[:50] = W[100:150]                                                                          # rows that ARE codes
    xh, xs = ops.rownorm(_t(x, dev)); wh, ws = ops.rownorm(_t(W, dev))
    i1, d1, i2, d2 = both_paths(xh, xs, wh, ws, 5)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    io, do = oracle.topk_search(xh.cpu().numpy(), xs.cpu().numpy(), wh.cpu().numpy(), ws.cpu().numpy(), 5)
    assert np.array_equal(i2.cpu().numpy(), io) and np.array_equal(d2.cpu().numpy(), do)
    assert (io[:50, 0] == np.arange(100, 150)).all() and (io[:50, 1] == np.arange(2000, 2050)).all()


def test_filter_out_of_range_norms_fall_back(oracle, dev):
    """Un-normalised operands (|x|^2 or |e|^2 > 4) leave the range the error bound assumes: exact path takes over."""
    from medtok_amd import ops
    rng = np.random.default_rng(9)
    x = rng.standard_normal((600, 64), dtype=np.float32) * 3
    W = rng.standard_normal((2048, 64), dtype=np.float32)
    xd, Wd = _t(x, dev), _t(W, dev)
    _, xs = ops.rownorm(xd, normalize=False); wh, ws = ops.rownorm(Wd)
    i1, d1, i2, d2 = both_paths(xd, xs, wh, ws, 5)            # big rows, unit codes
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    xh, xs = ops.rownorm(xd); _, ws = ops.rownorm(Wd * 5, normalize=False)
    i1, d1, i2, d2 = both_paths(xh, xs, (Wd * 5).contiguous(), ws, 5)   # unit rows, big codes
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    # some rows huge, some unit, some zero
    xm = x.copy(); xm[::3] /= np.linalg.norm(xm[::3], axis=1, keepdims=True); xm[5] = 0
    xmd = _t(xm, dev); _, xs = ops.rownorm(xmd, normalize=False)
    i1, d1, i2, d2 = both_paths(xmd, xs, wh, ops.rownorm(wh, normalize=False)[1], 5)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


def test_auto_path_matches_exact_in_modules(dev):
    """The drop-in classes use PATH_AUTO; forcing the exact path must not change a single output bit."""
    from medtok_amd import ops
    from medtok_amd.inference import quantize_pooled
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(0)
    v = VectorQuantizer(3 * 2048, 128, 0.25, 0.0, True, False, [128, 128]).to(dev).eval()
    h = torch.randn(3000, 256, device=dev); pt = torch.randn(3000, 128, device=dev); pg = torch.randn(3000, 128, device=dev)
    a = quantize_pooled(v, h, pt, pg)
    v.search_path = ops.PATH_F32_MFMA
    b = quantize_pooled(v, h, pt, pg)
    for u, w in zip(a, b):
        assert torch.equal(u, w)


def test_filter_fallback_sizes(oracle, dev):
    """Few rows falling back (split-code redo) and many rows falling back (> FB_ROWS: plain redo) both stay exact."""
    from medtok_amd import ops
    rng = np.random.default_rng(21)
    d, k = 64, 2048
    W = rng.standard_normal((k, d), dtype=np.float32)
    wh, ws = ops.rownorm(_t(W, dev))
    for n, n_big in ((3000, 7), (20000, 12000)):
        x = rng.standard_normal((n, d), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        sel = rng.choice(n, n_big, replace=False)
        x[sel] *= 3.0                                   # |x|^2 = 9 > 4: these rows must take the exact path
        xd = _t(x, dev)
        _, xs = ops.rownorm(xd, normalize=False)
        i1, d1, i2, d2 = both_paths(xd, xs, wh, ws, 5)
        assert torch.equal(i1, i2) and torch.equal(d1, d2)


def test_filter_race_screen(dev):
    """Repeated large launches against the exact path: the LDS-DMA ring is ordered only by counted vmcnt waits and
    raw barriers, and a misplaced wait shows up as rare wrong tiles that come and go with shape and memory load."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    shapes = [(150000, 8192, 768), (90000, 16384, 768), (300000, 4096, 256), (60000, 49152, 768), (200000, 21000, 64)]
    for rep in range(3):
        for n, k, d in shapes:
            x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
            xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
            i2, d2 = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F16_FILTER)
            i1, d1 = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
            assert torch.equal(i1, i2) and torch.equal(d1, d2), (rep, n, k, d)


def _filter_path(**plan):
    """PATH_F16_FILTER with launch-plan branches forced for that one call (MEDTOK_PLAN_* bits of `path`; no process state)."""
    from medtok_amd import ops
    return ops.plan_path(ops.PATH_F16_FILTER, **plan)


@pytest.mark.parametrize("splits", [2, 4, 8])
def test_xcd_block_order_gives_the_same_bits(dev, splits):
    """The XCD-aware block order (default from 1024 row tiles up) forced at a small size: every (row tile, split) pair must be
    visited exactly once -- ids and distances equal the exact path, including a row count that leaves XCD chunks partly empty."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(splits)
    n, K, D = 256 * 37 + 11, 256 * 24, 256
    xh, xs = ops.rownorm(torch.randn(n, D, device=dev, generator=g))
    wh, ws = ops.rownorm(torch.randn(K, D, device=dev, generator=g))
    i_ref, d_ref = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
    i_x, d_x = ops.topk_search(xh, xs, wh, ws, 5, _filter_path(filter_splits=splits, filter_xcd=True))
    assert torch.equal(i_x, i_ref) and torch.equal(d_x, d_ref)


@pytest.mark.parametrize("case", ["plain", "all_rows_to_exact_path", "some_rows_overflow", "strided_out"])
def test_fused_assignment_equals_separate_kernels(dev, case):
    """One-call forward on the filter path: the re-score kernel does the soft assignment itself (rows the filter hands to the
    exact kernel get it through the device-side row list).  Everything must equal the exact path + stand-alone kernel bit for bit."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    n, K, D = 3000, 2048, 128
    x = torch.randn(n, D, device=dev, generator=g)
    W = torch.randn(K, D, device=dev, generator=g)
    what, wsq = ops.rownorm(W)
    if case == "all_rows_to_exact_path":
        what = what * 3.0                                    # |e|^2 = 9 > the range the error bound covers
        wsq = ops.rownorm(what, normalize=False)[1]
    if case == "some_rows_overflow":
        what[:1500] = what[0]                                # 1500 identical codes: rows near them overflow their candidate lists
        wsq = ops.rownorm(what, normalize=False)[1]
        x[::3] = what[0] + 0.01 * x[::3]
    out_a = out_b = None
    if case == "strided_out":
        out_a = torch.zeros(n, 3 * D, device=dev); out_b = torch.zeros(n, 3 * D, device=dev)
    a = ops.soft_vq_forward(x, what, wsq, 5, ops.PATH_F16_FILTER, want_sqerr=False, out=None if out_a is None else out_a[:, D:2 * D])
    b = ops.soft_vq_forward(x, what, wsq, 5, ops.PATH_F32_MFMA, want_sqerr=False, out=None if out_b is None else out_b[:, D:2 * D])
    for k in ("idx", "dist", "w", "zq", "xhat"):
        assert torch.equal(a[k], b[k]), k
    if out_a is not None:
        assert torch.equal(out_a, out_b) and not out_a[:, :D].any() and not out_a[:, 2 * D:].any()
    # with the squared error requested the stand-alone kernel runs: same values again
    c = ops.soft_vq_forward(x, what, wsq, 5, ops.PATH_F16_FILTER, want_sqerr=True)
    assert torch.equal(c["zq"], b["zq"] if out_b is None else b["zq"]) and torch.equal(c["w"], b["w"])


@pytest.mark.parametrize("splits,tiles", [(2, 131), (4, 70), (1, 260)])
def test_tail_launch_gives_the_same_bits(dev, splits, tiles):
    """Large searches launch the row tiles of the last, partly filled round of blocks separately with more code splits (their own
    candidate lists, a second region the re-score kernel reads by row range).  Forced here at a small size."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(tiles)
    n, K, D = 256 * tiles - 77, 256 * 40, 128
    xh, xs = ops.rownorm(torch.randn(n, D, device=dev, generator=g))
    wh, ws = ops.rownorm(torch.randn(K, D, device=dev, generator=g))
    i_ref, d_ref = ops.topk_search(xh, xs, wh, ws, 5, ops.PATH_F32_MFMA)
    for xcd in (False, True):
        path = _filter_path(filter_splits=splits, filter_tail=True, filter_xcd=xcd)
        i_t, d_t = ops.topk_search(xh, xs, wh, ws, 5, path)
        assert torch.equal(i_t, i_ref) and torch.equal(d_t, d_ref), xcd
    r = ops.soft_vq_forward(xh, wh, ws, 5, path, want_sqerr=False)      # fused assignment over both regions
    r0 = ops.soft_vq_forward(xh, wh, ws, 5, ops.PATH_F32_MFMA, want_sqerr=False)
    assert all(torch.equal(r[k], r0[k]) for k in ("idx", "dist", "w", "zq"))


@pytest.mark.parametrize("n,K,D,topk,splits", [(256 * 5 + 3, 256 * 7, 512, 5, 1), (256 * 9 + 100, 256 * 26 - 5, 768, 5, 2),
                                               (256 * 3, 256 * 4, 512, 1, 1), (256 * 4 + 17, 256 * 6, 1024, 8, 1), (256 * 2 + 1, 256 * 3 - 200, 640, 3, 1)])
def test_scan_hit_path_with_several_passing_values_per_quad(oracle, dev, n, K, D, topk, splits):
    """The scan's hit path is one exec-masked instruction sequence that appends a quad's MAXIMUM; a second passing value in the
    same quad (4 consecutive codes of a lane) raises a flag and the tile is revisited (filter_scan_rest).  Random codebooks almost
    never take that branch, so here every code comes in a run of near-copies: whole quads pass together, including exact ties
    (the revisit must skip exactly the position the hit path took).  Many code tiles per block and every k-list length (1, 5, 8
    slots).  ids and distances must equal the exact path (and the oracle)."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(n + K)
    xh, xs = ops.rownorm(torch.randn(n, D, device=dev, generator=g))
    base = torch.randn((K + 5) // 6, D, device=dev, generator=g)
    W = base.repeat_interleave(6, 0)[:K].clone()                       # runs of 6: a run straddles quads and 32-code groups
    W[1::6] += 1e-3 * torch.randn(W[1::6].shape, device=dev, generator=g)
    W[2::6] += 1e-4 * torch.randn(W[2::6].shape, device=dev, generator=g)
    W[4::6] += 1e-2 * torch.randn(W[4::6].shape, device=dev, generator=g)      # 0, 3, 5: exact copies
    wh, ws = ops.rownorm(W)
    i_ref, d_ref = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA)
    ops.SEARCH_STATS = {}
    try:
        i_s, d_s = ops.topk_search(xh, xs, wh, ws, topk, _filter_path(filter_splits=splits))
        assert ops.SEARCH_STATS["fallback_rows"] < n // 4        # the filter path decided these rows, not the exact redo
    finally:
        ops.SEARCH_STATS = None
    assert torch.equal(i_s, i_ref) and torch.equal(d_s, d_ref)
    sub = slice(0, 300)
    io, do = oracle.topk_search(xh[sub].cpu().numpy(), xs[sub].cpu().numpy(), wh.cpu().numpy(), ws.cpu().numpy(), topk)
    assert np.array_equal(i_s[sub].cpu().numpy(), io) and np.array_equal(d_s[sub].cpu().numpy(), do)


@pytest.mark.parametrize("n,K,D,topk,splits", [
    (4000, 256, 64, 5, 1),            # one code tile: it is learnt from and revisited (W = nct = 1)
    (70000, 2048, 64, 5, 0),          # default plan: 4 splits of 2 tiles -> every tile is a learning tile
    (3000, 256 * 5 + 40, 64, 5, 1),   # nct = 6 < R64_LEARN, ragged last tile
    (3000, 256 * 20, 64, 1, 1),       # learning tiles + scanned tiles + revisit, k-list of 1
    (3000, 256 * 20 + 8, 60, 8, 2),   # D < 64 (zero-padded to 64), k-list of 8, two splits
    (129, 256 * 9, 36, 3, 1),         # one partly filled row tile (forced onto the filter path below its size threshold)
    (600000, 21000, 64, 5, 0),        # the reference's shape: the size at which a scheduling hazard of the learning step showed
])
def test_rows64_kernel_equals_exact_path_and_general_kernel(oracle, dev, n, K, D, topk, splits):
    """filter_rows64_kernel (rows of <= 64 elements: x in registers, learning tiles revisited last) against the exact fp32 path,
    against the general filter kernel on the same search, and against the oracle on a slice: ids and distances, bit for bit."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(n + K + D)
    xh, xs = ops.rownorm(torch.randn(n, D, device=dev, generator=g))
    wh, ws = ops.rownorm(torch.randn(K, D, device=dev, generator=g))
    i_ref, d_ref = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA)
    ops.SEARCH_STATS = {}
    try:
        i_r, d_r = ops.topk_search(xh, xs, wh, ws, topk, _filter_path(filter_splits=splits, filter_rows64=True))
        assert ops.SEARCH_STATS["fallback_rows"] <= n // 100      # the filter decided the rows, not the exact redo
    finally:
        ops.SEARCH_STATS = None
    i_g, d_g = ops.topk_search(xh, xs, wh, ws, topk, _filter_path(filter_splits=splits, filter_rows64=False))
    i_w, d_w = ops.topk_search(xh, xs, wh, ws, topk, _filter_path(filter_splits=splits, filter_rows64="wide"))     # 128 x 64 wave tiles
    assert torch.equal(i_r, i_ref) and torch.equal(d_r, d_ref)
    assert torch.equal(i_g, i_ref) and torch.equal(d_g, d_ref)
    assert torch.equal(i_w, i_ref) and torch.equal(d_w, d_ref)
    sub = slice(0, 200)
    io, do = oracle.topk_search(xh[sub].cpu().numpy(), xs[sub].cpu().numpy(), wh.cpu().numpy(), ws.cpu().numpy(), topk)
    assert np.array_equal(i_r[sub].cpu().numpy(), io) and np.array_equal(d_r[sub].cpu().numpy(), do)


def test_rows64_kernel_with_near_copies_of_codes(oracle, dev):
    """Runs of near-identical codes at D = 64: whole quads pass together in the scanned tiles AND in the revisited learning tiles
    (append without insert, the revisit of a quad with several passing values), with exact ties."""
    from medtok_amd import ops
    n, K, D = 2500, 256 * 14, 64
    g = torch.Generator(device=dev).manual_seed(5)
    xh, xs = ops.rownorm(torch.randn(n, D, device=dev, generator=g))
    base = torch.randn((K + 5) // 6, D, device=dev, generator=g)
    W = base.repeat_interleave(6, 0)[:K].clone()
    W[1::6] += 1e-3 * torch.randn(W[1::6].shape, device=dev, generator=g)
    W[2::6] += 1e-4 * torch.randn(W[2::6].shape, device=dev, generator=g)
    W[4::6] += 1e-2 * torch.randn(W[4::6].shape, device=dev, generator=g)
    wh, ws = ops.rownorm(W)
    for topk in (1, 5, 8):
        i_ref, d_ref = ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA)
        for kind in ("wide", True):
            i_r, d_r = ops.topk_search(xh, xs, wh, ws, topk, _filter_path(filter_splits=1, filter_rows64=kind))
            assert torch.equal(i_r, i_ref) and torch.equal(d_r, d_ref), (topk, kind)
    io, do = oracle.topk_search(xh[:200].cpu().numpy(), xs[:200].cpu().numpy(), wh.cpu().numpy(), ws.cpu().numpy(), 8)
    assert np.array_equal(i_r[:200].cpu().numpy(), io) and np.array_equal(d_r[:200].cpu().numpy(), do)


def test_rows_of_at_most_32_elements_repeated_searches(dev):
    """D <= 32 used to be ONE 32-deep stage per code tile of the general kernel, where nothing guaranteed that the start values of a
    tile (copied by LDS-DMA two tiles ahead) had landed before they were read: a wave's 64 rows wrong in about 1 % of the searches
    (tools/fuzz_search.py found it; fixed by padding such rows to 64 columns, i.e. two stages or the narrow-row kernel).  A timing
    bug shows up only in repetition: 4 shapes x 12 seeds x 3 searches each on both filter kernels, every one against the exact path."""
    from medtok_amd import ops
    bad = []
    for seed in range(12):
        for (n, K, D, k, near, env) in ((4097, 20001, 32, 1, True, {}), (20000, 20001, 4, 1, False, dict(filter_splits=8, filter_xcd=True)),
                                        (4097, 20001, 32, 5, True, dict(filter_rows64=False)), (4097, 8191, 16, 1, False, {}),
                                        (4097, 20001, 32, 5, True, dict(filter_rows64="wide"))):
            g = torch.Generator(device=dev).manual_seed(seed * 7 + D)
            x = torch.randn(n, D, device=dev, generator=g); W = torch.randn(K, D, device=dev, generator=g)
            if near:
                x = x * 0.01 + W[torch.randint(0, K, (n,), device=dev, generator=g)]
            xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
            i0, d0 = ops.topk_search(xh, xs, wh, ws, k, ops.PATH_F32_MFMA)
            for rep in range(3):
                i1, d1 = ops.topk_search(xh, xs, wh, ws, k, _filter_path(**env))
                if not (torch.equal(i0, i1) and torch.equal(d0, d1)):
                    bad.append((seed, n, K, D, k, rep, int((i0 != i1).any(1).sum())))
    assert not bad, bad


@pytest.mark.parametrize("n,K,D,topk", [(20000, 3 * 4096, 768, 5), (9000, 3 * 7000, 64, 5), (5000, 3 * 2731, 100, 1), (70001, 3 * 1500, 36, 8)])
def test_prepared_codebook_searches_equal_the_unprepared_ones(dev, n, K, D, topk):
    """ops.prepare_codebook (normalisation + fp16 image + start values + largest norm of the three regions, two launches, once per weight
    version) against searches that prepare their region themselves: every output of soft_vq_forward, bit for bit -- regions that
    start inside the codebook and end at its last row, widths the image pads, both filter kernels; also from rows that were
    normalised earlier (the image made from `what`)."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(n + K + D)
    W = torch.randn(K, D, device=dev, generator=g)
    W[5] = 0
    regions = {"text": (0, K // 3), "graph": (K - K // 3, K), "shared": (0, K)}
    what0, wsq0 = ops.rownorm(W)
    what, wsq, prepared = ops.prepare_codebook(W, regions)
    assert torch.equal(what, what0) and torch.equal(wsq, wsq0)
    _, _, prepared2 = ops.prepare_codebook(None, regions, normalised=(what0, wsq0))
    x = torch.randn(n, D, device=dev, generator=g)
    for name, (lo, hi) in regions.items():
        assert ops.takes_filter_path(n, hi - lo, D, topk, ops.PATH_F16_FILTER)
        ref = ops.soft_vq_forward(x, what[lo:hi], wsq[lo:hi].contiguous(), topk, ops.PATH_F16_FILTER, want_sqerr=False)
        for prep in (prepared, prepared2):
            assert torch.equal(prep[name]["wsqp"], prepared[name]["wsqp"]) and torch.equal(prep[name]["en_max"], prepared[name]["en_max"])
            got = ops.soft_vq_forward(x, what[lo:hi], wsq[lo:hi].contiguous(), topk, ops.PATH_F16_FILTER, want_sqerr=False, prepared=prep[name])
            for k in ("xhat", "idx", "dist", "w", "zq"):
                assert torch.equal(got[k], ref[k]), (name, k)
        exact = ops.soft_vq_forward(x, what[lo:hi], wsq[lo:hi].contiguous(), topk, ops.PATH_F32_MFMA, want_sqerr=False, prepared=prepared[name])
        assert torch.equal(exact["idx"], ref["idx"]) and torch.equal(exact["dist"], ref["dist"])       # (the exact path ignores `prepared`)
    with pytest.raises(ValueError):
        ops.soft_vq_forward(x, what[: K // 3], wsq[: K // 3].contiguous(), topk, ops.PATH_F16_FILTER, want_sqerr=False, prepared=prepared["shared"])


def test_forward_with_and_without_a_prepared_codebook(dev, monkeypatch):
    """VectorQuantizer.forward (eval) at a size whose searches take the fp16 shortlist: PREPARED_CODEBOOK on / off, the same bits; the
    cache entry made by a caller without use for the image (a small search) is upgraded, and a weight update invalidates it."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd import ops
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(11)
    dim, bsz = 128, 6000
    vq = VectorQuantizer(3 * 2048, dim, 0.25, 0.0, True, False, [dim, dim], k=5).to(dev).eval()
    vq.search_path = ops.PATH_F16_FILTER
    g = torch.Generator(device=dev).manual_seed(2)
    h = torch.randn(bsz, 2 * dim, device=dev, generator=g)
    text = torch.randn(bsz, 6, dim, device=dev, generator=g)
    mask = torch.ones(bsz, 6, dtype=torch.int64, device=dev)
    n_nodes = torch.randint(1, 4, (bsz,), device=dev, generator=g)
    batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
    nodes = torch.randn(int(n_nodes.sum()), dim, device=dev, generator=g)
    keys = ("shared_text_embedding", "shared_graph_embedding", "specific_embedding_text", "specific_embedding_graph", "text_tokens", "graph_tokens",
            "shared_text_tokens", "shared_graph_tokens", "text_tokens_weights", "shared_graph_tokens_weights")
    with torch.no_grad():
        monkeypatch.setattr(vqmod, "PREPARED_CODEBOOK", False)
        ref = vq(h, text, nodes, mask, batch)
        assert getattr(vq._norm_cache[1], "prepared", None) is None
        monkeypatch.setattr(vqmod, "PREPARED_CODEBOOK", True)
        got = vq(h, text, nodes, mask, batch)                       # upgrades the entry the first forward left
        assert vq._norm_cache[1].prepared is not None
        for k in keys:
            assert torch.equal(got[k], ref[k]), k
        vq.invalidate_codebook_cache()
        got = vq(h, text, nodes, mask, batch)                       # built in one go
        for k in keys:
            assert torch.equal(got[k], ref[k]), k
        with torch.no_grad():
            vq.codebook.weight.mul_(-1.0)                             # a new weight version: nothing stale may be read
        new = vq(h, text, nodes, mask, batch)
        monkeypatch.setattr(vqmod, "PREPARED_CODEBOOK", False)
        vq.invalidate_codebook_cache()
        new_ref = vq(h, text, nodes, mask, batch)
        for k in keys:
            assert torch.equal(new[k], new_ref[k]), k
        assert not torch.equal(new["text_tokens"], ref["text_tokens"])
