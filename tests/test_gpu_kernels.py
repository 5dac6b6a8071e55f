"""HIP kernels vs the CPU oracle through the C ABI (bit-exact ids and distances)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("n,d", [(1, 4), (7, 64), (130, 768), (33, 100), (5, 1028),
                                 (1000, 64), (37, 60), (19, 8), (50, 32), (17, 68)])   # d <= 64: sixteen lanes per row
def test_rownorm_bit_exact(oracle, dev, n, d):
    from medtok_amd import ops
    rng = np.random.default_rng(n * 1000 + d)
    x = rng.standard_normal((n, d), dtype=np.float32) * 3
    if n > 2:
        x[1] = 0.0          # eps clamp branch
    xh_o, sq_o = oracle.rownorm(x, True)
    xh, sq = ops.rownorm(_t(x, dev), True)
    assert np.array_equal(xh.cpu().numpy(), xh_o)
    assert np.array_equal(sq.cpu().numpy(), sq_o)
    _, sq2_o = oracle.rownorm(x, False)
    _, sq2 = ops.rownorm(_t(x, dev), False)
    assert np.array_equal(sq2.cpu().numpy(), sq2_o)


@pytest.mark.parametrize("n,k,d,topk", [
    (256, 384, 64, 5),       # reference-like small shape
    (300, 1000, 768, 5),     # ragged row/code tails, code split path
    (129, 257, 36, 1),       # argmin, D tail inside a BK block
    (5000, 512, 64, 5),      # many row tiles
    (64, 8192, 768, 8),      # 8-slot list
    (1, 16, 4, 3),           # minimum sizes
    (70000, 300, 32, 5),     # single split (FINAL kernel), ragged everything
    (140001, 700, 32, 5),    # >= 1024 row tiles: unsplit main rounds + the last round's row tiles as a split tail launch
])
def test_search_bit_exact(oracle, dev, n, k, d, topk):
    from medtok_amd import ops
    rng = np.random.default_rng(n + 7 * k + 13 * d)
    x = rng.standard_normal((n, d), dtype=np.float32)
    W = rng.standard_normal((k, d), dtype=np.float32)
    xh, xs = oracle.rownorm(x)
    wh, ws = oracle.rownorm(W)
    idx_o, dist_o = oracle.topk_search(xh, xs, wh, ws, topk)
    idx, dist = ops.topk_search(_t(xh, dev), _t(xs, dev), _t(wh, dev), _t(ws, dev), topk)
    torch.cuda.synchronize()
    assert np.array_equal(dist.cpu().numpy(), dist_o), "distances must be bit-identical to the fmaf-chain oracle"
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    if n > 100000:           # the per-call plan hook takes the split-everything plan: same bits
        idx2, dist2 = ops.topk_search(_t(xh, dev), _t(xs, dev), _t(wh, dev), _t(ws, dev), topk, ops.plan_path(search_max_splits=64))
        assert torch.equal(idx2, idx) and torch.equal(dist2, dist)


def test_search_ties_lowest_index(oracle, dev, golden):
    from medtok_amd import ops
    g = golden("f8_ties")
    W, x = g["W"], g["x"]
    Wp = np.zeros((W.shape[0], 8), np.float32); Wp[:, :W.shape[1]] = W
    xp = np.zeros((x.shape[0], 8), np.float32); xp[:, :x.shape[1]] = x
    xh, xs = oracle.rownorm(xp); wh, ws = oracle.rownorm(Wp)
    idx_o, _ = oracle.topk_search(xh, xs, wh, ws, 5)
    idx, _ = ops.topk_search(_t(xh, dev), _t(xs, dev), _t(wh, dev), _t(ws, dev), 5)
    idx = idx.cpu().numpy()
    assert np.array_equal(idx, idx_o)
    rule = g["build_rule_idx"]
    for r in range(rule.shape[0]):
        want = [v for v in rule[r] if v >= 0]
        assert list(idx[r, :len(want)]) == want
    # exact duplicates in a big codebook: every duplicate pair must come out lowest-index first
    rng = np.random.default_rng(3)
    W = rng.standard_normal((2048, 64), dtype=np.float32)
    W[1024:] = W[:1024]
    x = rng.standard_normal((200, 64), dtype=np.float32)
    xh, xs = oracle.rownorm(x); wh, ws = oracle.rownorm(W)
    idx_o, dist_o = oracle.topk_search(xh, xs, wh, ws, 5)
    idx, dist = ops.topk_search(_t(xh, dev), _t(xs, dev), _t(wh, dev), _t(ws, dev), 5)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert np.array_equal(dist.cpu().numpy(), dist_o)
    assert (idx_o[:, 0] < 1024).all() and (idx_o[:, 1] == idx_o[:, 0] + 1024).all()


@pytest.mark.parametrize("n,k,d,topk,hard", [(200, 300, 64, 5, False), (65, 128, 768, 5, False), (77, 64, 32, 1, True), (9, 40, 100, 3, False)])
def test_soft_assign(oracle, dev, n, k, d, topk, hard):
    from medtok_amd import ops
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d), dtype=np.float32)
    W = rng.standard_normal((k, d), dtype=np.float32)
    xh, xs = oracle.rownorm(x); wh, ws = oracle.rownorm(W)
    idx, dist = oracle.topk_search(xh, xs, wh, ws, topk)
    xref = xh if hard else x
    ii = idx[:, 0] if hard else idx
    w_o, zq_o, se_o = oracle.soft_assign(xref, wh, ii, dist, hard)
    w, zq, se = ops.soft_assign(_t(xref, dev), _t(wh, dev), _t(ii, dev), _t(dist, dev), hard)
    # tolerance: 1e-5 relative (north_star).  expf differs from glibc by <= 2 ulp, and the
    # straight-through form x + (zq - x) re-rounds at ulp(|x|), which is larger than ulp(|zq|).
    assert rel(w.cpu().numpy(), w_o) <= 1e-6
    assert rel(zq.cpu().numpy(), zq_o) <= 1e-5
    if not hard:
        _, zr_o, _ = oracle.soft_assign(xref, wh, ii, dist, hard, raw=True)
        _, zr, _ = ops.soft_assign(_t(xref, dev), _t(wh, dev), _t(ii, dev), _t(dist, dev), hard, raw=True)
        assert rel(zr.cpu().numpy(), zr_o) <= 1e-6
    assert np.allclose(se.cpu().numpy(), se_o, rtol=1e-5, atol=0)
    if hard:
        assert np.array_equal(zq.cpu().numpy(), zq_o)


def test_sum_scale(dev):
    from medtok_amd import ops
    rng = np.random.default_rng(0)
    v = rng.random(100003, dtype=np.float32)
    out = ops.sum_scale(_t(v, dev), 1.0 / v.size).item()
    assert abs(out - v.astype(np.float64).mean()) <= 1e-7
    assert ops.sum_scale(_t(v[:0], dev), 1.0).item() == 0.0


@pytest.mark.parametrize("n,k,d", [(512, 64, 32), (1000, 300, 768), (5, 70000, 8), (3000, 2, 64), (100000, 8192, 64)])
def test_ema_stats_bit_exact(oracle, dev, n, k, d):
    from medtok_amd import ops
    rng = np.random.default_rng(n + k + d)
    z = rng.standard_normal((n, d), dtype=np.float32)
    idx = rng.integers(0, k, n).astype(np.int64)
    if n > 100:
        idx[: n // 3] = k - 1       # one long segment
    bins_o, es_o = oracle.ema_stats(z, idx, k)
    bins, es = ops.ema_stats(_t(z, dev), _t(idx, dev), k)
    assert np.array_equal(bins.cpu().numpy(), bins_o)
    assert np.array_equal(es.cpu().numpy(), es_o), "row-ordered segmented sum must match the oracle bit for bit"


@pytest.mark.parametrize("k,d", [(64, 32), (300, 768), (5, 100)])
def test_ema_apply_bit_exact(oracle, dev, k, d):
    from medtok_amd import ops
    rng = np.random.default_rng(k + d)
    E = oracle.rownorm(rng.standard_normal((k, d), dtype=np.float32))[0]
    cs = rng.random(k, dtype=np.float32) * 10
    bins = rng.integers(0, 4, k).astype(np.float32)
    es = rng.standard_normal((k, d), dtype=np.float32) * bins[:, None]
    E_o, cs_o = E.copy(), cs.copy()
    oracle.ema_apply(E_o, cs_o, bins, es, 0.99)
    E_d, cs_d = _t(E, dev), _t(cs, dev)
    ops.ema_apply_(E_d, cs_d, _t(bins, dev), _t(es, dev), 0.99)
    assert np.array_equal(cs_d.cpu().numpy(), cs_o)
    assert np.array_equal(E_d.cpu().numpy(), E_o)
    cs2_o = cs.copy(); oracle.ema_cluster_size(cs2_o, bins, 0.99)
    cs2 = _t(cs, dev); ops.ema_cluster_size_(cs2, _t(bins, dev), 0.99)
    assert np.array_equal(cs2.cpu().numpy(), cs2_o)


def test_usage_window(oracle, dev, golden):
    from medtok_amd import ops
    g = golden("f10_usage")
    n_e, wlen = int(g["n_e"]), int(g["window"])
    win_o = np.zeros(wlen, np.float32)
    win = torch.zeros(wlen, device=dev)
    for c in range(4):
        ids = g[f"c{c}.ids"]
        u_o = oracle.usage_update(win_o, ids, n_e)
        cnt = ops.usage_update_(win, _t(ids, dev), n_e)
        assert cnt.item() / n_e == u_o == float(g[f"c{c}.usage"])
    assert np.array_equal(win.cpu().numpy(), win_o)
    assert np.array_equal(win_o[-4000:], g["final_tail"])
    # more ids than the window holds: keep the newest
    small_o = np.zeros(10, np.float32); small = torch.zeros(10, device=dev)
    ids = np.arange(25, dtype=np.int64) % 7
    assert ops.usage_update_(small, _t(ids, dev), 7).item() / 7 == oracle.usage_update(small_o, ids, 7)
    assert np.array_equal(small.cpu().numpy(), small_o)


def test_soft_vq_forward_one_call(oracle, dev):
    from medtok_amd import ops
    rng = np.random.default_rng(11)
    x = rng.standard_normal((500, 128), dtype=np.float32) * 2
    W = rng.standard_normal((700, 128), dtype=np.float32)
    r_o = oracle.specific_search(x, W, 5)
    what, wsq = ops.rownorm(_t(W, dev))
    r = ops.soft_vq_forward(_t(x, dev), what, wsq, 5)
    assert np.array_equal(r["idx"].cpu().numpy(), r_o["idx"])
    assert np.array_equal(r["dist"].cpu().numpy(), r_o["dist"])
    assert np.array_equal(r["xhat"].cpu().numpy(), r_o["xhat"])
    assert rel(r["zq"].cpu().numpy(), r_o["zq"]) <= 1e-5
    assert rel(r["w"].cpu().numpy(), r_o["w"]) <= 1e-6


def test_cpu_tensors_are_rejected():
    from medtok_amd import ops, _lib
    with pytest.raises(_lib.MedTokLibraryError):
        ops.rownorm(torch.zeros(4, 8))


def test_code_sharded_merge_is_bit_exact(oracle, dev):
    """8 code shards searched separately (either path) and merged == one search over the whole codebook."""
    from medtok_amd import ops
    from medtok_amd.distributed import code_shard
    rng = np.random.default_rng(33)
    n, k, d = 3000, 16384, 64
    x = rng.standard_normal((n, d), dtype=np.float32); W = rng.standard_normal((k, d), dtype=np.float32)
    W[9000] = W[17]                                          # cross-shard duplicate
    xh, xs = ops.rownorm(_t(x, dev)); wh, ws = ops.rownorm(_t(W, dev))
    idx, dist = ops.topk_search(xh, xs, wh, ws, 5)
    dparts, iparts = [], []
    for r in range(8):
        lo, hi = code_shard(k, r, 8)
        i_r, d_r = ops.topk_search(xh, xs, wh[lo:hi], ws[lo:hi], 5)
        dparts.append(d_r); iparts.append(i_r + lo)
    idx_m, dist_m = ops.merge_topk_lists(torch.stack(dparts), torch.stack(iparts))
    assert torch.equal(idx_m, idx) and torch.equal(dist_m, dist)


@pytest.mark.parametrize("exact_f32", [False, True])
@pytest.mark.parametrize("d,heads,seed", [(128, 4, 0), (768, 4, 1), (512, 2, 2), (384, 1, 3), (640, 4, 4), (256, 4, 5), (64, 4, 6)])
def test_shared_kv_attention_matches_oracle(oracle, dev, d, heads, seed, exact_f32):
    """Ragged attention core vs the oracle's restatement (double accumulation): ragged query/key counts that are not
    multiples of the 32-row tiles, empty query sets, single keys, a key spike that forces the online-softmax rescale.
    Both kernels: the inference default (three fp16 MFMAs per product over (hi, lo) pairs) and the exact fp32-MFMA one."""
    from medtok_amd import ops
    rng = np.random.default_rng(seed)
    q_len = np.array([0, 1, 31, 32, 33, 80, 4, 200 * heads % 97 + 1, 64], np.int64) * 1
    kv_len = np.array([5, 1, 33, 512, 31, 260, 7, 64, 96], np.int64)
    q_start = np.cumsum(q_len) - q_len + 3                   # rows before/between the codes must stay untouched
    kv_start = (np.cumsum(kv_len) - kv_len)[::-1].copy()     # key blocks in a different order than the codes
    kv_start = np.cumsum(kv_len[::-1])[::-1] - kv_len        # contiguous, reversed order
    nq, nk = int(q_start[-1] + q_len[-1]) + 2, int(kv_len.sum())
    q = (rng.standard_normal((nq, d)) * 0.3).astype(np.float32)
    kv = rng.standard_normal((nk, d)).astype(np.float32)
    # spike: one late key of code 3 aligned with one query row -> the running max jumps in a later chunk
    kv[kv_start[3] + 300] = q[q_start[3] + 5] * 40.0
    scale = (d // heads) ** -0.5
    want = oracle.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale)
    T = lambda a: torch.from_numpy(a).to(dev)
    got = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), scale, exact_f32).cpu().numpy()
    touched = ~np.isnan(want).all(1)
    assert touched.sum() == q_len.sum()
    err = np.abs(got[touched].astype(np.float64) - want[touched]).max() / np.abs(want[touched]).max()
    assert err <= 1e-5, err
    # deterministic
    again = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), scale, exact_f32).cpu().numpy()
    assert np.array_equal(got[touched], again[touched])


@pytest.mark.parametrize("d,rows,seed", [(768, 4, 0), (64, 4, 1), (128, 1, 2), (768, 8, 3), (640, 6, 4), (256, 3, 5)])
def test_shared_kv_attention_few_rows_per_code_matches_oracle(oracle, dev, d, rows, seed):
    """The text side's shape -- at most 8 query rows per code (one per head) against a code's few dozen keys -- runs on the
    one-wavefront-per-code fp32 kernel (shared_kv_attention_fewq_kernel): same oracle, same tolerance; ragged row counts up to
    `rows`, codes without queries or without keys, a key spike that moves the running maximum late, the (hi, lo) output images,
    and agreement with the matrix kernel that serves larger row counts."""
    from medtok_amd import ops
    rng = np.random.default_rng(seed)
    n_codes = 37
    q_len = rng.integers(0, rows + 1, n_codes).astype(np.int64)
    q_len[0], q_len[1] = rows, 0
    kv_len = rng.integers(1, 41, n_codes).astype(np.int64)
    kv_len[2], kv_len[3] = 0, 300                                      # no keys at all; many keys
    if q_len[2] == 0: q_len[2] = 1
    if q_len[3] == 0: q_len[3] = 1
    q_start = np.cumsum(q_len) - q_len + 2
    kv_start = np.cumsum(kv_len[::-1])[::-1] - kv_len
    nq, nk = int(q_start[-1] + q_len[-1]) + 3, int(kv_len.sum())
    q = (rng.standard_normal((nq, d)) * 0.3).astype(np.float32)
    kv = rng.standard_normal((nk, d)).astype(np.float32)
    kv[kv_start[3] + 250] = q[q_start[3]] * 40.0
    scale = 0.11
    want = oracle.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale)
    T = lambda a: torch.from_numpy(a).to(dev)
    args = (T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len))
    got = ops.shared_kv_attention(*args, int(q_len.max()), scale).cpu().numpy()
    touched = ~np.isnan(want).all(1)
    has_keys = np.repeat(kv_len > 0, q_len)                            # (the oracle leaves rows of key-less codes as 0 / nan: checked apart)
    rows_idx = np.concatenate([np.arange(s, s + l) for s, l in zip(q_start, q_len)])
    ok = rows_idx[has_keys]
    err = np.abs(got[ok].astype(np.float64) - want[ok]).max() / np.abs(want[ok]).max()
    assert err <= 1e-5, err
    assert not got[rows_idx[~has_keys]].any()                          # a code without keys attends to nothing: context 0
    # the matrix kernel on the same inputs (max_q_len = 9 takes it off the few-rows path): same function
    big = ops.shared_kv_attention(*args, 9, scale).cpu().numpy()
    assert np.abs(big[ok].astype(np.float64) - got[ok]).max() / np.abs(want[ok]).max() <= 1e-5
    # the (hi, lo) images of the result are the result
    hi, lo = ops.shared_kv_attention(*args, int(q_len.max()), scale, split_out=True)
    back = (hi.float() + lo.float()).cpu().numpy()
    assert np.abs(back[rows_idx].astype(np.float64) - got[rows_idx]).max() <= 2.0 ** -20 * np.abs(got[rows_idx]).max() + 2.0 ** -24
    again = ops.shared_kv_attention(*args, int(q_len.max()), scale).cpu().numpy()
    assert np.array_equal(got[rows_idx], again[rows_idx])


def test_shared_kv_attention_rejects_bad_shapes(dev):
    from medtok_amd import ops
    from medtok_amd._lib import MedTokLibraryError
    z = torch.zeros(1, dtype=torch.int64, device=dev)
    with pytest.raises(MedTokLibraryError):
        ops.shared_kv_attention(torch.zeros(4, 96, device=dev), z, z + 4, torch.zeros(4, 96, device=dev), z, z + 4, 4, 1.0)


@pytest.mark.parametrize("d,p", [(64, 0.0), (128, 0.1), (768, 0.1), (256, 0.5), (384, 0.0), (512, 0.1), (640, 0.1), (640, 0.0)])
def test_attention_train_forward_and_backward_match_oracle(oracle, dev, d, p):
    """Training-mode ragged attention core (dropout by the stateless hash mask, log-sum-exp) and its backward (dQ / dKV kernels)
    vs the C oracle -- itself pinned to torch autograd (tests/test_oracle_golden.py).  Ragged codes incl. one without keys, one
    without queries, lengths that are not multiples of the 32-row tiles; rows of dq / dkv that belong to no code must be zero."""
    from medtok_amd import ops
    rng = np.random.default_rng(d + int(100 * p))
    q_len = np.array([40, 7, 0, 64, 33, 5], np.int64); kv_len = np.array([50, 33, 12, 100, 0, 1], np.int64)
    slot_kv = kv_len + np.array([0, 3, 0, 0, 4, 0])                      # some codes' key slots are longer than their valid keys
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(slot_kv) - slot_kv
    nq, nk = int(q_len.sum()) + 2, int(slot_kv.sum())                    # + two query rows that belong to no code
    q = (rng.standard_normal((nq, d)) * 0.3).astype(np.float32); kv = rng.standard_normal((nk, d)).astype(np.float32)
    d_out = rng.standard_normal((nq, d)).astype(np.float32)
    scale, seed = 0.2, 99
    T = lambda a: torch.from_numpy(a).to(dev)
    args = (T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len))
    out, lse = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed)
    out_o, lse_o, dq_o, dkv_o = oracle.shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, scale, p, seed, d_out)
    assert np.abs(out.cpu().numpy() - out_o).max() <= 1e-5 * np.abs(out_o).max()
    own = np.zeros(nq, bool)
    for b in range(len(q_len)):
        own[q_start[b]: q_start[b] + q_len[b]] = True
    fin = own & np.isfinite(lse_o)
    assert np.allclose(lse.cpu().numpy()[fin], lse_o[fin], rtol=1e-5, atol=1e-5) and np.isinf(lse.cpu().numpy()[own & ~fin]).all()
    if p == 0.0:                                                          # no dropout: the inference kernel's output
        plain = ops.shared_kv_attention(*args, int(q_len.max()), scale, exact_f32=True)
        assert torch.equal(plain[T(own)], out[T(own)])
    dq, dkv = ops.shared_kv_attention_backward(*args, int(q_len.max()), int(kv_len.max()), scale, p, seed, out, lse, T(d_out))
    assert np.abs(dq.cpu().numpy() - dq_o).max() <= 1e-5 * np.abs(dq_o).max()
    assert np.abs(dkv.cpu().numpy() - dkv_o).max() <= 1e-5 * np.abs(dkv_o).max()
    assert not dq.cpu().numpy()[~own].any()
    # deterministic: same seed, same bits; another seed, another mask
    out2, _ = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed)
    assert torch.equal(out2, out)
    if p > 0:
        out3, _ = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed + 1)
        assert not torch.equal(out3, out)


@pytest.mark.parametrize("d,half", [(64, torch.bfloat16), (128, torch.float16), (768, torch.bfloat16), (768, torch.float16), (384, torch.bfloat16),
                                    (256, torch.bfloat16), (512, torch.float16), (640, torch.bfloat16)])
def test_attention_backward_in_one_half_precision_pass_tracks_the_fp32_backward(dev, d, half):
    """The autocast form of the attention backward (four matrix products as ONE fp16 / bf16 pass, fp32 accumulation, softmax rebuilt in
    fp32 from the forward's log-sum-exp) against the exact fp32 kernels on the same operands: the difference is the rounding of the
    operands to 11 / 8 significant bits -- 4e-3 / 4e-2 of the gradient's largest entry; rows that belong to no code stay zero; with dropout."""
    from medtok_amd import ops
    rng = np.random.default_rng(d)
    q_len = np.array([40, 7, 0, 64, 33, 5, 150], np.int64); kv_len = np.array([50, 33, 12, 100, 0, 1, 300], np.int64)
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
    nq, nk = int(q_len.sum()) + 2, int(kv_len.sum())
    T = lambda a: torch.from_numpy(a).to(dev)
    q, kv = T((rng.standard_normal((nq, d)) * 0.3).astype(np.float32)), T(rng.standard_normal((nk, d)).astype(np.float32))
    d_out = T(rng.standard_normal((nq, d)).astype(np.float32))
    for p in (0.0, 0.1):
        args = (q, T(q_start), T(q_len), kv, T(kv_start), T(kv_len))
        out, lse = ops.shared_kv_attention_train(*args, int(q_len.max()), 0.2, p, 7)
        dq0, dkv0 = ops.shared_kv_attention_backward(*args, int(q_len.max()), int(kv_len.max()), 0.2, p, 7, out, lse, d_out)
        dq1, dkv1 = ops.shared_kv_attention_backward(*args, int(q_len.max()), int(kv_len.max()), 0.2, p, 7, out, lse, d_out, half=half)
        tol = 4e-3 if half == torch.float16 else 4e-2
        for a, b, what in ((dq1, dq0, "dq"), (dkv1, dkv0, "dkv")):
            err = float((a - b).abs().max()) / float(b.abs().max())
            assert err <= tol, (what, p, err)
        assert not dq1[-2:].any()


@pytest.mark.parametrize("n,d", [(1, 4), (130, 768), (1000, 64), (7, 4096), (33, 132)])
def test_residual_layernorm_bit_exact(oracle, dev, n, d):
    """The tail of CrossAttentionLayer (residual + LayerNorm) in one kernel: the oracle restates its summation order."""
    from medtok_amd import ops
    rng = np.random.default_rng(n + d)
    a = rng.standard_normal((n, d), dtype=np.float32) * 3
    b = rng.standard_normal((n, d), dtype=np.float32)
    if n > 2:
        a[1] = 0.0; b[1] = 0.0              # constant row: variance 0, eps alone under the root
    g, h = rng.standard_normal(d, dtype=np.float32), rng.standard_normal(d, dtype=np.float32)
    y = ops.residual_layernorm(_t(a, dev), _t(b, dev), _t(g, dev), _t(h, dev), 1e-5).cpu().numpy()
    y_o = oracle.residual_layernorm(a, b, g, h, 1e-5)
    assert np.array_equal(y, y_o)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(a + b), (d,), torch.from_numpy(g), torch.from_numpy(h), 1e-5).numpy()
    assert np.abs(y - ref).max() <= 4e-6 * max(np.abs(ref).max(), 1.0)
    with pytest.raises(Exception):
        ops.residual_layernorm(_t(a[:, :d - 1].copy(), dev), _t(b[:, :d - 1].copy(), dev), _t(g[:d - 1].copy(), dev), _t(h[:d - 1].copy(), dev), 1e-5)


def test_segment_mean_bit_exact(oracle, dev):
    from medtok_amd import ops
    rng = np.random.default_rng(5)
    length = np.array([3, 0, 200, 1, 17, 0], np.int64)
    start = np.cumsum(length) - length
    for d in (64, 768, 1028):
        x = rng.standard_normal((int(length.sum()), d), dtype=np.float32)
        out = ops.segment_mean(_t(x, dev), _t(start, dev), _t(length, dev)).cpu().numpy()
        assert np.array_equal(out, oracle.segment_mean(x, start, length))
        assert not out[1].any() and not out[5].any()            # empty segments: zero rows
        assert np.abs(out[2] - x[start[2]:start[2] + 200].astype(np.float64).mean(0)).max() <= 1e-6
    assert ops.segment_mean(_t(x, dev), _t(start[:0], dev), _t(length[:0], dev)).shape == (0, 1028)


@pytest.mark.parametrize("topk", [1, 5])
def test_exact_search_on_pointers_that_are_only_4_byte_aligned(oracle, dev, topk):
    """The C ABI promises nothing beyond float alignment for the row matrices: views at an odd element offset (the LDS-DMA staging
    and the 16-byte |e|^2 loads of the exact kernel must cope)."""
    from medtok_amd import ops
    rng = np.random.default_rng(3)
    n, k, d = 300, 700, 64
    xh, xs = oracle.rownorm(rng.standard_normal((n, d), dtype=np.float32))
    wh, ws = oracle.rownorm(rng.standard_normal((k, d), dtype=np.float32))
    idx_o, dist_o = oracle.topk_search(xh, xs, wh, ws, topk)

    def odd(a):                      # the same values behind a pointer that is 4 bytes past a 16-byte boundary
        buf = torch.empty(a.size + 1, dtype=torch.float32, device=dev)
        view = buf[1:].view(a.shape)
        view.copy_(torch.from_numpy(a))
        assert view.data_ptr() % 16 == 4
        return view

    idx, dist = ops.topk_search(odd(xh), odd(xs), odd(wh), odd(ws), topk, ops.PATH_F32_MFMA)
    assert np.array_equal(dist.cpu().numpy(), dist_o) and np.array_equal(idx.cpu().numpy(), idx_o)


@pytest.mark.parametrize("n,k,d,topk,path_name", [(3000, 2048, 768, 1, "PATH_F16_FILTER"), (3000, 2048, 768, 5, "PATH_F16_FILTER"),
                                                  (257, 300, 64, 1, "PATH_F32_MFMA"), (70000, 4096, 256, 1, "PATH_AUTO")])
def test_normalized_search_equals_rownorm_plus_search(dev, n, k, d, topk, path_name):
    """The one-call head of NormEMAVectorQuantizer.forward (l2norm + nearest codes): same bits as the two calls it replaces."""
    from medtok_amd import ops
    path = getattr(ops, path_name)
    g = torch.Generator(device=dev).manual_seed(n + k)
    z = torch.randn(n, d, device=dev, generator=g) * 2
    z[1] = 0.0
    E, esq = ops.rownorm(torch.randn(k, d, device=dev, generator=g))
    zhat, zsq = ops.rownorm(z)
    idx, dist = ops.topk_search(zhat, zsq, E, esq, topk, path)
    zhat2, zsq2, idx2, dist2 = ops.normalized_search(z, E, esq, topk, path)
    assert torch.equal(zhat, zhat2) and torch.equal(zsq, zsq2)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)


@pytest.mark.parametrize("d,heads,seed,variant", [(128, 4, 0, 0), (768, 4, 1, 0), (768, 4, 1, 1), (512, 2, 2, 0), (384, 1, 3, 0), (256, 4, 5, 0),
                                                  (768, 4, 1, 2), (512, 2, 2, 2), (256, 4, 5, 2)])
def test_shared_kv_attention_split_matches_oracle(oracle, dev, d, heads, seed, variant):
    """The wide-batch attention core (64 query rows per block, keys copied into LDS by DMA from their (hi, lo) fp16 images, value
    operands by transposed LDS reads) vs the oracle: same ragged cases as above -- query counts around the 64-row tile, key counts
    around the 16-key chunk, empty query sets, single keys, the late key spike that forces the online-softmax rescale, key rows
    that belong to no code left UNINITIALISED in the images (the masked split skips them)."""
    from medtok_amd import ops
    rng = np.random.default_rng(seed)
    q_len = np.array([0, 1, 63, 64, 65, 130, 4, 200 * heads % 97 + 1, 16], np.int64)
    kv_len = np.array([5, 1, 33, 512, 15, 260, 16, 17, 96], np.int64)
    slot = 520                                                    # every code owns a slot of 520 key rows; only kv_len of them are valid
    q_start = np.cumsum(q_len) - q_len + 3
    kv_start = np.arange(len(kv_len), dtype=np.int64) * slot
    nq, nk = int(q_start[-1] + q_len[-1]) + 2, int(len(kv_len) * slot)
    q = (rng.standard_normal((nq, d)) * 0.3).astype(np.float32)
    kv = rng.standard_normal((nk, d)).astype(np.float32)
    kv[kv_start[3] + 300] = q[q_start[3] + 5] * 40.0
    scale = (d // heads) ** -0.5
    want = oracle.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale)
    T = lambda a: torch.from_numpy(a).to(dev)
    images = ops.split_half(T(kv), seg_len=T(kv_len), seg_rows=slot)
    # poison what the masked conversion skipped: nothing may read it
    hi, lo = images
    valid = (torch.arange(slot, device=dev)[None, :] < T(kv_len)[:, None]).reshape(-1)
    hi[~valid] = float("nan"); lo[~valid] = float("nan")
    got = ops.shared_kv_attention_split(T(q), T(q_start), T(q_len), images, T(kv_start), T(kv_len), int(q_len.max()), scale, variant=variant).cpu().numpy()
    touched = ~np.isnan(want).all(1)
    assert touched.sum() == q_len.sum()
    err = np.abs(got[touched].astype(np.float64) - want[touched]).max() / np.abs(want[touched]).max()
    assert err <= 1e-5, err
    again = ops.shared_kv_attention_split(T(q), T(q_start), T(q_len), images, T(kv_start), T(kv_len), int(q_len.max()), scale, variant=variant).cpu().numpy()
    assert np.array_equal(got[touched], again[touched])
    # and the 32-row kernel on the same problem: same function
    other = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), scale).cpu().numpy()
    assert np.abs(other[touched] - got[touched]).max() <= 2e-6 * np.abs(want[touched]).max()
    if variant == 2:
        # the same kernel fed the fp32 key rows (it forms the images of every chunk in LDS itself): the bits of the call on the images,
        # fp32 output and output images alike; key rows no code owns may hold anything
        kv_dev = T(kv).clone()
        kv_dev[~valid] = float("nan")
        own = ops.shared_kv_attention_split(T(q), T(q_start), T(q_len), kv_dev, T(kv_start), T(kv_len), int(q_len.max()), scale, variant=2)
        assert np.array_equal(own.cpu().numpy()[touched], got[touched])
        h1, l1 = ops.shared_kv_attention_split(T(q), T(q_start), T(q_len), images, T(kv_start), T(kv_len), int(q_len.max()), scale, split_out=True, variant=2)
        h2, l2 = ops.shared_kv_attention_split(T(q), T(q_start), T(q_len), kv_dev, T(kv_start), T(kv_len), int(q_len.max()), scale, split_out=True, variant=2)
        tt = torch.from_numpy(touched).to(dev)
        assert torch.equal(h1[tt], h2[tt]) and torch.equal(l1[tt], l2[tt])


@pytest.mark.parametrize("d", [64, 128, 768, 384])
def test_attention_split_output_images_equal_the_fp32_output(dev, d):
    """split_out: the inference kernels write the (hi, lo) fp16 images of the context themselves (through LDS, 16-byte pieces per row)
    for the dense product that follows; hi + lo must be the kernel's own fp32 output to the split's 2^-22."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(d)
    q_len = torch.tensor([70, 5, 0, 64, 33], device=dev); kv_len = torch.tensor([40, 17, 9, 100, 1], device=dev)
    q_start = torch.cumsum(q_len, 0) - q_len; kv_start = torch.cumsum(kv_len, 0) - kv_len
    q = torch.randn(int(q_len.sum()), d, device=dev, generator=g) * 0.3
    kv = torch.randn(int(kv_len.sum()), d, device=dev, generator=g)
    a = (q, q_start, q_len, kv, kv_start, kv_len, 70, 0.2)
    ref = ops.shared_kv_attention(*a)
    hi, lo = ops.shared_kv_attention(*a, split_out=True)
    assert float((hi.double() + lo.double() - ref.double()).abs().max()) <= 2.0 ** -21 * float(ref.abs().max()) + 2.0 ** -24
    if d in ops.ATTENTION_SPLIT_WIDTHS:
        img = ops.split_half(kv)
        b = (q, q_start, q_len, img, kv_start, kv_len, 70, 0.2)
        for variant in ((0, 1, 2) if d == 768 else (0,)):
            ref2 = ops.shared_kv_attention_split(*b, variant=variant)
            hi2, lo2 = ops.shared_kv_attention_split(*b, split_out=True, variant=variant)
            assert float((hi2.double() + lo2.double() - ref2.double()).abs().max()) <= 2.0 ** -21 * float(ref2.abs().max()) + 2.0 ** -24
            assert float((ref2 - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_attention_dropout_mask_rate_and_independence(dev):
    """nn.MultiheadAttention's dropout on the attention probabilities is a stateless hash mask of (seed, query row, key) here, not
    torch's RNG stream: same distribution, not the same bits.  With zero queries (uniform probabilities) and one-hot
    keys the output IS the mask (out[r, j] = keep[r, j] / (T (1 - p))), so its statistics can be checked directly: keep rate
    1 - p within 4 sigma, rows / keys / seeds mutually independent (agreement of two masks = (1-p)^2 + p^2, lag-1 correlations ~ 0),
    and no mask at p = 0."""
    from medtok_amd import ops
    T, R, D, p = 128, 4096, 128, 0.1
    q = torch.zeros(R, D, device=dev)
    kv = torch.eye(T, D, device=dev)
    z = torch.zeros(1, dtype=torch.int64, device=dev)
    a = (q, z, z + R, kv, z, z + T, R, 1.0)

    def mask(seed, prob=p):
        out, _ = ops.shared_kv_attention_train(*a, prob, seed)
        m = out * (T * (1.0 - prob))
        assert bool(((m - m.round()).abs() < 1e-4).all()) and bool(((m.round() == 0) | (m.round() == 1)).all())
        return m.round()
    m1, m2 = mask(11), mask(12)
    n = R * T
    sigma = (p * (1 - p) / n) ** 0.5
    for m in (m1, m2):
        assert abs(float(m.mean()) - (1 - p)) <= 4 * sigma
        assert float(m.mean(1).min()) > 0.7 and float(m.mean(0).min()) > 0.85          # no dead row, no dead key
    agree = float((m1 == m2).float().mean())
    assert abs(agree - ((1 - p) ** 2 + p ** 2)) <= 6 * (0.18 * 0.82 / n) ** 0.5       # two seeds: independent masks
    c = m1 - m1.mean()
    var = float((c * c).mean())
    assert abs(float((c[:, 1:] * c[:, :-1]).mean()) / var) < 0.01                      # neighbouring keys
    assert abs(float((c[1:] * c[:-1]).mean()) / var) < 0.01                            # neighbouring rows
    assert torch.equal(mask(11), m1)                                                   # stateless: same seed, same mask
    assert bool((mask(5, 0.0) == 1).all())


@pytest.mark.parametrize("mask_dtype", [torch.int64, torch.bool, torch.int32])
@pytest.mark.parametrize("lpt", [False, True])
def test_pack_codes_matches_the_torch_prologue(dev, mask_dtype, lpt):
    """medtok_pack_codes (three small launches) vs the torch ops it replaces in CrossAttention.pooled -- what the reference's loop
    reads per code with mask[idx].sum().item() and batch == idx (vector_quantization_soft_one_new.py:133-142): token counts, node
    counts and offsets, the launch lists of both attention sides (integer work: exact), the longest-first order as a permutation
    sorted by key count, and the stats of the one host read, including an unsorted batch vector and codes without nodes."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    B, L, heads = 1500, 77, 4
    tok = torch.randint(0, L + 1, (B,), device=dev, generator=g)
    mask = (torch.arange(L, device=dev)[None, :] < tok[:, None]).to(mask_dtype)
    n_nodes = torch.randint(0, 9, (B,), device=dev, generator=g)
    batch = torch.repeat_interleave(torch.arange(B, device=dev), n_nodes)
    r = ops.pack_codes(mask, batch, heads, lpt)
    assert torch.equal(r["valid_len"], tok) and torch.equal(r["counts"], n_nodes)
    starts = torch.cumsum(n_nodes, 0) - n_nodes
    assert torch.equal(r["starts"], starts)
    code = torch.arange(B, device=dev)
    assert torch.equal(r["t_start"], code * heads) and torch.equal(r["t_len"], torch.full_like(code, heads))
    order = r["tok_start"] // L
    assert torch.equal(torch.sort(order).values, code)                     # a permutation of the codes
    assert torch.equal(r["g_start"], starts[order] * heads) and torch.equal(r["g_len"], n_nodes[order] * heads)
    assert torch.equal(r["g_kv_len"], tok[order])
    if lpt:
        assert bool((r["g_kv_len"][1:] <= r["g_kv_len"][:-1]).all())       # longest key set first
    else:
        assert torch.equal(order, code)
    assert r["stats"].tolist() == [int(n_nodes.max()), int(batch.min()), int(batch.max()), 0]
    shuffled = batch[torch.randperm(batch.numel(), device=dev, generator=g)]
    r2 = ops.pack_codes(mask, shuffled, heads, lpt)
    assert torch.equal(r2["counts"], n_nodes) and r2["stats"].tolist()[3] == 1
    bad = batch.clone(); bad[5] = B + 3; bad[9] = -2
    assert ops.pack_codes(mask, bad, heads, lpt)["stats"].tolist()[1:3] == [-2, B + 3]
    empty = ops.pack_codes(mask, batch[:0], heads, lpt)
    assert int(empty["counts"].sum()) == 0 and empty["stats"].tolist()[0] == 0


@pytest.mark.parametrize("d,dp", [(768, 768), (200, 256), (64, 64)])
def test_residual_layernorm_split_images_equal_split_half_of_the_output(dev, d, dp):
    """The layer tail with the (hi, lo) fp16 images of its output as a by-product: y bit-equal to the plain kernel, the images
    bit-equal to ops.split_half(y, dp) (zero columns appended)."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(d)
    a, b = torch.randn(333, d, device=dev, generator=g), torch.randn(333, d, device=dev, generator=g)
    gamma, beta = torch.randn(d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
    y0 = ops.residual_layernorm(a, b, gamma, beta, 1e-5)
    y, (hi, lo) = ops.residual_layernorm(a, b, gamma, beta, 1e-5, split_dp=dp)
    h2, l2 = ops.split_half(y0, dp=dp)
    assert torch.equal(y, y0) and torch.equal(hi, h2) and torch.equal(lo, l2)


@pytest.mark.parametrize("d,half", [(64, None), (768, None), (768, torch.bfloat16), (256, torch.float16)])
def test_attention_backward_accumulates_key_gradient_in_place(dev, d, half):
    """medtok_shared_kv_attention_backward_acc_f32: the key gradient ADDED to a buffer that already holds one -- bit for bit held + this
    launch's on the rows the launch owns, every other row (slots past a code's keys, rows of no code) untouched; dq as without it."""
    from medtok_amd import ops
    rng = np.random.default_rng(d)
    q_len = np.array([40, 7, 0, 64, 33, 5], np.int64); kv_len = np.array([50, 33, 12, 100, 0, 1], np.int64)
    slot_kv = kv_len + np.array([0, 3, 0, 0, 4, 0])
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(slot_kv) - slot_kv
    nq, nk = int(q_len.sum()), int(slot_kv.sum()) + 3
    T = lambda a: torch.from_numpy(a).to(dev)
    q, kv = T((rng.standard_normal((nq, d)) * 0.3).astype(np.float32)), T(rng.standard_normal((nk, d)).astype(np.float32))
    d_out = T(rng.standard_normal((nq, d)).astype(np.float32))
    args = (q, T(q_start), T(q_len), kv, T(kv_start), T(kv_len))
    out, lse = ops.shared_kv_attention_train(*args, int(q_len.max()), 0.2, 0.1, 7)
    tail = (int(q_len.max()), int(kv_len.max()), 0.2, 0.1, 7, out, lse, d_out)
    dq0, dkv0 = ops.shared_kv_attention_backward(*args, *tail, half=half)
    held = T(rng.standard_normal((nk, d)).astype(np.float32))
    buf = held.clone()
    dq1, dkv1 = ops.shared_kv_attention_backward(*args, *tail, half=half, dkv_into=buf, accumulate=True)
    assert dkv1 is buf and torch.equal(dq1, dq0)
    assert torch.equal(buf, held + dkv0)               # (rows the launch does not own: dkv0 is zero there)
    owned = torch.zeros(nk, dtype=torch.bool, device=dev)
    for b in range(len(kv_len)):
        owned[kv_start[b]: kv_start[b] + kv_len[b]] = True
    assert torch.equal(buf[~owned], held[~owned])
    # dkv_into without accumulate: the plain result, in the caller's buffer
    buf2 = held.clone()
    _, dkv2 = ops.shared_kv_attention_backward(*args, *tail, half=half, dkv_into=buf2)
    assert dkv2 is buf2 and torch.equal(buf2, dkv0)
    with pytest.raises(ValueError):
        ops.shared_kv_attention_backward(*args, *tail, accumulate=True)
    with pytest.raises(ValueError):
        ops.shared_kv_attention_backward(*args, *tail, dkv_into=held[:, : d // 2], accumulate=True)


@pytest.mark.parametrize("d,p", [(256, 0.0), (512, 0.1), (768, 0.1), (768, 0.0), (256, 0.5)])
def test_attention_train_forward_on_split_products_matches_the_fp32_kernel(oracle, dev, d, p):
    """medtok_shared_kv_attention_train_split_f32 (the autocast trainer's forward: three-pass fp16 products, keys split in the kernel)
    against the exact fp32 training kernel and the C oracle: out and log-sum-exp within 1e-5, the SAME dropout mask (a dropped
    probability is an exact zero contribution: with one key per code the outputs of dropped rows are exactly zero in both), rows of no
    code zero / -inf, codes without keys or queries, odd tile counts (group 1 idle), and the backward fed from it tracks the oracle."""
    from medtok_amd import ops
    rng = np.random.default_rng(d + int(100 * p))
    q_len = np.array([40, 7, 0, 64, 33, 5, 130, 97], np.int64); kv_len = np.array([50, 33, 12, 100, 0, 1, 300, 17], np.int64)
    slot_kv = kv_len + np.array([0, 3, 0, 0, 4, 0, 0, 5])
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(slot_kv) - slot_kv
    nq, nk = int(q_len.sum()) + 2, int(slot_kv.sum())
    q = (rng.standard_normal((nq, d)) * 0.3).astype(np.float32); kv = rng.standard_normal((nk, d)).astype(np.float32)
    d_out = rng.standard_normal((nq, d)).astype(np.float32)
    scale, seed = 0.2, 99
    T = lambda a: torch.from_numpy(a).to(dev)
    args = (T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len))
    out0, lse0 = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed)
    out1, lse1 = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed, split=True)
    out_o, lse_o, dq_o, dkv_o = oracle.shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, scale, p, seed, d_out)
    assert np.abs(out1.cpu().numpy() - out_o).max() <= 1e-5 * np.abs(out_o).max()
    assert float((out1 - out0).abs().max()) <= 1e-5 * float(out0.abs().max())
    own = np.zeros(nq, bool)
    for b in range(len(q_len)):
        own[q_start[b]: q_start[b] + q_len[b]] = True
    fin = own & np.isfinite(lse_o)
    assert np.allclose(lse1.cpu().numpy()[fin], lse_o[fin], rtol=1e-5, atol=1e-5) and np.isinf(lse1.cpu().numpy()[own & ~fin]).all()
    assert not out1.cpu().numpy()[~own].any() and np.isinf(lse1.cpu().numpy()[~own]).all()
    # the code with ONE key: a row's output is that key (kept, scaled) or exactly zero (dropped) -- the same rows in both kernels
    one = slice(int(q_start[5]), int(q_start[5] + q_len[5]))
    assert torch.equal(out1[one].abs().sum(1) == 0, out0[one].abs().sum(1) == 0)
    dq, dkv = ops.shared_kv_attention_backward(*args, int(q_len.max()), int(kv_len.max()), scale, p, seed, out1, lse1, T(d_out))
    assert np.abs(dq.cpu().numpy() - dq_o).max() <= 2e-5 * np.abs(dq_o).max()
    assert np.abs(dkv.cpu().numpy() - dkv_o).max() <= 2e-5 * np.abs(dkv_o).max()
    out2, lse2 = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed, split=True)
    assert torch.equal(out2, out1) and torch.equal(lse2, lse1)
    with pytest.raises(Exception):
        ops.shared_kv_attention_train(T(q[:, :128].copy()), *args[1:3], T(kv[:, :128].copy()), *args[4:], int(q_len.max()), scale, p, seed, split=True)


@pytest.mark.parametrize("d,half", [(64, None), (768, None), (768, torch.bfloat16), (256, torch.float16)])
def test_key_gradient_of_several_attention_calls_in_one_launch(dev, d, half):
    """medtok_shared_kv_attention_dkv_multi_f32: the key gradient of two attention calls over the same keys (their own queries, upstream
    gradients, masks) in one launch against the sum of the two calls' own dKV (accumulation order differs: 1e-5 in fp32, the half
    forms' own tolerance otherwise); the dQ-only call that makes a source returns the dq of the full call bit for bit."""
    from medtok_amd import ops
    rng = np.random.default_rng(d + 1)
    kv_len = np.array([50, 33, 12, 100, 0, 1], np.int64)
    slot_kv = kv_len + np.array([0, 3, 0, 0, 4, 0])
    kv_start = np.cumsum(slot_kv) - slot_kv
    nk = int(slot_kv.sum()) + 2
    T = lambda a: torch.from_numpy(a).to(dev)
    kv = T(rng.standard_normal((nk, d)).astype(np.float32))
    sources, singles = [], []
    for i, q_len in enumerate((np.array([40, 7, 0, 64, 33, 5], np.int64), np.array([3, 70, 9, 0, 1, 31], np.int64))):
        q_start = np.cumsum(q_len) - q_len
        nq = int(q_len.sum())
        q = T((rng.standard_normal((nq, d)) * 0.3).astype(np.float32)); d_out = T(rng.standard_normal((nq, d)).astype(np.float32))
        args = (q, T(q_start), T(q_len), kv, T(kv_start), T(kv_len))
        p, seed, scale = (0.1, 7 + i, 0.2) if i else (0.0, 0, 0.15)
        out, lse = ops.shared_kv_attention_train(*args, int(q_len.max()), scale, p, seed)
        tail = (int(q_len.max()), int(kv_len.max()), scale, p, seed, out, lse, d_out)
        dq_full, dkv_one = ops.shared_kv_attention_backward(*args, *tail, half=half)
        dq, delta = ops.shared_kv_attention_backward_dq(*args, *tail, half=half)
        assert torch.equal(dq, dq_full)
        singles.append(dkv_one)
        sources.append(dict(q=q, d_out=d_out, lse=lse, delta=delta, q_start=args[1], q_len=args[2], scale=scale, dropout_p=p, seed=seed))
    both = ops.shared_kv_attention_dkv_multi(sources, kv, T(kv_start), T(kv_len), int(kv_len.max()), half=half)
    ref = singles[0] + singles[1]
    tol = 1e-5 if half is None else (4e-3 if half == torch.float16 else 4e-2)
    assert float((both - ref).abs().max()) <= tol * float(ref.abs().max())
    one = ops.shared_kv_attention_dkv_multi(sources[:1], kv, T(kv_start), T(kv_len), int(kv_len.max()), half=half)
    assert torch.equal(one, singles[0])                       # a single source: the plain dKV launch
    owned = torch.zeros(nk, dtype=torch.bool, device=dev)
    for b in range(len(kv_len)):
        owned[kv_start[b]: kv_start[b] + kv_len[b]] = True
    assert not both[~owned].any()
    with pytest.raises(ValueError):
        ops.shared_kv_attention_dkv_multi(sources * 3, kv, T(kv_start), T(kv_len), int(kv_len.max()))
