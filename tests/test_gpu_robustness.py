"""GPU: behaviour the reference gets for free from ATen and a fused path has to earn -- valid token ids for non-finite rows,
a codebook cache that cannot go stale silently, attention over empty key sets."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path_name", ["exact", "filter"])
@pytest.mark.parametrize("topk", [1, 5])
@pytest.mark.parametrize("D", [128, 64])          # (64: the filter path's kernel for rows of <= 64 elements)
def test_non_finite_rows_keep_token_ids_in_range(dev, path_name, topk, D):
    """NaN / Inf / all-zero input rows (an overflowed AMP activation, a pooled row of an empty graph): torch.topk and argmin
    return valid indices for them and the reference's GradScaler loop survives the step.  Here: ids in [0, K) on both search
    paths, no memory fault in the gathers (assignment, fused assignment, sparse backward), healthy rows untouched."""
    from medtok_amd import ops
    path = ops.PATH_F32_MFMA if path_name == "exact" else ops.PATH_F16_FILTER
    g = torch.Generator(device=dev).manual_seed(3)
    n, K = 6000, 4096
    x = torch.randn(n, D, device=dev, generator=g)
    W = torch.randn(K, D, device=dev, generator=g)
    bad_nan, bad_inf, bad_zero = [5, 257, 5999], [6, 300], [7, 4000]
    x_bad = x.clone()
    x_bad[bad_nan, 3] = float("nan"); x_bad[bad_inf, 0] = float("inf"); x_bad[bad_zero] = 0.0
    what, wsq = ops.rownorm(W)
    good = ops.soft_vq_forward(x, what, wsq, topk, path, want_sqerr=False)
    for want_sqerr in (False, True):                  # fused assignment in the re-score kernel / stand-alone kernel
        r = ops.soft_vq_forward(x_bad, what, wsq, topk, path, want_sqerr=want_sqerr)
        torch.cuda.synchronize()
        idx = r["idx"]
        assert int(idx.min()) >= 0 and int(idx.max()) < K
        keep = torch.ones(n, dtype=torch.bool, device=dev); keep[bad_nan + bad_inf + bad_zero] = False
        assert torch.equal(idx[keep], good["idx"][keep]) and torch.equal(r["zq"][keep], good["zq"][keep])
        assert torch.isfinite(r["zq"][bad_zero]).all()      # zero rows normalise to zero: a perfectly good query
    gx, gc = ops.soft_vq_backward(x_bad, r["xhat"], what, r["idx"], r["w"], g_zq=torch.ones_like(x))
    torch.cuda.synchronize()
    assert gx.shape == x.shape and gc.shape == (n * topk, D)
    # the argmin + EMA module on the same rows
    from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
    q = NormEMAVectorQuantizer(K, D, 0.25).to(dev).train()
    q.search_path = path
    with torch.no_grad():
        _, _, ids = q(x_bad[:, :, None, None])
    torch.cuda.synchronize()
    assert int(ids.min()) >= 0 and int(ids.max()) < K


def test_codebook_cache_follows_data_writes(dev):
    """`.data` writes do not bump autograd's version counter.  Training mode re-normalises every call; eval mode caches and is
    told through invalidate_codebook_cache() / load_state_dict / train()-eval() switches."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(0)
    D = 64
    v = VectorQuantizer(3 * 512, D, 0.25, 0.0, True, False, [D, D]).to(dev)
    x = torch.randn(300, D, device=dev)
    fresh = torch.randn_like(v.codebook.weight)

    def ids():
        with torch.no_grad():
            return v.specific_embedding(x, "text", return_tokens=True)[3].clone()

    v.train()
    a = ids()
    v.codebook.weight.data.copy_(fresh)                  # invisible to _version
    b = ids()
    w2 = VectorQuantizer(3 * 512, D, 0.25, 0.0, True, False, [D, D]).to(dev).train()
    w2.load_state_dict(v.state_dict())
    with torch.no_grad():
        ids2 = w2.specific_embedding(x, "text", return_tokens=True)[3]
    assert torch.equal(b, ids2) and not torch.equal(a, b)
    v.eval()
    c = ids()
    assert torch.equal(c, b)                             # the mode switch dropped the cache
    v.codebook.weight.data.mul_(-1.0)
    v.invalidate_codebook_cache()
    d = ids()
    assert not torch.equal(d, c)
    sd = {k: t.clone() for k, t in v.state_dict().items()}
    sd["codebook.weight"] = fresh
    v.load_state_dict(sd)
    assert torch.equal(ids(), b)


def test_attention_kernel_with_empty_key_sets_and_many_codes(oracle, dev):
    """Codes without key rows give a zero context (not 0/0); more than 65535 codes in one call (the old grid.y limit)."""
    from medtok_amd import ops
    rng = np.random.default_rng(0)
    d, n_codes = 128, 70000
    q_len = rng.integers(1, 4, n_codes).astype(np.int64)
    kv_len = rng.integers(0, 5, n_codes).astype(np.int64)
    kv_len[[0, 17, n_codes - 1]] = 0
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
    q = (rng.standard_normal((int(q_len.sum()), d)) * 0.3).astype(np.float32)
    kv = rng.standard_normal((max(int(kv_len.sum()), 1), d)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    out = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), 0.25).cpu().numpy()
    ref = oracle.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, 0.25)
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()
    empty_rows = np.concatenate([np.arange(q_start[b], q_start[b] + q_len[b]) for b in np.nonzero(kv_len == 0)[0]])
    assert not out[empty_rows].any()


def test_empty_inputs_through_every_row_op(dev):
    """Zero rows / zero segments / zero codes: every row-wise entry point returns empty results instead of launching anything."""
    from medtok_amd import ops
    d, k = 64, 96
    g = torch.Generator(device=dev).manual_seed(0)
    W, wsq = ops.rownorm(torch.randn(k, d, device=dev, generator=g))
    x0 = torch.empty(0, d, device=dev)
    xh, xs = ops.rownorm(x0)
    assert xh.shape == (0, d) and xs.shape == (0,)
    for path in (ops.PATH_AUTO, ops.PATH_F32_MFMA, ops.PATH_F16_FILTER):
        idx, dist = ops.topk_search(xh, xs, W, wsq, 5, path)
        assert idx.shape == (0, 5) and dist.shape == (0, 5)
        zh, zs, idx, dist = ops.normalized_search(x0, W, wsq, 1, path)
        assert zh.shape == (0, d) and idx.shape == (0, 1)
    r = ops.soft_vq_forward(x0, W, wsq, 5)
    assert r["zq"].shape == (0, d) and r["idx"].shape == (0, 5)
    y = ops.residual_layernorm(x0, x0, torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5)
    assert y.shape == (0, d)
    z64 = torch.zeros(0, dtype=torch.int64, device=dev)
    assert ops.segment_mean(torch.randn(3, d, device=dev, generator=g), z64, z64).shape == (0, d)
    out = ops.shared_kv_attention(x0, z64, z64, torch.randn(3, d, device=dev, generator=g), z64, z64, 0, 0.125)
    assert out.shape == (0, d)
    torch.cuda.synchronize()


def test_empty_batch_through_the_modules(dev):
    """B = 0 (an empty shard of a ragged last batch): the drop-in modules return empty results of the right shapes."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
    D = 64
    v = VectorQuantizer(96, D, 0.25, 0.0, True, True, [D, D]).to(dev).eval()
    with torch.no_grad():
        r = v(torch.empty(0, 2 * D, device=dev), torch.empty(0, 8, D, device=dev), torch.empty(0, D, device=dev),
              torch.empty(0, 8, dtype=torch.long, device=dev), torch.empty(0, dtype=torch.long, device=dev))
    assert r["text_tokens"].shape == (0, 5) and r["shared_text_tokens"].shape == (0, 5)
    q = NormEMAVectorQuantizer(32, D, 0.25).to(dev).eval()
    with torch.no_grad():
        zq, loss, idx = q(torch.empty(0, D, 1, 1, device=dev))
    assert zq.shape == (0, D, 1, 1) and idx.shape == (0,)


@pytest.mark.parametrize("bsz", [48, 640])
def test_two_threads_through_one_module_return_the_single_thread_bits(dev, bsz):
    """Per-call state travels in arguments and return values, not on the module ("pure w.r.t. inputs except documented
    in-place state"): two Python threads run forward() of ONE VectorQuantizer on different batches at the same time, many times;
    each must get exactly what a single-threaded call returns for its batch.  (show_usage=False: the usage window IS shared
    in-place state, as in the reference.)  bsz = 640 takes the side-stream path, whose streams the threads share."""
    import threading
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(5)
    D, L = 128, 24
    v = VectorQuantizer(3 * 256, D, 0.25, 0.0, True, False, [D, D]).to(dev).eval()

    def batch(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        tok = torch.randint(1, L + 1, (bsz,), device=dev, generator=g)
        mask = (torch.arange(L, device=dev)[None, :] < tok[:, None]).to(torch.int64)
        n_nodes = torch.randint(1, 7, (bsz,), device=dev, generator=g)
        return (torch.randn(bsz, 2 * D, device=dev, generator=g), torch.randn(bsz, L, D, device=dev, generator=g),
                torch.randn(int(n_nodes.sum()), D, device=dev, generator=g), mask, torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes))
    inputs = [batch(1), batch(2)]
    keys = ("shared_text_embedding", "shared_graph_embedding", "specific_embedding_text", "specific_embedding_graph", "text_tokens",
            "graph_tokens", "shared_text_tokens", "shared_graph_tokens", "text_tokens_weights", "shared_graph_tokens_weights")
    with torch.no_grad():
        want = [{k: v(*a)[k].clone() for k in keys} for a in inputs]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            with torch.no_grad():
                for _ in range(15):
                    r = v(*inputs[i])
                    got = {k: r[k].clone() for k in keys}
                    torch.cuda.synchronize()
                    for k in keys:
                        if not torch.equal(got[k], want[i][k]):
                            errors.append((i, k))
        except Exception as exc:            # (a thread's exception must fail the test, not vanish)
            errors.append((i, repr(exc)))
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
