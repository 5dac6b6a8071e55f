"""GPU: generic k and width (the reference takes any k and any e_dim, vector_quantization_soft_one_new.py:91,157,203).
k = 9 .. 16 runs as two passes of the exact kernel with lists of 8 (the 8 best, then the best among the codes behind the row's 8th
(distance, index) pair): bit-identical to the C oracle's single list of k, on every launch plan, across ties at the pass boundary.
The module-level checks against reference-generated fixtures (F20-F23) live in test_gpu_modules.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _search(dev, oracle, n, k_codes, d, topk, seed, path=None, dup=0):
    from medtok_amd import ops
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((k_codes, d), dtype=np.float32)
    if dup:                                   # every code `dup` times: exact ties, resolved by the lowest index, straddle the 8th entry
        W = np.repeat(W[: k_codes // dup], dup, axis=0)
        W = np.concatenate([W, W[: k_codes - W.shape[0]]]) if W.shape[0] < k_codes else W
    x = rng.standard_normal((n, d), dtype=np.float32)
    what, wsq = oracle.rownorm(W)
    xhat, xsq = oracle.rownorm(x)
    T = lambda a: torch.from_numpy(a).to(dev)
    idx, dist = ops.topk_search(T(xhat), T(xsq), T(what), T(wsq), topk, ops.PATH_AUTO if path is None else path)
    ref_idx, ref_dist = oracle.topk_search(xhat, xsq, what, wsq, topk)
    assert np.array_equal(idx.cpu().numpy(), ref_idx), (n, k_codes, d, topk)
    assert np.array_equal(dist.cpu().numpy(), ref_dist), (n, k_codes, d, topk)
    return x, W


@pytest.mark.parametrize("topk", [9, 12, 16])
@pytest.mark.parametrize("shape", [(700, 5000, 96), (257, 1300, 40), (33, 20000, 768), (5, 17, 8)])
def test_search_for_more_than_eight_codes_per_row_is_bit_exact(dev, oracle, topk, shape):
    n, k_codes, d = shape
    _search(dev, oracle, n, k_codes, d, topk, seed=topk + n)


@pytest.mark.parametrize("topk", [9, 16])
def test_wide_lists_with_exact_ties_at_the_pass_boundary(dev, oracle, topk):
    """every code three times: a row's 8th and 9th entries are copies of one code (equal distances, consecutive indices) for most
    rows -- the second pass must skip exactly the copies the first pass returned"""
    _search(dev, oracle, 300, 999, 64, topk, seed=3, dup=3)
    _search(dev, oracle, 300, 1000, 36, topk, seed=4, dup=5)          # (D % 32 != 0: the register-staged kernel form)


def test_wide_lists_on_every_launch_plan(dev, oracle):
    """one block per row tile walking all codes; code-range splits + merge (forced through the plan bits); the main + tail launches
    of a search with at least two full rounds of row tiles (>= 131 072 rows)"""
    from medtok_amd import ops
    _search(dev, oracle, 1500, 2000, 64, 12, seed=7, path=ops.plan_path(ops.PATH_F32_MFMA, search_max_splits=1))
    _search(dev, oracle, 1500, 2000, 64, 12, seed=8, path=ops.plan_path(ops.PATH_F32_MFMA, search_max_splits=7))
    _search(dev, oracle, 140000, 300, 32, 10, seed=9)
    # a shape whose k <= 8 searches take the fp16 shortlist: k > 8 must resolve to the exact path by itself
    assert not ops.takes_filter_path(4096, 8192, 768, 12)
    assert ops.takes_filter_path(4096, 8192, 768, 8)


@pytest.mark.parametrize("topk,d", [(12, 64), (16, 768), (9, 40)])
def test_soft_vq_forward_and_backward_with_wide_lists(dev, oracle, topk, d):
    from medtok_amd import ops
    rng = np.random.default_rng(100 + topk)
    n, k_codes = 200, 777
    x = rng.standard_normal((n, d), dtype=np.float32)
    W = rng.standard_normal((k_codes, d), dtype=np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    what, wsq = ops.rownorm(T(W))
    r = ops.soft_vq_forward(T(x), what, wsq, topk)
    ref = oracle.specific_search(x, W, topk)
    assert np.array_equal(r["idx"].cpu().numpy(), ref["idx"]) and np.array_equal(r["dist"].cpu().numpy(), ref["dist"])
    for key in ("w", "zq", "xhat", "row_sqerr"):
        a, b = r[key].cpu().numpy().astype(np.float64), ref[key].astype(np.float64)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), key
    g = rng.standard_normal((n, d), dtype=np.float32)
    gx, gc = ops.soft_vq_backward(T(x), r["xhat"], what, r["idx"], r["w"], g_zq=T(g))
    gx_o, gc_o = oracle.soft_vq_backward(x, ref["xhat"], oracle.rownorm(W)[0], ref["idx"], ref["w"], g_zq=g)
    assert np.abs(gx.cpu().numpy() - gx_o).max() <= 1e-5 * np.abs(gx_o).max()
    assert np.abs(gc.cpu().numpy() - gc_o).max() <= 1e-5 * np.abs(gc_o).max()


def test_quantizer_accepts_k_up_to_16_and_refuses_17(dev):
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    VectorQuantizer(300, 64, 0.25, 0.0, True, False, [64, 64], k=16)
    with pytest.raises(ValueError, match="k=17"):
        VectorQuantizer(300, 64, 0.25, 0.0, True, False, [64, 64], k=17)


def test_width_that_is_not_a_multiple_of_four_through_quantize_pooled(dev, oracle):
    """e_dim = 70: rows and codes get zero columns appended inside; ids equal the oracle's on the unpadded data, embeddings 1e-5"""
    from medtok_amd.inference import quantize_pooled
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(0)
    e, n_e, n = 70, 600, 500
    vq = VectorQuantizer(n_e, e, 0.25, 0.0, True, False, [e, e], num_head=2, k=5).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    h = torch.randn(n, 2 * e, device=dev, generator=g)
    pt, pg = torch.randn(n, e, device=dev, generator=g), torch.randn(n, e, device=dev, generator=g)
    emb, tok, wts = quantize_pooled(vq, h, pt, pg)
    assert emb.shape == (n, 4 * e) and tok.shape == (n, 4, 5)
    W = vq.codebook.weight.detach().cpu().numpy()
    region = n_e // 3
    with torch.no_grad():
        xs = (vq.proj_text(h[:, :e]), vq.proj_graph(h[:, e:]), pt, pg)
    for j, (x, Wr) in enumerate(zip(xs, (W[:region], W[-region:], W, W))):
        ref = oracle.specific_search(x.cpu().numpy(), Wr, 5)
        near = np.diff(np.sort(ref["dist"].astype(np.float64), axis=1), axis=1).min(axis=1) < 1e-5      # (projection round-off vs near-ties)
        assert np.array_equal(tok[:, j].cpu().numpy()[~near], ref["idx"][~near]), j
        a = emb[:, j * e:(j + 1) * e].cpu().numpy()[~near]
        assert np.abs(a - ref["zq"][~near]).max() <= 1e-5 * np.abs(ref["zq"]).max(), j


@pytest.mark.parametrize("d,q_max", [(1024, 40), (1024, 3)])
def test_attention_core_at_width_1024_matches_the_oracle(dev, oracle, d, q_max):
    """the image-form attention core (medtok_shared_kv_attention_split_f32) and the few-rows kernel at BERT-large width against
    the C oracle's restatement: 1e-5 of the output scale; ragged query / key counts incl. an empty key set"""
    from medtok_amd import ops
    rng = np.random.default_rng(d + q_max)
    q_len = np.array([q_max, 1, max(q_max // 2, 1), q_max, 2][: 5], np.int64)
    kv_len = np.array([50, 33, 0, 129, 16], np.int64)
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
    qa = (rng.standard_normal((int(q_len.sum()), d)) * 0.1).astype(np.float32)
    ka = rng.standard_normal((int(kv_len.sum()), d)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    ref = oracle.shared_kv_attention(qa, q_start, q_len, ka, kv_start, kv_len, 0.06)
    if q_max <= 4:
        out = ops.shared_kv_attention(T(qa), T(q_start), T(q_len), T(ka), T(kv_start), T(kv_len), int(q_len.max()), 0.06)
    else:
        images = ops.split_half(T(ka))
        out = ops.shared_kv_attention_split(T(qa), T(q_start), T(q_len), images, T(kv_start), T(kv_len), int(q_len.max()), 0.06, variant=2)
    err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err


@pytest.mark.parametrize("d,heads", [(1024, 4), (900, 4)])
def test_pooled_cross_attention_at_bert_large_width_vs_the_reference_loop(dev, d, heads):
    """CrossAttention.pooled at D = 1024 (and 900, zero-padded to 1024) against the reference's own per-code loop over
    nn.MultiheadAttention layers (vector_quantization_soft_one_new.py:133-142) run on the GPU in fp32: 1e-5 of the output scale;
    training at that width is refused with a clear error"""
    from medtok_amd import ops
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    torch.manual_seed(3)
    ca = CrossAttention(d, heads, dropout=0.1, layers=2).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(4)
    bsz, seq_len = 6, 24
    text = torch.randn(bsz, seq_len, d, device=dev, generator=g)
    tok = torch.tensor([24, 1, 7, 16, 24, 3], device=dev)
    mask = (torch.arange(seq_len, device=dev)[None, :] < tok[:, None]).to(torch.int64)
    n_nodes = torch.tensor([5, 1, 12, 3, 9, 2], device=dev)
    batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
    nodes = torch.randn(int(n_nodes.sum()), d, device=dev, generator=g)
    with torch.no_grad():
        pt, pg = ca.pooled(text, mask, nodes, batch)
        want_t, want_g = [], []
        for i in range(bsz):
            a, b = ca(text[i, : int(tok[i])], nodes[batch == i])
            want_t.append(a[0]); want_g.append(b.mean(0))
    want_t, want_g = torch.stack(want_t), torch.stack(want_g)
    for got, want, what in ((pt, want_t, "text"), (pg, want_g, "graph")):
        err = float((got.double() - want.double()).abs().max() / want.double().abs().max())
        assert err <= 1e-5, (what, err)
    ca.train()
    with pytest.raises(ops.MedTokLibraryError, match="inference only"):
        ca.pooled(text, mask, nodes, batch)
    with pytest.raises(ops.MedTokLibraryError, match="not supported"):
        ops.attention_width(1100)


@pytest.mark.parametrize("d,heads,l1,l2", [(64, 4, 12, 9), (768, 4, 300, 21), (128, 4, 1, 1), (70, 2, 17, 40), (256, 8, 40, 5)])
def test_per_pair_cross_attention_call_runs_on_the_kernels(dev, d, heads, l1, l2):
    """CrossAttention.forward(v1, v2) -- the reference's public per-pair call (vector_quantization_soft_one_new.py:53-88) -- at inference:
    the library's kernels (the pair as one code of the ragged core) against the stock nn.MultiheadAttention / LayerNorm modules the same
    object runs under autograd: 1e-5 of the output scale, full sequences of both directions"""
    from medtok_amd import ops
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    torch.manual_seed(d + l1)
    ca = CrossAttention(d, heads, dropout=0.1, layers=2).to(dev).eval()
    with torch.no_grad():
        for layer in ca.model:
            layer.multihead_attn.in_proj_bias.normal_(0, 0.2)
            layer.layer_norm.weight.normal_(1.0, 0.2)
            layer.layer_norm.bias.normal_(0, 0.2)
    g = torch.Generator(device=dev).manual_seed(1)
    v1, v2 = torch.randn(l1, d, device=dev, generator=g), torch.randn(l2, d, device=dev, generator=g)
    ops.profile_begin()
    with torch.no_grad():
        a1, a2 = ca(v1, v2)
    torch.cuda.synchronize()
    prof = ops.profile_end()
    assert prof["shared_kv_attention_kernel"]["launches"] >= 4          # (the kernels ran: two layers x two directions)
    b1, b2 = ca(v1, v2)                                                  # eval mode under autograd: the stock modules, dropout off
    for got, want in ((a1, b1), (a2, b2)):
        assert got.shape == want.shape
        err = float((got.double() - want.detach().double()).abs().max() / want.detach().double().abs().max())
        assert err <= 1e-5, err
