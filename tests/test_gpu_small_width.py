"""GPU: the two-launch cross-attention at the reference's own width (e_dim = 64, 4 heads; medtok_cross_attention_small_f32) against the
layer-by-layer product path, the padded torch comparator and the reference-generated forward fixtures; the forward without a host
read replayed from a HIP graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _ragged(dev, seed, bsz, seq_len, max_nodes, zero_nodes=(), zero_tokens=()):
    g = torch.Generator(device=dev).manual_seed(seed)
    text = torch.randn(bsz, seq_len, 64, device=dev, generator=g)
    tok = torch.randint(1, seq_len + 1, (bsz,), device=dev, generator=g)
    n_nodes = torch.randint(1, max_nodes + 1, (bsz,), device=dev, generator=g)
    for b in zero_nodes:
        n_nodes[b] = 0
    for b in zero_tokens:
        tok[b] = 0
    mask = (torch.arange(seq_len, device=dev)[None, :] < tok[:, None]).to(torch.int64)
    batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
    nodes = torch.randn(int(n_nodes.sum()), 64, device=dev, generator=g)
    return text, mask, nodes, batch


def _module(dev, seed=0):
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    torch.manual_seed(seed)
    ca = CrossAttention(64, 4, dropout=0.1, layers=2).to(dev).eval()
    with torch.no_grad():                       # non-trivial biases and LayerNorm parameters
        for layer in ca.model:
            layer.multihead_attn.in_proj_bias.normal_(0, 0.2)
            layer.multihead_attn.out_proj.bias.normal_(0, 0.2)
            layer.layer_norm.weight.normal_(1.0, 0.2)
            layer.layer_norm.bias.normal_(0, 0.2)
    return ca


def _close(a, b, what):
    err = float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)
    assert err <= RTOL, f"{what}: relative error {err:.3g}"


@pytest.mark.parametrize("shape", [(37, 50, 25), (256, 512, 40), (5, 7, 3), (64, 33, 1), (9, 300, 70)])
def test_small_width_path_equals_the_layer_by_layer_path(dev, shape):
    """pooled_small (two launches, fp32 FMA / fp32 MFMA) against pooled() (pack + seven launches per layer and side on the split-fp16
    kernels): 1e-5 of the output scale, incl. codes without nodes, codes without valid tokens, tiles that span many codes (one
    node per code), node counts that are not a multiple of the tile, and more nodes per code than one tile."""
    bsz, seq_len, max_nodes = shape
    ca = _module(dev)
    text, mask, nodes, batch = _ragged(dev, 11 + bsz, bsz, seq_len, max_nodes, zero_nodes=(1, bsz - 1) if bsz > 4 else (), zero_tokens=(2,) if bsz > 4 else ())
    with torch.no_grad():
        assert ca.small_eligible(text, nodes)
        both = ca.pooled_small(text, mask, nodes, batch)
        ca.check_small_status()
        pt, pg = ca.pooled(text, mask, nodes, batch)
    _close(both[:, 0], pt, "attended CLS rows")
    _close(both[:, 1], pg, "node means")


@pytest.mark.parametrize("shape", [(64, 512, 40), (9, 70, 33), (300, 17, 5), (2, 1, 1)])
def test_split_fp16_core_equals_the_fp32_matrix_pipe_core(dev, shape):
    """the product's attention core (three fp16 MFMA passes over (hi, lo) pairs, keys permuted so that the probabilities feed the
    value product from the registers they stand in) against the round-5 core on the fp32 matrix pipe (exact fmaf chains): 2e-6 of the
    output scale -- chunks that are not full, keys in every position of the permuted order, tiles that span codes"""
    from medtok_amd import ops
    bsz, seq_len, max_nodes = shape
    ca = _module(dev, 7)
    text, mask, nodes, batch = _ragged(dev, 31 + bsz, bsz, seq_len, max_nodes, zero_nodes=(1,) if bsz > 4 else (), zero_tokens=(2,) if bsz > 4 else ())
    heads = 4
    outs = []
    with torch.no_grad():
        for exact in (False, True):
            pooled = torch.empty(bsz, 2, 64, device=dev)
            ops.cross_attention_small(text, mask, nodes, batch, ca._small_weights(), len(ca.model), (64 // heads) ** -0.5, ca.model[0].layer_norm.eps,
                                      pooled, ca._status_word(dev), exact_f32=exact)
            outs.append(pooled)
    ca.check_small_status()
    err = float((outs[0].double() - outs[1].double()).abs().max()) / float(outs[1].double().abs().max())
    assert err <= 2e-6, err
    # a transposition or a wrong key permutation would not survive asymmetric data: keys scaled by their index
    scale = 1.0 + 0.01 * torch.arange(seq_len, device=dev, dtype=torch.float32)
    text2 = text * scale[None, :, None]
    outs = []
    with torch.no_grad():
        for exact in (False, True):
            pooled = torch.empty(bsz, 2, 64, device=dev)
            ops.cross_attention_small(text2, mask, nodes, batch, ca._small_weights(), len(ca.model), 0.25, ca.model[0].layer_norm.eps, pooled,
                                      ca._status_word(dev), exact_f32=exact)
            outs.append(pooled)
    err = float((outs[0].double() - outs[1].double()).abs().max()) / float(outs[1].double().abs().max())
    assert err <= 2e-6, err


def test_small_width_path_equals_the_padded_torch_comparator(dev):
    ca = _module(dev, 3)
    text, mask, nodes, batch = _ragged(dev, 5, 23, 40, 12)
    with torch.no_grad():
        both = ca.pooled_small(text, mask, nodes, batch)
        pt, pg = ca.pooled_reference(text, mask, nodes, batch, fold=False)          # nn.MultiheadAttention on padded batches
    _close(both[:, 0], pt, "attended CLS rows vs nn.MultiheadAttention")
    _close(both[:, 1], pg, "node means vs nn.MultiheadAttention")


def test_small_width_path_flags_an_unsorted_batch_vector(dev):
    ca = _module(dev)
    text, mask, nodes, batch = _ragged(dev, 7, 16, 20, 6)
    perm = torch.randperm(batch.numel(), device=dev)
    with torch.no_grad():
        ca.pooled_small(text, mask, nodes[perm], batch[perm])
    with pytest.raises(ValueError, match="non-decreasing"):
        ca.check_small_status()
    ca.check_small_status()                       # cleared
    bad = batch.clone()
    bad[-1] = 16
    with torch.no_grad():
        ca.pooled_small(text, mask, nodes, bad)
    with pytest.raises(ValueError, match="outside"):
        ca.check_small_status()


@pytest.mark.parametrize("show_usage", [False, True])
@pytest.mark.parametrize("bsz", [256, 3000])
def test_forward_takes_any_batch_vector_like_the_reference(dev, show_usage, bsz):
    """The reference selects a code's nodes with `batch == idx` (vector_quantization_soft_one_new.py:135): any batch vector works.  The
    two-launch cross-attention needs a sorted one; by default the forward verifies the device-side check and runs an unsorted vector
    again on the stably sorted nodes -- every output, the usage values AND the usage window equal the forward on nodes the caller
    sorted himself, bit for bit (the first, flagged pass must not have written the window).  bsz = 256: the small-batch form (status
    read with the usage counts / at the end); bsz = 3000: the general form (status read behind the cross-attention).
    assume_sorted_batch = True restores the unchecked behaviour: the flag stays on the device for check_status()."""
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention, UNSORTED_BATCH_MESSAGE
    vq_a, vq_b = _quantizer(dev, show_usage), _quantizer(dev, show_usage)
    z, text, nodes, mask, batch = _forward_inputs(dev, bsz=bsz, seq_len=64, max_nodes=12)
    perm = torch.randperm(batch.numel(), device=dev)
    nodes_p, batch_p = nodes[perm], batch[perm]
    nodes_s, batch_s = CrossAttention.sort_by_code(nodes_p, batch_p)
    with torch.no_grad():
        for step in range(2):                    # twice: the window of step 1 feeds the usage values of step 2
            got = vq_a(z, text, nodes_p, mask, batch_p)
            want = vq_b(z, text, nodes_s, mask, batch_s)
            for k, v in want.items():
                if isinstance(v, torch.Tensor):
                    assert torch.equal(got[k], v), (k, step)
                elif isinstance(v, float):
                    assert got[k] == v, (k, step, got[k], v)
            if show_usage:
                assert torch.equal(vq_a.codebook_used, vq_b.codebook_used), step
        vq_a.cross_attn.check_status()           # nothing left behind
        # ids outside [0, B) still raise what pooled() raises
        bad = batch_s.clone()
        bad[-1] = bsz
        with pytest.raises(ValueError, match="outside"):
            vq_a(z, text, nodes_s, mask, bad)
            vq_a.cross_attn.check_status()
        # the opt-out: nothing is verified, the flag waits for check_status()
        vq_a.cross_attn.assume_sorted_batch = True
        try:
            if show_usage:
                with pytest.raises(ValueError, match="non-decreasing"):
                    vq_a(z, text, nodes_p, mask, batch_p)
            else:
                vq_a(z, text, nodes_p, mask, batch_p)
                with pytest.raises(ValueError, match="non-decreasing"):
                    vq_a.cross_attn.check_status()
        finally:
            vq_a.cross_attn.assume_sorted_batch = False
    assert "non-decreasing" in UNSORTED_BATCH_MESSAGE


def test_a_forced_search_path_is_not_replaced_by_the_batched_small_searches(dev):
    """A module whose search path was forced (bench --path, plan bits) runs the kernel the caller named, whatever the batch size: the
    batched exact small-batch searches are taken under PATH_AUTO / PATH_F32_MFMA only (same ids either way: the paths agree bit for bit)."""
    from medtok_amd import ops
    vq = _quantizer(dev)
    inp = _forward_inputs(dev)
    with torch.no_grad():
        auto = vq(*inp)
        ops.profile_begin()
        vq.search_path = ops.PATH_F16_FILTER
        try:
            forced = vq(*inp)
        finally:
            vq.search_path = ops.PATH_AUTO
        torch.cuda.synchronize()
        prof = ops.profile_end()
    assert prof["filter_f16_kernel"]["launches"] >= 3, prof["filter_f16_kernel"]
    for k in ("text_tokens", "graph_tokens", "shared_text_tokens", "shared_graph_tokens", "specific_embedding_text", "shared_text_embedding"):
        assert torch.equal(auto[k], forced[k]), k


def _quantizer(dev, show_usage=False):
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(1)
    return VectorQuantizer(21000, 64, 0.25, 0.0, True, show_usage, [64, 64], k=5).to(dev).eval()


def _forward_inputs(dev, bsz=256, seq_len=512, max_nodes=40):
    text, mask, nodes, batch = _ragged(dev, 21, bsz, seq_len, max_nodes)
    g = torch.Generator(device=dev).manual_seed(22)
    return torch.randn(bsz, 128, device=dev, generator=g), text, nodes, mask, batch


def test_forward_small_width_equals_the_layer_by_layer_forward(dev):
    """VectorQuantizer.forward at the reference's default shape and batch through the two-launch cross-attention + ONE shared search
    over the interleaved rows, against the round-4 form (layer-by-layer cross-attention, two shared searches): embeddings 1e-5; token
    ids equal wherever the pooled rows' 1e-6 differences do not meet a near-tie (counted, bounded)."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    vq = _quantizer(dev)
    inp = _forward_inputs(dev)
    with torch.no_grad():
        new = vq(*inp)
        vq.cross_attn.check_small_status()
        keep = (vqmod.SMALL_WIDTH_FUSED, vqmod.MERGE_SHARED_SEARCHES)
        vqmod.SMALL_WIDTH_FUSED = vqmod.MERGE_SHARED_SEARCHES = False
        try:
            old = vq(*inp)
        finally:
            vqmod.SMALL_WIDTH_FUSED, vqmod.MERGE_SHARED_SEARCHES = keep
    for k in ("text_tokens", "graph_tokens", "text_tokens_weights", "graph_tokens_weights", "specific_embedding_text", "specific_embedding_graph"):
        assert torch.equal(new[k], old[k]), k                    # the modality-specific searches do not see the cross-attention
    for k in ("shared_text_tokens", "shared_graph_tokens"):
        differ = int((new[k] != old[k]).any(1).sum())
        assert differ <= 2, f"{k}: {differ} of {new[k].shape[0]} rows pick different codes"
        same = (new[k] == old[k]).all(1)
        e = "shared_text_embedding" if "text" in k else "shared_graph_embedding"
        _close(new[e][same], old[e][same], e)


def test_forward_replayed_from_a_hip_graph_is_bit_equal_to_eager(dev):
    """No host read anywhere in the eval forward at the reference's shape (show_usage = False): it captures into a HIP graph, and the
    replay -- on NEW input values in the captured buffers -- equals the eager forward bit for bit."""
    vq = _quantizer(dev)
    inp = [t.clone() for t in _forward_inputs(dev)]
    keys = ("shared_text_embedding", "shared_graph_embedding", "specific_embedding_text", "specific_embedding_graph", "text_tokens", "graph_tokens",
            "shared_text_tokens", "shared_graph_tokens", "text_tokens_weights", "graph_tokens_weights", "shared_text_tokens_weights",
            "shared_graph_tokens_weights")
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # warm-up on the side stream (caches, workspaces), as torch.cuda.graph wants it
            for _ in range(3):
                vq(*inp)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = vq(*inp)
        # new values in the captured input buffers (same shapes, same batch vector)
        g = torch.Generator(device=dev).manual_seed(99)
        inp[0].copy_(torch.randn(inp[0].shape, device=dev, generator=g))
        inp[1].copy_(torch.randn(inp[1].shape, device=dev, generator=g))
        inp[2].copy_(torch.randn(inp[2].shape, device=dev, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        replayed = {k: out[k].clone() for k in keys}
        eager = vq(*inp)
        vq.cross_attn.check_small_status()
    for k in keys:
        assert torch.equal(replayed[k], eager[k]), k


@pytest.mark.parametrize("d,topk", [(64, 5), (64, 1), (768, 5), (40, 8), (128, 2)])
def test_batched_small_searches_equal_the_single_calls(dev, d, topk):
    """ops.soft_vq_forward_multi (three launches for all searches) against ops.soft_vq_forward per search: every output, bit for bit --
    ragged code counts, a region slice of a larger codebook, widths that are not a multiple of the k block, a [2B, d] view output."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(100 + d + topk)
    W = torch.randn(3000 if d > 64 else 21000, d, device=dev, generator=g)
    what, wsq = ops.rownorm(W)
    region = W.shape[0] // 3
    shapes = [(512, 0, W.shape[0]), (256, 0, region), (256, W.shape[0] - region, W.shape[0]), (1, 5, 5 + max(topk, 130)), (77, 0, 129)]
    emb = torch.empty(256, 2 * d, device=dev)
    searches, singles = [], []
    for i, (n, lo, hi) in enumerate(shapes):
        x = torch.randn(n, d, device=dev, generator=g)
        assert ops.multi_search_eligible(n, hi - lo, d, topk)
        out = emb.view(512, d) if i == 0 else None
        searches.append(dict(x=x, what=what[lo:hi], wsq=wsq[lo:hi].contiguous(), out=out))
    res = ops.soft_vq_forward_multi(searches, topk)
    assert all(r["row_sqerr"] is None for r in res)
    zq0 = res[0]["zq"].clone()
    res_train = ops.soft_vq_forward_multi(searches, topk, want_sqerr=True)      # the training forward's call: + per-row squared errors
    for q, r, rt in zip(searches, res, res_train):
        one = ops.soft_vq_forward(q["x"], q["what"], q["wsq"], topk, want_sqerr=False)
        one_train = ops.soft_vq_forward(q["x"], q["what"], q["wsq"], topk, want_sqerr=True)
        assert torch.equal(one_train["row_sqerr"], rt["row_sqerr"]), ("row_sqerr", q["x"].shape)
        for key in ("xhat", "idx", "dist", "w"):
            assert torch.equal(one[key], r[key]), (key, q["x"].shape, q["what"].shape)
            assert torch.equal(one[key], rt[key]), (key, "want_sqerr", q["x"].shape)
        assert torch.equal(one["zq"], r["zq"] if q["out"] is None else zq0), ("zq", q["x"].shape)
        exact = ops.topk_search(*ops.rownorm(q["x"]), q["what"], q["wsq"], topk, ops.PATH_F32_MFMA)
        assert torch.equal(exact[0], r["idx"]) and torch.equal(exact[1], r["dist"])


def test_batched_usage_updates_equal_the_updates_one_by_one(dev):
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    for wlen, sizes, n_codes in ((300000, (2560, 1280, 1280), 21000), (1000, (300, 800, 1500, 7, 1), 50), (64, (10, 0, 3), 9)):
        win_a = torch.randint(0, n_codes, (wlen,), device=dev, generator=g).float()
        win_a[: wlen // 3] = 0.0
        win_b = win_a.clone()
        ids = [torch.randint(0, n_codes, (m,), device=dev, generator=g) for m in sizes]
        one_by_one = [int(ops.usage_update_(win_a, t, n_codes).item()) for t in ids]
        together = ops.usage_update_multi_(win_b, ids, n_codes).cpu().tolist()
        assert together == one_by_one, (wlen, sizes)
        assert torch.equal(win_a, win_b)


@pytest.mark.parametrize("aug", [False, True])
def test_small_batch_forward_equals_the_general_forward(dev, aug):
    """the B = 256 forward through ONE batched search call + ONE usage call against the per-search form: every tensor of the dict bit
    for bit, the usage floats and the window too (show_usage = True: the reference's default, tokenizer.py:72)."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    inp = list(_forward_inputs(dev))
    if aug:
        inp.append(inp[0] + 0.05 * torch.randn_like(inp[0]))
    outs, wins = [], []
    for batched in (True, False):
        vq = _quantizer(dev, show_usage=True)
        keep = vqmod.BATCHED_SMALL_SEARCHES
        vqmod.BATCHED_SMALL_SEARCHES = batched
        try:
            with torch.no_grad():
                outs.append([vq(*inp), vq(*inp)][1])           # (twice: the window carries state from call to call)
        finally:
            vqmod.BATCHED_SMALL_SEARCHES = keep
        wins.append(vq.codebook_used.clone())
    a, b = outs
    assert list(a) == list(b)
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y), k
        elif isinstance(x, tuple):
            for i, (p, q) in enumerate(zip(x, y)):
                assert torch.equal(p, q), (k, i)
        else:
            assert x == y, k
    assert torch.equal(wins[0], wins[1])


def test_small_batch_forward_at_d768_equals_the_general_forward(dev):
    """a serving batch at BASELINE's width (B = 200, D = 768, n_e = 49152: every search on the exact path): the batched search / usage
    calls behind the layer-by-layer cross-attention (the two-launch path is e_dim = 64 only) against the per-search form, bit for bit"""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd import ops
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    bsz, d = 200, 768
    g = torch.Generator(device=dev).manual_seed(8)
    text = torch.randn(bsz, 64, d, device=dev, generator=g)
    tok = torch.randint(1, 65, (bsz,), device=dev, generator=g)
    mask = (torch.arange(64, device=dev)[None, :] < tok[:, None]).to(torch.int64)
    n_nodes = torch.randint(1, 12, (bsz,), device=dev, generator=g)
    batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
    nodes = torch.randn(int(n_nodes.sum()), d, device=dev, generator=g)
    z = torch.randn(bsz, 2 * d, device=dev, generator=g)
    assert ops.multi_search_eligible(2 * bsz, 49152, d, 5) and ops.multi_search_eligible(bsz, 16384, d, 5)
    outs = []
    for batched in (True, False):
        torch.manual_seed(2)
        vq = VectorQuantizer(49152, d, 0.25, 0.0, True, True, [d, d], k=5).to(dev).eval()
        keep = vqmod.BATCHED_SMALL_SEARCHES
        vqmod.BATCHED_SMALL_SEARCHES = batched
        try:
            with torch.no_grad():
                outs.append(vq(z, text, nodes, mask, batch))
        finally:
            vqmod.BATCHED_SMALL_SEARCHES = keep
    a, b = outs
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y), k
        elif isinstance(x, tuple):
            for i, (p, q) in enumerate(zip(x, y)):
                assert torch.equal(p, q), (k, i)
        else:
            assert x == y, k
