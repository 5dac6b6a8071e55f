"""GPU: CrossAttention.max_nodes_bound -- pooled() / VectorQuantizer.forward at ANY width without the host read of the batch's largest
node count: the same bits as the default forward, the batch checks on the device word, and the eval forward replayed from a HIP graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dev, seed, bsz, seq_len, max_nodes, dim):
    g = torch.Generator(device=dev).manual_seed(seed)
    text = torch.randn(bsz, seq_len, dim, device=dev, generator=g)
    tok = torch.randint(1, seq_len + 1, (bsz,), device=dev, generator=g)
    n_nodes = torch.randint(1, max_nodes + 1, (bsz,), device=dev, generator=g)
    n_nodes[0] = max_nodes
    if bsz > 3:
        n_nodes[2] = 0
    mask = (torch.arange(seq_len, device=dev)[None, :] < tok[:, None]).to(torch.int64)
    batch = torch.repeat_interleave(torch.arange(bsz, device=dev), n_nodes)
    nodes = torch.randn(int(n_nodes.sum()), dim, device=dev, generator=g)
    return text, mask, nodes, batch


def _cross_attention(dev, dim, heads=4):
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    torch.manual_seed(3)
    ca = CrossAttention(dim, heads, dropout=0.1, layers=2).to(dev).eval()
    with torch.no_grad():
        for layer in ca.model:
            layer.multihead_attn.in_proj_bias.normal_(0, 0.2)
            layer.layer_norm.weight.normal_(1.0, 0.2)
    return ca


@pytest.mark.parametrize("dim,bsz,seq_len,max_nodes", [(768, 300, 96, 24), (256, 130, 200, 40), (128, 64, 50, 9), (100, 17, 33, 5)])
def test_pooled_with_a_node_bound_equals_the_default_pooled(dev, monkeypatch, dim, bsz, seq_len, max_nodes):
    """launches sized from the bound (equal to, and larger than, the batch's largest node count) against launches sized from the host
    read: the same bits -- wide batches on the tile-pair kernels, small ones on the few-rows kernels, a width that is zero-padded."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    monkeypatch.setattr(vqmod, "SMALL_WIDTH_FUSED", False)
    ca = _cross_attention(dev, dim)
    text, mask, nodes, batch = _inputs(dev, dim + bsz, bsz, seq_len, max_nodes, dim)
    with torch.no_grad():
        ref = ca.pooled(text, mask, nodes, batch)
        for bound in (max_nodes, max_nodes + 23):
            ca.max_nodes_bound = bound
            got = ca.pooled(text, mask, nodes, batch)
            ca.check_status()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), bound


def test_node_bound_violations_are_flagged_on_the_device(dev, monkeypatch):
    """what the host read would have rejected or repaired: a code with more nodes than the bound, an unsorted batch vector, an id out of
    range -- raised by check_status(), which clears the word; training ignores the bound (it reads the count as before)."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    monkeypatch.setattr(vqmod, "SMALL_WIDTH_FUSED", False)
    ca = _cross_attention(dev, 128)
    text, mask, nodes, batch = _inputs(dev, 5, 40, 30, 12, 128)
    with torch.no_grad():
        ca.max_nodes_bound = 11
        ca.pooled(text, mask, nodes, batch)
        with pytest.raises(ValueError, match="max_nodes_bound"):
            ca.check_status()
        ca.check_status()                                        # cleared
        ca.max_nodes_bound = 12
        perm = torch.randperm(batch.numel(), device=dev)
        ca.pooled(text, mask, nodes[perm], batch[perm])
        with pytest.raises(ValueError, match="non-decreasing"):
            ca.check_status()
        bad = batch.clone(); bad[-1] = 40
        ca.pooled(text, mask, nodes, bad)
        with pytest.raises(ValueError, match=r"outside \[0, B\)"):
            ca.check_status()
        with pytest.raises(ValueError, match="positive"):
            ca.max_nodes_bound = 0
            ca.pooled(text, mask, nodes, batch)
        # the default path still repairs an unsorted vector
        ca.max_nodes_bound = None
        ref = ca.pooled(text, mask, nodes, batch)
        got = ca.pooled(text, mask, nodes[perm], batch[perm])
        assert torch.allclose(got[1], ref[1], rtol=0, atol=2e-5 * float(ref[1].abs().max()))
    ca.train()
    ca.max_nodes_bound = 3                                       # (below the batch's counts: ignored under autograd)
    out = ca.pooled(text, mask, nodes.requires_grad_(True), batch)
    out[1].sum().backward()
    assert nodes.grad is not None and torch.isfinite(nodes.grad).all()


@pytest.mark.parametrize("dim", [256, 768])
def test_wide_forward_replayed_from_a_hip_graph_is_bit_equal_to_eager(dev, monkeypatch, dim):
    """With the bound there is no host read in the eval forward at any width (show_usage = False): it records into a HIP graph and the
    replay -- on NEW values in the captured buffers -- equals the eager forward, with and without the bound, bit for bit."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    monkeypatch.setattr(vqmod, "SIDE_STREAM_MIN_CODES", 0)       # one stream inside the capture
    torch.manual_seed(7)
    vq = VectorQuantizer(3 * 1024, dim, 0.25, 0.0, True, False, [dim, dim], k=5).to(dev).eval()
    bsz, seq_len, max_nodes = 150, 80, 20
    text, mask, nodes, batch = _inputs(dev, 21, bsz, seq_len, max_nodes, dim)
    g = torch.Generator(device=dev).manual_seed(1)
    h = torch.randn(bsz, 2 * dim, device=dev, generator=g)
    inp = [h, text, nodes, mask, batch]
    keys = ("shared_text_embedding", "shared_graph_embedding", "specific_embedding_text", "specific_embedding_graph", "text_tokens", "graph_tokens",
            "shared_text_tokens", "shared_graph_tokens", "shared_text_tokens_weights", "shared_graph_tokens_weights")
    with torch.no_grad():
        vq.cross_attn.max_nodes_bound = max_nodes
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                vq(*inp)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = vq(*inp)
        g = torch.Generator(device=dev).manual_seed(99)
        for t in (h, text, nodes):
            t.copy_(torch.randn(t.shape, device=dev, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        replayed = {k: out[k].clone() for k in keys}
        eager = vq(*inp)
        vq.cross_attn.check_status()
        vq.cross_attn.max_nodes_bound = None
        default = vq(*inp)
    for k in keys:
        assert torch.equal(replayed[k], eager[k]), k
        assert torch.equal(default[k], eager[k]), k
