"""GPU: BASELINE config 4 shape -- a train step through the tokenizer call sites under bf16 autocast.

The reference's BERT / GAT encoders are out of scope; plain-torch stand-ins of the same
interface feed the quantiser here (text encoder -> [B, L, 768] token features, 2-layer GAT over a synthetic
PrimeKG-shaped batch -> node features).  Parity of the quantiser / losses themselves in train mode, gradients
included, is pinned by the reference fixtures in test_gpu_modules.py; this test checks the assembled step."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


class ToyTextEncoder(nn.Module):
    def __init__(self, vocab=1000, dim=768):
        super().__init__()
        self.emb = nn.Embedding(vocab, dim)
        self.ff = nn.Linear(dim, dim)

    def forward(self, input_ids, attention_mask):
        return torch.tanh(self.ff(self.emb(input_ids)))


def synthetic_batch(bsz, dev, seed=0, max_len=64):
    from medtok_amd.synthetic import primekg_shaped_batch
    return primekg_shaped_batch(bsz, dev, seed=seed, max_len=max_len, vocab=1000)


def test_train_step_bf16_autocast(dev):
    from medtok_amd import loss as L
    from medtok_amd.tokenizer import MultimodalTokenizer
    torch.manual_seed(0)
    from medtok_amd.synthetic import StandInGAT as ToyGAT
    model = MultimodalTokenizer(ToyTextEncoder(), ToyGAT(), text_dim=768, graph_out_channels=64, codebook_size=21000,
                                codebook_embed_dim=64).to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    inputs = synthetic_batch(256, dev)
    model.train()
    w0 = model.quantize.codebook.weight.detach().clone()
    losses = []
    for _ in range(2):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            r = model(inputs)
            loss, parts = L.total_loss(r, 0.1, 0.1)
        assert torch.isfinite(loss)
        loss.float().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    g = model.quantize.codebook.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert (g.abs().sum(1) > 0).sum() <= 256 * 6 * 5           # sparse: only selected codes carry gradient
    assert float(model.graph_encoder.w[0].weight.grad.abs().sum()) > 0 and float(model.text_mapped.weight.grad.abs().sum()) > 0
    assert not torch.equal(model.quantize.codebook.weight.detach(), w0)
    for key in ("shared_codebook_usage", "text_specific_usage", "graph_specific_usage"):
        assert isinstance(r[key], float) and 0 < r[key] <= 1
    # eval / inference surface after training
    model.eval()
    with torch.no_grad():
        emb, tok, w = model(inputs)
        emb2 = model.tokenize(inputs)
    assert emb.shape == (256, 256) and tok.shape == (256, 4, 5) and w.shape == (256, 4, 5)
    # the toy GAT accumulates with float atomics (index_add_), so two encoder passes agree only to round-off
    assert emb2.shape == emb.shape and float((emb - emb2).abs().max()) < 1e-2


def test_train_step_fp16_autocast_with_grad_scaler(dev):
    """The reference's DEFAULT training mode (train_MedTok.py:99,212,240-247,394: --mixed-precision fp16): forward under fp16
    autocast, scaler.scale(loss).backward(), unscale_, clip_grad_norm_(1.0), scaler.step, scaler.update.  The searches, their
    sparse backward and the losses are fp32 kernels whatever autocast says; the scaled upstream gradient must pass through the
    custom autograd functions linearly: after unscale_ the gradients equal those of the same step without a scaler (the
    forward is identical: same autocast), and no inf/nan makes the scaler skip the step."""
    from medtok_amd import loss as L
    from medtok_amd.synthetic import StandInGAT
    from medtok_amd.tokenizer import MultimodalTokenizer

    def build():
        torch.manual_seed(0)
        m = MultimodalTokenizer(ToyTextEncoder(), StandInGAT(), text_dim=768, graph_out_channels=64, codebook_size=21000,
                                codebook_embed_dim=64).to(dev).train()
        for mod in m.modules():                       # identical forwards in both runs: no dropout noise
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, nn.MultiheadAttention):
                mod.dropout = 0.0
        return m
    inputs = synthetic_batch(128, dev, seed=3)
    watched = lambda m: [m.quantize.codebook.weight, m.quantize.proj_text.weight, m.quantize.cross_attn.model[1].multihead_attn.in_proj_weight,
                         m.text_mapped.weight, m.graph_encoder.w[0].weight]
    # (a) the reference's loop
    model = build()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    # (a moderate initial scale: at the default 2^16 the stand-in encoders' fp16 backward overflows in the first iterations, which
    # the scaler answers by skipping those steps -- part (c) below runs that protocol)
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 10)
    w0 = model.quantize.codebook.weight.detach().clone()
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        r = model(inputs)
        loss, _ = L.total_loss(r, 0.1, 0.1)
    assert torch.isfinite(loss)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    grads_scaled = [p.grad.detach().clone() for p in watched(model)]
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    scale_before = scaler.get_scale()
    scaler.step(opt)
    scaler.update()
    assert torch.isfinite(norm) and scaler.get_scale() >= scale_before            # no overflow: the step was taken, the scale kept
    assert not torch.equal(model.quantize.codebook.weight.detach(), w0)
    # (b) the same forward, plain backward
    model2 = build()
    with torch.autocast("cuda", dtype=torch.float16):
        r2 = model2(inputs)
        loss2, _ = L.total_loss(r2, 0.1, 0.1)
    loss2.float().backward()
    assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))
    for got, p in zip(grads_scaled, watched(model2)):
        ref = p.grad
        # the graph encoder accumulates with float atomics and the scaled pass rounds fp16 intermediates at a different exponent:
        # agreement to fp16 resolution relative to the largest entry
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-8, float((got - ref).abs().max()) / float(ref.abs().max())
    assert (grads_scaled[0].abs().sum(1) > 0).sum() <= 128 * 6 * 5                  # the codebook gradient stays sparse
    # (c) the reference's default scaler (init 2^16, train_MedTok.py:99): overflowing iterations are skipped and halve the scale;
    # within a few iterations steps are taken and the codebook moves
    model3 = build()
    opt3 = torch.optim.Adam([p for p in model3.parameters() if p.requires_grad], lr=1e-4)
    scaler3 = torch.amp.GradScaler("cuda", enabled=True)
    w3 = model3.quantize.codebook.weight.detach().clone()
    taken = 0
    for _ in range(10):
        opt3.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            l3, _ = L.total_loss(model3(inputs), 0.1, 0.1)
        assert torch.isfinite(l3)
        scaler3.scale(l3).backward()
        scaler3.unscale_(opt3)
        torch.nn.utils.clip_grad_norm_(model3.parameters(), 1.0)
        before = scaler3.get_scale()
        scaler3.step(opt3)
        scaler3.update()
        taken += int(scaler3.get_scale() >= before)
    assert taken >= 3 and not torch.equal(model3.quantize.codebook.weight.detach(), w3)
    assert torch.isfinite(model3.quantize.codebook.weight).all()


def _pooled_case(dev, d=128, heads=4, bsz=7, seq_len=24, seed=0):
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    g = torch.Generator(device="cpu").manual_seed(seed)
    ca = CrossAttention(d, heads, dropout=0.1).to(dev).train()
    counts = torch.tensor([3, 0, 9, 1, 40, 2, 5])[:bsz]
    batch = torch.repeat_interleave(torch.arange(bsz), counts).to(dev)
    valid = torch.tensor([24, 5, 1, 17, 24, 9, 2])[:bsz]
    mask = (torch.arange(seq_len)[None, :] < valid[:, None]).long().to(dev)
    text = torch.randn(bsz, seq_len, d, generator=g).to(dev)
    nodes = torch.randn(int(counts.sum()), d, generator=g).to(dev)
    probes = [torch.randn(bsz, d, generator=g).to(dev) for _ in range(3)]
    return ca, text, mask, nodes, batch, probes


@pytest.mark.parametrize("autocast", [None, torch.bfloat16])
def test_key_gradient_sink_gives_the_gradients_of_plain_autograd(dev, autocast):
    """KEY_GRADIENT_SINK: the layers' dKV kernels add into one buffer and the CLS gradients join it in place -- the gradients of the
    text rows, the nodes and every parameter equal those of plain autograd accumulation (same kernels, same dropout masks; only the
    order of the three-term sums on the CLS rows differs).  Also: a use of the text tensor outside the protocol, a backward that
    does not reach the text (torch.autograd.grad for the nodes alone) followed by a full one, and two backwards of one graph."""
    import contextlib
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    ca, text0, mask, nodes0, batch, (pa, pb, pc) = _pooled_case(dev)

    def run(sink, mode):
        old = vqmod.KEY_GRADIENT_SINK
        vqmod.KEY_GRADIENT_SINK = sink
        try:
            ca.zero_grad(set_to_none=True)
            text, nodes = text0.clone().requires_grad_(), nodes0.clone().requires_grad_()
            torch.manual_seed(11)                     # (the dropout seeds come from the host generator)
            ctx = torch.autocast("cuda", dtype=autocast) if autocast is not None else contextlib.nullcontext()
            with ctx:
                t = vqmod.fan_out_text(text) if mode == "fan-first" else text
                fan = getattr(t, "_medtok_fan", None)
                assert (fan is not None) == (sink and mode == "fan-first")
                pt, pg = ca.pooled(t, mask, nodes, batch)
                cls = t[:, 0] if fan is None else fan.cls
                loss = (pt.float() * pa).sum() + (pg.float() * pb).sum() + (cls * pc).sum() + t[:, 1].pow(2).sum()
            if mode == "partial-then-full":
                g_nodes, = torch.autograd.grad(loss, [nodes], retain_graph=True)
                loss.backward()
                assert torch.allclose(g_nodes, nodes.grad, rtol=0, atol=0)
            elif mode == "twice":
                loss.backward(retain_graph=True)
                first = text.grad.clone()
                text.grad = None
                nodes.grad = None
                ca.zero_grad(set_to_none=True)
                loss.backward()
                assert torch.equal(first, text.grad)
            else:
                loss.backward()
            return [text.grad.clone(), nodes.grad.clone()] + [p.grad.clone() for p in ca.parameters()]
        finally:
            vqmod.KEY_GRADIENT_SINK = old
    ref = run(False, "plain")
    assert float(ref[0].abs().max()) > 0 and float(ref[0][:, 2:].abs().max()) > 0      # the key gradients are there
    for mode in ("plain", "fan-first", "partial-then-full", "twice"):
        got = run(True, mode)
        for a, b in zip(got, ref):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6), mode


def test_tokenizer_training_forward_maps_only_the_cls_row_of_the_aug_text(dev):
    """MultimodalTokenizer.forward in training: the aug text view is only read at its CLS row (reference tokenizer.py:163), so only that
    row goes through text_mapped; the CLS rows of the main view come from the gradient fan-out.  Outputs and the gradients of
    text_mapped / the codebook equal the form that maps every token of both views (sink off, full aug map)."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd import loss as L
    from medtok_amd.tokenizer import MultimodalTokenizer, make_inputs
    bsz, seq_len, d = 12, 16, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    counts = torch.randint(1, 9, (bsz,), generator=g)
    inputs = make_inputs(batch=torch.repeat_interleave(torch.arange(bsz), counts).to(dev),
                         attention_mask=(torch.arange(seq_len)[None, :] < torch.randint(1, seq_len + 1, (bsz, 1), generator=g)).long().to(dev),
                         text_features=torch.randn(bsz, seq_len, 96, generator=g).to(dev),
                         text_features_aug=torch.randn(bsz, seq_len, 96, generator=g).to(dev),
                         graph_node_features=torch.randn(int(counts.sum()), d, generator=g).to(dev).requires_grad_(),
                         graph_node_features_aug=torch.randn(int(counts.sum()), d, generator=g).to(dev).requires_grad_())
    torch.manual_seed(3)
    model = MultimodalTokenizer(None, None, text_dim=96, graph_out_channels=d, codebook_size=3000, codebook_embed_dim=d).to(dev).train()
    for mod in model.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    old = (vqmod.KEY_GRADIENT_SINK, vqmod.TRAIN_SPLIT_TEXT_MAPPING)

    def run(sink, full_aug):
        vqmod.KEY_GRADIENT_SINK, vqmod.TRAIN_SPLIT_TEXT_MAPPING = sink, True
        model.zero_grad(set_to_none=True)
        inputs.graph_node_features.grad = inputs.graph_node_features_aug.grad = None
        if full_aug:                                   # the reference's form: every token of the aug view through the Linear
            x = make_inputs(**vars(inputs))
            x.text_features_aug = None
            text_aug = model._map_text(inputs.text_features_aug)
            text = model._map_text(inputs.text_features)
            from medtok_amd.tokenizer import global_mean_pool
            r = model.quant(text, inputs.graph_node_features, global_mean_pool(inputs.graph_node_features, inputs.batch, bsz), text_aug,
                            inputs.graph_node_features_aug, global_mean_pool(inputs.graph_node_features_aug, inputs.batch, bsz),
                            inputs.attention_mask, inputs.batch)
        else:
            r = model(inputs)
        loss, _ = L.total_loss(r, 0.1, 0.1)
        loss.backward()
        return float(loss), [model.text_mapped.weight.grad.clone(), model.text_mapped.bias.grad.clone(), model.quantize.codebook.weight.grad.clone(),
                             inputs.graph_node_features.grad.clone(), model.quantize.cross_attn.model[0].multihead_attn.in_proj_weight.grad.clone()]
    try:
        loss_ref, ref = run(False, True)
        loss_new, got = run(True, False)
    finally:
        vqmod.KEY_GRADIENT_SINK, vqmod.TRAIN_SPLIT_TEXT_MAPPING = old
    assert abs(loss_new - loss_ref) <= 1e-5 * abs(loss_ref)
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("autocast", [None, torch.float16])
def test_two_sided_attention_node_equals_the_two_nodes(dev, autocast):
    """TWO_SIDED_ATTENTION_NODE: both directions of a training layer under one autograd node (outputs and dQ in row ranges of one
    matrix) against the two-node form (split, two _RaggedAttentionFunction nodes, concatenation): the same kernels on the same rows --
    outputs and every gradient bit for bit (dropout off: the two forms draw their mask seeds differently)."""
    import contextlib
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    ca, text0, mask, nodes0, batch, (pa, pb, pc) = _pooled_case(dev, d=256, seed=4)
    for layer in ca.model:
        layer.multihead_attn.dropout = 0.0
        layer.dropout.p = 0.0

    def run(two_sided):
        old = vqmod.TWO_SIDED_ATTENTION_NODE
        vqmod.TWO_SIDED_ATTENTION_NODE = two_sided
        try:
            ca.zero_grad(set_to_none=True)
            text, nodes = text0.clone().requires_grad_(), nodes0.clone().requires_grad_()
            ctx = torch.autocast("cuda", dtype=autocast) if autocast is not None else contextlib.nullcontext()
            with ctx:
                pt, pg = ca.pooled(text, mask, nodes, batch)
                loss = (pt.float() * pa).sum() + (pg.float() * pb).sum()
            loss.backward()
            return [pt.detach().clone(), pg.detach().clone(), text.grad.clone(), nodes.grad.clone()] + [p.grad.clone() for p in ca.parameters()]
        finally:
            vqmod.TWO_SIDED_ATTENTION_NODE = old
    a, b = run(True), run(False)
    assert float(a[2].abs().max()) > 0 and float(a[3].abs().max()) > 0
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_batched_training_searches_switch_gives_the_same_bits(dev):
    """TRAIN_BATCHED_SEARCHES (off by default: slower at BASELINE's codebook sizes): all searches of a training forward in one batched call
    (ops.soft_vq_forward_multi with per-row squared errors) -- every output of VectorQuantizer.forward and every gradient bit for bit
    those of the search-by-search form."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(2)
    bsz, seq_len, d = 9, 12, 128
    vq = VectorQuantizer(n_e=3000, e_dim=d, beta=0.25, entropy_loss_ratio=0.0, l2_norm=True, show_usage=True, split=[d, d]).to(dev).train()
    for mod in vq.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    g = torch.Generator(device="cpu").manual_seed(7)
    counts = torch.randint(1, 7, (bsz,), generator=g)
    batch = torch.repeat_interleave(torch.arange(bsz), counts).to(dev)
    mask = (torch.arange(seq_len)[None, :] < torch.randint(1, seq_len + 1, (bsz, 1), generator=g)).long().to(dev)
    z, z_aug = torch.randn(bsz, 2 * d, generator=g).to(dev), torch.randn(bsz, 2 * d, generator=g).to(dev)
    text0, nodes0 = torch.randn(bsz, seq_len, d, generator=g).to(dev), torch.randn(int(counts.sum()), d, generator=g).to(dev)
    state = {k: v.clone() for k, v in vq.state_dict().items()}

    def run(batched):
        old = vqmod.TRAIN_BATCHED_SEARCHES
        vqmod.TRAIN_BATCHED_SEARCHES = batched
        try:
            vq.load_state_dict(state)
            vq.zero_grad(set_to_none=True)
            text, nodes, zz = text0.clone().requires_grad_(), nodes0.clone().requires_grad_(), z.clone().requires_grad_()
            r = vq(zz, text, nodes, mask, batch, z_aug)
            loss = sum(t for t in r["shared_embed_loss"][:2]) + r["text_specific_loss"][0] + r["graph_specific_loss"][1] \
                + r["shared_text_embedding"].sum() * 0.01 + r["specific_embedding_graph_aug"].pow(2).sum() * 0.01
            loss.backward()
            outs = [loss.detach().clone(), r["shared_text_tokens"].clone(), r["text_tokens_weights"].clone(), r["specific_embedding_text"].detach().clone()]
            return outs + [zz.grad.clone(), text.grad.clone(), nodes.grad.clone(), vq.codebook.weight.grad.clone()]
        finally:
            vqmod.TRAIN_BATCHED_SEARCHES = old
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # ... and TRAIN_STACK_SEARCHES_OF_A_REGION (the searches of one region on their rows stacked; on by default): the same bits as off
    keep = vqmod.TRAIN_STACK_SEARCHES_OF_A_REGION
    try:
        vqmod.TRAIN_STACK_SEARCHES_OF_A_REGION = False
        c = run(False)
    finally:
        vqmod.TRAIN_STACK_SEARCHES_OF_A_REGION = keep
    for x, y in zip(a, c):
        assert torch.equal(x, y)


def test_max_nodes_bound_in_training_matches_the_host_read_and_flags_bad_batches(dev):
    """CrossAttention.max_nodes_bound in a TRAINING forward (round 6): launches sized from the bound, no host read in pooled() -- outputs and
    gradients equal those of the form that reads the largest node count back; a code above the bound and an unsorted batch vector are
    flagged on the device and raised with the usage counts' read (VectorQuantizer.forward) or by check_status()."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer, UNSORTED_BATCH_MESSAGE
    torch.manual_seed(4)
    bsz, seq_len, d = 10, 16, 128
    vq = VectorQuantizer(n_e=3000, e_dim=d, beta=0.25, entropy_loss_ratio=0.0, l2_norm=True, show_usage=True, split=[d, d]).to(dev).train()
    for mod in vq.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    g = torch.Generator(device="cpu").manual_seed(9)
    counts = torch.randint(1, 12, (bsz,), generator=g)
    batch = torch.repeat_interleave(torch.arange(bsz), counts).to(dev)
    mask = (torch.arange(seq_len)[None, :] < torch.randint(1, seq_len + 1, (bsz, 1), generator=g)).long().to(dev)
    z, z_aug = torch.randn(bsz, 2 * d, generator=g).to(dev), torch.randn(bsz, 2 * d, generator=g).to(dev)
    text0, nodes0 = torch.randn(bsz, seq_len, d, generator=g).to(dev), torch.randn(int(counts.sum()), d, generator=g).to(dev)
    state = {k: v.clone() for k, v in vq.state_dict().items()}

    def run(bound, batch_vec=batch, nodes_in=nodes0):
        vq.load_state_dict(state)
        vq.zero_grad(set_to_none=True)
        vq.cross_attn.max_nodes_bound = bound
        try:
            text, nodes = text0.clone().requires_grad_(), nodes_in.clone().requires_grad_()
            r = vq(z, text, nodes, mask, batch_vec, z_aug)
            loss = r["shared_embed_loss"][0] + r["shared_embed_loss"][1] + r["shared_text_embedding"].sum() * 0.01 + r["shared_graph_embedding"].pow(2).sum() * 0.01
            loss.backward()
            return [loss.detach().clone(), r["shared_graph_tokens"].clone(), text.grad.clone(), nodes.grad.clone(), vq.cross_attn.model[0].multihead_attn.in_proj_weight.grad.clone()]
        finally:
            vq.cross_attn.max_nodes_bound = None
    ref = run(None)
    for bound in (int(counts.max()), int(counts.max()) + 37):
        got = run(bound)
        for a, b in zip(got, ref):
            assert torch.equal(a, b) if a.dtype == torch.int64 else float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6)
    assert not torch.equal(vq.codebook_used, state["codebook_used"])            # (a clean forward slides the usage window)
    with pytest.raises(ValueError, match="max_nodes_bound"):
        run(int(counts.max()) - 1)
    assert torch.equal(vq.codebook_used, state["codebook_used"])                 # (run() restores the state; the flagged forward did not touch the window)
    perm = torch.randperm(batch.numel(), generator=g).to(dev)
    with pytest.raises(ValueError) as e:
        run(int(counts.max()), batch_vec=batch[perm], nodes_in=nodes0[perm])
    assert UNSORTED_BATCH_MESSAGE[:40] in str(e.value)
    assert run(int(counts.max()))[0].isfinite()                                  # (the word was cleared: the next forward is clean)
    # pooled() alone: the caller's check_status()
    vq.cross_attn.max_nodes_bound = int(counts.max()) - 1
    try:
        vq.cross_attn.pooled(text0.clone().requires_grad_(), mask, nodes0.clone().requires_grad_(), batch)
        with pytest.raises(ValueError, match="max_nodes_bound"):
            vq.cross_attn.check_status()
    finally:
        vq.cross_attn.max_nodes_bound = None


def test_prepacked_prologue_gives_the_same_forward_and_is_taken_only_for_its_own_inputs(dev):
    """CrossAttention.prepack(mask, batch): pooled() on the same two tensors takes the early prologue (outputs equal those without it, in
    training and in eval); a pooled() call on OTHER tensors ignores it; an in-place change of the batch vector in between invalidates it."""
    ca, text0, mask, nodes0, batch, _ = _pooled_case(dev, d=128)
    for layer in ca.model:
        layer.multihead_attn.dropout = 0.0
        layer.dropout.p = 0.0
    for train in (True, False):
        ca.train(train)
        with torch.set_grad_enabled(train):
            ref = ca.pooled(text0, mask, nodes0, batch)
            ca.prepack(mask, batch)
            assert ca._prepacked is not None
            got = ca.pooled(text0, mask, nodes0, batch)
            assert ca._prepacked is None                                  # taken
            assert all(torch.equal(a, b) for a, b in zip(got, ref))
            # other tensors: ignored (and dropped)
            ca.prepack(mask, batch)
            other = ca.pooled(text0, mask.clone(), nodes0, batch)
            assert all(torch.equal(a, b) for a, b in zip(other, ref)) and ca._prepacked is None
            # the batch vector changed in place after prepack(): its version no longer matches
            b2 = batch.clone()
            ca.prepack(mask, b2)
            b2.add_(0)
            again = ca.pooled(text0, mask, nodes0, b2)
            assert all(torch.equal(a, b) for a, b in zip(again, ref))


@pytest.mark.parametrize("d", [64, 96, 128, 256, 384, 512, 640, 768])
def test_training_cross_attention_matches_the_torch_comparator_at_every_width(dev, d):
    """CrossAttention.pooled in TRAINING at every kernel width (and one that is padded: 96) against the plain-torch comparator
    (pooled_reference: nn.MultiheadAttention on padded batches): outputs and the gradients of text, nodes and in_proj -- 1e-5 in fp32,
    2e-2 under bf16 autocast.  (Round 6: the fp32 dKV kernel was wrong at D = 640 and nothing ran that width under autograd.)"""
    import contextlib
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    torch.manual_seed(d)
    ca = CrossAttention(d, 4, dropout=0.0).to(dev).train()
    bsz, seq_len = 6, 40
    counts = torch.tensor([3, 0, 9, 1, 70, 2])
    batch = torch.repeat_interleave(torch.arange(bsz), counts).to(dev)
    mask = (torch.arange(seq_len)[None, :] < torch.tensor([40, 5, 1, 17, 33, 9])[:, None]).long().to(dev)
    text0, nodes0 = torch.randn(bsz, seq_len, d, device=dev), torch.randn(int(counts.sum()), d, device=dev)
    pa, pb = torch.randn(bsz, d, device=dev), torch.randn(bsz, d, device=dev)
    keep = (counts > 0).to(dev)                # (a code without nodes: the kernels give a zero context, the padded comparator a softmax over nothing)

    def run(fn, autocast=None):
        ca.zero_grad(set_to_none=True)
        t, n = text0.clone().requires_grad_(), nodes0.clone().requires_grad_()
        with (torch.autocast("cuda", dtype=autocast) if autocast is not None else contextlib.nullcontext()):
            pt, pg = fn(t, mask, n, batch)
        ((pt.float() * pa)[keep].sum() + (pg.float() * pb)[keep].sum()).backward()
        return [pt.float()[keep].detach(), pg.float()[keep].detach(), t.grad[keep].clone(), n.grad.clone(), ca.model[0].multihead_attn.in_proj_weight.grad.clone()]
    ref = run(ca.pooled_reference)
    for autocast, tol in ((None, 1e-5), (torch.bfloat16, 2e-2)):
        for a, b in zip(run(ca.pooled, autocast), ref):
            assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-9), (d, autocast)
