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
