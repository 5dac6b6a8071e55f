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
