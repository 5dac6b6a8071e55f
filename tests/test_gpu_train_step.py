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


class ToyGAT(nn.Module):
    """2-layer single-head graph attention over an edge list (what tokenizer.py's GraphEncoder is configured for)."""
    def __init__(self, n_nodes=130000, dim=64):
        super().__init__()
        self.emb = nn.Embedding(n_nodes, dim)
        self.w = nn.ModuleList([nn.Linear(dim, dim, bias=False) for _ in range(2)])
        self.a = nn.ParameterList([nn.Parameter(torch.randn(2 * dim) * 0.1) for _ in range(2)])

    def forward(self, x, edge_index, rel_index):
        h = self.emb(x)
        src, dst = edge_index
        for w, a in zip(self.w, self.a):
            z = w(h).float()
            e = torch.nn.functional.leaky_relu((torch.cat([z[src], z[dst]], -1) * a).sum(-1), 0.2)
            e = torch.exp(e - e.max())
            denom = torch.zeros(h.shape[0], device=h.device, dtype=e.dtype).index_add_(0, dst, e) + 1e-9
            h = torch.relu(torch.zeros_like(z).index_add_(0, dst, z[src] * (e / denom[dst]).unsqueeze(-1)) + z)
        return [h]


def synthetic_batch(bsz, dev, seed=0, max_len=64):
    from medtok_amd.tokenizer import make_inputs
    g = torch.Generator().manual_seed(seed)
    n_nodes = torch.clamp(torch.exp(torch.randn(bsz, generator=g) * 0.7 + 3.0), 3, 200).long()       # median ~20
    batch = torch.repeat_interleave(torch.arange(bsz), n_nodes)
    total = int(n_nodes.sum())
    x = torch.randint(0, 130000, (total,), generator=g)
    starts = torch.cumsum(n_nodes, 0) - n_nodes
    def edges():
        src = torch.randint(0, 1 << 30, (4 * total,), generator=g)
        dst = torch.randint(0, 1 << 30, (4 * total,), generator=g)
        owner = torch.randint(0, bsz, (4 * total,), generator=g)
        return torch.stack([starts[owner] + src % n_nodes[owner], starts[owner] + dst % n_nodes[owner]])
    lens = torch.randint(4, max_len + 1, (bsz,), generator=g)
    mask = (torch.arange(max_len)[None] < lens[:, None]).long()
    ids = torch.randint(0, 1000, (bsz, max_len), generator=g)
    e, ea = edges(), edges()
    return make_inputs(input_ids=ids.to(dev), attention_mask=mask.to(dev), x=x.to(dev), edge_index=e.to(dev), rel_index=None,
                       edge_index_aug=ea.to(dev), rel_index_aug=None, batch=batch.to(dev), code_indices=torch.arange(bsz))


def test_train_step_bf16_autocast(dev):
    from medtok_amd import loss as L
    from medtok_amd.tokenizer import MultimodalTokenizer
    torch.manual_seed(0)
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
