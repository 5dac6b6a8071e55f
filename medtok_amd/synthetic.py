"""Synthetic PrimeKG-shaped batches and plain-torch STAND-IN encoders for BASELINE config 4 (the train step).

The reference's BERT text encoder and GCN/GAT graph encoder are upstream of the VQ path and out of scope (SURVEY.md
section 2.1); SURVEY 8d asks for "the build's own plain-torch BERT-shaped + 2-layer GAT encoders" so that the quantise call
sites of tokenizer.py / train_MedTok.py:207-250 can be driven end to end at the real shapes (B = 256 codes, 512 tokens,
subgraphs of ~20 nodes).  Used by bench.py --workload cfg4 and the cfg-4 tests; not part of the product path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .tokenizer import make_inputs


class StandInTextEncoder(nn.Module):
    """BERT-base-shaped encoder (hidden 768, 12 heads, FFN 3072; `layers` of them, 12 = BERT-base): token + position
    embeddings, post-LN transformer layers, returns [B, L, 768] token features like `last_hidden_state`."""

    def __init__(self, layers: int = 12, vocab: int = 30522, dim: int = 768, heads: int = 12, ffn: int = 3072, max_len: int = 512):
        super().__init__()
        self.tok = nn.Embedding(vocab, dim)
        self.pos = nn.Embedding(max_len, dim)
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.TransformerEncoderLayer(dim, heads, ffn, dropout=0.1, activation="gelu", batch_first=True)
                                     for _ in range(layers)])

    def forward(self, input_ids, attention_mask):
        x = self.norm(self.tok(input_ids) + self.pos(torch.arange(input_ids.shape[1], device=input_ids.device))[None])
        pad = ~attention_mask.bool()
        for layer in self.layers:
            x = layer(x, src_key_padding_mask=pad)
        return x


class StandInGAT(nn.Module):
    """2-layer single-head graph attention over an edge list (the shape tokenizer.py's GraphEncoder is configured for:
    node-id embedding table of 130 000 PrimeKG nodes, tokenizer.py:84)."""

    def __init__(self, n_nodes: int = 130000, dim: int = 64):
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


def primekg_shaped_batch(bsz: int, dev, seed: int = 0, max_len: int = 512, vocab: int = 30522):
    """SURVEY 8d cfg 4: subgraph sizes ~ clipped log-normal (median ~ 20, max 200 nodes), edges ~ 4 x nodes, node ids uniform in
    [0, 130000), `max_len` text tokens with a random valid length; an augmented edge set for the second view."""
    g = torch.Generator().manual_seed(seed)
    n_nodes = torch.clamp(torch.exp(torch.randn(bsz, generator=g) * 0.7 + 3.0), 3, 200).long()
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
    ids = torch.randint(0, vocab, (bsz, max_len), generator=g)
    e, ea = edges(), edges()
    return make_inputs(input_ids=ids.to(dev), attention_mask=mask.to(dev), x=x.to(dev), edge_index=e.to(dev), rel_index=None,
                       edge_index_aug=ea.to(dev), rel_index_aug=None, batch=batch.to(dev), code_indices=torch.arange(bsz))
