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


class _EncoderLayer(nn.Module):
    """post-LN transformer layer written out in plain ops: every GEMM is a 2-D [B*L, d] x [d, *] product and the attention is
    F.scaled_dot_product_attention without a mask.  (nn.TransformerEncoderLayer's inference fast path packs the unpadded tokens
    into a nested tensor; the odd row counts it produces hit a memory fault in this image's hipBLASLt bf16 stream-K kernels.
    The stand-in therefore lets padded positions attend like any other token -- its outputs there are never read: the VQ path
    masks them as keys and only uses the CLS row as a query.)"""

    def __init__(self, dim, heads, ffn):
        super().__init__()
        self.heads = heads
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.ff1, self.ff2 = nn.Linear(dim, ffn), nn.Linear(ffn, dim)
        self.ln1, self.ln2 = nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x):
        b, l, d = x.shape
        qkv = self.qkv(x.reshape(b * l, d)).view(b, l, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        att = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        x = self.ln1(x + self.proj(att.transpose(1, 2).reshape(b * l, d)).view(b, l, d))
        return self.ln2(x + self.ff2(torch.nn.functional.gelu(self.ff1(x.reshape(b * l, d)))).view(b, l, d))


class StandInTextEncoder(nn.Module):
    """BERT-base-shaped encoder (hidden 768, 12 heads, FFN 3072; `layers` of them, 12 = BERT-base): token + position
    embeddings, post-LN transformer layers, returns [B, L, 768] token features like `last_hidden_state`."""

    def __init__(self, layers: int = 12, vocab: int = 30522, dim: int = 768, heads: int = 12, ffn: int = 3072, max_len: int = 512):
        super().__init__()
        self.tok = nn.Embedding(vocab, dim)
        self.pos = nn.Embedding(max_len, dim)
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([_EncoderLayer(dim, heads, ffn) for _ in range(layers)])

    def forward(self, input_ids, attention_mask):
        x = self.norm(self.tok(input_ids) + self.pos(torch.arange(input_ids.shape[1], device=input_ids.device))[None])
        for layer in self.layers:
            x = layer(x)
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
