"""Deterministic synthetic tensors shared by the fixture generator and tests.

TEST INFRASTRUCTURE ONLY (see oracle/medtok_oracle.c header).

Tensors are keyed by name so a module's parameters can be regenerated on any
box without shipping multi-megabyte state dicts: torch's CPU generator
(mt19937 + Box-Muller) is deterministic for a given torch build, and the dev
container and the GPU box run the same image.
"""
from __future__ import annotations

import zlib

import torch


def det_randn(name: str, shape, scale: float = 1.0, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def det_state_dict(module: torch.nn.Module, prefix: str, seed: int = 0) -> dict:
    """A full state dict for `module` made of named deterministic tensors.
    Weights ~N(0, 1/sqrt(fan_in)), biases small, LayerNorm weight near 1,
    embeddings N(0,1) (nn.Embedding's default, soft VQ codebook)."""
    out = {}
    for k, v in module.state_dict().items():
        name = f"{prefix}.{k}"
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        if k.endswith("codebook_used") or k.endswith("cluster_size") or k.endswith("initted"):
            out[k] = v.clone()
        elif "layer_norm.weight" in k:
            out[k] = 1.0 + det_randn(name, v.shape, 0.05, seed)
        elif k.endswith("bias") or "layer_norm.bias" in k:
            out[k] = det_randn(name, v.shape, 0.02, seed)
        elif k.endswith("codebook.weight"):
            out[k] = det_randn(name, v.shape, 1.0, seed)
        elif v.dim() >= 2:
            out[k] = det_randn(name, v.shape, 1.0 / (v.shape[-1] ** 0.5), seed)
        else:
            out[k] = det_randn(name, v.shape, 1.0, seed)
    return out


def ragged_batch(name: str, bsz: int, max_len: int, max_nodes: int, dim: int, seed: int = 0):
    """Synthetic (text token features, mask, graph node features, batch vector)
    shaped like what MultimodalTokenizer.quant() hands the quantizer
    (tokenizer.py:160-166,199): text [B,L,D] with a left-aligned attention
    mask, graph nodes packed [sum n_i, D] with a PyG-style batch vector."""
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 104729 * seed) & 0x7FFFFFFF)
    lens = torch.randint(2, max_len + 1, (bsz,), generator=g)
    nodes = torch.randint(1, max_nodes + 1, (bsz,), generator=g)
    text = torch.randn(bsz, max_len, dim, generator=g)
    mask = (torch.arange(max_len)[None, :] < lens[:, None]).long()
    node_feat = torch.randn(int(nodes.sum()), dim, generator=g)
    batch = torch.repeat_interleave(torch.arange(bsz), nodes)
    return text, mask, node_feat, batch
