"""Inference driver: embeddings -> (quantised embedding, token ids, weights) -> three .npy files.

Keeps the output contract of the reference's inference.py (:105-138):
  embeddings_all.npy [num_codes, 4*e_dim] fp32
  tokens_all.npy     [num_codes, 4, k]    int64   (text, graph, shared-text, shared-graph)
  weights_all.npy    [num_codes, 4, k]    fp32
ordered by the dataset's code index (the reference sorts by the last batch's indices
only, :119-121; the evident intent -- order by code_indices -- is what happens here).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional

import numpy as np
import torch

from .tokenizer import MultimodalTokenizer
from .vector_quantization_soft_one_new import VectorQuantizer


@torch.no_grad()
def quantize_pooled(vq: VectorQuantizer, h: torch.Tensor, pooled_text: torch.Tensor, pooled_graph: torch.Tensor):
    """The four searches of VectorQuantizer.forward for inputs whose cross-attention pooling
    is already done (BASELINE config 3): h [N, 2*e_dim] -> specific text/graph searches over
    their codebook thirds; pooled_* [N, e_dim] -> shared searches over the whole codebook.
    Returns (embedding [N, 4*e_dim], tokens [N, 4, k], weights [N, 4, k])."""
    was_training = vq.training
    vq.eval()
    try:
        h_text, h_graph = torch.split(h, vq.split, dim=-1)
        e = vq.e_dim
        # the four searches write straight into their column block of the result (the reference's torch.cat, :246)
        embedding = torch.empty((h.shape[0], 4 * e), dtype=torch.float32, device=h.device)
        _, _, _, _, idx_t, w_t = vq._search(vq.proj_text(h_text), "text", False, out=embedding[:, 0:e])
        _, _, _, _, idx_g, w_g = vq._search(vq.proj_graph(h_graph), "graph", False, out=embedding[:, e:2 * e])
        _, _, _, _, idx_st, w_st = vq._search(pooled_text, "shared", False, out=embedding[:, 2 * e:3 * e])
        _, _, _, _, idx_sg, w_sg = vq._search(pooled_graph, "shared", False, out=embedding[:, 3 * e:4 * e])
    finally:
        vq.train(was_training)
    tokens = torch.stack((idx_t, idx_g, idx_st, idx_sg), dim=1)
    weights = torch.stack((w_t, w_g, w_st, w_sg), dim=1)
    return embedding, tokens, weights


def save_outputs(out_dir, embeddings: np.ndarray, tokens: np.ndarray, weights: np.ndarray) -> None:
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    np.save(out / "embeddings_all.npy", embeddings.astype(np.float32, copy=False))
    np.save(out / "tokens_all.npy", tokens.astype(np.int64, copy=False))
    np.save(out / "weights_all.npy", weights.astype(np.float32, copy=False))


@torch.no_grad()
def run_inference(model: MultimodalTokenizer, batches: Iterable, out_dir: Optional[str] = None, device=None):
    """Loop of inference.py:105-115 over `batches` (objects with the fields MultimodalTokenizer.forward
    reads plus `code_indices`), then order by code index and optionally write the three arrays."""
    model.eval()
    embs, toks, wts, order = [], [], [], []
    for x in batches:
        if device is not None and hasattr(x, "to"):
            x = x.to(device)
        e, t, w = model(x)
        embs.append(e.cpu()); toks.append(t.cpu()); wts.append(w.cpu())
        order.append(torch.as_tensor(x.code_indices).reshape(-1).cpu())
    order = torch.cat(order)
    perm = torch.argsort(order, stable=True)
    embeddings = torch.cat(embs)[perm].numpy()
    tokens = torch.cat(toks)[perm].numpy()
    weights = torch.cat(wts)[perm].numpy()
    if out_dir is not None:
        save_outputs(out_dir, embeddings, tokens, weights)
    return embeddings, tokens, weights
