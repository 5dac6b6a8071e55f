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


# quantize_pooled: the four searches are independent (vector_quantization_soft_one_new.py:238-271).  With this set (> 0: from that many
# rows up) they are issued on TWO HIP streams, a short (region) and a long (whole-codebook) search on each in opposite order -- the
# round-6 experiment of running one stream's re-score / projection / row-norm passes under the other's shortlist kernel.  Measured
# 3.4 % SLOWER at cfg 3 (139.4 vs 134.9 ms, profiles/r06_two_stream_searches.json): the shortlist kernel's two waves per SIMD hold
# 480 of its 512 registers, no re-score wave co-resides, the CUs are merely time-sliced.  Off (0); bit-identical either way.
TWO_STREAM_MIN_ROWS = 0


@torch.no_grad()
def quantize_pooled(vq: VectorQuantizer, h: torch.Tensor, pooled_text: torch.Tensor, pooled_graph: torch.Tensor):
    """The four searches of VectorQuantizer.forward for inputs whose cross-attention pooling
    is already done (BASELINE config 3): h [N, 2*e_dim] -> specific text/graph searches over
    their codebook thirds; pooled_* [N, e_dim] -> shared searches over the whole codebook.
    Returns (embedding [N, 4*e_dim], tokens [N, 4, k], weights [N, 4, k])."""
    from . import vector_quantization_soft_one_new as vqmod
    was_training = vq.training
    vq.eval()
    try:
        h_text, h_graph = torch.split(h, vq.split, dim=-1)
        e = vq.e_dim
        # the four searches write straight into their column block of the result (the reference's torch.cat, :246)
        embedding = torch.empty((h.shape[0], 4 * e), dtype=torch.float32, device=h.device)
        if h.is_cuda and 0 < TWO_STREAM_MIN_ROWS <= h.shape[0]:
            norm = vq._normalised_codebook(prepare=True)          # built once, on this stream, before the fork
            side, cur = vqmod._side_stream(h.device, 1)
            vqmod._lend(side, h, pooled_text, embedding)
            with torch.cuda.stream(side):
                _, _, _, _, idx_st, w_st = vq._search(pooled_text, "shared", False, out=embedding[:, 2 * e:3 * e], norm=norm)
                _, _, _, _, idx_g, w_g = vq._search(vq.project(h_graph, "graph"), "graph", False, out=embedding[:, e:2 * e], norm=norm)
            _, _, _, _, idx_t, w_t = vq._search(vq.project(h_text, "text"), "text", False, out=embedding[:, 0:e], norm=norm)
            _, _, _, _, idx_sg, w_sg = vq._search(pooled_graph, "shared", False, out=embedding[:, 3 * e:4 * e], norm=norm)
            vqmod._join_side(side, cur, (idx_st, w_st, idx_g, w_g))
        else:
            _, _, _, _, idx_t, w_t = vq._search(vq.project(h_text, "text"), "text", False, out=embedding[:, 0:e])
            _, _, _, _, idx_g, w_g = vq._search(vq.project(h_graph, "graph"), "graph", False, out=embedding[:, e:2 * e])
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
    reads plus `code_indices`), then order by code index and optionally write the three arrays.

    Every batch's results leave the GPU as they are made (asynchronous copies into pinned host memory, like the reference's
    per-batch `.cpu()`, :112-114): the table of all codes (several GB at 600k codes x 4 e_dim) is never resident in HBM.

    Multi-rank (reference inference.py:66-93: DistributedSampler, one process per GPU): every rank passes ITS batches; the
    per-rank results are gathered ON RANK 0 ONLY, in bounded chunks over the process group (RCCL on GPUs), ordered by code index
    -- a sampler that pads the last round by repeating codes leaves duplicates: the first copy is kept (multi-rank only; a
    single process keeps what its caller passed, like the reference) -- and rank 0 writes the files and returns the arrays; the
    other ranks return None.  (The reference lets every rank write its own shard over the same three files; the evident intent
    is one table of all codes, which is what the downstream readers of embeddings_all.npy expect.)
    `model` must be the bare module, not a DistributedDataParallel wrapper: inference synchronises no gradients, and DDP's
    per-forward buffer broadcast would hang when the ranks hold different numbers of batches."""
    from . import distributed as mdist
    model.eval()
    embs, toks, wts, order = [], [], [], []

    # Results leave the GPU batch by batch through TWO reusable pinned staging buffers per output (an event per buffer: the copy of
    # batch i overlaps the forward of batch i + 1) into pageable host tensors -- not one freshly page-locked tensor per output and
    # batch, all kept to the end (at 600k codes x 3072 floats that was 7+ GB of pinned memory and one hipHostMalloc per tensor).
    class _Stage:
        def __init__(self):
            self.buf, self.event, self.pending = [None, None], [None, None], [None, None]
            self.turn = 0

        def drain(self, i, sink):
            if self.pending[i] is not None:
                self.event[i].synchronize()
                shape, n = self.pending[i]
                sink.append(self.buf[i][:n].view(shape).clone())
                self.pending[i] = None

        def push(self, t, sink):
            if not t.is_cuda:
                sink.append(t)
                return
            i = self.turn
            self.turn ^= 1
            self.drain(i, sink)                      # the copy issued two batches ago from this buffer
            n = t.numel()
            if self.buf[i] is None or self.buf[i].numel() < n or self.buf[i].dtype != t.dtype:
                self.buf[i] = torch.empty(max(n, 1), dtype=t.dtype, pin_memory=True)
                self.event[i] = torch.cuda.Event()
            self.buf[i][:n].view(t.shape).copy_(t, non_blocking=True)
            self.event[i].record(torch.cuda.current_stream(t.device))
            self.pending[i] = (tuple(t.shape), n)

        def flush(self, sink):
            for i in (self.turn, self.turn ^ 1):     # oldest first: batch order is kept
                self.drain(i, sink)
    stages = (_Stage(), _Stage(), _Stage(), _Stage())
    dev = device
    for x in batches:
        # the code indices are taken BEFORE x.to(device); indices that already live on the GPU leave it like the results do -- through
        # the pinned double buffer, without a blocking copy per batch
        ci = torch.as_tensor(x.code_indices).reshape(-1)
        if ci.is_cuda:
            stages[3].push(ci.to(torch.int64), order)
        else:
            stages[3].flush(order)                   # (a caller that mixes host and device indices: batch order is kept)
            order.append(ci.to(torch.int64))
        if device is not None and hasattr(x, "to"):
            x = x.to(device)
        e, t, w = model(x)
        dev = e.device
        for st, val, sink in zip(stages, (e, t, w), (embs, toks, wts)):
            st.push(val, sink)
    for st, sink in zip(stages, (embs, toks, wts, order)):
        st.flush(sink)
    quant = getattr(model, "quantize", None)
    if quant is not None and hasattr(quant, "cross_attn"):
        quant.cross_attn.check_small_status()        # (the small-width path validates the batch vectors on the device: read it once, here)
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    if dev is None:
        dev = next(model.parameters()).device
    if not embs:
        if not multi:
            raise ValueError("run_inference: no batches")
        k, e_dim = model.quantize.k, model.quantize.e_dim
        embs, toks = [torch.zeros(0, 4 * e_dim)], [torch.zeros(0, 4, k, dtype=torch.int64)]
        wts, order = [torch.zeros(0, 4, k)], [torch.zeros(0, dtype=torch.int64)]
    emb, tok, wt, order = torch.cat(embs), torch.cat(toks), torch.cat(wts), torch.cat(order)
    if multi:
        emb, tok, wt, order = (mdist.gather_ragged_to_rank0(t, dev) for t in (emb, tok, wt, order))
        if torch.distributed.get_rank() != 0:
            return None
    perm = torch.argsort(order, stable=True)
    if multi and order.numel() > 1:                  # drop repeated codes (a padding sampler's wrap-around): keep the first copy
        sorted_ids = order[perm]
        keep = torch.ones_like(sorted_ids, dtype=torch.bool)
        keep[1:] = sorted_ids[1:] != sorted_ids[:-1]
        perm = perm[keep]
    embeddings = emb[perm].numpy()
    tokens = tok[perm].numpy()
    weights = wt[perm].numpy()
    if out_dir is not None:
        save_outputs(out_dir, embeddings, tokens, weights)
    return embeddings, tokens, weights
