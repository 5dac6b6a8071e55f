#!/usr/bin/env python3
"""Headline benchmark: codes/sec tokenized (text+graph embeddings -> VQ token ids) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2] [--rows R]

Workload (BASELINE.json configs[2], the one the >= 600k codes/s target is quoted on): the full MedTok
soft VQ -- per code 4 nearest-code searches (text / graph over their 16384-row codebook thirds, shared
text / graph over all 49152 rows), top-5, softmax weights, weighted code mix, straight-through value --
on R = 600000 synthetic codes per GPU, D = 768, fp32, cross-attention pooling already applied.
A "step" is one pass of that path over the rank's R rows, inputs resident in HBM.  Rows are
independent, so N GPUs run N row shards with no data-path collective ("weak" scaling: R per GPU).

One JSON line on rank 0 (contract in the task statement) plus:
  roofline      -- the dominant kernel (fp32-MFMA search) timed live with HIP events on its launch
                   stream; achieved = algorithmic flops (2*rows*K*D per launch) / kernel time.
  cpu_baseline  -- the reference's op sequence (oracle/torch_port.py, pinned to golden vectors) timed
                   on this box's host cores on a bounded row sample of the same workload.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from medtok_amd import distributed as mdist  # noqa: E402
from medtok_amd import ops  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, dense f32-input MFMA
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md, dense BF16/FP16 MFMA (the 5 PF headline figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--text-layers", type=int, default=12, help="cfg4: transformer layers of the BERT-shaped stand-in text encoder (12 = BERT-base)")
    ap.add_argument("--workload", choices=["cfg3", "cfg2", "cfg4", "cfg5", "full", "codeshard", "refdefault", "fullref"], default="cfg3",
                    help="cfg3: full soft VQ (headline); full: cfg3 plus the ragged cross-attention of get_shared_info in front of it; cfg2: argmin+EMA train step, 100k rows, K=8192; cfg4: BASELINE config 4 -- one train step (stand-in BERT-shaped text encoder + 2-layer GAT -> soft VQ with aug view -> loss.py -> backward -> clip -> Adam) on 256 codes/GPU under bf16 autocast, L=512; codeshard: one K=49152 soft top-5 search with the CODEBOOK sharded over the GPUs (every rank scores all rows "
                         "against its slice; all-gather of the k-lists + exact merge); cfg5: the same step on 600k rows "
                         "TOTAL (split over the GPUs: strong scaling), K=16384, with the RCCL all-reduce of the EMA statistics")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default 600000 for cfg3, 100000 for cfg2)")
    ap.add_argument("--path", type=int, default=ops.PATH_AUTO)
    ap.add_argument("--cpu-rows", type=int, default=None, help="row sample for the CPU baseline (0 disables)")
    ap.add_argument("--no-collectives", action="store_true", help="N > 1, headline workload: skip the extra cfg5 / codeshard strong-scaling block")
    ap.add_argument("--exact-steps", type=int, default=1, help="extra steps on the exact fp32-MFMA path for comparison (0 disables)")
    ap.add_argument("--one-stream", action="store_true", help="full workload: the whole run on ONE HIP stream (no side streams)")
    ap.add_argument("--precomputed-encoders", action="store_true",
                    help="cfg4: no encoders -- the tokenizer reads pre-computed text / node features (MultimodalTokenizer(None, None)): the step is "
                         "the VQ side alone (text mapping, cross-attention, 6 searches, loss.py, backward, clip, AdamW)")
    ap.add_argument("--no-half-text-pass", action="store_true",
                    help="full workload: skip the extra pass with fp16 text features (what a caller under fp16 autocast hands over)")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not run the shader-clock probe beside the timed region")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run (cfg3, 1 GPU): skip the short runs of the other workloads appended as extra.workloads")
    ap.add_argument("--data", choices=["gaussian", "near_codes", "clustered_codebook", "heavy_tail"], default="gaussian",
                    help="cfg3 / refdefault: the synthetic distribution (BASELINE prescribes i.i.d. Gaussian rows and codes; the others probe the "
                         "shortlist's data-dependent cost: rows near codes, a codebook of near-copies, Student-t rows)")
    ap.add_argument("--no-one-stream-pass", action="store_true",
                    help="full workload: skip the second, one-stream pass that times the kernels for the roofline object (timeline captures)")
    return ap.parse_args()


def cpu_protocol(run, rows_full, unit_rows, what):
    """SURVEY 8d / BASELINE.md section 3 protocol, bounded to ~20 s: `run(rows, threads)` executes the reference op sequence
    on the first `rows` rows of a 16384-row chunk.  Thread count = the best of {16, 64, all host threads} on a short probe
    (all threads oversubscribe this op mix on a 256-thread host); then one warm-up + the median of 3 timed passes over the
    chunk -- cut to what fits ~6 s per pass when the host is slow, and the cut is stated."""
    import statistics
    ncpu = os.cpu_count() or 1
    probe_rows = min(1024, rows_full)
    cands = sorted({min(16, ncpu), min(64, ncpu), ncpu})
    probe = {}
    for th in cands:
        run(min(256, probe_rows), th)                      # page in / spin up the pool
        t0 = time.perf_counter(); run(probe_rows, th); probe[th] = probe_rows / (time.perf_counter() - t0)
    best = max(probe, key=probe.get)
    rows = int(min(rows_full, max(probe_rows, probe[best] * 6.0)))
    run(min(rows, 2048), best)                             # warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter(); run(rows, best); times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    info = " ".join(torch.__config__.parallel_info().split())[:400]
    return dict(value=rows / dt * unit_rows, unit="codes/s", cores=best, kind="port",
                sample=(f"{rows} rows of a {rows_full}-row chunk of the same workload ({what}); reference op sequence in CPU PyTorch "
                        f"(oracle/torch_port.py); warm-up + median of 3 passes ({dt:.2f} s each); threads = best of "
                        + ", ".join(f"{t}: {v:.0f} codes/s" for t, v in probe.items()) + f" on a {probe_rows}-row probe; host has {ncpu} threads"),
                parallel_info=info)


class Cfg3:
    """Full soft VQ, one codebook of n_e = 3 * 16384 rows (text third, middle, graph third)."""
    name = "cfg3"
    D, REGION, TOPK = 768, 16384, 5
    N_E = 3 * REGION

    def __init__(self, rows, dev, seed, path, data="gaussian"):
        from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
        self.rows, self.dev, self.data = rows, dev, data
        D = self.D
        torch.manual_seed(1234)                                                   # same weights on every rank
        self.vq = VectorQuantizer(self.N_E, D, 0.25, 0.0, True, False, [D, D], k=self.TOPK).to(dev).eval()
        self.vq.search_path = path
        g = torch.Generator(device=dev).manual_seed(seed)
        self.h = torch.randn(rows, 2 * D, device=dev, generator=g)              # [CLS text | pooled graph] (tokenizer.py:162-166)
        self.pooled_text = torch.randn(rows, D, device=dev, generator=g)        # cross-attention outputs (bypassed; stated)
        self.pooled_graph = torch.randn(rows, D, device=dev, generator=g)
        self.description = (f"{self.name} full MedTok soft-VQ: {rows} codes/GPU x 4 searches (2 x K={self.REGION} regions + 2 x K={self.N_E} shared), "
                            f"D={D}, k={self.TOPK}, eval, fp32; proj Linear + normalise + search + softmax/mix/STE per code")
        self.data_note = None
        if data != "gaussian":
            self._shape_data(data, g)

    def _shape_data(self, data, g):
        """Distributions off the i.i.d. Gaussian that BASELINE prescribes: the shortlist's cost depends on the data (how many codes a
        row's window admits), a trained codebook is not i.i.d..  The modality-specific searches see their rows
        through proj_text / proj_graph (a random Linear here), so the shaping applies to what reaches a search directly: the pooled
        rows of the two shared searches, and the codebook itself."""
        rows, D, dev = self.rows, self.D, self.dev
        W = self.vq.codebook.weight
        if data == "near_codes":
            # x = w_hat[c] + sigma n, |sigma n| ~ 0.3: every row sits next to one code (what a trained quantiser sees)
            what = torch.nn.functional.normalize(W.detach(), dim=-1)
            for name in ("pooled_text", "pooled_graph"):
                c = torch.randint(0, self.N_E, (rows,), device=dev, generator=g)
                noise = torch.randn(rows, D, device=dev, generator=g) * (0.3 / D ** 0.5)
                setattr(self, name, what[c] + noise)
            self.data_note = "shared searches: rows = normalised code + noise of norm ~0.3; codebook and specific searches as gaussian"
        elif data == "clustered_codebook":
            # 512 clusters x (n_e / 512) near-copies (relative spread 1e-2): runs of near-identical codes fill a row's candidate lists
            per = self.N_E // 512
            centers = torch.randn(512, D, device=dev, generator=g)
            Wc = centers.repeat_interleave(per, dim=0)
            Wc = torch.cat([Wc, torch.randn(self.N_E - Wc.shape[0], D, device=dev, generator=g)]) if Wc.shape[0] < self.N_E else Wc
            Wc = Wc + 1e-2 * torch.randn(self.N_E, D, device=dev, generator=g)
            perm = torch.randperm(self.N_E, device=dev, generator=g)          # (cluster members spread over the code range)
            with torch.no_grad():
                W.copy_(Wc[perm])
            self.vq.invalidate_codebook_cache()
            for name in ("pooled_text", "pooled_graph"):                        # rows near cluster centres: the near-copies all compete
                c = torch.randint(0, 512, (rows,), device=dev, generator=g)
                setattr(self, name, centers[c] + 0.05 * torch.randn(rows, D, device=dev, generator=g))
            self.data_note = (f"codebook = 512 clusters x {per} near-copies (spread 1e-2 per element), shuffled over the code range; shared-search rows = "
                              "cluster centre + 0.05 noise per element")
        elif data == "heavy_tail":
            # Student-t (2 degrees of freedom) rows: a few huge coordinates dominate a row's direction
            def student(n, d):
                z = torch.randn(n, d, device=dev, generator=g)
                chi = torch.randn(n, d, 2, device=dev, generator=g).pow(2).sum(-1) / 2.0
                return z / chi.sqrt()
            self.pooled_text, self.pooled_graph = student(rows, D), student(rows, D)
            self.h = student(rows, 2 * D)
            self.data_note = "all rows Student-t with 2 degrees of freedom (element-wise); codebook gaussian"
        else:
            raise ValueError(data)
        self.description += f"; data = {data}: {self.data_note}"

    def flops_per_code(self):
        return 2.0 * self.D * (2 * self.REGION + 2 * self.N_E)

    def bytes_per_code(self):
        # 4 searches x (x row in + zq row out) + ids (8 B) and weights (4 B) for 4 x k tokens  ~ 24.8 KB
        return 4 * (4 * self.D + 4 * self.D) + 4 * self.TOPK * 12

    def set_path(self, path):
        self.vq.search_path = path

    def paths_agree(self):
        self.set_path(ops.PATH_AUTO)
        a = self.step()
        self.set_path(ops.PATH_F32_MFMA)
        b = self.step()
        self.set_path(ops.PATH_AUTO)
        return all(torch.equal(u, v) for u, v in zip(a, b))

    def step(self):
        from medtok_amd.inference import quantize_pooled
        self.vq._norm_cache = None      # re-normalise the codebook every call, like the reference (:148,198,200)
        return quantize_pooled(self.vq, self.h, self.pooled_text, self.pooled_graph)

    def cpu_baseline(self, sample_rows):
        """Reference op sequence on the host cores, `sample_rows` rows of this workload."""
        from oracle import torch_port as P
        g = torch.Generator().manual_seed(0)
        D = self.D
        W = torch.randn(self.N_E, D, generator=g)
        xs = [torch.randn(sample_rows, D, generator=g) for _ in range(4)]

        def run(rows, threads):
            torch.set_num_threads(threads)
            P.full_tokenize(*[x[:rows] for x in xs], W, self.TOPK)
        return cpu_protocol(run, sample_rows, 1, "4 searches per code, K=16384/49152, D=768")


class RefDefault(Cfg3):
    """The four searches at the shape MedTok ships (train_MedTok.py:363-368): e_dim = 64, n_e = 21 000 (regions of 7 000
    codes), k = 5 -- cross-attention pooling already applied, as in cfg3."""
    name = "refdefault"
    D, REGION, TOPK = 64, 7000, 5
    N_E = 21000

    def __init__(self, rows, dev, seed, path, data="gaussian"):
        super().__init__(rows, dev, seed, path, data)
        self.description = (f"refdefault: the four searches of the soft VQ at the reference's default shape: {rows} codes/GPU, e_dim = 64, "
                            f"n_e = 21000 (2 x K=7000 regions + 2 x K=21000 shared), k=5, eval, fp32; proj Linear + normalise + search + softmax/mix/STE per code"
                            + (f"; data = {data}: {self.data_note}" if self.data_note else ""))

    def cpu_baseline(self, sample_rows):
        from oracle import torch_port as P
        g = torch.Generator().manual_seed(0)
        W = torch.randn(self.N_E, self.D, generator=g)
        xs = [torch.randn(sample_rows, self.D, generator=g) for _ in range(4)]

        def run(rows, threads):
            torch.set_num_threads(threads)
            P.full_tokenize(*[x[:rows] for x in xs], W, self.TOPK)
        return cpu_protocol(run, sample_rows, 1, "4 searches per code, K=7000/21000, D=64")


class Full(Cfg3):
    """VectorQuantizer.forward end to end (vector_quantization_soft_one_new.py:238-271): ragged cross-attention over the text tokens
    and graph nodes of every code -> pooled rows -> the four searches of cfg3.  Inputs are the encoders' outputs, resident in HBM."""
    name = "full"
    L, MAX_NODES = 512, 40

    def __init__(self, rows, dev, seed, path):
        super().__init__(rows, dev, seed, path)
        D = self.D
        g = torch.Generator(device=dev).manual_seed(seed + 77)
        self.text = torch.randn(rows, self.L, D, device=dev, generator=g)
        self.tok = torch.randint(1, self.L + 1, (rows,), device=dev, generator=g)                       # valid tokens per code
        self.mask = (torch.arange(self.L, device=dev)[None, :] < self.tok[:, None]).to(torch.int64)
        self.n_nodes = torch.randint(1, self.MAX_NODES + 1, (rows,), device=dev, generator=g)
        self.batch = torch.repeat_interleave(torch.arange(rows, device=dev), self.n_nodes)
        self.nodes = torch.randn(int(self.n_nodes.sum()), D, device=dev, generator=g)
        heads = self.vq.cross_attn.model[0].multihead_attn.num_heads
        layers = len(self.vq.cross_attn.model)
        # algorithmic flops of the attention core per step: 2 products x 2 flops x D per (query row, key) pair
        pairs = float((self.n_nodes * heads * self.tok).sum()) + float((heads * self.n_nodes).sum())
        self.attention_flops = 4.0 * D * pairs * layers
        self.attention_f16x3 = True          # inference: medtok_shared_kv_attention_f32(exact_f32 = 0)
        # algorithmic bytes of the attention core per step: per layer the graph side reads a code's valid
        # token rows ONCE as (hi, lo) fp16 images (4 B per element; every query tile of the code shares them), reads its folded
        # fp32 query rows (nodes x heads) and writes their context as (hi, lo) images; the text side reads the code's node rows
        # (fp32) and reads / writes its `heads` CLS query rows
        q_rows = float((self.n_nodes * heads).sum())
        self.attention_bytes = layers * 4.0 * D * (float(self.tok.sum()) + 2.0 * q_rows + float(self.n_nodes.sum()) + 2.0 * heads * rows)
        self.description = (f"full VectorQuantizer.forward: {rows} codes/GPU/step, ragged cross-attention (<= {self.L} tokens x <= "
                            f"{self.MAX_NODES} nodes per code, 4 heads, 2 layers per direction, D=768) + the 4 searches of cfg3 "
                            f"(n_e = 49152, k=5), eval, fp32")

    def step(self):
        self.vq._norm_cache = None
        with torch.no_grad():
            return self.vq(self.h, self.text, self.nodes, self.mask, self.batch)

    def paths_agree(self):
        outs = []
        for path in (ops.PATH_AUTO, ops.PATH_F32_MFMA):
            self.set_path(path)
            r = self.step()
            outs.append([r[k] for k in sorted(r) if isinstance(r[k], torch.Tensor)])
        self.set_path(ops.PATH_AUTO)
        return all(torch.equal(u, v) for u, v in zip(*outs))

    def no_host_read(self, steps):
        """The same forward with CrossAttention.max_nodes_bound set to the generator's largest subgraph: the attention launches are
        sized from the bound, nothing is read back (the batch checks stay on the device), so the forward also records into a HIP
        graph.  Eager and replayed, on ONE stream (side streams off for this pass), outputs compared with the default forward."""
        import medtok_amd.vector_quantization_soft_one_new as vqmod
        ca = self.vq.cross_attn
        keep_streams, keep_bound = vqmod.SIDE_STREAM_MIN_CODES, ca.max_nodes_bound
        ref = self.step()
        out = {}
        try:
            vqmod.SIDE_STREAM_MIN_CODES = 0
            ca.max_nodes_bound = self.MAX_NODES
            r = self.step()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize(self.dev)
            dt = time.perf_counter() - t0
            ca.check_status()
            same = all(torch.equal(r[k], ref[k]) for k in r if isinstance(r[k], torch.Tensor))
            out["eager_one_stream"] = {"value": self.rows * steps / dt, "unit": "codes/s", "ms_per_step": dt / steps * 1e3, "equals_default_forward": bool(same)}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g_out = self.step()
            for _ in range(2):
                graph.replay()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize(self.dev)
            dt = time.perf_counter() - t0
            ca.check_status()
            same = all(torch.equal(g_out[k], ref[k]) for k in g_out if isinstance(g_out[k], torch.Tensor))
            out["hip_graph_replay"] = {"value": self.rows * steps / dt, "unit": "codes/s", "ms_per_step": dt / steps * 1e3, "replay_equals_default_forward": bool(same)}
        finally:
            vqmod.SIDE_STREAM_MIN_CODES, ca.max_nodes_bound = keep_streams, keep_bound
        out["note"] = (f"cross_attn.max_nodes_bound = {self.MAX_NODES} (not in the reference: an upper bound on the nodes of one code, here the generator's); "
                       "show_usage = False; rank-local; `value` above is the default forward (one host read per call, any batch vector)")
        return out

    def cpu_baseline(self, sample_rows):
        """The reference's own form on the host cores: per-code Python loop over nn.MultiheadAttention layers (:133-142), then the
        dense searches."""
        from oracle import torch_port as P
        # the loop issues thousands of small ops: more than a few threads only adds fork/join cost
        threads = min(16, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        n = min(sample_rows, self.rows)
        ca = copy.deepcopy(self.vq.cross_attn).cpu().eval()
        text, tok, nn_, nodes = self.text[:n].cpu(), self.tok[:n].cpu(), self.n_nodes[:n].cpu(), self.nodes[: int(self.n_nodes[:n].sum())].cpu()
        W = self.vq.codebook.weight.detach().cpu()
        h = self.h[:n].cpu()
        t0 = time.perf_counter()
        with torch.no_grad():
            pt, pg, off = [], [], 0
            for i in range(n):
                a, b = ca(text[i, : int(tok[i])], nodes[off: off + int(nn_[i])])
                off += int(nn_[i])
                pt.append(a[0]); pg.append(b.mean(0))
                if time.perf_counter() - t0 > 20.0:          # bounded sample: stop after ~20 s of loop
                    break
            n = len(pt)
            P.full_tokenize(h[:n, : self.D], h[:n, self.D:], torch.stack(pt), torch.stack(pg), W, self.TOPK)
        dt = time.perf_counter() - t0
        return dict(value=n / dt, unit="codes/s", cores=threads, kind="port",
                    sample=f"{n} codes: per-code cross-attention loop (the reference's form) + 4 dense searches in CPU PyTorch, {dt:.1f} s")


class FullRefDefault(Full):
    """VectorQuantizer.forward end to end at the reference's default shape and per-GPU batch (train_MedTok.py:363-368,387: e_dim = 64,
    n_e = 21 000, B = 256, <= 512 tokens): ragged cross-attention + the four searches."""
    name = "fullref"
    D, REGION, TOPK = 64, 7000, 5
    N_E = 21000

    def __init__(self, rows, dev, seed, path):
        super().__init__(rows, dev, seed, path)
        from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
        # the reference builds its quantiser with show_usage = True (tokenizer.py:72,126): every forward slides the 300 000-entry usage
        # window three times and returns three Python floats (one host synchronisation per forward: part of the reference's interface)
        torch.manual_seed(1234)
        self.vq = VectorQuantizer(self.N_E, self.D, 0.25, 0.0, True, True, [self.D, self.D], k=self.TOPK).to(dev).eval()
        self.vq.search_path = path
        self.description = (f"full VectorQuantizer.forward at the reference's default shape: {rows} codes/GPU/step (256 = its per-GPU batch), "
                            f"ragged cross-attention (<= {self.L} tokens x <= {self.MAX_NODES} nodes per code, 4 heads, 2 layers per direction) + 4 "
                            f"searches, e_dim = 64, n_e = 21000, k=5, show_usage = True (the reference's default: three usage-window "
                            f"updates and one host read per forward), eval, fp32")

    def graph_replay(self, steps):
        """the same forward without the usage window (no host read), captured into a HIP graph and replayed: (codes/s, ms per forward)"""
        from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
        torch.manual_seed(1234)
        vq = VectorQuantizer(self.N_E, self.D, 0.25, 0.0, True, False, [self.D, self.D], k=self.TOPK).to(self.dev).eval()
        args = (self.h, self.text, self.nodes, self.mask, self.batch)
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    vq(*args)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = vq(*args)
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize(self.dev)
            dt = time.perf_counter() - t0
            eager = vq(*args)
            same = all(torch.equal(out[k], eager[k]) for k in out if isinstance(out[k], torch.Tensor))
            vq.cross_attn.check_small_status()
        return {"value": self.rows * steps / dt, "unit": "codes/s", "ms_per_step": dt / steps * 1e3, "replay_equals_eager": bool(same),
                "note": "VectorQuantizer.forward (eval, show_usage = False: no host read) captured into a HIP graph and replayed; rank-local"}


class CodeShard:
    """SURVEY 8e variant: the codebook is sharded over the ranks, every rank holds all rows.  A step = local top-k over the
    rank's slice (HIP search) -> all-gather of (distance, global id) lists over RCCL -> exact merge -> soft assignment."""
    name = "codeshard"
    D, K, TOPK = 768, 49152, 5

    def __init__(self, rows, dev, seed, path, rank, world):
        self.rows, self.dev, self.path = rows, dev, path
        g = torch.Generator(device=dev).manual_seed(4321)                       # the SAME rows and codebook on every rank
        self.x = torch.randn(rows, self.D, device=dev, generator=g)
        W = torch.randn(self.K, self.D, device=dev, generator=g)
        self.what, self.wsq = ops.rownorm(W)
        self.lo, self.hi = mdist.code_shard(self.K, rank, world)
        self.description = (f"one soft top-5 search, {rows} rows (replicated) x K=49152 codes sharded x{world} "
                            f"({self.hi - self.lo} codes/GPU), D=768; all-gather of the k-lists + exact merge")

    def flops_per_code(self):
        return 2.0 * self.D * self.K

    def bytes_per_code(self):
        return 8 * self.D + self.TOPK * 12 + self.K * self.D * 4 / max(self.rows, 1)

    def set_path(self, path):
        self.path = path

    def step(self):
        xhat, xsq = ops.rownorm(self.x)
        idx, dist_ = mdist.code_sharded_search(xhat, xsq, self.what[self.lo:self.hi], self.wsq[self.lo:self.hi].contiguous(), self.lo,
                                               self.TOPK, search_fn=lambda a, b, c, d, k: ops.topk_search(a, b, c, d, k, self.path))
        w, zq, _ = ops.soft_assign(self.x, self.what, idx, dist_, want_sqerr=False)
        return idx, dist_, w, zq

    def paths_agree(self):
        outs = []
        for path in (ops.PATH_AUTO, ops.PATH_F32_MFMA):
            self.set_path(path)
            outs.append(self.step())
        self.set_path(ops.PATH_AUTO)
        return all(torch.equal(u, v) for u, v in zip(*outs))

    def cpu_baseline(self, sample_rows):
        from oracle import torch_port as P
        g = torch.Generator().manual_seed(0)
        W = torch.randn(self.K, self.D, generator=g); x = torch.randn(sample_rows, self.D, generator=g)

        def run(rows, threads):
            torch.set_num_threads(threads)
            P.soft_search(x[:rows], W, self.TOPK)
        return cpu_protocol(run, sample_rows, 1, "one dense K=49152 search + soft assignment")


class Cfg2:
    """Single-modality argmin + EMA train step (NormEMAVectorQuantizer), K = 8192."""
    name = "cfg2"
    D, K = 768, 8192

    def __init__(self, rows, dev, seed, path, k_codes=None):
        if k_codes:
            self.K = k_codes
        from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
        self.rows, self.dev = rows, dev
        g = torch.Generator(device=dev).manual_seed(seed)
        self.z = torch.randn(rows, self.D, 1, 1, device=dev, generator=g)
        torch.manual_seed(1234)
        self.q = NormEMAVectorQuantizer(self.K, self.D, 0.25).to(dev).train()   # picks RCCL all-reduce when WORLD_SIZE > 1
        self.q.search_path = path
        self.description = f"{self.name} NormEMA argmin + EMA codebook update (train): {rows} rows/GPU, D=768, K={self.K}, fp32"

    def flops_per_code(self):
        return 2.0 * self.K * self.D

    def bytes_per_code(self):
        # SURVEY 8d cfg 2: z in + z_q out + id, plus the codebook-sized state (3 K D + 2 K floats) amortised over the rows
        return 8 * self.D + 8 + (3 * self.K * self.D * 4 + 2 * self.K * 4) / max(self.rows, 1)

    def set_path(self, path):
        self.q.search_path = path

    def paths_agree(self):
        keep = (self.q.embedding.weight.data.clone(), self.q.cluster_size.clone())
        outs = []
        for path in (ops.PATH_AUTO, ops.PATH_F32_MFMA):
            self.q.embedding.weight.data.copy_(keep[0]); self.q.cluster_size.copy_(keep[1])
            self.set_path(path)
            zq, loss, idx = self.step()
            outs.append((zq.clone(), loss.clone(), idx.clone(), self.q.embedding.weight.data.clone(), self.q.cluster_size.clone()))
        self.set_path(ops.PATH_AUTO)
        return all(torch.equal(u, v) for u, v in zip(*outs))

    def step(self):
        with torch.no_grad():
            return self.q(self.z)

    def cpu_baseline(self, sample_rows):
        from oracle import torch_port as P
        g = torch.Generator().manual_seed(0)
        E = torch.nn.functional.normalize(torch.randn(self.K, self.D, generator=g), dim=-1)
        z = torch.randn(sample_rows, self.D, generator=g)

        def run(rows, threads):
            torch.set_num_threads(threads)
            P.norm_ema_forward(z[:rows], E.clone(), torch.zeros(self.K), 0.25, 0.99, True)
        return cpu_protocol(run, sample_rows, 1, f"argmin + EMA train step, K={self.K}, D=768")


class Cfg4:
    """BASELINE config 4: the train step of train_MedTok.py:207-250 -- zero_grad, bf16 autocast forward of the tokenizer
    (text + graph encoders, the aug view, VectorQuantizer.forward: cross-attention + 6 searches), the loss assembly of loss.py,
    backward, gradient clipping, optimizer step -- on B = 256 synthetic PrimeKG-shaped codes per GPU, 512 text tokens.
    The encoders are plain-torch stand-ins of the reference's shapes (BERT-base-shaped, frozen as in tokenizer.py:80-81; 2-layer
    GAT): they are upstream of the path this package rebuilds, so their time is reported separately (`encoders_ms_per_step`)."""
    name = "cfg4"
    D, N_E, L, TOPK = 768, 49152, 512, 5

    def __init__(self, rows, dev, seed, path, text_layers=12, precomputed=False):
        from medtok_amd.synthetic import StandInGAT, StandInTextEncoder, primekg_shaped_batch
        from medtok_amd.tokenizer import MultimodalTokenizer
        self.rows, self.dev, self.precomputed = rows, dev, precomputed
        torch.manual_seed(1234)
        self.inputs = primekg_shaped_batch(rows, dev, seed=seed, max_len=self.L)
        if precomputed:
            import medtok_amd.vector_quantization_soft_one_new as vqmod
            vqmod.TRAIN_SPLIT_TEXT_MAPPING = True           # every dense product of the step on the library's own kernels
            # the VQ side alone: the encoders' outputs are inputs (the node features carry requires_grad, as they would coming out of
            # a trainable graph encoder, so that the backward reaches them through the cross-attention's dKV kernel)
            self.model = MultimodalTokenizer(None, None, text_dim=768, graph_out_channels=self.D, codebook_size=self.N_E,
                                             codebook_embed_dim=self.D).to(dev).train()
            g = torch.Generator(device=dev).manual_seed(seed + 5)
            n_nodes = int(self.inputs.batch.numel())
            self.inputs.text_features = torch.randn(rows, self.L, 768, device=dev, generator=g)
            self.inputs.text_features_aug = self.inputs.text_features + 0.01 * torch.randn(rows, self.L, 768, device=dev, generator=g)
            self.inputs.graph_node_features = torch.randn(n_nodes, self.D, device=dev, generator=g).requires_grad_()
            self.inputs.graph_node_features_aug = torch.randn(n_nodes, self.D, device=dev, generator=g).requires_grad_()
        else:
            self.model = MultimodalTokenizer(StandInTextEncoder(layers=text_layers), StandInGAT(dim=self.D), text_dim=768, graph_out_channels=self.D,
                                             codebook_size=self.N_E, codebook_embed_dim=self.D).to(dev).train()
            for p in self.model.text_model.parameters():
                p.requires_grad = False                                     # tokenizer.py:80-81
        self.model.quantize.search_path = path
        self.opt = torch.optim.AdamW([p for p in self.model.parameters() if p.requires_grad], lr=1e-4)
        # (query row, key) pairs of the cross-attention per layer: every node x head against its code's valid tokens, and the
        # CLS row x head against the code's nodes; forward 2 products (4 D flop per pair), backward 10 D per pair and kernel
        heads, layers = 4, 2
        tok = self.inputs.attention_mask.sum(1).to(torch.float64)
        n_nodes = torch.bincount(self.inputs.batch, minlength=rows).to(torch.float64)
        pairs = float((n_nodes * heads * tok).sum()) + float((heads * n_nodes).sum())
        self.attention_flops = 4.0 * self.D * pairs * layers
        self._attention_bwd_flops = 20.0 * self.D * pairs * layers
        self.enc_ms, self.enc_events = 0.0, []
        self.autocast_one_pass = True                      # (step() runs under bf16 autocast: one-pass dense products and attention backward)
        self.description = (f"cfg4 train step: {rows} codes/GPU, {self.L} text tokens, PrimeKG-shaped subgraphs (median ~20 nodes), stand-in "
                            f"BERT-shaped text encoder ({text_layers} layers, frozen) + 2-layer GAT -> soft VQ (n_e=49152, D=768, k=5, aug view, "
                            f"cross-attention) -> loss.py -> backward -> clip -> AdamW; bf16 autocast")

    def flops_per_code(self):
        # the six searches of a train-mode forward (2 shared over n_e, 2 + 2 aug over a third)
        return 2.0 * self.D * (2 * self.N_E + 4 * (self.N_E // 3))

    def bytes_per_code(self):
        return 6 * (8 * self.D) + 6 * self.TOPK * 12 + self.L * self.D * 4

    def set_path(self, path):
        self.model.quantize.search_path = path

    def paths_agree(self):
        return True          # (a train step mutates the weights: the two paths are compared at fixed weights by the tests)

    def step(self):
        from medtok_amd import loss as L
        m = self.model
        self.opt.zero_grad(set_to_none=True)
        if self.precomputed:
            self.inputs.graph_node_features.grad = self.inputs.graph_node_features_aug.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            r = m(self.inputs)
            loss, _ = L.total_loss(r, 0.1, 0.1)
        loss.float().backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        self.opt.step()
        return (loss.detach(),)

    def attention_backward_flops(self):
        return self._attention_bwd_flops

    def encoder_ms(self, steps=3):
        """time of the stand-in encoders alone (both views, as forward() runs them), outside the timed region"""
        if self.precomputed:
            return 0.0
        m, x = self.model, self.inputs
        def enc():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                for aug in (False, True):            # (the text mapping is the VQ side's, as in --precomputed-encoders)
                    m.tokenize_text(x, aug=aug); m.tokenize_graph(x, aug=aug)
        enc(); torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            enc()
        torch.cuda.synchronize(self.dev)
        return (time.perf_counter() - t0) / steps * 1e3

    def cpu_baseline(self, sample_rows):
        """the VQ + loss part of the step in CPU PyTorch on the reference's op sequence (dense N x K distances, autograd through
        them), at the same B: the encoders are identical torch modules on either side and are left out"""
        from oracle import torch_port as P
        g = torch.Generator().manual_seed(0)
        B, D = min(sample_rows, self.rows), self.D
        W = torch.randn(self.N_E, D, generator=g, requires_grad=True)
        xs = [torch.randn(B, D, generator=g, requires_grad=True) for _ in range(6)]
        region = self.N_E // 3

        def run(rows, threads):
            torch.set_num_threads(threads)
            tot = 0.0
            for i, x in enumerate(xs):
                Wr = W if i < 2 else (W[:region] if i % 2 == 0 else W[-region:])
                xn = torch.nn.functional.normalize(x[:rows], dim=-1); wn = torch.nn.functional.normalize(Wr, dim=-1)
                d = P.distance_matrix(xn, wn)
                vals, idx = torch.topk(d, self.TOPK, largest=False)
                zq = (torch.softmax(-vals, 1).unsqueeze(-1) * wn[idx]).sum(1)
                tot = tot + ((zq - x[:rows].detach()) ** 2).mean() + 0.25 * ((zq.detach() - x[:rows]) ** 2).mean()
            tot.backward()
        return cpu_protocol(run, B, 1, "6 dense searches of a train-mode forward + their autograd backward, n_e=49152, D=768; encoders excluded")


def pmc_traffic(workload, kernel, rows):
    """(bytes, source): fabric bytes per launch of `kernel`.  PMC counters cannot be read from inside this process (they need their
    own rocprofv3 --pmc passes, tools/pmc_traffic.sh), so the figure is the one RECORDED in profiles/pmc_traffic.json -- returned
    only when this run has the workload and row count that file was taken at, and always with its provenance; otherwise null."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(f.read_text())
        meta = rec.get("_meta", {})
        if meta.get("rows", {}).get(workload) != rows:
            return None, None
        val = rec.get(workload, {}).get(kernel)
        if val is None:
            return None, None
        # the recorded figure is per DISPATCH; `launches_timed` / `avg_launch_ms` of the line count what the library brackets with one
        # event pair -- for the shortlist kernel a whole search (main + tail dispatch at 600 000 rows): same unit here
        per = meta.get("dispatches_per_timed_launch", {}).get(workload, {}).get(kernel, 1)
        return val * per, (f"recorded, not measured in this run: profiles/pmc_traffic.json ({meta.get('taken', '?')}; rocprofv3 --pmc FETCH_SIZE x2 + "
                           f"WRITE_SIZE per MI355X_MICROARCH.md; {workload} at {rows} rows/GPU); per timed launch = {per} dispatch(es) of {val / 1e9:.2f} GB")
    except Exception:
        return None, None


def step_fabric_bytes(workload, rows):
    """recorded fabric bytes of ALL kernels of one step (profiles/pmc_traffic.json `_meta.step_fabric_bytes`), or None"""
    try:
        meta = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get("_meta", {})
        if meta.get("rows", {}).get(workload) != rows:
            return None
        return meta.get("step_fabric_bytes", {}).get(workload)
    except Exception:
        return None


def issue_roofs(workload, kernel, rows, avg_launch_ms, clock_ghz):
    """The D <= 64 filter is not priced by the f16 MFMA roof alone: its scan is VALU work of the same order as its matrix work.  From
    the instruction counts RECORDED in profiles/pipe_counters.json (own rocprofv3 --pmc passes) and THIS run's launch time and clock:
    the time the kernel's VALU instructions need at one per 4 cycles and SIMD, the time its MFMAs need at 32 cycles each, and the
    fraction of each roof the launch reaches (1.0 = that pipe never idles)."""
    try:
        rec = json.loads((ROOT / "profiles" / "pipe_counters.json").read_text())
        if rec.get("_meta", {}).get("rows", {}).get(workload) != rows or not clock_ghz:
            return None
        c = rec[workload][kernel]
        simd_hz = 1024 * clock_ghz * 1e9
        valu_ms = c["valu_instructions"] * 4 / simd_hz * 1e3
        mfma_ms = c["mfma_instructions"] * 32 / simd_hz * 1e3
        simd_cycles = c["grbm_gui_active"] / 8 * 1024
        scores = c["mfma_instructions"] * 1024 / 4
        return {"kernel": c["kernel"], "valu_issue_roof_ms": valu_ms, "frac_of_valu_issue_roof": valu_ms / avg_launch_ms,
                "mfma_roof_ms_at_this_clock": mfma_ms, "frac_of_mfma_roof_at_this_clock": mfma_ms / avg_launch_ms,
                "valu_lane_ops_per_score": c["valu_instructions"] * 64 / scores,
                "recorded_busy": {"valu": c["valu_instructions"] * 4 / simd_cycles, "matrix_pipe": c["valu_mfma_busy_cycles"] / simd_cycles,
                                  "both_at_once": c["valu_mfma_coexec_cycles"] / simd_cycles,
                                  "waves_per_simd": c["wave_cycles_x4"] * 4 / simd_cycles,
                                  "wave_cycles_waiting_on_s_waitcnt": c["wait_inst_any_x4"] / c["wave_cycles_x4"]},
                "clock_ghz": clock_ghz,
                "source": "instruction counts recorded, not measured in this run: profiles/pipe_counters.json (" + rec["_meta"]["taken"] + "); launch time and clock from this run",
                "reading": "neither pipe binds alone: the matrix pipe and the VALU are each busy about half of the SIMD cycles and overlap in a sixth -- the kernel is bound by how "
                           "well three to four waves per SIMD interleave their MFMA phase with each other's scan (DESIGN 6.0)"}
    except Exception:
        return None


def kernel_roofline(wl, prof, steps):
    """(kname, kp, view, f16x3, hbm_bound, prof): the roofline view of the dominant kernel = the library kernel with the most TIME in
    `prof` (HIP events recorded by the library around its matrix-pipe launches over `steps` steps); its flops come from the library
    where the launch knows them, from the workload where the ragged counts live on the device, and are null (no roofline fraction)
    where neither does."""
    prof = {k: dict(v) for k, v in prof.items()}
    if hasattr(wl, "attention_flops"):
        prof["shared_kv_attention_kernel"]["flops"] = wl.attention_flops * steps
    if hasattr(wl, "attention_backward_flops"):
        prof["shared_kv_attention_backward_kernels"]["flops"] = wl.attention_backward_flops() * steps
    kname = max(prof, key=lambda k: prof[k]["ms"])
    kp = prof[kname]
    # the split kernels run a product as three fp16 MFMA passes: their algorithmic (fp32-equivalent) flops are priced against a
    # third of the dense fp16 peak
    # (a workload under torch.autocast -- cfg4 -- runs its dense products and its attention backward as ONE half-precision pass:
    # priced against the dense f16 / bf16 peak itself)
    one_pass = getattr(wl, "autocast_one_pass", False) and kname in ("split_gemm_kernel", "shared_kv_attention_backward_kernels")
    f16x3 = not one_pass and (kname == "split_gemm_kernel" or (kname == "shared_kv_attention_kernel" and getattr(wl, "attention_f16x3", False)))
    peak = (F16_MFMA_PEAK_TFLOPS if (kname == "filter_f16_kernel" or one_pass) else
            (F16_MFMA_PEAK_TFLOPS / 3.0 if f16x3 else FP32_MFMA_PEAK_TFLOPS))
    achieved = (kp["flops"] / (kp["ms"] * 1e-3) / 1e12) if (kp["ms"] > 0 and kp["flops"] > 0) else None
    # which roof binds THIS kernel: its arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge of the pipe it
    # runs on (peak flop rate / 8 TB/s).  The searches sit at K/4 flop/B, far right of every ridge; the attention core of the
    # `full` workload (raw key rows shared by all heads, D = 768) sits at ~50 flop/B, LEFT of the split-fp16 ridge (~104): HBM-bound.
    kbytes = wl.attention_bytes * steps if (kname == "shared_kv_attention_kernel" and hasattr(wl, "attention_bytes")) else None
    intensity = (kp["flops"] / kbytes) if (kbytes and kp["flops"] > 0) else None
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    hbm_bound = intensity is not None and intensity < ridge
    mfma_view = {"achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": (achieved / peak) if achieved is not None else None}
    if hbm_bound:
        k_gbs = kbytes / (kp["ms"] * 1e-3) / 1e9
        view = {"bound": "hbm", "achieved": k_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": kbytes / max(kp["launches"], 1), "mfma_view": mfma_view}
    else:
        view = dict(mfma_view, bound="mfma")
    view["arithmetic_intensity_flop_per_byte"] = intensity
    view["ridge_flop_per_byte"] = ridge
    view["one_half_precision_pass"] = bool(one_pass)
    return kname, kp, view, f16x3, hbm_bound, prof


def extra_workloads(args, dev):
    """The default run (`bench.py`, cfg3 on one GPU) also times a few steps of the other workloads this build makes claims about, so
    that those claims are measured by whoever runs the default command and not only by the builder: {name: {value, unit, ms_per_step,
    dominant_kernel, bound, frac, clock_ghz, ...}}.  Each is a shortened `bench.py --workload <name>` (same classes, same timing
    protocol: warm-up, synchronise, K steps, synchronise); kernel durations of the multi-stream forwards come from a one-stream pass."""
    import medtok_amd.vector_quantization_soft_one_new as vqmod
    out = {}

    def timed(step, steps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    def run(name, make, steps, warmup, streams=False, clock=True, rows_note=None, more=None):
        try:
            wl = make()
            for _ in range(warmup):
                wl.step()
            import gc
            gc.collect()
            dt = timed(wl.step, steps)
            keep = vqmod.SIDE_STREAM_MIN_CODES
            try:
                if streams:                      # kernel durations on ONE stream (event pairs of overlapping streams include each other)
                    vqmod.SIDE_STREAM_MIN_CODES = 0
                    wl.step()
                torch.cuda.synchronize(dev)
                ops.profile_begin()
                dt1 = timed(wl.step, steps)
                prof = ops.profile_end()
                ghz = None
                if clock and dt1 / steps >= 0.004:
                    csteps = max(2, min(steps, int(0.3 / (dt1 / steps))))
                    with ops.ClockProbe(dev, max_seconds=3.0 * dt1 / steps * csteps + 5.0) as probe:
                        for _ in range(csteps):
                            wl.step()
                        torch.cuda.current_stream(dev).synchronize()
                    ghz = probe.result().get("ghz_mean")
                    torch.cuda.synchronize(dev)
            finally:
                vqmod.SIDE_STREAM_MIN_CODES = keep
            kname, kp, view, f16x3, hbm_bound, _ = kernel_roofline(wl, prof, steps)
            rec = {"value": wl.rows * steps / dt, "unit": "codes/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                   "dominant_kernel": ("filter_rows64n_kernel" if kname == "filter_f16_kernel" and wl.D <= 64 else kname) if kp["launches"] else None,
                   "bound": view["bound"], "frac": view["frac"], "achieved": view["achieved"], "peak": view["peak"], "roof_unit": view["unit"],
                   "avg_launch_ms": kp["ms"] / max(kp["launches"], 1), "launches_timed": kp["launches"],
                   "clock_ghz": ghz, "workload": wl.description}
            if streams:
                rec["one_stream_ms_per_step"] = dt1 / steps * 1e3
            if more is not None:
                rec.update(more(wl))
            out[name] = rec
            del wl
        except Exception as exc:                 # (extras: the headline line is printed regardless)
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        torch.cuda.empty_cache()

    run("refdefault", lambda: RefDefault(600000, dev, 0, args.path), 3, 1)
    run("cfg2", lambda: Cfg2(100000, dev, 0, args.path), 5, 2)
    run("full", lambda: Full(4096, dev, 0, args.path), 5, 2, streams=True)

    def bound_replay(wl):
        # the reference's per-GPU batch at BASELINE's width is launch-bound in eager mode (~60 launches of under-filled grids); with an
        # upper bound on the nodes of one code the forward makes no host read and replays from a HIP graph (CrossAttention.max_nodes_bound)
        try:
            r = wl.no_host_read(10)
            return {"no_host_read": {k: {kk: v[kk] for kk in v if kk in ("value", "unit", "ms_per_step", "equals_default_forward", "replay_equals_default_forward")}
                                     for k, v in r.items() if isinstance(v, dict)}}
        except Exception as exc:
            return {"no_host_read": {"error": f"{type(exc).__name__}: {exc}"[:300]}}
    run("full_rows256", lambda: Full(256, dev, 0, args.path), 10, 3, streams=True, clock=False, more=bound_replay)

    def replay(wl):
        try:
            r = wl.graph_replay(20)
            return {"hip_graph_replay": {k: r[k] for k in ("value", "unit", "ms_per_step", "replay_equals_eager")}}
        except Exception as exc:
            return {"hip_graph_replay": {"error": f"{type(exc).__name__}: {exc}"[:300]}}
    run("fullref", lambda: FullRefDefault(256, dev, 0, args.path), 20, 3, clock=False, more=replay)
    keep_map = vqmod.TRAIN_SPLIT_TEXT_MAPPING
    try:
        run("cfg4_vq_only", lambda: Cfg4(256, dev, 0, args.path, precomputed=True), 10, 4, clock=False)   # (the first steps grow the allocator's pools)
    finally:
        vqmod.TRAIN_SPLIT_TEXT_MAPPING = keep_map
    out["note"] = ("short runs of `bench.py --workload <name>` inside the default command, after its timed region (same classes and timing protocol; "
                   "`full*`: value from the multi-stream forward as shipped, kernel durations / frac from a one-stream pass; cfg4_vq_only = "
                   "--workload cfg4 --precomputed-encoders); rank-local")
    return out


def collective_block(args, rank, world, dev):
    """N > 1 only: the two partitionings of BASELINE config 5 that DO exchange data, so that the driver's unchanged command line
    (`bench.py --gpus N`, whose headline workload shards rows with no data-path collective) also measures RCCL over xGMI.
      cfg5_ema_step     600k rows in total row-sharded over the ranks, K = 16384: argmin + EMA statistics + ONE all-reduce of
                        [embed_sum | bins] (50.4 MB) + the fused codebook update -- strong scaling;
      codeshard_search  one K = 49152 top-5 search of 600k replicated rows with the CODEBOOK sharded over the ranks: ONE packed
                        all-gather of the k-lists (n * k * 8 B per rank) + the exact merge -- strong scaling.
    Each is timed like the headline (warm-up, barrier + synchronize on both sides, max over ranks); the collectives are
    bracketed by HIP events on the stream they are enqueued on (medtok_amd.distributed.COLLECTIVE_TIMER)."""
    out = {}
    total_rows = 600000

    def timed(wl, units):
        for _ in range(max(args.warmup, 1)):
            wl.step()
        torch.cuda.synchronize(dev); mdist.barrier(); torch.cuda.synchronize(dev)
        mdist.COLLECTIVE_TIMER = []
        try:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize(dev); mdist.barrier(); torch.cuda.synchronize(dev)
            dt = mdist.max_over_ranks(time.perf_counter() - t0, dev)
            recs = mdist.COLLECTIVE_TIMER
        finally:
            mdist.COLLECTIVE_TIMER = None
        ms = [e0.elapsed_time(e1) for _, e0, e1, _ in recs]
        coll_ms = mdist.max_over_ranks(sum(ms) / max(args.steps, 1), dev)
        nbytes = sum(b for _, _, _, b in recs) / max(args.steps, 1)
        return dict(value=units * args.steps / dt, unit="rows/s (whole job)", ms_per_step=dt / args.steps * 1e3, collective=recs[0][0] if recs else None,
                    collective_ms_per_step=coll_ms, collective_bytes_per_step=nbytes, collectives_per_step=len(recs) / max(args.steps, 1),
                    # ring-equivalent bus bandwidth: all-reduce moves 2 (N-1)/N of the buffer per rank, all-gather (N-1)/N of the result
                    busbw_gbs=((2.0 if recs and recs[0][0] == "all_reduce" else 1.0) * (world - 1) / world * nbytes / (coll_ms * 1e-3) / 1e9
                               if coll_ms > 0 else None))
    lo, hi = mdist.row_shard(total_rows, rank, world)
    wl = Cfg2(hi - lo, dev, seed=rank, path=args.path, k_codes=16384)
    out["cfg5_ema_step"] = dict(timed(wl, total_rows), workload=f"{total_rows} rows row-sharded x{world}, D=768, K=16384, train step")
    del wl
    torch.cuda.empty_cache()
    wl = CodeShard(total_rows, dev, seed=0, path=args.path, rank=rank, world=world)
    out["codeshard_search"] = dict(timed(wl, total_rows), workload=f"{total_rows} rows replicated, K=49152 sharded x{world}, top-5 + soft assignment")
    del wl
    torch.cuda.empty_cache()
    out["backend"] = torch.distributed.get_backend()
    return out


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh rank processes (one per GPU, env://
    rendezvous on 127.0.0.1 -- the launch contract of MedTok/utils/distributed.py:20-58), relay rank 0's JSON line and
    return the worst exit code.  This parent never touches the GPU (no torch.cuda call that initialises HIP), and the
    children are new processes, not an exec of this one."""
    import socket
    import subprocess
    backend = os.environ.get("MEDTOK_DIST_BACKEND") or "nccl"
    if backend == "nccl":
        # RCCL needs one GPU per rank: refuse up front.  The count is read in a short-lived CHILD -- even counting devices can
        # open the driver, and this parent must stay a process that has never touched the GPU.
        try:
            have = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                      capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1])
        except Exception:
            have = None         # could not tell: let the ranks report it (init_distributed() fails cleanly when LOCAL_RANK has no GPU)
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible (RCCL needs one GPU per rank)", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                      # the ranks' stdout: only the JSON line goes to ours
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to benchmark)")
    rank, local, world = mdist.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.one_stream:
        import medtok_amd.vector_quantization_soft_one_new as vqmod
        vqmod.SIDE_STREAM_MIN_CODES = 0
    if args.workload == "cfg5":
        # BASELINE config 5 (EMA variant): 600k rows in total, row-sharded over the GPUs, one all-reduce of [embed_sum | bins] per step
        total_rows = args.rows or 600000
        lo, hi = mdist.row_shard(total_rows, rank, world)
        rows = hi - lo
        wl = Cfg2(rows, dev, seed=rank, path=args.path, k_codes=16384)
        wl.name = "cfg5"
        wl.description = (f"cfg5 NormEMA argmin + EMA update (train): {total_rows} rows total row-sharded x{world} ({rows}/GPU), D=768, "
                          f"K=16384, one all-reduce of [embed_sum | bins] = {16384 * 769 * 4 / 1e6:.1f} MB per step")
    elif args.workload == "codeshard":
        rows = args.rows or 600000
        wl = CodeShard(rows, dev, seed=0, path=args.path, rank=rank, world=world)
    elif args.workload == "cfg4":
        rows = args.rows or 256
        wl = Cfg4(rows, dev, seed=rank, path=args.path, text_layers=args.text_layers, precomputed=args.precomputed_encoders)
        if args.precomputed_encoders:
            wl.description = wl.description.replace("stand-in BERT-shaped text encoder", "NO encoders (pre-computed text / node features); was: stand-in BERT-shaped text encoder")
    else:
        rows = args.rows or {"cfg3": 600000, "full": 4096, "refdefault": 600000, "fullref": 256}.get(args.workload, 100000)
        cls = {"cfg3": Cfg3, "full": Full, "refdefault": RefDefault, "fullref": FullRefDefault}.get(args.workload, Cfg2)
        if args.data != "gaussian":
            if args.workload not in ("cfg3", "refdefault"):
                raise SystemExit("--data applies to --workload cfg3 / refdefault")
            wl = cls(rows, dev, seed=rank, path=args.path, data=args.data)
        else:
            wl = cls(rows, dev, seed=rank, path=args.path)

    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize(dev)
    # A training step is ~180 library launches, and the event pairs around its 40 dense products alone cost 0.27 ms of a 10.5 ms step
    # (tools/r06/ab_cfg4_profile_cost.py): cfg4's timed region runs WITHOUT the library's events; the kernel durations of its roofline
    # object come from the same K steps repeated right behind it with the events on (like the multi-stream forwards' one-stream pass)
    events_behind = args.workload == "cfg4"
    import gc
    gc.collect()                         # (a generation-2 collection in the middle of 5 - 20 timed steps is 80 ms on this host)
    mdist.barrier()
    torch.cuda.synchronize(dev)
    if not events_behind:
        ops.profile_begin()              # library brackets each search-kernel launch with HIP events on its stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize(dev)
    mdist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = mdist.max_over_ranks(time.perf_counter() - t0, dev)
    events_pass_ms = None
    if events_behind:
        ops.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize(dev)
        events_pass_ms = (time.perf_counter() - t1) / max(args.steps, 1) * 1e3
    prof = ops.profile_end()
    # The shader clock the steps run at (the chip clocks to its power budget, and boxes differ by a few per cent -- as much as a round's
    # kernel work moves the headline): one idle wavefront per XCD on a stream of its own counts shader cycles against the 100 MHz
    # counter over `clock_steps` FURTHER steps right behind the timed region (same state of the chip; the timed region itself runs
    # with nothing beside it).  Only where the work is on ONE stream (a probe stream that lands on a hardware queue of a side stream
    # would serialise with it; the `full` workload takes it in its one-stream pass below), not for cfg4 (torch's encoders / autograd
    # use queues of their own: measured 107 -> 132 ms per step with the probe beside them), not where a step is too short to matter.
    clock = None
    step_s = elapsed / max(args.steps, 1)
    single_stream = (args.workload not in ("full", "fullref", "cfg4") or (args.one_stream and args.workload != "cfg4")) and step_s >= 0.005
    if single_stream and not args.no_clock_probe:
        clock_steps = max(2, min(args.steps, int(0.5 / max(step_s, 1e-4))))
        with ops.ClockProbe(dev, max_seconds=3.0 * step_s * clock_steps + 5.0) as probe:
            for _ in range(clock_steps):
                wl.step()
            torch.cuda.current_stream(dev).synchronize()
        clock = dict(probe.result(), region=f"{clock_steps} further steps right behind the timed region")
        torch.cuda.synchronize(dev)
    # what the shortlist left behind, read once AFTER the timed region from one further step (device-side reduction of the candidate
    # counts; the count of rows handed to the exact kernel is the library's own device counter `fb_count`)
    fallback = None
    if args.workload in ("cfg3", "refdefault") and args.path in (ops.PATH_AUTO, ops.PATH_F16_FILTER):
        try:
            ops.FILTER_STATS = []
            wl.step()
            stats = [ops.filter_stats(e) for e in ops.FILTER_STATS]
            fallback = {"rows_handed_to_the_exact_kernel_per_step": sum(t["fallback_rows"] for t in stats),
                        "searches": stats, "read": "once, from one further step behind the timed region (ops.filter_stats)"}
        except Exception as exc:
            fallback = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finally:
            ops.FILTER_STATS = None
    prof_note = None
    if events_behind:
        prof_note = (f"the same {args.steps} steps repeated right behind the timed region with the library's event pairs on ({events_pass_ms:.2f} ms per "
                     f"step there): a training step is ~180 library launches, the pairs around its dense products alone cost ~0.27 ms per step")
    one_stream_elapsed = None
    if args.workload in ("full", "fullref") and not args.no_one_stream_pass and not args.one_stream:
        # The forward enqueues on several HIP streams: an event pair around a launch then also covers the other streams' kernels that
        # share the device with it (a 0.05 ms text-side product is "1.3 ms" beside the graph side's attention).  Kernel durations for
        # the roofline object come from a second pass of the same steps on ONE stream; `value` is the multi-stream timed region above.
        import medtok_amd.vector_quantization_soft_one_new as vqmod
        keep = vqmod.SIDE_STREAM_MIN_CODES
        vqmod.SIDE_STREAM_MIN_CODES = 0
        try:
            wl.step()
            torch.cuda.synchronize(dev)
            ops.profile_begin()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize(dev)
            one_stream_elapsed = time.perf_counter() - t1
            prof = ops.profile_end()
            if not args.no_clock_probe and step_s >= 0.005:
                with ops.ClockProbe(dev, max_seconds=3.0 * step_s * args.steps + 5.0) as probe1:
                    for _ in range(max(2, args.steps)):
                        wl.step()
                    torch.cuda.current_stream(dev).synchronize()
                clock = dict(probe1.result(), region="further one-stream steps behind the one-stream pass")
                torch.cuda.synchronize(dev)
        finally:
            vqmod.SIDE_STREAM_MIN_CODES = keep
        prof_note = ("kernel durations from a second pass of the same steps on one stream (event pairs of overlapping streams include each "
                     "other's kernels); value / ms_per_step are the multi-stream timed region")

    # full workload, extra: the same forward with the text features in fp16 -- what a caller under fp16 autocast hands over (the
    # reference's default mode, train_MedTok.py:212,394): no image pass, half the key bytes, two matrix passes per product
    half_text = None
    if args.workload == "full" and not args.no_half_text_pass:
        text32 = wl.text
        try:
            wl.text = text32.half()
            wl.step(); wl.step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            half_text = {"value": float(rows) * world * args.steps / dt, "unit": "codes/s", "ms_per_step": dt / args.steps * 1e3,
                         "note": "the same steps with the text features handed over in fp16 (rank-local timing; every other input fp32)"}
        finally:
            wl.text = text32

    # the exact fp32-MFMA path on the same workload (1 step): the filter path returns the same bits, faster
    exact = None
    if args.path == ops.PATH_AUTO and args.exact_steps > 0 and args.workload != "cfg4":
        agree = wl.paths_agree()             # same state, same inputs, both paths: every output tensor must be bit-identical
        wl.set_path(ops.PATH_F32_MFMA)
        wl.step()
        torch.cuda.synchronize(dev)
        ops.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.exact_steps):
            wl.step()
        torch.cuda.synchronize(dev)
        e_elapsed = time.perf_counter() - t1
        e_prof = ops.profile_end()["search_f32_kernel"]      # (for the `full` workload this block reports the search part only)
        e_ach = e_prof["flops"] / (e_prof["ms"] * 1e-3) / 1e12 if e_prof["ms"] > 0 else 0.0
        exact = {"value": rows * args.exact_steps / e_elapsed, "unit": "codes/s per GPU", "steps": args.exact_steps,
                 "outputs_bit_identical_to_default_path": bool(agree),
                 "roofline": {"bound": "mfma", "kernel": "search_f32_kernel", "achieved": e_ach, "peak": FP32_MFMA_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": e_ach / FP32_MFMA_PEAK_TFLOPS,
                              "avg_launch_ms": e_prof["ms"] / max(e_prof["launches"], 1)}}
        wl.set_path(args.path)

    kname, kp, bound_view, f16x3, hbm_bound, prof = kernel_roofline(wl, prof, args.steps)
    achieved = bound_view.get("mfma_view", bound_view).get("achieved")
    traffic, traffic_source = pmc_traffic(wl.name, kname, rows)
    alg_bytes_step = float(wl.bytes_per_code()) * rows                # SURVEY 8d per-code figure x the codes one step processes (per GPU)
    hbm_gbs = alg_bytes_step * args.steps / elapsed / 1e9

    # N > 1 under the headline workload: also time the two partitionings that exchange data (RCCL all-reduce / all-gather)
    strong = None
    if world > 1 and args.workload == "cfg3" and not args.no_collectives:
        del wl.h, wl.pooled_text, wl.pooled_graph
        torch.cuda.empty_cache()
        try:
            strong = collective_block(args, rank, world, dev)
        except Exception as exc:                 # (the headline line must still be printed: this block is an extra)
            strong = {"error": f"{type(exc).__name__}: {exc}"[:500]}

    if rank == 0:
        total_codes = (float(args.rows or 600000) if args.workload in ("cfg5", "codeshard") else float(rows) * world) * args.steps
        value = total_codes / elapsed
        line = {
            "metric": "codes_per_sec_tokenized",
            "value": value,
            "unit": "codes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.workload in ("cfg5", "codeshard") else "weak",
            "vs_baseline": None,
            "dtype": ("f32 results (bit-identical to the fp32-MFMA path); f16-MFMA shortlist + exact f32 re-score"
                      if kname == "filter_f16_kernel" else
                      "f32 in / f32 out; products as three f16 MFMA passes over (hi, lo) pairs (~2^-22 relative), f32 softmax and accumulation" if f16x3 else "f32"),
            "data": "synthetic",
            "n_ranks_seen": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
            "config": {"workload": wl.description, "rows_per_gpu": rows, "D": wl.D, "search_path": args.path,
                       "search": ("fp16-MFMA shortlist with a proven error bound + exact fp32 re-score: token ids and distances are "
                                  "bit-identical to the fp32-MFMA path (tests/test_gpu_filter.py)" if kname == "filter_f16_kernel"
                                  else "exact fp32 MFMA"),
                       "parallelism": (f"code-shard x{world}: all-gather of the per-rank k-lists (n*k*12 B per rank) + exact merge"
                                       if args.workload == "codeshard" else
                                       f"row-shard x{world}, codebook replicated, no data-path collective")},
            "roofline": {**bound_view,
                         "traffic": traffic, "traffic_source": traffic_source,
                         # fabric bytes of the WHOLE step (every kernel) over its SURVEY-8d algorithmic bytes: re-reads through the L2s /
                         # Infinity Cache (the fp16 codebook image is re-streamed per 256-row block) -- recorded, like `traffic`
                         "fabric_over_algorithmic": (step_fabric_bytes(wl.name, rows) / alg_bytes_step) if step_fabric_bytes(wl.name, rows) else None,
                         # (rows of <= 64 elements take the fp16 filter's narrow-row kernel: the name rocprofv3 shows)
                         "kernel": ("filter_rows64n_kernel" if kname == "filter_f16_kernel" and wl.D <= 64 else kname),
                         "hbm_frac": hbm_gbs / HBM_PEAK_GBS, "hbm_achieved_gbs": hbm_gbs, "hbm_peak_gbs": HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg_bytes_step,
                         "binding_roof": ("hbm for this kernel (intensity left of the ridge of the pipe it runs on); hbm_frac below is the whole STEP's SURVEY-8d bytes"
                                          if hbm_bound else
                                          "mfma (arithmetic intensity K/4 flop/B >> ridge; the HBM fraction is reported because BASELINE.json asks for it)"),
                         "peak_note": ("dense f16 / bf16 MFMA (one half-precision pass under autocast)" if bound_view.get("one_half_precision_pass") else
                                       "dense f16 MFMA" if kname == "filter_f16_kernel" else
                                       "dense f16 MFMA / 3: every product is three fp16 passes over (hi, lo) pairs" if f16x3 else "dense f32-input MFMA") + " (MI355X_MICROARCH.md)",
                         "launches_timed": kp["launches"], "avg_launch_ms": kp["ms"] / max(kp["launches"], 1), "timed_in": prof_note or "the timed region",
                         "algorithmic_flops_per_launch": (kp["flops"] / max(kp["launches"], 1)) if kp["flops"] > 0 else None,
                         "kernel_share_of_step": kp["ms"] / (elapsed * 1e3),
                         "achieved_over_fp32_mfma_peak": (achieved / FP32_MFMA_PEAK_TFLOPS) if achieved is not None else None,
                         "whole_step_tflops": wl.flops_per_code() * rows * args.steps / elapsed / 1e12,
                         "other_kernels": {k: {"ms_per_step": v["ms"] / args.steps, "launches": v["launches"],
                                               "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0)}
                                           for k, v in prof.items() if k != kname and v["launches"]},
                         **({"events_pass_ms_per_step": events_pass_ms} if events_pass_ms is not None else {})},
            "exact_fp32_path": exact,
            # shader clock of further steps of the same workload right behind the timed region (ops.ClockProbe): separates the box (its power
            # budget / silicon) from the code when two lines differ by a few per cent; the dense-MFMA peaks above are quoted at 2.4 GHz
            "clock": clock,
        }
        ir = issue_roofs(wl.name, kname, rows, kp["ms"] / max(kp["launches"], 1), (clock or {}).get("ghz_mean"))
        if ir is not None:
            line["roofline"]["issue_roofs"] = ir
        if strong is not None:
            line["extra"] = {"strong_scaling": strong}
        if fallback is not None:
            line["fallback_rows"] = fallback
        if args.workload in ("full", "fullref"):
            # both stream settings in one line: `value` is the forward as shipped (side streams from 512 codes up) unless --one-stream
            line["config"]["streams"] = "one (--one-stream)" if args.one_stream else "main + side streams (modality-specific searches, text side; the image pass over fp32 text rows has its own only where the attention kernel does not split the keys itself)"
            if half_text is not None:
                line["half_precision_text"] = half_text
            if args.workload == "full" and not args.no_one_stream_pass:
                try:
                    line["no_host_read"] = wl.no_host_read(max(args.steps, 5))
                except Exception as exc:           # (an extra: the headline line is printed regardless)
                    line["no_host_read"] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
            if args.workload == "fullref" and hasattr(wl, "graph_replay"):
                try:
                    line["hip_graph_replay"] = wl.graph_replay(max(args.steps, 20))
                except Exception as exc:           # (an extra: the headline line is printed regardless)
                    line["hip_graph_replay"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            if one_stream_elapsed is not None:
                line["one_stream"] = {"value": float(rows) * world * args.steps / one_stream_elapsed, "unit": "codes/s",
                                      "ms_per_step": one_stream_elapsed / args.steps * 1e3,
                                      "note": "the same steps with every launch on ONE HIP stream (per-kernel event timing on; rank-local)"}
        if args.workload == "cfg4":
            enc = wl.encoder_ms()
            line["dtype"] = "bf16 autocast (encoders, projections); the searches, their backward and the losses run in f32"
            line["config"]["encoders_ms_per_step"] = enc
            line["config"]["vq_path_ms_per_step"] = max(line["ms_per_step"] - enc, 0.0)
            line["config"]["note"] = ("stand-in encoders (out of scope, upstream of the path); vq_path = cross-attention + 6 searches + "
                                      "loss.py + backward + clip + AdamW = step - encoders")
            # the same steps with CrossAttention.max_nodes_bound set to the batch's largest subgraph (a trainer knows its dataset's): the
            # cross-attention then sizes its launches from the bound -- one of the step's two host reads is gone (an extra, not `value`)
            try:
                ca = wl.model.quantize.cross_attn
                ca.max_nodes_bound = int(torch.bincount(wl.inputs.batch).max())
                for _ in range(2):
                    wl.step()
                torch.cuda.synchronize(dev)
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    wl.step()
                torch.cuda.synchronize(dev)
                line["with_max_nodes_bound"] = {"ms_per_step": (time.perf_counter() - t2) / max(args.steps, 1) * 1e3, "max_nodes_bound": ca.max_nodes_bound,
                                                "note": "CrossAttention.max_nodes_bound set: no host read of the node counts in pooled(); the usage counts' read remains"}
            except Exception as exc:
                line["with_max_nodes_bound"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            finally:
                wl.model.quantize.cross_attn.max_nodes_bound = None
        if world == 1 and args.workload == "cfg3" and args.data == "gaussian" and args.rows is None and not args.no_extra_workloads:
            del wl.h, wl.pooled_text, wl.pooled_graph
            torch.cuda.empty_cache()
            line.setdefault("extra", {})["workloads"] = extra_workloads(args, dev)
        cpu_rows = args.cpu_rows if args.cpu_rows is not None else {"full": 512, "fullref": 512, "cfg4": 256}.get(args.workload, 16384)
        if world == 1 and cpu_rows > 0:
            line["cpu_baseline"] = wl.cpu_baseline(cpu_rows)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
