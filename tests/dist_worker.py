"""Rank program of the multi-rank GPU tests (tests/test_gpu_distributed.py): started by torch.distributed.run with two ranks
that share this box's one GPU over gloo (MEDTOK_DIST_BACKEND=gloo; RCCL itself needs one GPU per rank).  Every mode runs the
PRODUCT path -- the HIP kernels through the C ABI -- and writes what it computed to <out>/<mode>_r<rank>.npz; the test process
compares that with its own single-rank run of the same seeded problem.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_worker.py <mode> <out_dir> [sizes...]
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from medtok_amd import distributed as D  # noqa: E402


def ema_problem(n, k, d, dev):
    """Seeded rows and a normalised start codebook, generated on the device (same GPU type: same values in every process)."""
    g = torch.Generator(device=dev).manual_seed(20240)
    z = torch.randn(n, d, device=dev, generator=g)
    e0 = torch.nn.functional.normalize(torch.randn(k, d, device=dev, generator=g), dim=-1)
    return z, e0


def ema_step(z_shard, e0, beta=0.25, decay=0.99, steps=2):
    """`steps` train forwards of the HIP NormEMAVectorQuantizer on this rank's rows; the module picks its all-reduce at
    construction time (norm_ema_quantizer.py:155-159), i.e. after init_process_group."""
    from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
    k, d = e0.shape
    q = NormEMAVectorQuantizer(k, d, beta, decay).to(z_shard.device).train()
    q.embedding.weight.data.copy_(e0)
    ids, losses = [], []
    with torch.no_grad():
        for _ in range(steps):
            _, loss, idx = q(z_shard[:, :, None, None])
            ids.append(idx.clone()); losses.append(loss.clone())
    return q, ids, losses


def mode_ema(out, rank, world, dev, n, k, d):
    z, e0 = ema_problem(n, k, d, dev)
    lo, hi = D.row_shard(n, rank, world)
    q, ids, losses = ema_step(z[lo:hi].contiguous(), e0)
    del z
    ids_all = [D.gather_rows(i, n).cpu().numpy() for i in ids]
    np.savez(out / f"ema_r{rank}.npz", ids0=ids_all[0], ids1=ids_all[1], cluster_size=q.cluster_size.cpu().numpy(),
             weight=q.embedding.weight.data.cpu().numpy(), lo=lo, hi=hi, loss=np.array([float(x) for x in losses]))


def codeshard_problem(n, k, d, dev):
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(777)
    x = torch.randn(n, d, device=dev, generator=g)
    W = torch.randn(k, d, device=dev, generator=g)
    W[k // 2 + 3] = W[5]                       # a duplicate that straddles the shard boundary: the tie rule across ranks
    xh, xs = ops.rownorm(x)
    wh, ws = ops.rownorm(W)
    return xh, xs, wh, ws


def mode_codeshard(out, rank, world, dev, n, k, d):
    xh, xs, wh, ws = codeshard_problem(n, k, d, dev)
    lo, hi = D.code_shard(k, rank, world)
    idx, dist_ = D.code_sharded_search(xh, xs, wh[lo:hi], ws[lo:hi].contiguous(), lo, 5)     # HIP search + HIP merge kernel
    np.savez(out / f"codeshard_r{rank}.npz", idx=idx.cpu().numpy(), dist=dist_.cpu().numpy())


def ddp_model(dev, n_e=768, dim=64):
    """MultimodalTokenizer around the HIP VectorQuantizer with the stand-in encoders (one text layer), seeded."""
    from medtok_amd.synthetic import StandInGAT, StandInTextEncoder
    from medtok_amd.tokenizer import MultimodalTokenizer
    torch.manual_seed(4242)
    m = MultimodalTokenizer(StandInTextEncoder(layers=1, dim=128, heads=4, ffn=256, max_len=32), StandInGAT(n_nodes=500, dim=dim), text_dim=128,
                            graph_out_channels=dim, codebook_size=n_e, codebook_embed_dim=dim).to(dev).train()
    for p in m.text_model.parameters():
        p.requires_grad = False                                   # tokenizer.py:80-81
    for layer in m.quantize.cross_attn.model:                    # deterministic step: the dropout masks are per-process random
        layer.multihead_attn.dropout = 0.0
        layer.dropout.p = 0.0
    return m


def ddp_batch(bsz, dev, seed):
    from medtok_amd.synthetic import primekg_shaped_batch
    b = primekg_shaped_batch(bsz, dev, seed=seed, max_len=32)
    b.x = b.x % 500
    return b


def ddp_step(model, batch):
    """forward -> loss.py assembly -> backward (train_MedTok.py:207-240), fp32"""
    from medtok_amd import loss as L
    r = model(batch)
    loss, _ = L.total_loss(r, 0.1, 0.1)
    loss.backward()
    return loss.detach()


def mode_ddp(out, rank, world, dev, bsz):
    from torch.nn.parallel import DistributedDataParallel as DDP
    m = ddp_model(dev)
    ddp = DDP(m, device_ids=[dev.index], find_unused_parameters=True)       # train_MedTok.py:185
    loss = ddp_step(ddp, ddp_batch(bsz, dev, seed=100 + rank))
    q = m.quantize
    np.savez(out / f"ddp_r{rank}.npz", loss=float(loss), g_codebook=q.codebook.weight.grad.cpu().numpy(),
             g_proj=q.proj_text.weight.grad.cpu().numpy(), g_inproj=q.cross_attn.model[0].multihead_attn.in_proj_weight.grad.cpu().numpy(),
             g_text_mapped=m.text_mapped.weight.grad.cpu().numpy(), used=q.codebook_used.cpu().numpy()[-2048:])


def inference_model(dev):
    m = ddp_model(dev).eval()
    return m


def inference_batches(n_codes, bsz, dev):
    """The dataset as a list of batches with their code indices (shuffled: the driver must order by index)."""
    out = []
    perm = torch.randperm(n_codes, generator=torch.Generator().manual_seed(5))
    for b0 in range(0, n_codes, bsz):
        ids = perm[b0: b0 + bsz]
        b = ddp_batch(len(ids), dev, seed=1000 + b0)
        b.code_indices = ids
        out.append(b)
    return out


def mode_inference(out, rank, world, dev, n_codes, bsz):
    from medtok_amd.inference import run_inference
    m = inference_model(dev)
    batches = inference_batches(n_codes, bsz, dev)
    mine = batches[rank::world]                                  # DistributedSampler-style split of the batches
    res = run_inference(m, mine, out_dir=out / "sharded", device=dev)
    if rank == 0:
        assert res is not None
    np.savez(out / f"inference_r{rank}.npz", done=1)


def main():
    mode, out = sys.argv[1], Path(sys.argv[2])
    sizes = [int(a) for a in sys.argv[3:]]
    rank, local, world = D.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    {"ema": mode_ema, "codeshard": mode_codeshard, "ddp": mode_ddp, "inference": mode_inference}[mode](out, rank, world, dev, *sizes)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
