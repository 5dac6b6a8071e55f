"""Row-sharded multi-GPU driver for the VQ path: one process per GPU, RCCL over xGMI.

The path shards by rows (codes to tokenize): every rank holds the whole codebook
(25-151 MB) and searches its own slice of the input with NO communication, so
token ids are identical to a single-GPU run by construction.  The only exchange
step is the EMA training update, where the per-rank statistics [embed_sum | bins]
are summed with one all-reduce (the reference issues two: norm_ema_quantizer.py:195,203)
before every rank applies the same codebook update.

Launch contract (reference MedTok/utils/distributed.py:20-58): env:// rendezvous,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from torchrun.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

# dmabuf IPC is the only kind the host driver supports; must be in the environment before the HIP runtime starts,
# i.e. before the first torch.cuda call of the process (importing this module is early enough; launchers export it too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size); initialises the default process group when
    WORLD_SIZE > 1.  backend defaults to nccl (= RCCL on ROCm) with a GPU, gloo without."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        use_gpu = torch.cuda.is_available()
        # MEDTOK_DIST_BACKEND=gloo: run the N > 1 code paths on a single-GPU box (RCCL needs one GPU per rank)
        backend = backend or os.environ.get("MEDTOK_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        if use_gpu:
            n_dev = torch.cuda.device_count()
            if backend == "gloo":
                local = local % n_dev                   # several ranks may share one GPU under gloo only
            elif local >= n_dev:
                raise RuntimeError(f"LOCAL_RANK={local} but only {n_dev} GPU(s) are visible: RCCL needs one GPU per rank "
                                   f"(oversubscribed launch or wrong LOCAL_RANK)")
            torch.cuda.set_device(local)
        kw = {}
        if backend == "gloo":
            # gloo picks its interface by resolving the machine's hostname, which a container may not be able to do (a rendezvous
            # that then hangs until its default 30-minute timeout): on a loopback rendezvous bind to loopback, and give up early
            if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            # ... and give up EARLY on a rendezvous that never completes (300 s, not torch's 30 minutes) -- the rendezvous only: see below
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("MEDTOK_DIST_TIMEOUT_S", "300")))
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
        if "timeout" in kw:
            # The timeout handed to init_process_group also bounds every later collective and barrier, and a rank may legitimately
            # work alone for longer than 300 s (rank 0 sorting and writing the inference table while the others wait at the barrier,
            # a k-means init, a checkpoint write): once the group stands, collectives get torch's default back.
            dist.barrier()
            try:
                dist.distributed_c10d._set_pg_timeout(dist.distributed_c10d.default_pg_timeout)
            except Exception:           # (a private hook: without it the short timeout simply stays, as before)
                pass
    elif torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)
    return rank, local, world


def row_shard(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n rows for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Measurement hook (bench.py): when set to a list, every data-path collective below is bracketed by HIP events on the current
# stream -- torch's synchronous collectives make the current stream wait for the communicator's stream, so the pair encloses the
# collective -- and (name, start, stop, payload_bytes) is appended.  None = no overhead.
COLLECTIVE_TIMER = None


def _timed(name: str, nbytes: int, fn):
    timer = COLLECTIVE_TIMER
    if timer is None or not torch.cuda.is_available():
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    timer.append((name, e0, e1, int(nbytes)))
    return out


def all_reduce_sum(t: torch.Tensor) -> None:
    """torch.distributed.all_reduce(t) (SUM, in place, default group) -- what NormEMAVectorQuantizer binds at construction time
    (norm_ema_quantizer.py:155-159), visible to the measurement hook above."""
    _timed("all_reduce", t.numel() * t.element_size(), lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))


def all_reduce_stats(stats: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM of the fused EMA statistics buffer across ranks (no-op single-process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _timed("all_reduce", stats.numel() * stats.element_size(),
               lambda: dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group))
    return stats


def max_over_ranks(value: float, device) -> float:
    """Timing helper for bench.py: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":        # name the device: RCCL otherwise guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank row shards (made with row_shard) back into [n_total, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [row_shard(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((longest,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def gather_ragged(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank tensors whose FIRST dimension differs (rank order) on every rank: one all-gather of the lengths,
    one of the rows padded to the longest shard.  Single-process: returns `local`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(lens, n_local, group=group)
    lens = [int(x.item()) for x in lens]
    longest = max(lens)
    pad = local.new_zeros((longest,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    _timed("all_gather", pad.numel() * pad.element_size() * world, lambda: dist.all_gather(parts, pad, group=group))
    return torch.cat([p[:m] for p, m in zip(parts, lens)], dim=0)


def gather_ragged_to_rank0(local: torch.Tensor, device, chunk_bytes: int = 64 << 20, group=None):
    """Concatenate per-rank HOST tensors whose first dimension differs (rank order) on rank 0 only; the other ranks get None.
    The rows travel in chunks of at most `chunk_bytes` per rank (staged through `device` when the backend is RCCL, which only
    moves device memory), so neither the senders nor rank 0 ever hold more than world x chunk_bytes of it on a GPU -- an
    inference table of 600k codes x 3072 floats is several GB per rank.  Single-process: returns `local`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    stage = torch.device(device) if on_gpu else torch.device("cpu")
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=stage)
    lens = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(lens, n_local, group=group)
    lens = [int(x.item()) for x in lens]
    row_shape = tuple(local.shape[1:])
    row_bytes = local.element_size() * math.prod(row_shape)
    step = max(int(chunk_bytes) // row_bytes, 1)
    out = [torch.empty((m,) + row_shape, dtype=local.dtype) for m in lens] if rank == 0 else None
    for lo in range(0, max(lens), step):
        send = torch.zeros((step,) + row_shape, dtype=local.dtype, device=stage)
        mine = local[lo: lo + step]
        send[: mine.shape[0]].copy_(mine)
        parts = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, parts, dst=0, group=group)
        if rank == 0:
            for r, m in enumerate(lens):
                take = min(max(m - lo, 0), step)
                if take:
                    out[r][lo: lo + take].copy_(parts[r][:take])
    return torch.cat(out, dim=0) if rank == 0 else None


def code_shard(k_codes: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the codebook for `rank` (same balancing rule as row_shard)."""
    return row_shard(k_codes, rank, world)


def code_sharded_search(xhat: torch.Tensor, xsq: torch.Tensor, what_local: torch.Tensor, wsq_local: torch.Tensor,
                        code_lo: int, topk: int, group=None, search_fn=None, merge_fn=None):
    """The variant north_star names: every rank holds a slice [code_lo, code_lo + K_local) of the codebook and ALL rows.
    Local top-k over the slice -> ONE all-gather of the packed (distance, global id) lists (n * k * 8 bytes per rank over
    xGMI) -> exact (d, index) merge.  Identical bits to a single-GPU search because the contract's dot-product order does
    not depend on where a code lives.  `search_fn` / `merge_fn` default to the HIP ops; the CPU gloo test passes the
    oracle's so the exchange logic runs without a GPU."""
    if search_fn is None or merge_fn is None:
        from . import ops
        search_fn = search_fn or (lambda a, b, c, d, k: ops.topk_search(a, b, c, d, k))
        merge_fn = merge_fn or ops.merge_topk_lists
    idx, d_local = search_fn(xhat, xsq, what_local, wsq_local, topk)
    idx = idx + code_lo
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return idx, d_local
    world = dist.get_world_size(group)
    n, k = idx.shape
    # ONE collective: a (distance, id) pair travels as one int64 -- the fp32 bits of the distance in the high word, the global
    # code id (< 2^32) in the low word: n * k * 8 bytes per rank instead of two gathers of 4 + 8
    packed = (d_local.contiguous().view(torch.int32).to(torch.int64) << 32) | (idx & 0xFFFFFFFF)
    gathered = torch.empty((world * n, k), dtype=torch.int64, device=packed.device)
    _timed("all_gather", gathered.numel() * 8, lambda: dist.all_gather_into_tensor(gathered, packed, group=group))
    gathered = gathered.view(world, n, k)
    d_parts = (gathered >> 32).to(torch.int32).view(torch.float32)
    i_parts = gathered & 0xFFFFFFFF
    return merge_fn(d_parts, i_parts)
