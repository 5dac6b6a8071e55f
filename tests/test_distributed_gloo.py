"""CPU, 2 processes over gloo: the N>1 path (row shards + one all-reduce of the EMA statistics).

The kernels cannot run here, so each rank computes its shard's statistics with the oracle; what is
under test is the host logic in medtok_amd/distributed.py: the row partition, the fused
[embed_sum | bins] all-reduce, and that every rank ends with the same codebook as a single process
(tolerance 1e-6: fp32 partial sums are associated differently)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, k, d, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from medtok_amd import distributed as D
    from oracle import oracle as O
    r, _, w = D.init_distributed("gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((n, d), dtype=np.float32)
    E0 = O.rownorm(rng.standard_normal((k, d), dtype=np.float32))[0]
    lo, hi = D.row_shard(n, rank, world)
    zh, zs = O.rownorm(z[lo:hi])
    _, es = O.rownorm(E0, False)
    idx, _ = O.topk_search(zh, zs, E0, es, 1)
    bins, esum = O.ema_stats(zh, idx[:, 0], k)
    stats = torch.from_numpy(np.concatenate([esum.reshape(-1), bins]))
    D.all_reduce_stats(stats)
    esum_g = stats[: k * d].view(k, d).numpy().copy(); bins_g = stats[k * d:].numpy().copy()
    E, cs = E0.copy(), np.zeros(k, np.float32)
    O.ema_apply(E, cs, bins_g, esum_g, 0.99)
    ids_all = D.gather_rows(torch.from_numpy(idx[:, 0]), n).numpy()
    slow = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    D.barrier()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), E=E, cs=cs, ids=ids_all, slow=slow)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_ema_step_equals_single_process(tmp_path, oracle):
    n, k, d, world = 1001, 64, 32, 2
    mp.spawn(_worker, args=(world, _free_port(), n, k, d, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((n, d), dtype=np.float32)
    E = oracle.rownorm(rng.standard_normal((k, d), dtype=np.float32))[0]
    cs = np.zeros(k, np.float32)
    _, _, ids = oracle.norm_ema_forward(z, E, cs, 0.25, 0.99, True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["ids"], ids) and np.array_equal(r1["ids"], ids)      # ids: bit-exact
    assert np.array_equal(r0["E"], r1["E"]) and np.array_equal(r0["cs"], r1["cs"])  # ranks agree exactly
    assert np.array_equal(r0["cs"], cs)                                           # integer counts: exact
    assert np.abs(r0["E"] - E).max() <= 1e-6
    assert float(r0["slow"]) == 2.0 == float(r1["slow"])


def _code_shard_worker(rank, world, port, n, k, d, topk, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from medtok_amd import distributed as D
    from oracle import oracle as O
    D.init_distributed("gloo")
    rng = np.random.default_rng(1)
    xh, xs = O.rownorm(rng.standard_normal((n, d), dtype=np.float32))
    W = rng.standard_normal((k, d), dtype=np.float32)
    W[k // 2 + 3] = W[5]                                    # a duplicate that straddles the shard boundary: tie rule across ranks
    wh, ws = O.rownorm(W)
    lo, hi = D.code_shard(k, rank, world)

    def search(a, b, c, e, kk):
        i, dd = O.topk_search(a.numpy(), b.numpy(), c.numpy(), e.numpy(), kk)
        return torch.from_numpy(i), torch.from_numpy(dd)

    def merge(dp, ip):                                       # numpy stand-in for the HIP merge kernel: (d, index) lexicographic
        P, nn, kk = dp.shape
        dflat = dp.permute(1, 0, 2).reshape(nn, P * kk).numpy(); iflat = ip.permute(1, 0, 2).reshape(nn, P * kk).numpy()
        order = np.lexsort((iflat, dflat), axis=1)[:, :kk]
        return torch.from_numpy(np.take_along_axis(iflat, order, 1)), torch.from_numpy(np.take_along_axis(dflat, order, 1))

    idx, dist_ = D.code_sharded_search(torch.from_numpy(xh), torch.from_numpy(xs), torch.from_numpy(wh[lo:hi]), torch.from_numpy(ws[lo:hi]),
                                       lo, topk, search_fn=search, merge_fn=merge)
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), idx=idx.numpy(), dist=dist_.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_code_sharded_search_equals_single_process(tmp_path, oracle):
    n, k, d, topk, world = 300, 1000, 64, 5, 2
    mp.spawn(_code_shard_worker, args=(world, _free_port(), n, k, d, topk, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(1)
    xh, xs = oracle.rownorm(rng.standard_normal((n, d), dtype=np.float32))
    W = rng.standard_normal((k, d), dtype=np.float32); W[k // 2 + 3] = W[5]
    wh, ws = oracle.rownorm(W)
    idx, dd = oracle.topk_search(xh, xs, wh, ws, topk)
    for r in range(world):
        got = np.load(tmp_path / f"c{r}.npz")
        assert np.array_equal(got["idx"], idx) and np.array_equal(got["dist"], dd)       # bit-exact, tie across the boundary included


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from medtok_amd import distributed as D
    D.init_distributed("gloo")
    # ragged shards (rank 1 holds nothing of the int64 table: the zero-batch branch of run_inference), chunks far smaller than a shard
    rows = [37, 0][rank] if world == 2 else 5
    a = (torch.arange(rows * 6, dtype=torch.float32).view(rows, 2, 3) + 1000 * rank)
    b = torch.arange(rows, dtype=torch.int64) * 2 + rank
    ga = D.gather_ragged_to_rank0(a, "cpu", chunk_bytes=5 * 24)
    gb = D.gather_ragged_to_rank0(b, "cpu", chunk_bytes=64)
    if rank == 0:
        np.savez(os.path.join(out_dir, "g.npz"), a=ga.numpy(), b=gb.numpy())
    else:
        assert ga is None and gb is None
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_chunked_gather_to_rank0(tmp_path):
    """run_inference's table gather: per-rank host tensors of different lengths end up concatenated in rank order on rank 0 only,
    moved in bounded chunks (medtok_amd.distributed.gather_ragged_to_rank0)."""
    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "g.npz")
    assert np.array_equal(g["a"], np.arange(37 * 6, dtype=np.float32).reshape(37, 2, 3))
    assert np.array_equal(g["b"], np.arange(37, dtype=np.int64) * 2)
