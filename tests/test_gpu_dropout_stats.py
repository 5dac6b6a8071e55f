"""GPU: the attention-probability dropout of the training kernels as a random mask.  The reference draws its mask from Philox inside
nn.MultiheadAttention (vector_quantization_soft_one_new.py:21,30); the kernels use a stateless hash of (seed, packed query row, key) so
that the forward and both backward kernels regenerate the same mask (attention_kernels.h: att_keep).  Train-mode parity with the
reference holds with dropout off on both sides; what must hold with it on is statistical: the keep rate is 1 - p, the mask
is independent across keys, across the heads of a node (adjacent packed rows), across nodes and across seeds, and kept probabilities
are scaled by 1 / (1 - p).  The mask is read off the kernel's own output: with one-hot value rows, out[r, j] = P[r, j] M[r, j] / (1 - p)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _masks(dev, p, seeds, n_codes=64, q_len=64, T=128, D=128):
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    rows = n_codes * q_len
    q = 0.1 * torch.randn(rows, D, device=dev, generator=g)
    kv = torch.eye(T, D, device=dev)                                     # key j = e_j: scores = scale * q[:, j], values pick column j
    q_start = torch.arange(n_codes, device=dev) * q_len
    q_lens = torch.full((n_codes,), q_len, device=dev, dtype=torch.long)
    kv_start = torch.zeros(n_codes, device=dev, dtype=torch.long)          # every code attends to the same T keys
    kv_len = torch.full((n_codes,), T, device=dev, dtype=torch.long)
    full, _ = ops.shared_kv_attention_train(q, q_start, q_lens, kv, kv_start, kv_len, q_len, 0.3, 0.0, 0)
    assert bool((full[:, :T] > 0).all())
    outs = [ops.shared_kv_attention_train(q, q_start, q_lens, kv, kv_start, kv_len, q_len, 0.3, p, s)[0][:, :T] for s in seeds]
    return full[:, :T], outs


def _corr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / (a.norm() * b.norm()))


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_mask_rate_and_independence(dev, p):
    seeds = (1, 2, 12345, 2 ** 31 - 5)
    full, outs = _masks(dev, p, seeds)
    keeps = [(o != 0) for o in outs]
    n = keeps[0].numel()
    sigma = math.sqrt(p * (1 - p) / n)
    bound = 4.0 / math.sqrt(n)                       # |correlation| of independent masks: ~ N(0, 1/n)
    for s, keep, o in zip(seeds, keeps, outs):
        rate = float(keep.double().mean())
        assert abs(rate - (1 - p)) <= 3 * sigma, (s, rate)                                   # rate within 3 sigma of 1 - p
        assert abs(_corr(keep[:, :-1], keep[:, 1:])) <= bound, ("adjacent keys", s)
        assert abs(_corr(keep[:-1], keep[1:])) <= bound, ("adjacent packed rows = heads of one node", s)
        assert abs(_corr(keep[:-4], keep[4:])) <= bound, ("the same head of the next node", s)
        assert abs(_corr(keep[:-64], keep[64:])) <= bound, ("the same row of the next code", s)
        # kept counts per row are Binomial(T, 1 - p): mean and variance over 4096 rows
        cnt = keep.double().sum(1)
        T = keep.shape[1]
        assert abs(float(cnt.mean()) - T * (1 - p)) <= 4 * math.sqrt(T * p * (1 - p) / cnt.numel())
        assert abs(float(cnt.var()) / (T * p * (1 - p)) - 1.0) <= 0.15, float(cnt.var())
        # ... and per key over the rows (no key is favoured)
        col = keep.double().mean(0)
        assert float((col - (1 - p)).abs().max()) <= 5 * math.sqrt(p * (1 - p) / keep.shape[0])
        # kept probabilities are the full softmax's, scaled by 1 / (1 - p)
        err = ((o - full / (1 - p)).abs() * keep).max() / full.max()
        assert float(err) <= 2e-6
    for i in range(len(seeds)):
        for j in range(i + 1, len(seeds)):
            assert abs(_corr(keeps[i], keeps[j])) <= bound, ("seeds", seeds[i], seeds[j])
    # the same seed gives the same mask (the backward kernels rely on it)
    _, again = _masks(dev, p, seeds[:1])
    assert torch.equal(again[0], outs[0])
