"""GPU: the training half of the path through the C ABI -- sparse soft-VQ backward, normalize backward, InfoNCE.

Checkers: the oracle's restatements (pinned to the reference's own gradients in tests/test_oracle_golden.py) and the
golden fixtures directly.  Floats: 1e-5 relative (north_star), 2e-5 for gradients that pass through exp/log twice.
"""
import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


RTOL = 1e-5        # north_star: embeddings / losses within 1e-5 relative fp32


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))


def _case(oracle, n, d, K, k, seed):
    x = synth.det_randn(f"bw.x.{seed}", (n, d), 1.5, seed).numpy()
    W = synth.det_randn(f"bw.W.{seed}", (K, d), 1.0, seed).numpy()
    xhat, xs = oracle.rownorm(x)
    what, ws = oracle.rownorm(W)
    idx, dist = oracle.topk_search(xhat, xs, what, ws, k)
    w, _, _ = oracle.soft_assign(x, what, idx, dist)
    g = {nm: synth.det_randn(f"bw.{nm}.{seed}", (n, d), 1.0, seed).numpy() for nm in ("g_zq", "g_xhat", "g_out")}
    return x, W, xhat, what, idx, w, g


@pytest.mark.parametrize("n,d,K,k,seed", [(37, 64, 96, 5, 1), (64, 768, 300, 5, 2), (5, 4, 7, 1, 3), (130, 1540, 64, 8, 4)])
def test_soft_vq_backward_matches_oracle(oracle, dev, n, d, K, k, seed):
    from medtok_amd import ops
    x, W, xhat, what, idx, w, g = _case(oracle, n, d, K, k, seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vs, cs = 2.0 / (n * d), 0.5 / (n * d)
    for use in (("g_zq", "g_xhat", "g_out"), ("g_out",), ("g_zq",), ()):
        kw = {nm: g[nm] for nm in use}
        gx_o, gc_o = oracle.soft_vq_backward(x, xhat, what, idx, w, g_vq=0.7, g_commit=-1.3, vq_scale=vs, commit_scale=cs, **kw)
        gx, gc = ops.soft_vq_backward(T(x), T(xhat), T(what), T(idx), T(w), g_vq=torch.tensor(0.7, device=dev),
                                      g_commit=torch.tensor(-1.3, device=dev), vq_scale=vs, commit_scale=cs,
                                      **{nm: T(v) for nm, v in kw.items()})
        assert rel(gx, gx_o) <= 1e-5, use
        assert rel(gc, gc_o) <= 1e-5, use
    # no scalar gradients at all, only one output wanted
    gx_o, gc_o = oracle.soft_vq_backward(x, xhat, what, idx, w, g_zq=g["g_zq"])
    gx, none = ops.soft_vq_backward(T(x), T(xhat), T(what), T(idx), T(w), g_zq=T(g["g_zq"]), want_g_code=False)
    assert none is None and rel(gx, gx_o) <= 1e-5
    none, gc = ops.soft_vq_backward(T(x), T(xhat), T(what), T(idx), T(w), g_zq=T(g["g_zq"]), want_gx=False)
    assert none is None and rel(gc, gc_o) <= 1e-5
    # normalize backward
    gW = ops.normalize_backward(T(g["g_zq"][:min(n, K)]), T(what[:min(n, K)]), T(W[:min(n, K)]))
    assert rel(gW, oracle.normalize_backward(g["g_zq"][:min(n, K)], what[:min(n, K)], W[:min(n, K)])) <= 1e-5


def test_code_gradient_is_deterministic_and_sparse(oracle, dev):
    """The per-code sum runs through the stable sort + segmented sum: bit-identical across runs, zero off the selected codes."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    D, n_e, N = 64, 600, 512
    v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D]).to(dev).train()
    x = synth.det_randn("det.x", (N, D), 1.0, 3).to(dev)
    grads = []
    for _ in range(3):
        v.zero_grad()
        zq, (vq, cm, xhat, _), _ = v.specific_embedding(x.clone().requires_grad_(True), types="graph")
        (vq + cm + zq.square().mean()).backward()
        grads.append(v.codebook.weight.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    touched = grads[0].abs().sum(1) > 0
    assert not touched[: n_e - n_e // 3].any() and 0 < int(touched.sum()) <= N * 5


@pytest.mark.parametrize("name", ["f11_info_nce", "f11_info_nce_wide"])
def test_info_nce_matches_reference_fixture(golden, dev, name):
    from medtok_amd import loss as L
    g = golden(name)
    q = torch.from_numpy(g["q"]).to(dev).requires_grad_(True)
    k = torch.from_numpy(g["k"]).to(dev).requires_grad_(True)
    loss = L.info_nce_loss(q, k, float(g["temperature"]))
    (loss * float(g["upstream"])).backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert rel(q.grad, g["grad_q"]) <= 2e-5 and rel(k.grad, g["grad_k"]) <= 2e-5


@pytest.mark.parametrize("b,d,temp", [(256, 1536, 0.07), (256, 768, 0.07), (3, 4, 1.0), (100, 130, 0.5), (1, 64, 0.07)])
def test_info_nce_matches_oracle(oracle, dev, b, d, temp):
    from medtok_amd import loss as L
    q = synth.det_randn(f"nce.q.{b}.{d}", (b, d), 1.0, 5)
    k = (0.5 * q + synth.det_randn(f"nce.k.{b}.{d}", (b, d), 1.0, 6)).contiguous()
    lo, gq_o, gk_o = oracle.info_nce(q.numpy(), k.numpy(), temp, 0.9)
    qd, kd = q.to(dev).requires_grad_(True), k.to(dev).requires_grad_(True)
    loss = L.info_nce_loss(qd, kd, temp)
    (loss * 0.9).backward()
    assert abs(float(loss) - lo) <= 1e-5 * max(abs(lo), 1e-3)
    if b > 1:
        assert rel(qd.grad, gq_o) <= 2e-5 and rel(kd.grad, gk_o) <= 2e-5
    else:
        assert float(qd.grad.abs().max()) <= 1e-6       # one row: the loss is identically zero


def test_losses_match_reference_fixture_on_device(golden, dev):
    """loss.py surface against fixture F7 (values and gradients of the assembled total, train_MedTok.py:215-238)."""
    from medtok_amd import loss as L
    g = golden("f7_losses")
    t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("z1", "z2", "x1", "x2", "z1_aug", "z2_aug", "z1_c", "z2_c")}
    s = L.shared_loss(t["z1_c"], t["z2_c"], t["x1"], t["x2"])
    p = L.specific_loss(t["z1"], t["z1_aug"], t["z2"], t["z2_aug"], t["z1_c"], t["z2_c"])
    assert rel(torch.stack(s), g["shared"]) <= 1e-5
    assert rel(torch.stack(p), g["specific"]) <= 1e-5
    total = float(g["codebook_loss"]) + (s[0] - 0.1 * s[1]) + (s[2] - 0.1 * s[3]) + (p[0] + 0.1 * p[1]) + (p[2] + 0.1 * p[3])
    assert abs(float(total) - float(g["total"])) <= 1e-5 * abs(float(g["total"]))
    total.backward()
    for k, v in t.items():
        assert rel(v.grad, g[f"grad.{k}"]) <= 2e-5, k
    assert abs(float(L.info_nce_loss(t["z1"], t["z2"])) - float(g["nce_z1_z2"])) <= 1e-5 * float(g["nce_z1_z2"])


def test_total_loss_assembly_on_device(golden, dev):
    from medtok_amd import loss as L
    g = golden("f7_losses")
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("z1", "z2", "x1", "x2", "z1_aug", "z2_aug", "z1_c", "z2_c")}
    c = torch.tensor(float(g["codebook_loss"]) / 6, device=dev)
    r = {"shared_embed_loss": (c, c), "text_specific_loss": (c, c), "graph_specific_loss": (c, c),
         "shared_text_embedding": t["z1_c"], "shared_graph_embedding": t["z2_c"], "text_feature": t["x1"], "graph_feature": t["x2"],
         "specific_embedding_text": t["z1"], "specific_embedding_text_aug": t["z1_aug"],
         "specific_embedding_graph": t["z2"], "specific_embedding_graph_aug": t["z2_aug"]}
    loss, parts = L.total_loss(r)
    assert abs(float(loss) - float(g["total"])) <= 1e-5 * abs(float(g["total"]))
    assert set(parts) >= {"codebook_loss", "shared_loss", "specific_loss"}


@pytest.mark.parametrize("b,d1,d2", [(16, 64, 64), (256, 768, 768), (37, 100, 36), (5, 4, 8), (77, 132, 68), (200, 96, 260), (64, 33, 31)])
def test_alignment_and_orthogonal_kernels_match_oracle(oracle, dev, b, d1, d2):
    """alignment_loss / orthogonal_loss (loss.py:59-83) on the gfx950 kernels vs the C oracle: row dots and every entry of
    z^T z* bit-identical (one fmaf chain per entry), the Frobenius norm to fp32 round-off; all three GEMM orientations the
    backward uses."""
    from medtok_amd import ops
    rng = np.random.default_rng(b + d1)
    z = rng.standard_normal((b, d1), dtype=np.float32); zs = rng.standard_normal((b, d2), dtype=np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    if d1 == d2:
        assert np.array_equal(ops.row_dot(T(z), T(zs)).cpu().numpy(), oracle.row_dot(z, zs))
    m = ops.small_gemm(T(z), T(zs), trans_a=True)
    m_o = oracle.small_gemm(z, zs, trans_a=True)
    assert np.array_equal(m.cpu().numpy(), m_o)
    assert np.array_equal(ops.small_gemm(T(zs), m, trans_b=True).cpu().numpy(), oracle.small_gemm(zs, m_o, trans_b=True))
    assert np.array_equal(ops.small_gemm(T(z), m).cpu().numpy(), oracle.small_gemm(z, m_o))
    if d2 % 4 == 0:
        f, f_o = float(ops.frobenius(m)), float(oracle.frobenius(m_o))
        assert abs(f - f_o) <= 1e-6 * f_o
        assert abs(f_o - np.linalg.norm(z.astype(np.float64).T @ zs.astype(np.float64))) <= 1e-5 * f_o


def test_alignment_and_orthogonal_losses_match_reference_fixture(golden, dev):
    """Values and autograd gradients of the two regularisers alone, against the reference's (F7)."""
    from medtok_amd import loss as L
    g = golden("f7_losses")
    T = lambda k: torch.from_numpy(g[k]).to(dev).requires_grad_(True)
    x1, x2 = T("x1"), T("x2")
    al = L.alignment_loss(x1, x2)
    assert abs(float(al) - float(g["align"])) <= RTOL * abs(float(g["align"]))
    (al * 3.0).backward()
    assert float((x1.grad - 3.0 * x2.detach() / x1.shape[0]).abs().max()) <= 1e-6
    assert float((x2.grad - 3.0 * x1.detach() / x1.shape[0]).abs().max()) <= 1e-6
    z, zc = T("z1"), T("z1_c")
    orth = L.orthogonal_loss(z, zc)
    assert abs(float(orth) - float(g["orth"])) <= RTOL * float(g["orth"])
    orth.backward()
    zr, zcr = torch.from_numpy(g["z1"]).double().requires_grad_(True), torch.from_numpy(g["z1_c"]).double().requires_grad_(True)
    torch.linalg.matrix_norm(zr.t() @ zcr).backward()
    assert float((z.grad.cpu().double() - zr.grad).abs().max()) <= RTOL * float(zr.grad.abs().max())
    assert float((zc.grad.cpu().double() - zcr.grad).abs().max()) <= RTOL * float(zcr.grad.abs().max())


def test_normalize_backward_skips_rows_known_to_be_zero(dev):
    """ops.normalize_backward(live=...): rows whose `live` entry is 0 hold an all-zero gradient and are written as zeros without being
    read -- bit for bit the full kernel's output (the codebook's gradient through F.normalize: a step touches a few thousand of 49 152 codes)."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(0)
    n, d = 5000, 192
    v = torch.randn(n, d, device=dev, generator=g)
    v[7] = 0.0                                                       # (a zero row: |v| clamps to 1e-12)
    vhat, _ = ops.rownorm(v)
    live = (torch.rand(n, device=dev, generator=g) < 0.1).float() * torch.randint(1, 5, (n,), device=dev, generator=g).float()
    grad = torch.randn(n, d, device=dev, generator=g) * (live > 0).float()[:, None]
    full = ops.normalize_backward(grad, vhat, v)
    sparse = ops.normalize_backward(grad, vhat, v, live=live)
    assert torch.equal(full, sparse)
    assert not sparse[live == 0].any() and sparse[live > 0].abs().sum() > 0
    with pytest.raises(ValueError):
        ops.normalize_backward(grad, vhat, v, live=live[:-1])
