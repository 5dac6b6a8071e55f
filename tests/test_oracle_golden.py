"""CPU: pins the oracle (oracle/medtok_oracle.c) to golden vectors produced by the reference itself.

Tolerances: ids bit-exact (every fixture row has an fp64 top-(k+1) gap well above fp32
round-off, asserted below); floats 1e-5 relative, the bar BASELINE.json:north_star states.
"""
import numpy as np
import pytest

from oracle import synth

RTOL = 1e-5


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", ["f1_specific_d64", "f2_specific_d768", "f20_specific_k12", "f21_specific_k16_d768"])
def test_specific_embedding(oracle, golden, name):
    g = golden(name)
    n_e, D, k, seed = int(g["n_e"]), int(g["e_dim"]), int(g["k"]), int(g["seed"])
    W = synth.det_randn(f"{name}.codebook.weight", (n_e, D), 1.0, seed).numpy()
    region = n_e // 3
    for t, Wr in (("text", W[:region]), ("graph", W[-region:])):
        xp = g[f"{t}.x_proj"]
        r = oracle.specific_search(xp, Wr, k)
        assert np.diff(g[f"{t}.gap64"], axis=1).min() > 1e-6, "fixture has a near-tie; regenerate with another seed"
        assert np.array_equal(r["idx"], g[f"{t}.idx"])
        assert rel(r["dist"], g[f"{t}.dist"]) <= RTOL
        assert rel(r["w"], g[f"{t}.w"]) <= RTOL
        assert rel(r["xhat"], g[f"{t}.eval.xhat"]) <= RTOL
        assert rel(r["zq"], g[f"{t}.eval.zq"]) <= RTOL
        vq = r["row_sqerr"].astype(np.float64).sum() / xp.size
        assert abs(vq - g[f"{t}.train.vq"]) <= RTOL * g[f"{t}.train.vq"]
        assert abs(float(g["beta"]) * vq - g[f"{t}.train.commit"]) <= RTOL * g[f"{t}.train.commit"]
        assert float(g[f"{t}.eval.vq"]) == 0.0      # eval returns tensor(0.0) (:210-212)


@pytest.mark.parametrize("name", ["f5_normema_d32", "f5_normema_d768", "f6_normema_zero_usage"])
def test_norm_ema_steps(oracle, golden, name):
    g = golden(name)
    E = g["E0"].copy()
    cs = np.zeros(E.shape[0], np.float32)
    beta, decay = float(g["beta"]), float(g["decay"])
    for s in range(int(g["steps"])):
        assert np.diff(g[f"s{s}.gap64"], axis=1).min() > 1e-6
        zq, loss, idx = oracle.norm_ema_forward(g[f"s{s}.z"], E, cs, beta, decay, True)
        assert np.array_equal(idx, g[f"s{s}.idx"])
        assert rel(zq, g[f"s{s}.zq"]) <= RTOL
        assert abs(loss - g[f"s{s}.loss"]) <= RTOL * g[f"s{s}.loss"]
        assert rel(E, g[f"s{s}.E"]) <= RTOL
        assert rel(cs, g[f"s{s}.cluster_size"]) <= RTOL
    if name == "f6_normema_zero_usage":
        assert (cs == 0).sum() > 50              # the zero_mask branch really ran
    zq, loss, idx = oracle.norm_ema_forward(g["eval.z"], E, cs, beta, decay, False)
    assert np.array_equal(idx, g["eval.idx"])
    assert rel(E, g["eval.E"]) <= RTOL          # eval leaves the codebook alone
    assert rel(cs, g["eval.cluster_size"]) <= RTOL


def test_usage_window(oracle, golden):
    g = golden("f10_usage")
    win = np.zeros(int(g["window"]), np.float32)
    for c in range(4):
        assert oracle.usage_update(win, g[f"c{c}.ids"], int(g["n_e"])) == float(g[f"c{c}.usage"])
    assert np.array_equal(win[-4000:], g["final_tail"])


def test_tie_rule(oracle, golden):
    g = golden("f8_ties")
    xh, xs = oracle.rownorm(g["x"]); wh, ws = oracle.rownorm(g["W"])
    idx, dist = oracle.topk_search(xh, xs, wh, ws, 5)
    rule = g["build_rule_idx"]
    for r in range(rule.shape[0]):
        want = [v for v in rule[r] if v >= 0]
        assert list(idx[r, :len(want)]) == want
    assert np.array_equal(idx[:, 0], g["torch_argmin"])     # argmin's first-index rule agrees


def test_search_matches_full_matrix_and_fp64(oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((40, 96), dtype=np.float32); W = rng.standard_normal((333, 96), dtype=np.float32)
    xh, xs = oracle.rownorm(x); wh, ws = oracle.rownorm(W)
    idx, dist = oracle.topk_search(xh, xs, wh, ws, 7)
    full = oracle.distance(xh, xs, wh, ws)
    order = np.lexsort((np.broadcast_to(np.arange(333), full.shape), full), axis=1)[:, :7]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.take_along_axis(full, order, 1))
    d64 = (xh.astype(np.float64) ** 2).sum(1, keepdims=True) + (wh.astype(np.float64) ** 2).sum(1) - 2 * xh.astype(np.float64) @ wh.astype(np.float64).T
    assert np.abs(full - d64).max() < 5e-6


def test_edge_cases(oracle):
    # zero row -> eps clamp, finite output; k == K; n == 0
    x = np.zeros((2, 8), np.float32); x[1, 0] = 3.0
    xh, xs = oracle.rownorm(x)
    assert np.all(xh[0] == 0) and xs[0] == 0 and abs(xs[1] - 1) < 1e-6
    W = np.eye(8, dtype=np.float32)[:5]
    idx, dist = oracle.topk_search(xh, xs, W, np.ones(5, np.float32), 5)
    assert sorted(idx[1]) == [0, 1, 2, 3, 4] and idx[1, 0] == 0
    assert list(idx[0]) == [0, 1, 2, 3, 4]     # all-equal distances: lowest index first
    idx, dist = oracle.topk_search(xh[:0], xs[:0], W, np.ones(5, np.float32), 3)
    assert idx.shape == (0, 3)


# ---- the torch op-sequence port used as bench.py's cpu_baseline is pinned to the same vectors
@pytest.mark.parametrize("name", ["f1_specific_d64", "f2_specific_d768", "f20_specific_k12", "f21_specific_k16_d768"])
def test_torch_port_specific(golden, name):
    import torch
    from oracle import torch_port as P
    g = golden(name)
    n_e, D, k, seed = int(g["n_e"]), int(g["e_dim"]), int(g["k"]), int(g["seed"])
    W = synth.det_randn(f"{name}.codebook.weight", (n_e, D), 1.0, seed)
    region = n_e // 3
    for t, Wr in (("text", W[:region]), ("graph", W[-region:])):
        r = P.soft_search(torch.from_numpy(g[f"{t}.x_proj"]), Wr, k)
        assert np.array_equal(r["idx"].numpy(), g[f"{t}.idx"])
        assert rel(r["w"].numpy(), g[f"{t}.w"]) <= RTOL
        assert rel(r["zq"].numpy(), g[f"{t}.eval.zq"]) <= RTOL
        assert abs(float(r["mse"]) - g[f"{t}.train.vq"]) <= RTOL * g[f"{t}.train.vq"]


@pytest.mark.parametrize("name", ["f5_normema_d32", "f6_normema_zero_usage"])
def test_torch_port_norm_ema(golden, name):
    import torch
    from oracle import torch_port as P
    g = golden(name)
    E = torch.from_numpy(g["E0"].copy()); cs = torch.zeros(E.shape[0])
    for s in range(int(g["steps"])):
        zq, loss, idx = P.norm_ema_forward(torch.from_numpy(g[f"s{s}.z"]), E, cs, float(g["beta"]), float(g["decay"]), True)
        assert np.array_equal(idx.numpy(), g[f"s{s}.idx"])
        assert rel(E.numpy(), g[f"s{s}.E"]) <= RTOL and rel(cs.numpy(), g[f"s{s}.cluster_size"]) <= RTOL
        assert abs(float(loss) - g[f"s{s}.loss"]) <= RTOL * g[f"s{s}.loss"]


@pytest.mark.parametrize("name", ["f1_specific_d64", "f2_specific_d768", "f20_specific_k12", "f21_specific_k16_d768"])
def test_soft_backward_matches_reference_gradients(oracle, golden, name):
    """oracle.soft_vq_backward + segment sum + normalize_backward == the gradients the reference's autograd
    produced for loss = vq + commit + (zq_ste * probe).sum() / N (oracle/gen_golden.py::fixture_specific)."""
    g = golden(name)
    n_e, D, k, seed, beta = int(g["n_e"]), int(g["e_dim"]), int(g["k"]), int(g["seed"]), float(g["beta"])
    W = synth.det_randn(f"{name}.codebook.weight", (n_e, D), 1.0, seed).numpy()
    region = n_e // 3
    N = g["x"].shape[0]
    probe = synth.det_randn(name + ".probe", (N, D), 1.0, seed).numpy()
    for t, lo in (("text", 0), ("graph", n_e - region)):
        Wr = W[lo:lo + region]
        Wp = synth.det_randn(f"{name}.proj_{t}.weight", (D, D), 1.0 / D ** 0.5, seed).numpy()
        xp = g[f"{t}.x_proj"]
        xhat, _ = oracle.rownorm(xp)
        what, _ = oracle.rownorm(Wr)
        gx, gc = oracle.soft_vq_backward(xp, xhat, what, g[f"{t}.idx"], g[f"{t}.w"], g_out=probe / N, g_vq=1.0, g_commit=1.0,
                                         vq_scale=2.0 / (N * D), commit_scale=2.0 * beta / (N * D))
        assert rel(gx.astype(np.float64).sum(0), g[f"{t}.train.grad_proj_b"]) <= RTOL
        assert rel(gx.astype(np.float64) @ Wp.astype(np.float64), g[f"{t}.train.grad_x"]) <= RTOL
        _, g_what = oracle.ema_stats(gc, g[f"{t}.idx"].reshape(-1), region)
        gW = oracle.normalize_backward(g_what, what, Wr)
        ref = g[f"{t}.train.grad_codebook"]
        assert rel(gW, ref[lo:lo + region]) <= RTOL          # (measured 2e-7 .. 7e-7 on F1 / F2)
        outside = np.ones(n_e, bool); outside[lo:lo + region] = False
        assert not ref[outside].any()            # codes outside the searched region get no gradient


@pytest.mark.parametrize("name", ["f11_info_nce", "f11_info_nce_wide"])
def test_info_nce_matches_reference(oracle, golden, name):
    g = golden(name)
    loss, gq, gk = oracle.info_nce(g["q"], g["k"], float(g["temperature"]), float(g["upstream"]))
    assert abs(loss - float(g["loss"])) <= RTOL * abs(float(g["loss"]))
    assert rel(gq, g["grad_q"]) <= 2e-5
    assert rel(gk, g["grad_k"]) <= 2e-5


def test_oracle_attention_backward_matches_torch_autograd(oracle):
    import torch
    """The oracle's training-mode attention core (dropout by the stateless hash mask) and its backward against torch autograd of the
    dense formula.  The mask is data-independent, so it is read off the oracle itself with one-hot keys and zero queries
    (uniform probabilities: out[r, j] = M[r, j] / ((1 - p) n_keys))."""
    rng = np.random.default_rng(3)
    d, p, seed, scale = 16, 0.25, 1234, 0.4
    q_len = np.array([5, 0, 7], np.int64); kv_len = np.array([6, 3, 9], np.int64)
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
    nq, nk = int(q_len.sum()), int(kv_len.sum())
    # the mask
    kv_hot = np.zeros((nk, d), np.float32)
    for b in range(3):
        for j in range(kv_len[b]):
            kv_hot[kv_start[b] + j, j] = 1.0
    out_hot, _ = oracle.shared_kv_attention_train(np.zeros((nq, d), np.float32), q_start, q_len, kv_hot, kv_start, kv_len, scale, p, seed)
    q = rng.standard_normal((nq, d)).astype(np.float32); kv = rng.standard_normal((nk, d)).astype(np.float32)
    d_out = rng.standard_normal((nq, d)).astype(np.float32)
    out, lse, dq, dkv = oracle.shared_kv_attention_train(q, q_start, q_len, kv, kv_start, kv_len, scale, p, seed, d_out)
    qt, kt = torch.from_numpy(q).double().requires_grad_(True), torch.from_numpy(kv).double().requires_grad_(True)
    outs = []
    kept = 0
    for b in range(3):
        if q_len[b] == 0:
            continue
        qs, ks, n = int(q_start[b]), int(kv_start[b]), int(kv_len[b])
        mask = torch.from_numpy(out_hot[qs: qs + q_len[b], :n] > 0).double() / (1.0 - np.float32(p))
        kept += int((mask > 0).sum())
        s = scale * qt[qs: qs + q_len[b]] @ kt[ks: ks + n].t()
        outs.append((torch.softmax(s, -1) * mask) @ kt[ks: ks + n])
        assert np.allclose(lse[qs: qs + q_len[b]], torch.logsumexp(s, -1).detach().numpy(), rtol=1e-5, atol=1e-6)
    ref = torch.cat(outs)
    assert 0.6 < kept / float((q_len * kv_len).sum()) < 0.9            # P(keep) = 0.75
    ref.backward(torch.from_numpy(d_out).double())
    assert np.abs(out - ref.detach().numpy()).max() <= 1e-5
    assert np.abs(dq - qt.grad.numpy()).max() <= 1e-5 * np.abs(qt.grad.numpy()).max()
    assert np.abs(dkv - kt.grad.numpy()).max() <= 1e-5 * np.abs(kt.grad.numpy()).max()


def test_oracle_layer_tail_and_node_mean_match_torch(oracle):
    """The restatements of CrossAttentionLayer's residual + LayerNorm (vector_quantization_soft_one_new.py:47-50) and of the
    node mean (:140-141) against torch's own ops on the CPU (summation orders differ: 1e-6)."""
    import torch
    rng = np.random.default_rng(11)
    for n, d in ((37, 768), (5, 64), (3, 12)):
        a = rng.standard_normal((n, d), dtype=np.float32) * 2
        b = rng.standard_normal((n, d), dtype=np.float32)
        ln = torch.nn.LayerNorm(d)
        with torch.no_grad():
            ln.weight.copy_(torch.from_numpy(rng.standard_normal(d, dtype=np.float32)))
            ln.bias.copy_(torch.from_numpy(rng.standard_normal(d, dtype=np.float32)))
            ref = ln(torch.from_numpy(a) + torch.from_numpy(b)).numpy()
        got = oracle.residual_layernorm(a, b, ln.weight.detach().numpy(), ln.bias.detach().numpy(), ln.eps)
        assert np.abs(got - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0) * 4
    x = rng.standard_normal((50, 64), dtype=np.float32)
    start, length = np.array([0, 7, 7, 30], np.int64), np.array([7, 0, 23, 20], np.int64)
    got = oracle.segment_mean(x, start, length)
    for b in range(4):
        ref = x[start[b]:start[b] + length[b]].astype(np.float64).mean(0) if length[b] else np.zeros(64)
        assert np.abs(got[b] - ref).max() <= 1e-6


TAU = 1e-5      # ids must be identical wherever the fp64 gap between neighbouring top-(k+1) distances exceeds this


def census_check(idx_got, dist_got, idx_ref, gapmin, near_rows, near_d64, rows, what):
    """The tau protocol on the rows `rows` (array of row numbers): ids identical outside tau; inside tau the distances must
    still be the reference's (each of the k sorted distances within tau of the fp64 list).  Returns (rows inside tau, of which
    ids differ)."""
    inside = gapmin[rows] <= TAU
    bad = [int(r) for r, got, ref, ins in zip(rows, idx_got, idx_ref[rows], inside) if not ins and not np.array_equal(got, ref)]
    assert not bad, f"{what}: ids differ from the reference OUTSIDE tau on rows {bad[:10]}"
    where = {int(r): i for i, r in enumerate(near_rows)}
    differ = 0
    for j, r in enumerate(rows):
        if inside[j]:
            d64 = near_d64[where[int(r)]][: idx_got.shape[1]]
            assert np.abs(dist_got[j].astype(np.float64) - d64).max() <= TAU, f"{what}: row {int(r)} inside tau picked a code that is not a near-tie"
            differ += int(not np.array_equal(idx_got[j], idx_ref[int(r)]))
    return int(inside.sum()), differ


def test_cfg3_slice_oracle_vs_reference(oracle, golden):
    """F14: the reference's VectorQuantizer at n_e = 49152, D = 768, k = 5 (BASELINE config 3's codebook).  The C oracle on the
    first rows of each of the four searches; the GPU test runs all 16 384."""
    g = golden("f14_cfg3_slice")
    n_e, D, k, seed, N = int(g["n_e"]), int(g["e_dim"]), int(g["k"]), int(g["seed"]), int(g["N"])
    name = "f14_cfg3_slice"
    W = synth.det_randn(f"{name}.codebook.weight", (n_e, D), 1.0, seed).numpy()
    region = n_e // 3
    h = synth.det_randn(name + ".h", (N, 2 * D), 1.0, seed)[:48]
    import torch
    lin = {t: (synth.det_randn(f"{name}.proj_{t}.weight", (D, D), 1.0 / D ** 0.5, seed), synth.det_randn(f"{name}.proj_{t}.bias", (D,), 0.02, seed))
           for t in ("text", "graph")}
    xs = {"text": torch.nn.functional.linear(h[:, :D], *lin["text"]).numpy(), "graph": torch.nn.functional.linear(h[:, D:], *lin["graph"]).numpy(),
          "shared_text": synth.det_randn(name + ".pt", (N, D), 1.0, seed)[:24].numpy(),
          "shared_graph": synth.det_randn(name + ".pg", (N, D), 1.0, seed)[:24].numpy()}
    for key, x in xs.items():
        Wr = W[:region] if key == "text" else (W[-region:] if key == "graph" else W)
        r = oracle.specific_search(np.ascontiguousarray(x), Wr, k)
        rows = np.arange(x.shape[0])
        census_check(r["idx"], r["dist"], g[f"{key}.idx"].astype(np.int64), g[f"{key}.gapmin"], g[f"{key}.near_rows"], g[f"{key}.near_d64"], rows, key)
        assert rel(r["w"], g[f"{key}.w_head"][: x.shape[0]]) <= 1e-4       # softmax weights move by the distances' round-off (a few 1e-7 of d ~ 2)


def test_cfg2_slice_oracle_vs_reference(oracle, golden):
    """F15: one train step of the reference's NormEMAVectorQuantizer at K = 8192, D = 768 on 16 384 rows.  The oracle's ids on
    the first rows; the whole step (ids -> exact counts -> EMA codebook) is checked on the GPU."""
    g = golden("f15_cfg2_slice")
    K, D, N, seed = int(g["K"]), int(g["D"]), int(g["N"]), int(g["seed"])
    name = "f15_cfg2_slice"
    E0 = oracle.rownorm(synth.det_randn(name + ".E", (K, D), 1.0, seed).numpy())[0]
    z = synth.det_randn(name + ".z", (N, D), 1.0, seed)[:128].numpy()
    zh, zs = oracle.rownorm(z)
    _, es = oracle.rownorm(E0, False)
    idx, dist = oracle.topk_search(zh, zs, E0, es, 1)
    census_check(idx, dist, g["idx"].astype(np.int64)[:, None], g["gapmin"], g["near_rows"], g["near_d64"], np.arange(128), "cfg2")
