"""CPU: host-side mirror of the reference interfaces (no kernel launches)."""
import numpy as np
import pytest
import torch

from oracle import synth


def rel(a, b):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_state_dict_keys_match_reference(golden):
    from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    q = NormEMAVectorQuantizer(64, 32, 0.25)
    assert sorted(q.state_dict().keys()) == list(golden("f5_normema_d32")["state_dict_keys"])
    v = VectorQuantizer(96, 16, 0.25, 0.0, True, True, [16, 16])
    # keys, order and shapes as read off the reference's own module (fixture F16, oracle/gen_golden.py: fixture_state_dict_keys)
    ref = golden("f16_soft_state_dict")
    sd = v.state_dict()
    assert list(sd.keys()) == [str(k) for k in ref["keys"]]
    assert [",".join(str(int(n)) for n in t.shape) for t in sd.values()] == [str(x) for x in ref["shapes"]]
    assert v.state_dict()["codebook.weight"].shape == (96, 16)
    assert v.state_dict()["codebook_used"].shape == (300000,)
    assert torch.allclose(q.embedding.weight.norm(dim=-1), torch.ones(64), atol=1e-6)   # l2norm(randn) init (:69-70)
    assert float(q.embedding.initted) == 1.0 and q.embedding.update is True


def test_constructor_signatures_match_reference():
    import inspect
    from medtok_amd.norm_ema_quantizer import EmbeddingEMA, NormEMAVectorQuantizer
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    assert list(inspect.signature(VectorQuantizer.__init__).parameters)[1:] == [
        "n_e", "e_dim", "beta", "entropy_loss_ratio", "l2_norm", "show_usage", "split", "kmeans", "num_head", "k"]
    assert list(inspect.signature(NormEMAVectorQuantizer.__init__).parameters)[1:] == [
        "n_embed", "embedding_dim", "beta", "decay", "eps", "statistic_code_usage", "kmeans_init", "codebook_init_path"]
    assert list(inspect.signature(EmbeddingEMA.__init__).parameters)[1:] == [
        "num_tokens", "codebook_dim", "decay", "eps", "kmeans_init", "codebook_init_path"]
    assert list(inspect.signature(VectorQuantizer.forward).parameters)[1:] == [
        "z", "text_features", "graph_node_features", "text_attention_mask", "batch", "z_aug"]


def test_losses_refuse_cpu_tensors_and_oracle_matches_reference_fixture(golden, oracle):
    """Every term of loss.py runs on HIP kernels (checked on the GPU, tests/test_gpu_train_kernels.py): on CPU tensors they must
    refuse loudly.  The oracle's restatements of the alignment / orthogonality terms are pinned to the reference's values here."""
    from medtok_amd import loss as L
    from medtok_amd._lib import MedTokLibraryError
    g = golden("f7_losses")
    t = {k: torch.from_numpy(g[k]) for k in ("z1", "z2", "x1", "x2", "z1_c")}
    for fn, args in ((L.alignment_loss, ("x1", "x2")), (L.orthogonal_loss, ("z1", "z1_c")), (L.info_nce_loss, ("z1", "z2"))):
        with pytest.raises(MedTokLibraryError):
            fn(*[t[a] for a in args])
    align = float(oracle.row_dot(g["x1"], g["x2"]).astype(np.float64).mean())
    assert abs(align - float(g["align"])) <= 1e-5 * abs(float(g["align"]))
    m = oracle.small_gemm(g["z1"], g["z1_c"], trans_a=True)
    assert np.abs(m - g["z1"].astype(np.float64).T @ g["z1_c"].astype(np.float64)).max() <= 1e-5 * np.abs(m).max()
    assert abs(float(oracle.frobenius(m)) - float(g["orth"])) <= 1e-5 * float(g["orth"])


@pytest.mark.parametrize("name", ["f3_forward_d64", "f4_forward_d128", "f19_forward_b64"])
def test_batched_cross_attention_equals_reference_loop(golden, name):
    """CrossAttention.pooled (batched) vs the reference's per-code loop outputs (fixture)."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    g = golden(name)
    D, n_e = int(g["e_dim"]), int(g["n_e"])
    v = VectorQuantizer(n_e, D, 0.25, 0.0, True, True, [D, D])
    v.load_state_dict(synth.det_state_dict(v, name, int(g["seed"])), strict=True)
    v.eval()
    with torch.no_grad():
        pt, pg = v.cross_attn.pooled_reference(torch.from_numpy(g["text"]), torch.from_numpy(g["mask"]),
                                     torch.from_numpy(g["nodes"]), torch.from_numpy(g["batch"]))
    assert rel(pt, g["pooled_text"]) <= 1e-5
    assert rel(pg, g["pooled_graph"]) <= 1e-5
    # shuffled node order must not matter (nodes of a code need not be contiguous)
    perm = torch.randperm(g["nodes"].shape[0], generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        pt2, pg2 = v.cross_attn.pooled_reference(torch.from_numpy(g["text"]), torch.from_numpy(g["mask"]),
                                       torch.from_numpy(g["nodes"])[perm], torch.from_numpy(g["batch"])[perm])
    assert rel(pt2, g["pooled_text"]) <= 1e-5 and rel(pg2, g["pooled_graph"]) <= 1e-5
    # packed inference path (D % 128 == 0 only) with the oracle's restatement of the ragged attention core injected:
    # pins that restatement, and the packing logic around it, to the reference's per-code loop
    if D % 128 == 0 or D == 64:
        from oracle import oracle as O

        def core(q, qs, ql, kv, ks, kl, max_q_len, scale):
            assert int(ql.max()) <= max_q_len
            return torch.from_numpy(O.shared_kv_attention(q.numpy(), qs.numpy(), ql.numpy(), kv.numpy(), ks.numpy(), kl.numpy(), scale))
        with torch.no_grad():
            pt4, pg4 = v.cross_attn.pooled_reference(torch.from_numpy(g["text"]), torch.from_numpy(g["mask"]),
                                           torch.from_numpy(g["nodes"])[perm], torch.from_numpy(g["batch"])[perm], fold=True, core=core)
        assert rel(pt4, g["pooled_text"]) <= 1e-5 and rel(pg4, g["pooled_graph"]) <= 1e-5
    # both forms of the graph side (projected keys / projections folded into the queries) are the same function
    for fold in (False, True):
        with torch.no_grad():
            pt3, pg3 = v.cross_attn.pooled_reference(torch.from_numpy(g["text"]), torch.from_numpy(g["mask"]),
                                           torch.from_numpy(g["nodes"]), torch.from_numpy(g["batch"]), fold=fold)
        assert rel(pt3, g["pooled_text"]) <= 1e-5 and rel(pg3, g["pooled_graph"]) <= 1e-5, fold


def test_regions_match_reference_slicing():
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    v = VectorQuantizer(21000, 64, 0.25, 0.0, True, False, [64, 64])
    assert v._region("text") == (0, 7000) and v._region("graph") == (14000, 21000) and v._region("shared") == (0, 21000)
    v = VectorQuantizer(8192, 64, 0.25, 0.0, True, False, [64, 64])
    assert v._region("text") == (0, 2730) and v._region("graph") == (8192 - 2730, 8192)
    assert torch.equal(v.global_token_ids(torch.tensor([0, 5]), "graph"), torch.tensor([5462, 5467]))


def test_no_cpu_fallback():
    from medtok_amd import _lib
    from medtok_amd.norm_ema_quantizer import NormEMAVectorQuantizer
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    v = VectorQuantizer(96, 16, 0.25, 0.0, True, True, [16, 16]).eval()
    with pytest.raises(_lib.MedTokLibraryError):
        v.specific_embedding(torch.randn(4, 16), "text")
    q = NormEMAVectorQuantizer(64, 32, 0.25)
    with pytest.raises(_lib.MedTokLibraryError):
        q(torch.randn(8, 32, 1, 1))


def test_row_shard_partition():
    from medtok_amd.distributed import row_shard
    for n in (0, 1, 7, 600000, 600001):
        for world in (1, 2, 3, 8):
            parts = [row_shard(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_lookup_tokenizer_surface(tmp_path):
    from medtok_amd.inference import save_outputs
    from medtok_amd.tokenizer import MedTokLookup
    codes = ["E11.9", "I10", "J45.909"]
    emb = np.arange(3 * 8, dtype=np.float32).reshape(3, 8)
    tok = np.arange(3 * 4 * 5, dtype=np.int64).reshape(3, 4, 5)
    w = np.full((3, 4, 5), 0.2, np.float32)
    save_outputs(tmp_path, emb, tok, w)
    lk = MedTokLookup.from_dir(tmp_path, codes, region_offsets=(0, 100, 0, 0))
    assert lk.tokenize("I10").shape == (4, 5) and lk.tokenize("I10")[1, 0] == tok[1, 1, 0] + 100
    ids, ws = lk.encode("E11.9")
    assert ids.shape == (20,) and ws.shape == (20,)
    assert np.array_equal(lk.embed(["J45.909", "E11.9"]), emb[[2, 0]])
    with pytest.raises(KeyError):
        lk.embed("nope")
    assert np.load(tmp_path / "tokens_all.npy").dtype == np.int64


def test_global_mean_pool():
    from medtok_amd.tokenizer import global_mean_pool
    x = torch.tensor([[1.0, 2], [3, 4], [5, 6]])
    out = global_mean_pool(x, torch.tensor([0, 0, 2]), 3)
    assert torch.equal(out, torch.tensor([[2.0, 3], [0, 0], [5, 6]]))


def test_pooled_edge_cases_empty_graph_and_bad_batch():
    """A code without graph nodes (or without a valid token) attends to nothing: finite outputs, every form agrees; a `batch`
    id outside [0, B) is rejected where it enters."""
    from medtok_amd.vector_quantization_soft_one_new import CrossAttention
    from oracle import oracle as O
    torch.manual_seed(0)
    D, B, L = 64, 5, 12
    ca = CrossAttention(D, 4).eval()
    text = torch.randn(B, L, D)
    mask = torch.ones(B, L, dtype=torch.long); mask[1, 3:] = 0; mask[4, :] = 0          # code 4: no valid token at all
    batch = torch.tensor([0, 0, 0, 1, 3, 3, 4])                                          # code 2: no nodes
    nodes = torch.randn(batch.numel(), D)

    def core(q, qs, ql, kv, ks, kl, max_q_len, scale):
        return torch.from_numpy(O.shared_kv_attention(q.numpy(), qs.numpy(), ql.numpy(), kv.numpy(), ks.numpy(), kl.numpy(), scale))
    with torch.no_grad():
        outs = [ca.pooled_reference(text, mask, nodes, batch, fold=f, core=c) for f, c in ((False, None), (True, None), (True, core))]
    for pt, pg in outs:
        assert torch.isfinite(pt).all() and torch.isfinite(pg).all()
        assert float(pg[2].abs().max()) == 0.0                        # mean over no nodes
    for pt, pg in outs[1:]:
        assert rel(pt, outs[0][0].numpy()) <= 1e-5 and rel(pg, outs[0][1].numpy()) <= 1e-5
    with pytest.raises(ValueError, match="batch"):
        ca.pooled_reference(text, mask, nodes, torch.tensor([0, 0, 0, 1, 3, 3, 5]))


def test_bench_parent_spawns_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: the parent starts torch.distributed.run children on 127.0.0.1 and relays
    their exit code; with RCCL it refuses up front when fewer GPUs are visible than ranks (here: none)."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("bench_mod", root / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    seen = {}

    class FakeProc:
        stdout = iter(['noise\n', '{"metric": "x"}\n'])
        def wait(self): return 7

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return FakeProc()
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    monkeypatch.setenv("MEDTOK_DIST_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.spawn_ranks(bench.parse())
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("MEDTOK_DIST_BACKEND")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: type("R", (), {"stdout": "1\n"})())     # the device count comes from a child
    assert bench.spawn_ranks(bench.parse()) == 2          # RCCL: one GPU per rank, checked before anything is started


def test_bench_recorded_counter_figures_carry_their_provenance():
    """What bench.py cannot measure inside its own process (PMC counters need their own rocprofv3 passes) comes from files under
    profiles/ -- only for the workload and row count they were taken at, always with the source spelled out, never as a live figure."""
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("bench_mod2", root / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    val, src = bench.pmc_traffic("cfg3", "filter_f16_kernel", 600000)
    assert val > 1e10 and "recorded, not measured in this run" in src
    assert bench.pmc_traffic("cfg3", "filter_f16_kernel", 1000) == (None, None)                 # other sizes: no figure
    ir = bench.issue_roofs("refdefault", "filter_f16_kernel", 600000, 1.35, 2.0)
    assert ir["kernel"] == "filter_rows64n_kernel" and "recorded, not measured in this run" in ir["source"]
    assert 0.3 < ir["frac_of_valu_issue_roof"] < 0.7 and 0.3 < ir["frac_of_mfma_roof_at_this_clock"] < 0.7
    assert abs(ir["recorded_busy"]["valu"] - 0.49) < 0.05 and abs(ir["recorded_busy"]["matrix_pipe"] - 0.43) < 0.05
    # the roofs scale with the clock of THIS run and the fractions with its launch time
    ir2 = bench.issue_roofs("refdefault", "filter_f16_kernel", 600000, 2.70, 1.0)
    assert abs(ir2["valu_issue_roof_ms"] / ir["valu_issue_roof_ms"] - 2.0) < 1e-9 and abs(ir2["frac_of_valu_issue_roof"] - ir["frac_of_valu_issue_roof"]) < 1e-9
    assert bench.issue_roofs("refdefault", "filter_f16_kernel", 1000, 1.35, 2.0) is None and bench.issue_roofs("cfg3", "filter_f16_kernel", 600000, 1.0, 2.0) is None
    assert bench.issue_roofs("refdefault", "filter_f16_kernel", 600000, 1.35, None) is None       # no clock probe: no figure


def test_unsupported_k_fails_at_construction():
    """k beyond the kernels' list length (or beyond a region's size) is refused when the module is built, not at the first search."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    for bad in (0, 17, 64):                # (k up to 16 since round 6: two exact passes of lists of 8)
        with pytest.raises(ValueError, match="k="):
            VectorQuantizer(96, 16, 0.25, 0.0, True, True, [16, 16], k=bad)
    VectorQuantizer(96, 16, 0.25, 0.0, True, True, [16, 16], k=16)
    with pytest.raises(ValueError, match="k="):
        VectorQuantizer(12, 16, 0.25, 0.0, True, True, [16, 16], k=5)          # regions of 4 codes
    VectorQuantizer(96, 16, 0.25, 0.0, True, True, [16, 16], k=8)
