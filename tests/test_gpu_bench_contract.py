"""GPU: bench.py prints ONE JSON line that carries the driver's contract fields plus `roofline` and `cpu_baseline`
(reduced row counts so the test takes seconds; the numbers themselves are not checked here)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("args", [["--rows", "20000", "--cpu-rows", "256"],
                                  ["--workload", "cfg2", "--rows", "20000", "--cpu-rows", "512"],
                                  ["--workload", "full", "--rows", "256", "--cpu-rows", "8"],
                                  ["--workload", "cfg4", "--rows", "32", "--text-layers", "1", "--cpu-rows", "16"]])
def test_bench_prints_one_contract_line(dev, args):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", *args],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "hbm_frac", "algorithmic_bytes_per_step"} <= set(r)
    assert r["bound"] in ("hbm", "mfma") and (r["traffic"] is None) == (r["traffic_source"] is None)
    assert r["traffic"] is None              # reduced row counts: the recorded PMC figure does not apply, so none is printed
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] in ("port", "reference") and c["value"] > 0
    e = d["exact_fp32_path"]
    assert e is None or e["outputs_bit_identical_to_default_path"] is True
    if "cfg4" in args:                       # the training step's extras: the same steps with the node-count read gone; kernel durations from the events pass
        assert d["with_max_nodes_bound"].get("ms_per_step", 0) > 0, d["with_max_nodes_bound"]
        assert r["events_pass_ms_per_step"] > 0 and "repeated right behind the timed region" in r["timed_in"]


def test_default_bench_line_carries_the_secondary_workloads(dev):
    """`python bench.py` as the driver runs it (cfg 3 at full size, one GPU): besides the contract fields the line carries
    extra.workloads -- short runs of the other workloads, so that the secondary claims are measured by whoever runs the default
    command -- the per-search traffic unit, the step's fabric / algorithmic ratio and the device-side fallback and candidate statistics.
    No entry may have failed."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-rows", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    r = d["roofline"]
    assert r["kernel"] == "filter_f16_kernel" and r["traffic"] is not None and r["traffic"] > 6.0e10          # per search: two dispatches of ~34 GB
    assert r["fabric_over_algorithmic"] is not None and 10.0 < r["fabric_over_algorithmic"] < 60.0
    fb = d["fallback_rows"]
    assert fb["rows_handed_to_the_exact_kernel_per_step"] == 0 and len(fb["searches"]) == 4
    assert all(40.0 < s["candidates_per_row"] < 400.0 and s["rows"] == 600000 for s in fb["searches"])
    w = d["extra"]["workloads"]
    for name in ("refdefault", "cfg2", "full", "full_rows256", "fullref", "cfg4_vq_only"):
        assert name in w and "error" not in w[name], (name, w.get(name))
        assert w[name]["value"] > 0 and w[name]["ms_per_step"] > 0 and w[name]["dominant_kernel"], name
    assert w["fullref"]["hip_graph_replay"]["replay_equals_eager"] is True
    assert w["full_rows256"]["no_host_read"]["hip_graph_replay"]["replay_equals_default_forward"] is True


def test_bench_data_distributions_off_the_gaussian(dev):
    """--data: rows next to codes, Student-t rows and a codebook of near-copies (reduced row count: seconds); the line says how many
    rows the shortlist handed to the exact kernel and how many candidates it kept per row"""
    for data in ("near_codes", "clustered_codebook", "heavy_tail"):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "1", "--rows", "20000", "--cpu-rows", "0",
                              "--exact-steps", "1", "--data", data], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
        assert data in d["config"]["workload"]
        assert d["exact_fp32_path"]["outputs_bit_identical_to_default_path"] is True, data
        fb = d["fallback_rows"]
        assert "error" not in fb and len(fb["searches"]) == 4 and 0 <= fb["rows_handed_to_the_exact_kernel_per_step"] <= 4 * 20000, (data, fb)
        if data != "clustered_codebook":                   # (only near-identical codes push rows over the shortlist's capacities)
            assert fb["rows_handed_to_the_exact_kernel_per_step"] == 0, (data, fb)


def test_build_then_smoke_in_one_process(dev):
    """The driver may call build() and smoke() from the same interpreter: the library is then loaded before anything
    touched torch.cuda, which once left two HIP runtimes fighting over the device."""
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "smoke ok" in out.stdout, (out.stdout + out.stderr)[-2000:]


@pytest.mark.parametrize("workload", ["cfg3", "cfg5", "codeshard"])
def test_two_rank_bench_on_one_gpu(dev, workload):
    """The N > 1 code paths of bench.py (barrier, max-over-ranks timing, the EMA all-reduce, the k-list all-gather) with two
    ranks sharing this box's GPU over gloo (MEDTOK_DIST_BACKEND: RCCL itself needs one GPU per rank)."""
    import os
    from conftest import run_with_retry
    env = dict(os.environ, MEDTOK_DIST_BACKEND="gloo")
    out = run_with_retry(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                       "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--workload", workload, "--rows", "20000"], cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["scaling"] == ("weak" if workload == "cfg3" else "strong")


def test_bench_gpus_2_starts_its_own_ranks(dev):
    """The driver's command line -- `python bench.py --gpus N`, no torchrun in front -- must launch the ranks itself (fresh child
    processes started before the parent touches the GPU).  Two ranks share this box's one GPU over gloo."""
    import os
    env = dict(os.environ, MEDTOK_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    from conftest import run_with_retry
    out = run_with_retry(lambda port: [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "20000"],
                         cwd=ROOT, env=env)              # (bench.py picks its own port; the argument is unused here)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rows_per_gpu"] == 20000
    # the unchanged driver command also times the two partitionings that exchange data (cfg 5): collective timings are in the line
    ss = d["extra"]["strong_scaling"]
    e, c = ss["cfg5_ema_step"], ss["codeshard_search"]
    assert e["collective"] == "all_reduce" and e["collectives_per_step"] == 1 and e["collective_bytes_per_step"] == 16384 * 769 * 4
    assert c["collective"] == "all_gather" and c["collectives_per_step"] == 1 and c["collective_bytes_per_step"] == 2 * 600000 * 5 * 8
    for r in (e, c):
        assert r["value"] > 0 and r["collective_ms_per_step"] > 0 and r["busbw_gbs"] > 0
