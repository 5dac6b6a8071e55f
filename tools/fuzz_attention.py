"""Dev tool: randomized ragged shapes through the attention core (eval and training forward) against the C oracle.
usage: python tools/fuzz_attention.py [cases] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from medtok_amd import ops
from oracle import oracle as O
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    d = int(rng.choice([64, 128, 256, 384, 512, 640, 768]))
    n_codes = int(rng.integers(1, 9))
    q_len = rng.integers(0, 200, n_codes).astype(np.int64)
    kv_len = rng.integers(0, 300, n_codes).astype(np.int64)
    if c % 7 == 0: kv_len[0] = 0
    if q_len.sum() == 0: q_len[0] = 5
    q_start, kv_start = np.cumsum(q_len) - q_len, np.cumsum(kv_len) - kv_len
    q = (rng.standard_normal((int(q_len.sum()), d)) * rng.choice([0.05, 0.3, 1.0])).astype(np.float32)
    kv = rng.standard_normal((max(int(kv_len.sum()), 1), d)).astype(np.float32)
    scale = float(rng.choice([0.07, 0.125, 0.25]))
    T = lambda a: torch.from_numpy(a).to(dev)
    ref = O.shared_kv_attention(q, q_start, q_len, kv, kv_start, kv_len, scale)
    err = 0.0
    for exact in (False, True):
        out = ops.shared_kv_attention(T(q), T(q_start), T(q_len), T(kv), T(kv_start), T(kv_len), int(q_len.max()), scale, exact).cpu().numpy()
        err = max(err, np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-30))
    if d in ops.ATTENTION_SPLIT_WIDTHS and kv_len.sum() > 0:
        # the wide-batch kernels on (hi, lo) images (every variant the width has), and variant 2 fed the fp32 rows themselves:
        # within tolerance of the oracle, and the in-kernel split bit-identical to the image pass
        img = ops.split_half(T(kv))
        a = (T(q), T(q_start), T(q_len))
        b = (T(kv_start), T(kv_len), int(q_len.max()), scale)
        touched = torch.from_numpy(~np.isnan(ref).all(1)).to(dev)
        for v in ((0, 1, 2) if d == 768 else ((0, 2) if d in ops.ATTENTION_HALF_KEY_WIDTHS else (0,))):
            out = ops.shared_kv_attention_split(*a, img, *b, variant=v)
            err = max(err, np.abs(out.cpu().numpy() - ref)[touched.cpu().numpy()].max() / max(np.abs(ref[touched.cpu().numpy()]).max(), 1e-30))
            if v == 2:
                own = ops.shared_kv_attention_split(*a, T(kv), *b, variant=2)
                if not torch.equal(own[touched], out[touched]):
                    err = float("inf")
    if not (err <= 1e-5):
        bad += 1
        print(f"MISMATCH case {c}: d={d} codes={n_codes} q_len={q_len.tolist()} kv_len={kv_len.tolist()} rel err {err:.3g}", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
