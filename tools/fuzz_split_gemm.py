"""Dev tool: random shapes through medtok_split_gemm_f16 (plain and grouped, both tile heights, every output combination) against
fp64.  usage: python tools/fuzz_split_gemm.py [cases] [seed]"""
import math, random, sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(cases):
    groups = rng.choice([1, 1, 1, 2, 4, 3])
    k_g = 32 * rng.randint(1, 24)
    n_g = 4 * rng.randint(1, 200) if rng.random() < 0.7 else rng.choice([64, 128, 192, 256, 384, 768])
    m = rng.choice([1, 7, 255, 256, 257, 1000, 4096, 5000, rng.randint(1, 70000)])
    a_cols = k_g if groups == 1 else k_g + 8 * rng.randint(0, 3)          # column stride between the groups' slices of A
    b_rows = n_g if groups == 1 else n_g + 4 * rng.randint(0, 5)
    lda = (groups - 1) * a_cols + k_g + 8 * rng.randint(0, 2)
    g = torch.Generator(device=dev).manual_seed(c)
    a = torch.randn(m, lda, device=dev, generator=g)
    w = torch.randn((groups - 1) * b_rows + n_g, k_g, device=dev, generator=g) / k_g ** 0.5
    bias = torch.randn(groups * n_g, device=dev, generator=g) if rng.random() < 0.6 else None
    amax = float(w.abs().max())
    scale = 2.0 ** (11 - math.floor(math.log2(amax)))
    ws = ops.split_half(w.contiguous(), dp=k_g, scale=scale)
    want_f32, want_split = rng.choice([(True, False), (False, True), (True, True)])
    cf, cs = ops.split_gemm(ops.split_half(a), ws, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=a_cols, b_group_rows=b_rows, bias=bias,
                            unscale=1.0 / scale, want_f32=want_f32, want_split=want_split)
    ref = torch.cat([a[:, h * a_cols: h * a_cols + k_g].double() @ w[h * b_rows: h * b_rows + n_g].double().t() for h in range(groups)], 1)
    if bias is not None:
        ref = ref + bias.double()
    sc = float(ref.abs().max()) + 1e-30
    errs = []
    if cf is not None: errs.append(float((cf.double() - ref).abs().max()) / sc)
    if cs is not None: errs.append(float((cs[0].double() + cs[1].double() - ref).abs().max()) / sc)
    e = max(errs)
    worst = max(worst, e)
    if not (e <= 4e-6):
        print("FAIL case", c, dict(m=m, n_g=n_g, k_g=k_g, groups=groups, a_cols=a_cols, b_rows=b_rows, lda=lda, bias=bias is not None, f32=want_f32, split=want_split), e)
        sys.exit(1)
print(f"{cases} cases ok, worst relative error {worst:.2e}")
