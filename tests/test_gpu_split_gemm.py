"""GPU: the split-fp16 dense products (medtok_split_gemm_f16, split_gemm.h) against fp64 -- the projections around the
cross-attention core.  These are tolerance items (north_star: embeddings within 1e-5 relative fp32); the bar written here is
3e-6 of the output scale, i.e. the split form must be as good as an fp32 GEMM, not merely inside 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _split_weights(w, ops):
    """power-of-two prescale so that max |w| lands in [2^11, 2^12) + the (hi, lo) images"""
    import math
    amax = float(w.abs().max())
    scale = 2.0 ** (11 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    return ops.split_half(w.contiguous(), dp=w.shape[1], scale=scale), 1.0 / scale


@pytest.mark.parametrize("m,n,k", [(1000, 768, 768), (256, 256, 32), (4097, 3072, 192), (37, 100, 64), (3000, 64, 3072)])
def test_plain_product_matches_fp64(dev, m, n, k):
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev, generator=g) * 1.7
    w = torch.randn(n, k, device=dev, generator=g) / k ** 0.5
    bias = torch.randn(n, device=dev, generator=g) * 0.1
    (wh, wl), unscale = _split_weights(w, ops)
    c, split = ops.split_gemm(ops.split_half(a), (wh, wl), n_g=n, k_g=k, bias=bias, unscale=unscale, want_f32=True, want_split=True)
    ref = a.double() @ w.double().t() + bias.double()
    scale = float(ref.abs().max())
    err = float((c.double() - ref).abs().max()) / scale
    ref32 = float((torch.nn.functional.linear(a, w, bias).double() - ref).abs().max()) / scale
    assert err <= 3e-6, (err, ref32)
    # the (hi, lo) images of the result are the result
    ch, cl = split
    assert float((ch.double() + cl.double() - ref).abs().max()) / scale <= 3e-6
    assert err <= max(4 * ref32, 1e-6), f"split product {err:.2e} vs the library's fp32 GEMM {ref32:.2e}"


def test_grouped_products_are_the_per_head_products(dev):
    """The two per-head products of the folded attention form, as one launch each: group g reads its own column block of A and
    row block of B and writes its own column block of C."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    R, H, hd, D = 777, 4, 192, 768
    q = torch.randn(R, H * hd, device=dev, generator=g)
    wk = torch.randn(H * D, hd, device=dev, generator=g) / hd ** 0.5          # B of group h: rows [h D, (h+1) D), depth hd
    (bh, bl), un = _split_weights(wk, ops)
    qf, _ = ops.split_gemm(ops.split_half(q), (bh, bl), n_g=D, k_g=hd, groups=H, a_group_cols=hd, b_group_rows=D, unscale=un)
    ref = torch.stack([q[:, h * hd:(h + 1) * hd].double() @ wk[h * D:(h + 1) * D].double().t() for h in range(H)], 1).reshape(R, H * D)
    assert float((qf.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    # value side: depth D per head, hd output features per head (not a multiple of the 256-feature tile: masked stores)
    ctx = torch.randn(R, H * D, device=dev, generator=g)
    wv = torch.randn(H * hd, D, device=dev, generator=g) / D ** 0.5
    bv = torch.randn(H * hd, device=dev, generator=g) * 0.05
    (vh, vl), un = _split_weights(wv, ops)
    out, _ = ops.split_gemm(ops.split_half(ctx), (vh, vl), n_g=hd, k_g=D, groups=H, a_group_cols=D, b_group_rows=hd, bias=bv, unscale=un)
    ref = torch.cat([ctx[:, h * D:(h + 1) * D].double() @ wv[h * hd:(h + 1) * hd].double().t() for h in range(H)], 1) + bv.double()
    assert float((out.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


def test_split_half_is_exact_to_22_bits_and_pads_with_zeros(dev):
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(513, 100, device=dev, generator=g) * torch.logspace(-3, 2, 100, device=dev)
    hi, lo = ops.split_half(x, dp=128)
    assert hi.shape == (513, 128) and not hi[:, 100:].any() and not lo[:, 100:].any()
    back = hi[:, :100].double() + lo[:, :100].double()
    err = (back - x.double()).abs()
    assert bool((err <= x.abs().double() * 2.0 ** -21 + 2.0 ** -24).all())       # 2^-22 relative, or the fp16 subnormal step
    # a strided view (a column block of a wider matrix) is taken as it is
    wide = torch.randn(64, 300, device=dev, generator=g)
    h2, l2 = ops.split_half(wide[:, 100:164])
    assert float((h2.double() + l2.double() - wide[:, 100:164].double()).abs().max()) <= 2.0 ** -20


def test_projection_of_the_specific_searches_matches_nn_linear(dev):
    """VectorQuantizer.project: proj_text / proj_graph (reference :190,192) on the split-fp16 product at inference at every batch
    size (a column block of a wider [N, 2D] tensor, as forward() hands it over); the nn.Linear itself in eval mode with autograd on
    and when the split products are switched off; the cached weight images follow weight versions."""
    from medtok_amd.vector_quantization_soft_one_new import VectorQuantizer
    torch.manual_seed(1)
    D = 768
    v = VectorQuantizer(3 * 512, D, 0.25, 0.0, True, False, [D, D]).to(dev).eval()
    z = torch.randn(5000, 2 * D, device=dev)
    zt, zg = torch.split(z, [D, D], dim=-1)
    with torch.no_grad():
        for x, types, lin in ((zt, "text", v.proj_text), (zg, "graph", v.proj_graph)):
            got = v.project(x, types)
            ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
            lib = lin(x)
            scale = float(ref.abs().max())
            assert float((got.double() - ref).abs().max()) <= 3e-6 * scale
            assert float((got.double() - ref).abs().max()) <= max(4 * float((lib.double() - ref).abs().max()), 1e-6 * scale)
            assert getattr(lin, "_medtok_split_cache", None) is not None
        small = v.project(zt[:100], "text")                      # a small batch: the same product (round 3 kept nn.Linear below 1024 rows)
        assert torch.equal(small, v.project(zt, "text")[:100])
        import medtok_amd.vector_quantization_soft_one_new as M
        keep, M.SPLIT_PRODUCTS = M.SPLIT_PRODUCTS, False
        try:
            assert torch.equal(v.project(zt[:100], "text"), v.proj_text(zt[:100]))      # switched off: the nn.Linear itself
        finally:
            M.SPLIT_PRODUCTS = keep
        v.proj_text.weight.mul_(2.0)                             # version bump: the images are rebuilt
        got2 = v.project(zt, "text")
        ref2 = torch.nn.functional.linear(zt.double(), v.proj_text.weight.double(), v.proj_text.bias.double())
        assert float((got2.double() - ref2).abs().max()) <= 3e-6 * float(ref2.abs().max())
    x = zt.clone().requires_grad_(True)                           # under autograd: the nn.Linear (its backward is torch's)
    v.project(x, "text").sum().backward()
    assert x.grad is not None


def test_random_shapes_plain_and_grouped(dev):
    """Random rows / features / depths / group strides through both tile heights (256- and 192-feature tiles), with and without
    bias, fp32 and / or split output: every case within 4e-6 of the fp64 product (the staged epilogue masks rows past M and
    features past the group; a tile's last stage is also the only one when the depth is 32)."""
    import math, random
    from medtok_amd import ops
    rng = random.Random(7)
    for c in range(40):
        groups = rng.choice([1, 1, 2, 4, 3])
        k_g = 32 * rng.randint(1, 24)
        n_g = 4 * rng.randint(1, 200) if rng.random() < 0.7 else rng.choice([64, 192, 256, 384, 768])
        m = rng.choice([1, 7, 255, 256, 257, 1000, 4096, rng.randint(1, 20000)])
        a_cols = k_g if groups == 1 else k_g + 8 * rng.randint(0, 3)
        b_rows = n_g if groups == 1 else n_g + 4 * rng.randint(0, 5)
        lda = (groups - 1) * a_cols + k_g + 8 * rng.randint(0, 2)
        g = torch.Generator(device=dev).manual_seed(c)
        a = torch.randn(m, lda, device=dev, generator=g)
        w = torch.randn((groups - 1) * b_rows + n_g, k_g, device=dev, generator=g) / k_g ** 0.5
        bias = torch.randn(groups * n_g, device=dev, generator=g) if rng.random() < 0.6 else None
        scale = 2.0 ** (11 - math.floor(math.log2(float(w.abs().max()))))
        ws = ops.split_half(w.contiguous(), dp=k_g, scale=scale)
        want_f32, want_split = rng.choice([(True, False), (False, True), (True, True)])
        cf, cs = ops.split_gemm(ops.split_half(a), ws, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=a_cols, b_group_rows=b_rows, bias=bias,
                                unscale=1.0 / scale, want_f32=want_f32, want_split=want_split)
        ref = torch.cat([a[:, h * a_cols: h * a_cols + k_g].double() @ w[h * b_rows: h * b_rows + n_g].double().t() for h in range(groups)], 1)
        if bias is not None:
            ref = ref + bias.double()
        sc = float(ref.abs().max()) + 1e-30
        what = dict(case=c, m=m, n_g=n_g, k_g=k_g, groups=groups)
        if cf is not None:
            assert float((cf.double() - ref).abs().max()) / sc <= 4e-6, what
        if cs is not None:
            assert float((cs[0].double() + cs[1].double() - ref).abs().max()) / sc <= 4e-6, what


@pytest.mark.parametrize("m,k,n,gscale", [(700, 768, 768, 1.0), (333, 768, 3072, 1e-8), (1030, 3072, 768, 3e4), (64, 64, 64, 1.0), (5, 100, 36, 1e-3),
                                              (9001, 768, 768, 1.0)])
def test_split_linear_forward_and_backward_match_fp64(dev, m, k, n, gscale):
    """The training-mode dense product (_SplitLinearFunction: forward, dX = dY W, dW = dY^T X on medtok_split_gemm_scaled_f16, the
    operands prescaled by powers of two taken from device-side |.|_max values) against torch in fp64: every result to 1e-5 of its
    own scale -- also for upstream gradients at 1e-8 (a mean over millions of elements) and at 3e4 (under a GradScaler), which an
    unscaled fp16 split would flush to zero / overflow.  Rows and depths that are not multiples of the 32-deep k blocks included."""
    from medtok_amd.vector_quantization_soft_one_new import split_linear
    g = torch.Generator(device=dev).manual_seed(m + n)
    x = torch.randn(m, k, device=dev, generator=g, requires_grad=True)
    w = (torch.randn(n, k, device=dev, generator=g) * 0.05).requires_grad_()
    b = torch.randn(n, device=dev, generator=g, requires_grad=True)
    dy = torch.randn(m, n, device=dev, generator=g) * gscale
    y = split_linear(x, w, b)
    y.backward(dy)
    x64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, w, b))
    y64 = x64 @ w64.t() + b64
    y64.backward(dy.double())
    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(y.detach(), y64.detach()) <= 1e-5
    assert rel(x.grad, x64.grad) <= 1e-5 and rel(w.grad, w64.grad) <= 1e-5 and rel(b.grad, b64.grad) <= 1e-5
    # no bias, no input gradient wanted: only dW
    x2 = x.detach()
    w2 = w.detach().clone().requires_grad_()
    split_linear(x2, w2).backward(dy)
    assert rel(w2.grad, w64.grad) <= 1e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_one_pass_half_precision_gemm_matches_fp32_accumulation(dev, dt):
    """medtok_half_gemm_f32 (one pass over fp16 / bf16 operands, fp32 accumulation: what torch.autocast makes of nn.Linear) against the
    same operands multiplied in fp64: products of two half-precision numbers are exact in fp32, so only the accumulation order
    differs -- 2e-6 of the row scale; plain and grouped, both tile heights, ragged row counts."""
    from medtok_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    for (m, n_g, k_g, groups) in ((1000, 768, 768, 1), (257, 192, 768, 4), (5000, 64, 64, 2), (1, 4, 32, 1), (4096, 3072, 96, 1)):
        lda, b_rows = groups * k_g, groups * n_g
        a = (torch.randn(m, lda, device=dev, generator=g)).to(dt)
        b = (torch.randn(b_rows, k_g, device=dev, generator=g) / k_g ** 0.5).to(dt)
        bias = torch.randn(groups * n_g, device=dev, generator=g)
        c = ops.half_gemm(a, b, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=k_g, b_group_rows=n_g, bias=bias, unscale=0.5)
        ref = torch.cat([a[:, h * k_g:(h + 1) * k_g].double() @ b[h * n_g:(h + 1) * n_g].double().t() for h in range(groups)], 1) * 0.5 + bias.double()
        err = float((c.double() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-6, (m, n_g, k_g, groups, err)
        # depths that are multiples of 64 run with 64-deep stages (round 6); the 32-deep form (the library's A/B switch) accumulates in
        # the same order: the same bits
        from medtok_amd import _lib
        lib = _lib.load()
        lib.medtok_debug_set_half_gemm_k32(1)
        try:
            c32 = ops.half_gemm(a, b, n_g=n_g, k_g=k_g, groups=groups, a_group_cols=k_g, b_group_rows=n_g, bias=bias, unscale=0.5)
        finally:
            lib.medtok_debug_set_half_gemm_k32(0)
        assert torch.equal(c, c32), (m, n_g, k_g, groups)


def test_autocast_linear_takes_the_one_pass_form_and_matches_torch(dev):
    """split_linear under torch.autocast: forward and all three gradients against F.linear under the same autocast (torch's half-precision
    GEMM, fp32 accumulation) -- the same precision class, so they agree to accumulation order (1e-2 of the scale covers bf16's own
    rounding of torch's half-precision OUTPUT; ours stays fp32)."""
    from medtok_amd.vector_quantization_soft_one_new import split_linear
    g = torch.Generator(device=dev).manual_seed(4)
    for dt in (torch.bfloat16, torch.float16):
        for (m, k, n) in ((5000, 768, 768), (300, 3072, 768), (4100, 64, 64)):
            x = torch.randn(m, k, device=dev, generator=g, requires_grad=True)
            w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).requires_grad_(True)
            b = torch.randn(n, device=dev, generator=g, requires_grad=True)
            go = torch.randn(m, n, device=dev, generator=g)
            with torch.autocast("cuda", dtype=dt):
                y = split_linear(x, w, b)
                yr = torch.nn.functional.linear(x, w, b)
            assert y.dtype == torch.float32
            gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
            rx, rw, rb = torch.autograd.grad(yr.float(), (x, w, b), go)
            for got, ref, what in ((y, yr.float(), "y"), (gx, rx, "dx"), (gw, rw, "dw"), (gb, rb, "db")):
                err = float((got.double() - ref.double()).abs().max()) / float(ref.double().abs().max())
                assert err <= 1.5e-2, (dt, m, k, n, what, err)
