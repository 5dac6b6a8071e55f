"""Dev tool (GPU box): does the half-precision product care how far apart the rows of its streamed operand are?  The text mapping's forward
(131072 x 768 -> 768, operand rows 1.5 KB apart) takes 223 us, its weight gradient (the same flops, both operands streamed with rows
9.5 - 265 KB apart) 481 us.  Here: the forward product with the activation matrix stored at row strides of 768 ... 16384 halves.
python tools/r06/gemm_row_stride.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from medtok_amd import ops
dev = torch.device("cuda:0")
m, k, n = 131072, 768, 768
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16).contiguous()
for lda in (768, 1024, 2048, 4096, 4736, 8192):
    a = torch.zeros(m, lda, device=dev, dtype=torch.bfloat16)
    a[:, :k] = torch.randn(m, k, device=dev).to(torch.bfloat16)
    for _ in range(3): ops.half_gemm(a, w, n_g=n, k_g=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ops.half_gemm(a, w, n_g=n, k_g=k)
    torch.cuda.synchronize()
    print(f"activation rows {lda * 2 / 1024:6.1f} KB apart: {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us per product", flush=True)
    del a
