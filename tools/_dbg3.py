import sys, torch
dev = torch.device("cuda:0")
def S(m): torch.cuda.synchronize(); print(m, flush=True)
torch.manual_seed(0)
which = sys.argv[1]
if which.startswith("rocblas"):
    torch.backends.cuda.preferred_blas_library("cublas")
    print(torch.backends.cuda.preferred_blas_library())
qf = torch.randn(256, 800, 768, device=dev).bfloat16(); kv = torch.randn(256, 512, 768, device=dev).bfloat16()
if which.endswith("bmm"): sc = torch.bmm(qf, kv.transpose(1, 2)); S(which + " bmm ok")
if which.endswith("linear"):
    x = torch.randn(256, 512, 768, device=dev).bfloat16(); w = torch.randn(2304, 768, device=dev).bfloat16(); b = torch.randn(2304, device=dev).bfloat16()
    y = torch.nn.functional.linear(x, w, b); S(which + " linear ok")
    y = torch.nn.functional.linear(x, w[:768], b[:768]); S(which + " linear768 ok")
