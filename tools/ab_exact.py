"""Dev tool: exact fp32-MFMA search kernel time for several builds of the library (each gets its own dlopen handle)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from medtok_amd import _lib, ops

def bench(lib_path, shapes, rounds=3):
    out = {}
    lib = ctypes.CDLL(lib_path, mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    _lib._lib = lib
    dev = torch.device("cuda:0")
    for (n, k, d, topk) in shapes:
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
        xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
        ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA); torch.cuda.synchronize()
        best = 1e9
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.topk_search(xh, xs, wh, ws, topk, ops.PATH_F32_MFMA); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[(n, k, d, topk)] = best
    return out

if __name__ == "__main__":
    shapes = [(600000, 16384, 768, 5), (100000, 8192, 768, 1)]
    libs = sys.argv[1:]
    res = {}
    for rnd in range(2):
        for l in libs:
            for k_, v in bench(l, shapes).items():
                res[(l, k_)] = min(v, res.get((l, k_), 1e9))
    for sh in shapes:
        n, k, d, t = sh
        print(sh, "  ".join(f"{l.split('/')[-1]}: {res[(l, sh)]:.2f} ms {2*n*k*d/res[(l, sh)]/1e9:.1f} TF" for l in libs), flush=True)
