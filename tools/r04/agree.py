import sys
sys.path.insert(0, ".")
import torch, bench
from medtok_amd import ops
import medtok_amd.vector_quantization_soft_one_new as vq
dev = torch.device("cuda:0")
w = bench.Full(4096, dev, 0, ops.PATH_AUTO)
def outs(path):
    w.set_path(path)
    r = w.step()
    torch.cuda.synchronize()
    return {k: r[k].clone() for k in sorted(r) if isinstance(r[k], torch.Tensor)}
def diff(a, b, tag):
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print(tag, "differ:", bad)
    for k in bad[:4]:
        x, y = a[k], b[k]
        if x.dtype.is_floating_point:
            d = (x.double() - y.double()).abs()
            print("   ", k, "max abs diff %.3e  rows differing %d of %d" % (float(d.max()), int((d.reshape(d.shape[0], -1).max(1).values > 0).sum()), d.shape[0]))
        else:
            print("   ", k, "rows differing", int((x != y).reshape(x.shape[0], -1).any(1).sum()))
for name, setting in (("shipped", {}), ("att0", dict(ATTENTION_VARIANT=0)), ("one_stream", dict(SIDE_STREAM_MIN_CODES=0))):
    keep = {k: getattr(vq, k) for k in setting}
    for k, v in setting.items(): setattr(vq, k, v)
    a1, a2 = outs(ops.PATH_AUTO), outs(ops.PATH_AUTO)
    diff(a1, a2, f"[{name}] AUTO vs AUTO")
    f1 = outs(ops.PATH_F32_MFMA)
    diff(a1, f1, f"[{name}] AUTO vs F32")
    f2 = outs(ops.PATH_F32_MFMA)
    diff(f1, f2, f"[{name}] F32 vs F32")
    for k, v in keep.items(): setattr(vq, k, v)
