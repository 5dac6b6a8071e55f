"""Dev tool: launch the search kernel a few times at one shape (for rocprofv3 --pmc passes)."""
import sys
sys.path.insert(0, ".")
import torch
from medtok_amd import ops
n, k, d, topk = [int(v) for v in sys.argv[1:5]]
path = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, d, device=dev, generator=g); W = torch.randn(k, d, device=dev, generator=g)
xh, xs = ops.rownorm(x); wh, ws = ops.rownorm(W)
for _ in range(2):
    ops.topk_search(xh, xs, wh, ws, topk, path)
torch.cuda.synchronize()
print("done")
