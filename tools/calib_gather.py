"""Dev tool: two kernels with KNOWN byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md,
section HBM: the x2 correction is established for wide streaming reads only):
  copy    -- a streaming float4 copy of 1.2 GB (reads 1.2 GB, writes 1.2 GB);
  gather  -- torch.index_select of 600 000 random 3 KB rows out of a 151 MB table (the re-score kernel's access pattern: whole
             rows of the codebook picked by id): reads 600 000 x 3072 B = 1.84 GB of rows (+ 4.8 MB of ids), writes 1.84 GB."""
import sys
sys.path.insert(0, ".")
import torch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(49152, 768, device=dev, generator=g)
ids = torch.randint(0, 49152, (600000,), device=dev, generator=g)
src = torch.randn(600000 * 768 // 2 * 1, device=dev, generator=g)          # 0.92 GB
out = torch.empty(600000, 768, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
    torch.index_select(table, 0, ids, out=out)
torch.cuda.synchronize()
print("copy bytes", src.numel() * 4, "gather row bytes", 600000 * 3072)
