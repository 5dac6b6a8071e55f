import sys, torch
sys.path.insert(0, ".")
from medtok_amd import loss as L
from medtok_amd.synthetic import StandInGAT, StandInTextEncoder, primekg_shaped_batch
from medtok_amd.tokenizer import MultimodalTokenizer
dev = torch.device("cuda:0")
torch.manual_seed(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
Lt = int(sys.argv[3]) if len(sys.argv) > 3 else 512
model = MultimodalTokenizer(StandInTextEncoder(layers=1), StandInGAT(dim=D), text_dim=768, graph_out_channels=D, codebook_size=ne, codebook_embed_dim=D).to(dev).train()
inputs = primekg_shaped_batch(256, dev, seed=0, max_len=Lt)
def S(msg):
    torch.cuda.synchronize(); print(msg, flush=True)
text = model.text_mapped(model.tokenize_text(inputs)); nodes = model.tokenize_graph(inputs); S("encoders")
from medtok_amd.tokenizer import global_mean_pool
pooled = global_mean_pool(nodes, inputs.batch, 256); S("pool")
q = model.quantize
pt, pg = q.cross_attn.pooled(text, inputs.attention_mask, nodes, inputs.batch); S("xattn")
r = model(inputs); S("forward")
loss, parts = L.total_loss(r, 0.1, 0.1); S("loss %f" % float(loss))
loss.backward(); S("backward")
model.zero_grad(set_to_none=True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    text = model.text_mapped(model.tokenize_text(inputs)); nodes = model.tokenize_graph(inputs); S("bf16 encoders %s %s" % (text.dtype, nodes.dtype))
    pt, pg = q.cross_attn.pooled(text, inputs.attention_mask, nodes, inputs.batch); S("bf16 xattn %s" % pt.dtype)
    zt, vq_t, cm, xh, idx, w = q._search(pt, "shared", True); S("bf16 search")
    r = model(inputs); S("bf16 forward")
    loss, parts = L.total_loss(r, 0.1, 0.1); S("bf16 loss %f" % float(loss))
loss.float().backward(); S("bf16 backward")
