"""Soft top-k vector quantiser of MedTok on MI355X.

Drop-in for the reference's MedTok/vector_quantization_soft_one_new.py:
same class names, constructor arguments, method names, return structures and
state_dict keys (`codebook.weight`, `codebook_used`, `proj_text.*`,
`proj_graph.*`, `cross_attn.model.{i}.{multihead_attn,layer_norm}.*`), so a
reference checkpoint loads with strict=True.

What runs where (every device computation is a gfx950 kernel of medtok_amd/csrc behind the C ABI; there is no eager-PyTorch
or library-GEMM path on the inference side at any batch size)
  - normalise / distance / top-k / softmax / code mix / straight-through / squared error / usage window: medtok_amd.ops
    (fp16-MFMA shortlist + exact fp32 re-score, or the exact fp32-MFMA kernel; small batches: all searches of a forward in
    one call of three launches, ops.soft_vq_forward_multi);
  - cross-attention (:17-88,133-142): batched over the codes instead of the reference's per-sample Python loop, the key / value
    projections folded into the queries.  e_dim = 64 with 4 heads (the reference's default): ops.cross_attention_small -- both
    layers, both directions and the node mean in two launches, no host read.  Other widths (<= 768): per layer the four dense
    products on the split-fp16 GEMM (ops.split_gemm, three MFMA passes over (hi, lo) images: fp32-accurate), the ragged
    attention core (ops.shared_kv_attention_split / shared_kv_attention), residual + LayerNorm in one kernel -- one C call per
    layer and side (ops.cross_attention_layer).  Training / autograd: the same packed rows through _SplitLinearFunction
    (three-pass products for fp32 callers, ONE half-precision pass under torch.autocast: ops.half_gemm) and
    _RaggedAttentionFunction (HIP forward with dropout + HIP dQ / dKV backward);
  - proj_text / proj_graph: the same GEMM kernels (project(), project_both());
  - backward of a search: autograd.Function around ONE sparse kernel that only touches the k selected codes per row (the
    reference back-propagates through a dense N x K matrix); the code gradients are summed per code without atomics.

Deviations from the reference, all additive or bug-compatible by intent
(SURVEY.md section 0):
  - forward() also returns the 8 token/weight entries that tokenizer.py:235-238
    reads but the reference never produced (R3);
  - codebook_used is a plain buffer (the reference wraps a Parameter in it and
    crashes in train mode on CPU, R5); a batch with more ids than the window
    keeps the newest ids instead of raising;
  - top-k ties resolve to the lowest index (torch.topk leaves it undefined).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

# The cross-attention layers run on combined weight products (one GEMM per side instead of in_proj -> per-head fold and per-head
# Wv -> out_proj) while the extra flops of that form, 4 R D^2 (heads - 2) per layer, stay below this: launch-bound sizes (eight
# launches of ~8 us saved per layer against ~70 us of fp32 GEMM at the bound).  Negative: never.
COMBINE_MAX_EXTRA_FLOPS = 8e9
LPT_ORDER = True
# Inference: the layers' dense products (and proj_text / proj_graph) run on the library's split-fp16 MFMA GEMMs
# (CrossAttention._folded_rows_split) from this many packed query rows up.  1: always -- no library GEMM on the inference path at any
# batch size (at the reference's own B = 256 the split form and the launch-bound combined-weight / hipBLASLt forms further down time
# the same: 0.78-0.80 ms per forward, tools/r04/ab_fullref.py).
SPLIT_PRODUCTS = True
SPLIT_MIN_ROWS = 1
SPLIT_ATTENTION = True       # the graph side's attention on ops.shared_kv_attention_split (keys from fp16 images, made once per forward)
SPLIT_ATTENTION_MIN_ROWS = 1024      # ... from this many query rows up (below it the image pass is not worth its launch)
# which form of the wide-batch attention core ops.shared_kv_attention_split runs (include/medtok_vq.h): 2 = two 32-row tiles of a
# code per block, one phase apart on one copy of the keys (D = 256 / 512 / 768; the others fall back to 0 inside the library)
ATTENTION_VARIANT = 2
# fp32 text rows go to the graph side's attention core as they are and become their (hi, lo) images inside the kernel, chunk by
# chunk (variant 2 at D = 256 / 512 / 768): no image pass over the whole text batch, no image buffers
KEYS_SPLIT_IN_KERNEL = True
# training / autograd: the cross-attention's dense products (and their backward) on the library's own split-fp16 GEMMs instead
# of nn.functional.linear / einsum (hipBLASLt)
TRAIN_SPLIT_PRODUCTS = True
# inference: a layer's seven launches (images, four dense products, attention core, residual + LayerNorm) through ONE call of the
# C ABI (medtok_cross_attention_layer_f32: the same kernels and bits; six fewer host round trips per layer and side)
FUSED_LAYER_CALL = True
# ... and the tokenizer's text mapping (tokenizer.py:118, a Linear over every token of the batch: 131 072 rows at B = 256, L = 512) too.
# Off by default: under autocast the library's half-precision GEMM is ~3x cheaper than the fp32-accurate three-pass product, and
# that layer is upstream of the quantiser (bench.py --precomputed-encoders turns it on for the all-own-kernels profile).
TRAIN_SPLIT_TEXT_MAPPING = False
# ... under torch.autocast those products run as ONE half-precision pass with fp32 accumulation (ops.half_gemm: the precision class
# autocast gives the reference's nn.Linear / nn.MultiheadAttention, train_MedTok.py:212,394) instead of the fp32-accurate three-pass
# form (3x the matrix work, device-side |x|_max prescales, lo images); fp32 callers keep the three-pass form
AUTOCAST_HALF_PRODUCTS = True
# ... and the attention forward of the graph side (many query rows per code) on the inference kernel's three-pass fp16 products with
# dropout and log-sum-exp added (attention_pp.h, TRAIN form) instead of the exact fp32 matrix pipe: 420 -> ~165 us per layer at cfg 4
AUTOCAST_SPLIT_ATTENTION_FORWARD = True
# training: the two directions of the cross-attention share every layer's weights, so their rows go through the layer's dense
# products in one launch per product (CrossAttention._pooled_packed)
MERGE_SIDES_IN_TRAINING = True
# training: all searches of a forward under one autograd node, so that the codebook receives ONE dense gradient (_SoftVQMultiFunction)
TRAIN_SINGLE_CODEBOOK_GRADIENT = True
# ... and their forward as ONE batched call (three launches: ops.soft_vq_forward_multi with per-row squared errors) where every search
# takes the exact path with at most 4096 rows.  Off: measured SLOWER at cfg 4 (six searches of 256 rows over 49 152 / 16 384 codes:
# 10.5 -> 10.9 ms per step, tools/r05/ab_cfg4_switch.py TRAIN_BATCHED_SEARCHES) -- the batched kernel's split plan is made for the
# few hundred codes of the reference's default codebook; it pays at e_dim = 64, n_e = 600 (fewer launches), not here
TRAIN_BATCHED_SEARCHES = False
# ... the searches that share a region (the two shared ones; text and its aug view; graph and its aug view) as ONE search on their rows stacked
TRAIN_STACK_SEARCHES_OF_A_REGION = True
# inference at the reference's own width (e_dim = 64, 4 heads): the whole cross-attention of a forward -- both layers, both
# directions, node mean -- in two launches with no host read (ops.cross_attention_small); needs a SORTED batch vector (PyG's are;
# the kernels flag anything else in CrossAttention.small_status, checked wherever the forward synchronises anyway)
SMALL_WIDTH_FUSED = True
# inference: the two shared searches of a forward as ONE search over the interleaved rows [text_0, graph_0, text_1, ...] (one
# codebook pass, twice the rows per launch; its [2 B, e] result IS the [B, 2 e] shared embedding)
MERGE_SHARED_SEARCHES = True
# inference on small batches (every search of the forward on the exact fp32 path with at most 4096 rows: ops.multi_search_eligible):
# the forward's three to five searches in ONE call of three launches (ops.soft_vq_forward_multi) and its three to five updates of
# the usage window in one call of two (ops.usage_update_multi_)
BATCHED_SMALL_SEARCHES = True
# inference: what every fp16-shortlist search derives from its codebook region (fp16 image, accumulator start values, largest norm)
# is made ONCE per weight version, for all three regions together with the normalisation (ops.prepare_codebook: two launches),
# instead of three passes over its region in each of a forward's three to five searches
PREPARED_CODEBOOK = True
# training under autocast: the row-major and the transposed 16-bit image of a product's input / upstream gradient from ONE pass over it
# (ops.half_image_pair) instead of two
FUSE_IMAGE_PAIRS = True
# ... and a Linear's bias gradient (the column sums of its upstream gradient) from that same pass -- per 64-row tile in the kernel, over the
# tiles in one small reduction -- instead of a reduction of its own over the gradient (104 us at 131 072 x 768)
BIAS_GRADIENT_FROM_IMAGE_PASS = True
# training: the text rows are read three ways -- as the keys of every cross-attention layer (:83,86: always the ORIGINAL text), as the
# CLS query of the text side, as the CLS half of h (tokenizer.py:162) -- and autograd would sum their four [B L, D] gradients with a
# zero fill and an add pass each (0.8 ms of a 12 ms step at B = 256, L = 512).  On: the layers' dKV kernels write into ONE buffer
# (the first zeroes and stores, the others add), the CLS gradients are added to its B rows in place (_TextFanOut)
KEY_GRADIENT_SINK = True
# ... and the layers' dKV launches deferred to the node that collects the key gradient: ONE launch over all layers' queries writes the
# [rows, D] matrix once (per-layer launches: a zero fill, a store and a read-add-store pass over it)
DEFERRED_KEY_GRADIENT = True
# training: both directions of a layer under ONE autograd node that writes their outputs (and dQ) into row ranges of one matrix, instead
# of a split in front of two nodes and a concatenation behind them (a copy of the [R heads, D] matrix each, forward and backward)
TWO_SIDED_ATTENTION_NODE = True
# CrossAttention.prepack(): pooled()'s prologue and the copy of its host-read values issued early by a caller that can (the tokenizer)
PREPACK_CODES = True
# training: the node mean of a code on the library's ordered segment mean under autograd (_SegmentMeanFunction) instead of a scatter into a
# zero [B, max_nodes, D] tensor and a sum over it
TRAIN_SEGMENT_MEAN = True
from .norm_ema_quantizer import EmbeddingEMA

USAGE_WINDOW = 300000   # vector_quantization_soft_one_new.py:118
UNSORTED_BATCH_MESSAGE = ("pooled(): the paths without a host read (assume_sorted_batch = True; max_nodes_bound) need a non-decreasing `batch` "
                          "vector (PyG-style); sort the nodes by code (CrossAttention.sort_by_code), or leave assume_sorted_batch = False / "
                          "max_nodes_bound = None")

# Inference forward: the work that does not depend on the graph side of the cross-attention -- the two (four with an aug view)
# modality-specific searches and the text side's attention chain, about an eighth of a forward, all launches of a few hundred
# blocks -- is enqueued on a second HIP stream and runs in the shadow of the graph side's chip-filling kernels; the streams join
# before the shared searches.  From this many codes per call (below it a forward is launch-bound and a second stream only adds
# host work); 0 turns it off.
SIDE_STREAM_MIN_CODES = 512
STREAM_PRIORITY = (-1, 0, 0)       # text side (high), modality-specific searches, text images
TEXT_CHAIN_AFTER_LAYER = 0         # the text side's launches are issued behind this graph-side layer (-1: in front of the graph side)
_side_streams = {}


def _side_stream(device, which=0):
    """(side, current): a per-device extra stream (`which`: 0 = the cross-attention's text side, 1 = the modality-specific searches,
    2 = the fp16 images of the text)
    that has just been made to wait for everything enqueued on the current one.
    All three are created together, in this fixed order, the first time any of them is needed on a device: HIP assigns streams to its
    few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and streams that share a queue serialise -- created
    lazily in use order (or re-created) the same forward ran at 410 k or 315 k codes/s depending on where they landed
    (tools/ab_streams.py)."""
    cur = torch.cuda.current_stream(device)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    side = _side_streams.get((dev_index, which))
    if side is None:
        # (the text side's stream has the higher priority: its kernels are small -- one query row per code and head -- and, at equal
        # priority, wait for slots behind the graph side's chip-filling launches until they END the forward instead of hiding in it)
        for w in range(len(STREAM_PRIORITY)):
            _side_streams.setdefault((dev_index, w), torch.cuda.Stream(device=device, priority=STREAM_PRIORITY[w]))
        side = _side_streams[(dev_index, which)]
    side.wait_stream(cur)
    return side, cur


def _lend(stream, *tensors):
    """tensors born on the current stream that kernels enqueued on `stream` read: tell the caching allocator, so that a block the
    host drops (or re-binds) right after the enqueue is not handed to later current-stream work while `stream` still reads it"""
    for t in tensors:
        if isinstance(t, (tuple, list)):
            _lend(stream, *t)
        elif isinstance(t, dict):
            _lend(stream, *t.values())
        elif isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(stream)


def _join_side(side, cur, tensors):
    """the current stream waits for the side stream; tensors born on the side stream are marked as used on the current one, so that
    the caching allocator does not hand their memory to later side-stream work while current-stream readers are still pending"""
    cur.wait_stream(side)
    _lend(cur, *tensors)


class _BuiltOn:
    """(event, stream) of a cache entry's build.  Copies of a module (copy.deepcopy, pickling) carry no mark: HIP events and
    streams belong to the process and device that made them."""
    __slots__ = ("event", "stream")

    def __init__(self, event, stream):
        self.event, self.stream = event, stream

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())


def _cached(holder, attr, key, build, device, rebuild=False):
    """A lazily built, per-(storage, version) cache entry that several HIP streams may read: (key, value, _BuiltOn) on `holder`.
    The entry is built on whatever stream first needs it; a reader on ANOTHER stream waits for the build's event and pins the
    memory (the tensors live in the building stream's allocator pool: when the entry is dropped they could otherwise be re-used
    by that stream while this one still reads).  Without this a forward that forks its side streams right after an invalidation
    (every bench step; every eval after an optimizer step) was ordered only by timing."""
    c = getattr(holder, attr, None)
    if not rebuild and c is not None and c[0] == key:
        # the hit on the building stream -- the common case, several times per forward -- without making a Stream object
        mark = c[2]
        if mark is None or device.type != "cuda" or mark.stream.cuda_stream == ops._stream_of(device):
            return c[1]
    cur = torch.cuda.current_stream(device) if device.type == "cuda" else None
    # Under HIP-graph capture no event of the outside world may be waited for (and none recorded here may be waited for outside):
    # a capture is preceded by a warm-up and a synchronisation (the caches are built and complete), so entries are read as they are;
    # one built inside a capture carries no mark.
    capturing = cur is not None and torch.cuda.is_current_stream_capturing()
    if rebuild or c is None or c[0] != key:
        value = build()
        mark = None
        if cur is not None and not capturing:
            mark = _BuiltOn(torch.cuda.Event(), cur)
            mark.event.record(cur)
        setattr(holder, attr, (key, value, mark))
        return value
    mark = c[2]
    if mark is not None and cur is not None and cur != mark.stream and not capturing:
        cur.wait_event(mark.event)
        _lend(cur, c[1])
    return c[1]


class _SegmentMeanFunction(torch.autograd.Function):
    """Mean of the attended nodes of every code (:140-141) under autograd: forward = the library's ordered segment mean (rows of a code
    are adjacent), backward = each node row receives its code's gradient / node count (one division, one gather).  The training path
    used to scatter the rows into a zero [B, max_nodes, D] tensor and sum it (157 MB at B = 256, max 200 nodes, D = 768)."""

    @staticmethod
    def forward(ctx, g, starts, counts, batch_sorted):
        ctx.save_for_backward(counts, batch_sorted)
        ctx.in_dtype = g.dtype
        return ops.segment_mean(g.detach().float().contiguous(), starts, counts)

    @staticmethod
    def backward(ctx, d_out):
        counts, batch_sorted = ctx.saved_tensors
        per_code = d_out.float() / counts.clamp(min=1).unsqueeze(-1).to(torch.float32)
        return per_code[batch_sorted].to(ctx.in_dtype), None, None, None


class _KeyGradSink:
    """Where the layers that share one key matrix leave its gradient during a backward (see _TextFanOut): either written by their dKV
    kernels at once (`buf`: the first stores, the others add), or -- DEFERRED_KEY_GRADIENT -- as `pending` sources (a layer's queries,
    upstream gradient, row statistics, mask parameters) that the collecting node turns into the gradient with ONE launch."""
    __slots__ = ("buf", "node", "pending", "common")

    def __init__(self):
        self.buf = None         # the [rows, D] gradient while a backward is under way
        self.node = None        # weak reference to the autograd node that collects it
        self.pending = []       # sources of ops.shared_kv_attention_dkv_multi
        self.common = None      # (kv, kv_start, kv_len, max_kv_len, half) of the pending sources


def _sink_key_gradient(sink, q, q_start, q_len, kv, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed, out, lse, d_out, half, dq_into=None):
    """dq of one attention call whose key gradient goes to `sink`: deferred as a source of the collecting node's one dKV launch where
    the keys are those of the sources already there, else written / added by this call's own dKV kernel."""
    common = (kv, kv_start, kv_len, max_kv_len, half)
    same = sink.common is not None and all(a is b or (torch.is_tensor(a) and torch.is_tensor(b) and a.data_ptr() == b.data_ptr() and a.shape == b.shape)
                                           or (not torch.is_tensor(a) and a == b) for a, b in zip(common, sink.common))
    if DEFERRED_KEY_GRADIENT and sink.buf is None and len(sink.pending) < ops.DKV_SOURCES_MAX and (not sink.pending or same):
        dq, delta = ops.shared_kv_attention_backward_dq(q, q_start, q_len, kv, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed, out, lse,
                                                        d_out, half=half, dq_into=dq_into)
        sink.pending.append(dict(q=q, d_out=d_out, lse=lse, delta=delta, q_start=q_start, q_len=q_len, scale=scale, dropout_p=dropout_p, seed=seed))
        sink.common = common
        return dq
    first = sink.buf is None
    if first:
        sink.buf = torch.empty_like(kv)
    dq, _ = ops.shared_kv_attention_backward(q, q_start, q_len, kv, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed, out, lse, d_out,
                                             half=half, dkv_into=sink.buf, accumulate=not first, dq_into=dq_into)
    return dq


class _TextFan:
    """carried by a text tensor that went through fan_out_text(): the CLS rows and the gradient sink of its keys"""
    __slots__ = ("sink", "cls")

    def __init__(self, sink, cls):
        self.sink, self.cls = sink, cls


class _TextFanOut(torch.autograd.Function):
    """text [B, L, D] -> (the same rows, their CLS rows [B, D]) with ONE gradient buffer behind both: the attention layers whose keys
    these rows are add their dKV into `sink` during the backward (and hand autograd no gradient for them), this node -- which the
    engine runs after every consumer of its outputs -- adds the CLS gradient to the buffer's B first rows and passes it on."""

    @staticmethod
    def forward(ctx, text, sink):
        ctx.sink = sink
        ctx.seq_len = text.shape[1]
        ctx.set_materialize_grads(False)
        return text.view_as(text), text[:, 0].contiguous()

    @staticmethod
    def backward(ctx, g_rows, g_cls):
        sink = ctx.sink
        buf, sink.buf = sink.buf, None
        pending, common, sink.pending, sink.common = sink.pending, sink.common, [], None
        if pending:                         # the deferred key gradients of all layers: one launch, one store of the [rows, D] matrix
            d = ops.shared_kv_attention_dkv_multi(pending, common[0], common[1], common[2], common[3], half=common[4])
            buf = d if buf is None else buf.add_(d)
        if g_rows is not None:              # (a consumer outside the sink protocol)
            g_rows = g_rows.float()
            buf = g_rows.clone() if buf is None else buf.view_as(g_rows).add_(g_rows)
        if g_cls is not None:
            if buf is None:
                buf = g_cls.new_zeros((g_cls.shape[0], ctx.seq_len, g_cls.shape[1]), dtype=torch.float32)
            buf = buf.view(g_cls.shape[0], -1, g_cls.shape[1])
            buf[:, 0].add_(g_cls)
        return (None if buf is None else buf.view(-1, ctx.seq_len, buf.shape[-1])), None


def fan_out_text(text):
    """The text rows [B, L, D] of a training step behind one gradient buffer (KEY_GRADIENT_SINK): returns the same values as a tensor
    that carries `_medtok_fan` = (sink, CLS rows).  CrossAttention.pooled() and MultimodalTokenizer.quant() read the CLS rows from
    there and route the key gradients of the attention layers into the sink; any other use of the returned tensor is ordinary
    autograd.  Anything it does not apply to (no autograd, not an fp32 [B, L, D] tensor on an MI355X) is returned as it came."""
    if getattr(text, "_medtok_fan", None) is not None:
        return text
    if not (KEY_GRADIENT_SINK and torch.is_tensor(text) and text.is_cuda and text.dim() == 3 and text.dtype == torch.float32
            and text.requires_grad and torch.is_grad_enabled() and text.is_contiguous() and text.shape[1] > 0
            and hasattr(torch._C, "_will_engine_execute_node")):
        return text
    import weakref
    sink = _KeyGradSink()
    rows, cls = _TextFanOut.apply(text, sink)
    sink.node = weakref.ref(rows.grad_fn)
    rows._medtok_fan = _TextFan(sink, cls)
    return rows


def _autocast_half():
    """the half-precision dtype of the attention backward's products under torch.autocast (None: exact fp32 kernels)"""
    if AUTOCAST_HALF_PRODUCTS and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") in (torch.float16, torch.bfloat16):
        return torch.get_autocast_dtype("cuda")
    return None


def _split_forward(half, width, max_q_len, q_rows, kv_rows):
    """the training forward on the three-pass fp16 kernel? (autocast callers, many query rows per code: see _RaggedAttentionFunction)"""
    return (half is not None and AUTOCAST_SPLIT_ATTENTION_FORWARD and width in ops.ATTENTION_TRAIN_SPLIT_WIDTHS and max_q_len > 8
            and q_rows > 0 and kv_rows > 0)


def _live_sink(sink, needs_kv_grad):
    """the sink if THIS backward will run the node that collects it (torch.autograd.grad() for other inputs does not), else None"""
    if sink is None or not needs_kv_grad:
        return None
    node = sink.node() if sink.node is not None else None
    return sink if (node is not None and torch._C._will_engine_execute_node(node)) else None


class _TwoSidedAttentionFunction(torch.autograd.Function):
    """Both directions of a training layer in ONE autograd node: query rows [0, cut) (the graph side: nodes x heads) attend to kv_a
    (the text rows), rows [cut, n) (the text side: CLS x heads) to kv_b (the nodes) -- the two launches of _RaggedAttentionFunction
    writing into row ranges of one output (and, backward, of one dQ) instead of a split in front and a concatenation behind, each a
    copy of the [R heads, D] matrix forward and backward.  la / lb = (q_start, q_len, kv_start, kv_len, max_q_len, max_kv_len) per side."""

    @staticmethod
    def forward(ctx, qf, kv_a, kv_b, la, lb, cut, scale, dropout_p, seed_a, seed_b, sink):
        q = qf.detach().float().contiguous()
        ka, kb = kv_a.detach().float().contiguous(), kv_b.detach().float().contiguous()
        ctx.half = _autocast_half()
        n = q.shape[0]
        out = torch.zeros_like(q)
        lse = torch.full((n,), float("-inf"), dtype=torch.float32, device=q.device)
        for lo, hi, kv, l, seed in ((0, cut, ka, la, seed_a), (cut, n, kb, lb, seed_b)):
            if hi > lo:
                ops.shared_kv_attention_train(q[lo:hi], l[0], l[1], kv, l[2], l[3], l[4], scale, dropout_p, seed,
                                              split=_split_forward(ctx.half, q.shape[1], l[4], hi - lo, kv.shape[0]), out=out[lo:hi], lse=lse[lo:hi])
        ctx.save_for_backward(q, ka, kb, out, lse, *la[:4], *lb[:4])
        ctx.cfg = (cut, la[4], la[5], lb[4], lb[5], scale, dropout_p, seed_a, seed_b, qf.dtype, kv_a.dtype, kv_b.dtype)
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, ka, kb, out, lse, *lists = ctx.saved_tensors
        cut, mq_a, mk_a, mq_b, mk_b, scale, dropout_p, seed_a, seed_b, qd, kad, kbd = ctx.cfg
        n = q.shape[0]
        d = d_out.float().contiguous()
        dq = torch.empty_like(q)
        dka = dkb = None
        sink = _live_sink(ctx.sink, ctx.needs_input_grad[1])
        if cut > 0:
            a = lists[:4]
            if sink is not None:
                _sink_key_gradient(sink, q[:cut], a[0], a[1], ka, a[2], a[3], mq_a, mk_a, scale, dropout_p, seed_a, out[:cut], lse[:cut], d[:cut], ctx.half,
                                   dq_into=dq[:cut])
            else:
                _, dka = ops.shared_kv_attention_backward(q[:cut], a[0], a[1], ka, a[2], a[3], mq_a, mk_a, scale, dropout_p, seed_a, out[:cut], lse[:cut],
                                                          d[:cut], half=ctx.half, dq_into=dq[:cut])
                dka = dka.to(kad)
        elif ctx.needs_input_grad[1] and sink is None:
            dka = torch.zeros_like(ka).to(kad)
        if n > cut:
            b = lists[4:]
            _, dkb = ops.shared_kv_attention_backward(q[cut:], b[0], b[1], kb, b[2], b[3], mq_b, mk_b, scale, dropout_p, seed_b, out[cut:], lse[cut:], d[cut:],
                                                      half=ctx.half, dq_into=dq[cut:])
            dkb = dkb.to(kbd)
        else:
            dkb = torch.zeros_like(kb).to(kbd)
        return dq.to(qd), dka, dkb, None, None, None, None, None, None, None, None


class _RaggedAttentionFunction(torch.autograd.Function):
    """The ragged attention core under autograd: forward = medtok_shared_kv_attention_train_f32 (dropout on the probabilities by
    a stateless hash mask, log-sum-exp kept per row), backward = medtok_shared_kv_attention_backward_f32 (dQ and dKV kernels that
    rebuild probabilities and mask; nothing of size rows x keys is stored).  fp32 whatever autocast says.
    `sink` (a _KeyGradSink, or None): the key gradient goes into the sink's buffer instead of back to autograd."""

    @staticmethod
    def forward(ctx, q, kv, q_start, q_len, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed, sink=None):
        qf, kvf = q.detach().float().contiguous(), kv.detach().float().contiguous()
        # under torch.autocast the backward's four matrix products run as ONE half-precision pass (the reference's class there) and
        # the forward on the three-pass fp16 products (fp32-accurate to ~1e-6: its log-sum-exp feeds the backward's softmax
        # rebuild) where more than a few query rows share a code's keys; fp32 callers keep the exact fp32 kernels on both sides
        ctx.half = _autocast_half()
        split = _split_forward(ctx.half, qf.shape[1], max_q_len, qf.shape[0], kvf.shape[0])
        out, lse = ops.shared_kv_attention_train(qf, q_start, q_len, kvf, kv_start, kv_len, max_q_len, scale, dropout_p, seed, split=split)
        ctx.save_for_backward(qf, kvf, out, lse, q_start, q_len, kv_start, kv_len)
        ctx.cfg = (max_q_len, max_kv_len, scale, dropout_p, seed, q.dtype, kv.dtype)
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, d_out):
        qf, kvf, out, lse, q_start, q_len, kv_start, kv_len = ctx.saved_tensors
        max_q_len, max_kv_len, scale, dropout_p, seed, qd, kd = ctx.cfg
        sink = _live_sink(ctx.sink, ctx.needs_input_grad[1])
        if sink is not None:
            dq = _sink_key_gradient(sink, qf, q_start, q_len, kvf, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed, out, lse,
                                    d_out.float().contiguous(), ctx.half)
            return dq.to(qd), None, None, None, None, None, None, None, None, None, None, None
        dq, dkv = ops.shared_kv_attention_backward(qf, q_start, q_len, kvf, kv_start, kv_len, max_q_len, max_kv_len, scale, dropout_p, seed,
                                                   out, lse, d_out.float().contiguous(), half=ctx.half)
        return dq.to(qd), dkv.to(kd), None, None, None, None, None, None, None, None, None, None


# split-K of the weight-gradient products (dW = dY^T X, contraction over the rows): at least this many rows per group.  A cfg 4 layer
# has ~5 600 rows: at 2048 its [768, 768] gradients ran as 18 blocks of 256 x 256 on 256 CUs (56 us), its block-diagonal ones as 72 (146 us)
SPLIT_K_MIN_ROWS = 512


def _pad32(n):
    return (int(n) + 31) // 32 * 32


class _SplitLinearFunction(torch.autograd.Function):
    """y = x W^T + b under autograd on the library's own dense product (medtok_split_gemm_scaled_f16: three fp16 MFMA passes over
    (hi, lo) pairs, fp32-accurate) -- forward, data gradient dX = dY W and weight gradient dW = dY^T X are all "A . B^T" products of
    split operands; the operands whose magnitude the host does not know (activations, upstream gradients) are prescaled by a power
    of two taken from a device-side |.|_max, so nothing is read back.  fp32 in and out whatever autocast says (the reference's
    projections, vector_quantization_soft_one_new.py:30,45, run in the autocast dtype: this is at least as accurate)."""

    @staticmethod
    def _weight_images(w, wf, npad_t):
        """(amax, images [n, pad32(k)], transposed images [k, pad32(n)]) of a weight, per (storage, version): a parameter is used by
        both attention directions of a step and by forward and backward -- one |w|_max and one split each instead of four.  Tensors
        built inside the graph (the block-diagonal per-head weights) are new objects every forward and are simply split again."""
        # (a slice of a parameter -- the q / k / v thirds of in_proj_weight -- is cached on the parameter it views)
        holder = w if isinstance(w, nn.Parameter) else (w._base if isinstance(getattr(w, "_base", None), nn.Parameter) else None)
        key = (w.data_ptr(), w._version, tuple(w.shape))
        cache = getattr(holder, "_medtok_train_images", None) if holder is not None else None
        if cache is not None and key[0] in cache and cache[key[0]][0] == key:
            return cache[key[0]][1]
        aw = ops.absmax(wf)
        val = (aw, ops.split_half_scaled(wf, _pad32(wf.shape[1]), aw), ops.split_half_scaled(wf, npad_t, aw, transpose=True))
        if holder is not None:
            if cache is None:
                cache = holder._medtok_train_images = {}
            cache[key[0]] = (key, val)
        return val

    @staticmethod
    def _half_images(w, dt, kp, npad):
        """(w [n, pad32(k)], w^T [k, pad32(n)]) as dt (fp16 / bf16) matrices, per (storage, version, dtype) for parameters"""
        holder = w if isinstance(w, nn.Parameter) else (w._base if isinstance(getattr(w, "_base", None), nn.Parameter) else None)
        key = (w.data_ptr(), w._version, tuple(w.shape), dt)
        cache = getattr(holder, "_medtok_half_images", None) if holder is not None else None
        if cache is not None and (key[0], dt) in cache and cache[(key[0], dt)][0] == key:
            return cache[(key[0], dt)][1]
        w16 = w.detach().to(dt)
        n, k = w16.shape
        val = (torch.nn.functional.pad(w16, (0, kp - k)).contiguous() if kp != k else w16.contiguous(),
               torch.nn.functional.pad(w16.t(), (0, npad - n)).contiguous())
        if holder is not None:
            if cache is None:
                cache = holder._medtok_half_images = {}
            cache[(key[0], dt)] = (key, val)
        return val

    @staticmethod
    def _forward_half(ctx, x, w, b, dt):
        """the autocast form: y = x16 w16^T + b, one half-precision pass with fp32 accumulation; the 16-bit operand images (and, in
        the backward, their transposes) come from the library's own cast / transpose kernels"""
        m, k = x.shape
        n = w.shape[0]
        kp, npad = _pad32(k), _pad32(n)
        xf = x.detach().float()
        xf = xf if xf.stride(1) == 1 and xf.stride(0) % 4 == 0 and xf.data_ptr() % 16 == 0 else xf.contiguous()
        w16, wt16 = _SplitLinearFunction._half_images(w, dt, kp, npad)
        # the weight gradient contracts over the rows: it reads x TRANSPOSED (split over the rows into `groups` chunks: see backward);
        # that image comes out of the same pass over x as the forward's operand, and is what the backward keeps of x
        groups, chunk, mp = _SplitLinearFunction._row_split(m, n, k)
        if ctx.needs_input_grad[1]:
            x16, xt16 = (ops.half_image_pair(xf, kp, mp, dt, group_cols=chunk) if FUSE_IMAGE_PAIRS else
                         (ops.half_image(xf, kp, dt), ops.half_image(xf, mp, dt, transpose=True, group_cols=chunk)))
        else:
            x16, xt16 = ops.half_image(xf, kp, dt), None
        y = ops.half_gemm(x16, w16, n_g=n, k_g=kp, bias=None if b is None else b.detach().float().contiguous())
        ctx.save_for_backward(xt16, wt16)
        ctx.shape = (m, k, n)
        ctx.half = dt
        ctx.dtypes = (x.dtype, w.dtype, None if b is None else b.dtype)
        return y

    @staticmethod
    def _row_split(m, n, k):
        """split-K of the weight-gradient product: (groups, rows per group (a multiple of 64), padded row count)"""
        tiles = ((n + 255) // 256) * ((k + 255) // 256)
        groups = max(1, min(256 // max(tiles, 1), m // SPLIT_K_MIN_ROWS))
        chunk = (-(-m // groups) + 63) // 64 * 64
        groups = -(-m // chunk)
        return groups, chunk, groups * chunk

    @staticmethod
    def _backward_half(ctx, dy):
        xt16, wt16 = ctx.saved_tensors
        dxt, dwt, dbt = ctx.dtypes
        m, k, n = ctx.shape
        dt = ctx.half
        npad = wt16.shape[1]
        dyf = dy.detach().float().contiguous()
        dx = dw = db = None
        groups, chunk, mp = _SplitLinearFunction._row_split(m, n, k)
        want_dx, want_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and xt16 is not None
        want_db = dbt is not None and ctx.needs_input_grad[2]
        if want_dx and want_dw and FUSE_IMAGE_PAIRS:              # both images of dY (and the bias gradient) from one pass over it
            if want_db and BIAS_GRADIENT_FROM_IMAGE_PASS:
                dy16, dyt16, db = ops.half_image_pair(dyf, npad, mp, dt, col_sums=True)
                db = db.to(dbt)
            else:
                dy16, dyt16 = ops.half_image_pair(dyf, npad, mp, dt)
        else:
            dy16 = ops.half_image(dyf, npad, dt) if want_dx else None
            if want_dw and want_db and BIAS_GRADIENT_FROM_IMAGE_PASS and m > 0:
                dyt16, db = ops.half_image(dyf, mp, dt, transpose=True, col_sums=True)
                db = db.to(dbt)
            else:
                dyt16 = ops.half_image(dyf, mp, dt, transpose=True) if want_dw else None
        if want_dx:                          # dX [m, k] = dY [m, n] . (W^T [k, n])^T
            dx = ops.half_gemm(dy16, wt16, n_g=k, k_g=npad).to(dxt)
        if want_dw:                          # dW [n, k] = dY^T [n, m] . (X^T [k, m])^T, split over the rows in one grouped launch
            dw = ops.half_gemm(dyt16, xt16, n_g=k, k_g=chunk, groups=groups, a_group_cols=chunk, b_group_rows=k)
            dw = (dw.view(n, groups, k).sum(1) if groups > 1 else dw).to(dwt)
        if want_db and db is None:
            db = dyf.sum(0).to(dbt)
        return dx, dw, db

    @staticmethod
    def forward(ctx, x, w, b):
        m, k = x.shape
        n = w.shape[0]
        ctx.half = None
        if (AUTOCAST_HALF_PRODUCTS and torch.is_autocast_enabled() and m > 0 and k % 8 == 0 and n % 4 == 0 and k % 4 == 0
                and torch.get_autocast_dtype("cuda") in (torch.float16, torch.bfloat16)):
            return _SplitLinearFunction._forward_half(ctx, x, w, b, torch.get_autocast_dtype("cuda"))
        xf, wf = x.detach().float().contiguous(), w.detach().float().contiguous()
        m, k = xf.shape
        n = wf.shape[0]
        kp = _pad32(k)
        if m == 0:
            ctx.shape = (0, k, n)
            ctx.dtypes = (x.dtype, w.dtype, None if b is None else b.dtype)
            return xf.new_zeros(0, n)
        ax = ops.absmax(xf)
        aw, w_img, wt_img = _SplitLinearFunction._weight_images(w, wf, _pad32(n))
        y = ops.split_gemm_scaled(ops.split_half_scaled(xf, kp, ax), w_img, n_g=n, k_g=kp,
                                  bias=None if b is None else b.detach().float().contiguous(), amax_a=ax, amax_b=aw)
        ctx.save_for_backward(xf, ax, aw, *wt_img)
        ctx.shape = (m, k, n)
        ctx.dtypes = (x.dtype, w.dtype, None if b is None else b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.half is not None:
            return _SplitLinearFunction._backward_half(ctx, dy)
        dxt, dwt, dbt = ctx.dtypes
        dyf = dy.float().contiguous()
        m, k, n = ctx.shape
        if m == 0:                           # no rows (a batch without graph nodes, an empty z): empty / zero gradients, like F.linear
            return (dyf.new_zeros(0, k).to(dxt) if ctx.needs_input_grad[0] else None,
                    dyf.new_zeros(n, k).to(dwt) if ctx.needs_input_grad[1] else None,
                    dyf.new_zeros(n).to(dbt) if (dbt is not None and ctx.needs_input_grad[2]) else None)
        xf, ax, aw, wt_hi, wt_lo = ctx.saved_tensors
        ad = ops.absmax(dyf)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:          # dX [m, k] = dY [m, n] . (W^T [k, n])^T
            npad = _pad32(n)
            dx = ops.split_gemm_scaled(ops.split_half_scaled(dyf, npad, ad), (wt_hi, wt_lo), n_g=k, k_g=npad, amax_a=ad, amax_b=aw).to(dxt)
        if ctx.needs_input_grad[1]:          # dW [n, k] = dY^T [n, m] . (X^T [k, m])^T: both operands transposed, contraction over the rows
            # split-K in one grouped launch: the rows are cut into G chunks, group g multiplies chunk g of both operands into its own
            # [n, k] block, the G blocks are summed (a [768, 768] gradient is nine 256 x 256 tiles: alone they would walk all the
            # rows on nine CUs)
            groups, chunk, mp = _SplitLinearFunction._row_split(m, n, k)
            dw = ops.split_gemm_scaled(ops.split_half_scaled(dyf, mp, ad, transpose=True),
                                       ops.split_half_scaled(xf, mp, ax, transpose=True, group_cols=chunk),
                                       n_g=k, k_g=chunk, amax_a=ad, amax_b=ax, groups=groups, a_group_cols=chunk, b_group_rows=k)
            dw = (dw.view(n, groups, k).sum(1) if groups > 1 else dw).to(dwt)
        if dbt is not None and ctx.needs_input_grad[2]:
            db = dyf.sum(0).to(dbt)
        return dx, dw, db


def split_linear(x, w, b=None):
    return _SplitLinearFunction.apply(x, w, b)


class _HeadBlockDiag(torch.autograd.Function):
    """The block-diagonal matrix of a [H hd, D] projection weight's per-head blocks -- transposed: blocks w_h^T, [H D, H hd] (the fold
    of W_k into the queries); else blocks w_h, [H hd, H D] (W_v on the per-head contexts) -- in two launches (zero fill + one
    strided copy) with a one-launch backward (the diagonal blocks of the gradient, gathered).  torch.block_diag over H slices is
    the same matrix in 5 launches forward and 11 backward (a zero [D, D] tensor, a copy and an add per slice)."""

    @staticmethod
    def forward(ctx, w, heads, transposed):
        hd, dim = w.shape[0] // heads, w.shape[1]
        ctx.cfg = (heads, hd, dim, transposed)
        blocks = w.detach().view(heads, hd, dim)
        if transposed:
            out = w.new_zeros(heads * dim, heads * hd)
            out.view(heads, dim, heads, hd).diagonal(dim1=0, dim2=2).copy_(blocks.permute(2, 1, 0))
        else:
            out = w.new_zeros(heads * hd, heads * dim)
            out.view(heads, hd, heads, dim).diagonal(dim1=0, dim2=2).copy_(blocks.permute(1, 2, 0))
        return out

    @staticmethod
    def backward(ctx, g):
        heads, hd, dim, transposed = ctx.cfg
        if transposed:
            gw = g.view(heads, dim, heads, hd).diagonal(dim1=0, dim2=2).permute(2, 1, 0)
        else:
            gw = g.view(heads, hd, heads, dim).diagonal(dim1=0, dim2=2).permute(2, 0, 1)
        return gw.reshape(heads * hd, dim), None, None


class CrossAttentionLayer(nn.Module):
    """Multi-head attention + residual + LayerNorm (reference :17-51)."""

    def __init__(self, embed_dim, num_heads, dropout=0.1):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.layer_norm = nn.LayerNorm(embed_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        attended, _ = self.multihead_attn(query, key, value, attn_mask=attn_mask,
                                          key_padding_mask=key_padding_mask, need_weights=False)
        return self.layer_norm(query + self.dropout(attended))


class CrossAttention(nn.Module):
    """Stack of cross-attention layers shared by both directions (reference :53-88).

    Quirks kept on purpose: every layer attends to the ORIGINAL other modality
    (:83,:86), and the same layers serve text->graph and graph->text.
    """

    def __init__(self, embed_dim, num_heads, dropout=0.1, layers=2):
        super().__init__()
        self.model = nn.ModuleList([CrossAttentionLayer(embed_dim, num_heads, dropout) for _ in range(layers)])
        # (not in the reference) An upper bound on the nodes of one code, e.g. the dataset's largest subgraph.  With it pooled()
        # sizes its attention launches from the bound instead of reading the batch's largest node count back: no host
        # synchronisation, so VectorQuantizer.forward (eval, show_usage = False) records into a HIP graph at any width, and a
        # training step loses one of its two host reads (0.23 ms of 9 at cfg 4).  `batch` must then be non-decreasing (PyG batch
        # vectors are); a batch that is not, holds ids outside [0, B) or exceeds the bound is flagged on the device (small_status)
        # and raised by check_status() / with the usage counts' read.  None: one host read per call, any batch vector.
        self.max_nodes_bound = None

    def forward(self, vector1, vector2, attn_mask=None):
        """The reference's per-pair call (:53-88): (vector1 attended by vector2, vector2 attended by vector1), every layer against the
        ORIGINAL other modality.  Inference on an MI355X (eval mode, no autograd, unbatched fp32 [L, D] operands, no mask, D <= 768):
        the library's own kernels -- folded projections on the split-fp16 GEMMs, the ragged attention core with the pair as ONE code,
        fused residual + LayerNorm -- within 1e-5 of nn.MultiheadAttention; anything else (training, autograd, masks, batched
        input, CPU) runs the stock modules as in the reference."""
        fast = self._forward_on_kernels(vector1, vector2, attn_mask)
        if fast is not None:
            return fast
        out1, out2 = vector1, vector2
        for layer in self.model:
            out1 = layer(out1, vector2, vector2, attn_mask)
        for layer in self.model:
            out2 = layer(out2, vector1, vector1, attn_mask)
        return out1, out2

    def _forward_on_kernels(self, vector1, vector2, attn_mask):
        mha = self.model[0].multihead_attn
        if (self.training or torch.is_grad_enabled() or attn_mask is not None or not torch.is_tensor(vector1) or not torch.is_tensor(vector2)
                or vector1.dim() != 2 or vector2.dim() != 2 or not (vector1.is_cuda and vector2.is_cuda) or torch.is_autocast_enabled()
                or vector1.dtype != torch.float32 or vector2.dtype != torch.float32 or vector1.shape[1] != vector2.shape[1]
                or vector1.shape[0] == 0 or vector2.shape[0] == 0 or mha.in_proj_bias is None or not mha._qkv_same_embed_dim):
            return None
        dim = vector1.shape[1]
        if dim > ops.ATTENTION_MAX_TRAIN_WIDTH or dim != mha.embed_dim:
            return None
        heads, scale = mha.num_heads, mha.head_dim ** -0.5
        pad = ops.attention_width(dim) - dim
        widen = (lambda t: torch.nn.functional.pad(t, (0, pad))) if pad else (lambda t: t)
        dev = vector1.device

        def attend_to(kv, n_q):
            kvw = widen(kv).contiguous()
            one = lambda v: torch.tensor([v], dtype=torch.long, device=dev)
            q_start, q_len, kv_start, kv_len = one(0), one(n_q * heads), one(0), one(kv.shape[0])

            def attend(qf, split_out=False, **_):
                wide_in = qf.shape[1] != dim                 # _folded_rows_split hands over (and takes back) rows at the kernel width
                q_in = qf if wide_in else widen(qf)
                if split_out:
                    return ops.shared_kv_attention(q_in, q_start, q_len, kvw, kv_start, kv_len, n_q * heads, scale, split_out=True)
                out = ops.shared_kv_attention(q_in.float(), q_start, q_len, kvw, kv_start, kv_len, n_q * heads, scale)
                return out[:, :dim].contiguous() if (pad and not wide_in) else out
            attend.library_core = True
            attend.core_args = lambda: dict(q_start=q_start, q_len=q_len, max_q_len=n_q * heads, kv=kvw, kv_split=None, kv_start=kv_start,
                                            kv_len=kv_len, scale=scale)
            return attend
        with torch.autocast(device_type="cuda", enabled=False):
            outs = []
            for rows, other in ((vector1, vector2), (vector2, vector1)):
                attend = attend_to(other, rows.shape[0])
                cur = rows.contiguous()
                for i, layer in enumerate(self.model):
                    cur = self._folded_rows(layer, cur, attend, next_split=i + 1 < len(self.model))
                outs.append(cur)
        return outs[0], outs[1]

    @staticmethod
    def _folded_layer(layer, query, kv, key_valid):
        """(Comparator form, used by pooled_reference only.)  One CrossAttentionLayer with the key/value projections folded into
        the query side, on PADDED batches in plain torch ops.

        nn.MultiheadAttention projects every key row: 4*T*D^2 flops per layer for T key rows.  With few
        queries against many keys (graph nodes attending to <= 512 text tokens, or the single CLS query
        attending to the nodes) it is far cheaper to move the projections across the dot products:
            q_h . (Wk_h t + bk_h) = (Wk_h^T q_h) . t + const      (const is the same for every key: softmax drops it)
            sum_j p_j (Wv_h t_j + bv_h) = Wv_h (sum_j p_j t_j) + bv_h
        so the keys/values are the RAW rows `kv`, shared by all heads, and the attention core is
        softmax(Qf kv^T) kv with Qf [rows*heads, D].  Same function as the reference layer (:17-51), re-associated;
        agreement is checked against the reference's per-code loop at 1e-5 (tests/test_host_logic.py).
        query [B, R, D], kv [B, T, D], key_valid [B, T] bool.
        """
        mha = layer.multihead_attn
        bsz, rows, dim = query.shape
        heads, hd = mha.num_heads, mha.head_dim
        wq, wk, wv = mha.in_proj_weight.chunk(3)
        bq, _, bv = mha.in_proj_bias.chunk(3)
        q = torch.nn.functional.linear(query, wq, bq).view(bsz, rows, heads, hd)
        qf = torch.einsum("brhd,hdk->brhk", q, wk.view(heads, hd, dim)).reshape(bsz, rows * heads, dim)
        # The attention core runs in fp32 also under autocast: PyTorch 2.10 / hipBLASLt on gfx950 takes a memory fault in bf16
        # strided-batched GEMMs at the training shape ([256, 800, 768] x [256, 512, 768]^T and the transposed forms autograd
        # derives from it; fp32 is fine), and fp32 scores are what the inference kernel computes anyway.
        with torch.autocast(device_type=query.device.type, enabled=False):
            qf32, kv32 = qf.float(), kv.float()
            scores = torch.bmm(qf32, kv32.transpose(1, 2)) * (hd ** -0.5)
            any_valid = key_valid.any(-1)[:, None, None]
            # a code with no valid key: its scores stay finite (0) through the softmax and its probabilities are zeroed after it --
            # a row of -inf would make softmax (and, in training, its gradient) NaN for the whole batch
            scores = torch.where(any_valid, scores.masked_fill(~key_valid[:, None, :], float("-inf")), torch.zeros_like(scores))
            prob = torch.softmax(scores, dim=-1) * any_valid.to(scores.dtype)
            if layer.training and mha.dropout > 0.0:
                prob = torch.nn.functional.dropout(prob, mha.dropout)
            ctx = torch.bmm(prob, kv32).view(bsz, rows, heads, dim)
        attended = torch.einsum("brhk,hdk->brhd", ctx, wv.view(heads, hd, dim)).reshape(bsz, rows, dim) + bv
        attended = mha.out_proj(attended)
        return layer.layer_norm(query + layer.dropout(attended))

    @staticmethod
    def _combined_weights(layer):
        """(Mq [D, heads*D], cq [heads*D], Mo [heads*D, D], co [D]) with
            qf[r, h, :]  = rows[r] @ Mq[:, hD:(h+1)D] + cq[hD:(h+1)D]      (= Wk_h^T (Wq_h x + bq_h))
            out_proj(concat_h(Wv_h ctx_h + bv_h)) = sum_h ctx[r, h, :] @ Mo[hD:(h+1)D, :] + co
        products formed in fp64, cached on the layer per (storage, version) of its four tensors.  A `.data` write that autograd's
        version counter does not see needs VectorQuantizer.invalidate_codebook_cache() (which drops this cache too)."""
        mha = layer.multihead_attn
        params = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias)
        key = tuple((t.data_ptr(), t._version, t.device) for t in params)

        def build():
            heads, hd = mha.num_heads, mha.head_dim
            wq, wk, wv = (t.double() for t in mha.in_proj_weight.detach().chunk(3))
            bq, _, bv = (t.double() for t in mha.in_proj_bias.detach().chunk(3))
            wo, bo = mha.out_proj.weight.detach().double(), mha.out_proj.bias.detach().double()
            sl = [slice(h * hd, (h + 1) * hd) for h in range(heads)]
            mq = torch.cat([wq[s].t() @ wk[s] for s in sl], dim=1)                   # [D, heads * D]
            cq = torch.cat([bq[s] @ wk[s] for s in sl])                              # [heads * D]
            mo = torch.cat([wv[s].t() @ wo[:, s].t() for s in sl], dim=0)            # [heads * D, D]
            co = bv @ wo.t() + bo
            return tuple(t.float().contiguous() for t in (mq, cq, mo, co))
        return _cached(layer, "_medtok_fold_cache", key, build, mha.in_proj_weight.device)

    @staticmethod
    def _split_weights(layer):
        """The layer's four weight matrices as (hi, lo) fp16 image pairs in the layouts medtok_split_gemm_f16 reads (k contiguous,
        depths padded to 32, the head slices padded to hd' = 32 ceil(hd / 32), the model width to Dw = ops.attention_width(D)), each
        prescaled by an exact power of two that brings its largest entry into [2^11, 2^12) -- cached on the layer per (storage,
        version) of its tensors like _combined_weights (a `.data` write needs VectorQuantizer.invalidate_codebook_cache()):
            wq [H hd', Dw]   rows h hd' + j = W_q[h hd + j]            q'  = rows . wq^T + bq'
            wk [H Dw, hd']   row h Dw + c, column j = W_k[h hd + j, c]   qf_h = q'_h . wk_h^T        (group h)
            wv [H hd', Dw]   rows as wq, from W_v                      att_h = ctx_h . wv_h^T + bv' (group h)
            wo [D, H hd']    column h hd' + j = W_o[:, h hd + j]        out = att . wo^T + bo"""
        import math
        mha = layer.multihead_attn
        params = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias)
        key = tuple((t.data_ptr(), t._version, t.device) for t in params)

        def build():
            heads, hd = mha.num_heads, mha.head_dim
            dim = heads * hd
            dw, hp = ops.attention_width(dim), (hd + 31) // 32 * 32
            dev = mha.in_proj_weight.device
            wq, wk, wv = (t.detach().float() for t in mha.in_proj_weight.chunk(3))
            bq, _, bv = (t.detach().float() for t in mha.in_proj_bias.chunk(3))
            wo, bo = mha.out_proj.weight.detach().float(), mha.out_proj.bias.detach().float().contiguous()

            def head_rows(w, b):                    # [D, D] -> [H hd', Dw] and its bias [H hd']
                out = torch.zeros(heads, hp, dw, device=dev)
                out[:, :hd, :dim] = w.view(heads, hd, dim)
                bias = torch.zeros(heads, hp, device=dev)
                bias[:, :hd] = b.view(heads, hd)
                return out.view(heads * hp, dw), bias.view(-1).contiguous()
            m_q, b_q = head_rows(wq, bq)
            m_v, b_v = head_rows(wv, bv)
            m_k = torch.zeros(heads, dw, hp, device=dev)
            m_k[:, :dim, :hd] = wk.view(heads, hd, dim).transpose(1, 2)
            m_o = torch.zeros(dim, heads, hp, device=dev)
            m_o[:, :, :hd] = wo.view(dim, heads, hd)

            def split(w):
                amax = float(w.abs().max())
                scale = 2.0 ** (11 - math.floor(math.log2(amax))) if amax > 0.0 and math.isfinite(amax) else 1.0
                return ops.split_half(w.contiguous(), dp=w.shape[1], scale=scale), 1.0 / scale
            out = dict(heads=heads, hd=hd, hp=hp, dim=dim, dw=dw, wq=split(m_q), bq=b_q, wk=split(m_k.view(heads * dw, hp)),
                       wv=split(m_v), bv=b_v, wo=split(m_o.view(dim, heads * hp)), bo=bo)
            return out
        return _cached(layer, "_medtok_split_cache", key, build, mha.in_proj_weight.device)

    @staticmethod
    def _folded_rows_split(layer, rows, attend, next_split=False):
        """_folded_rows at inference on the library's own dense products: the four projections of the layer run as split-fp16 MFMA
        GEMMs (medtok_split_gemm_f16: fp32-accurate, ~5x the fp32 matrix rate; the per-head fold and the per-head W_v product are
        one grouped launch each), every product hands its result to the next as (hi, lo) fp16 images, the tail is the fused
        residual + LayerNorm kernel.  No library GEMM, nothing padded.  `attend` maps qf [R heads, Dw] to the context [R heads, Dw]."""
        w = CrossAttention._split_weights(layer)
        heads, hp, dim, dw = w["heads"], w["hp"], w["dim"], w["dw"]
        n_rows = rows.shape[0]
        # (the previous layer's tail has already written the images of its output: `next_split`)
        x = getattr(rows, "_medtok_images", None)
        if x is not None and x[0].shape != (n_rows, dw):
            x = None
        core_args = getattr(attend, "core_args", None)
        ln = layer.layer_norm
        if FUSED_LAYER_CALL and core_args is not None:
            a = core_args()
            r = ops.cross_attention_layer(rows, x, w, a["q_start"], a["q_len"], a["max_q_len"], a["kv"], a["kv_split"], a["kv_start"], a["kv_len"],
                                          a["scale"], ATTENTION_VARIANT, ln.weight, ln.bias, ln.eps, want_images=next_split)
            if not next_split:
                return r
            y, images = r
            y._medtok_images = images
            return y
        if x is None:
            x = ops.split_half(rows, dp=dw)
        _, q = ops.split_gemm(x, w["wq"][0], n_g=heads * hp, k_g=dw, bias=w["bq"], unscale=w["wq"][1], want_f32=False, want_split=True)
        qf, _ = ops.split_gemm(q, w["wk"][0], n_g=dw, k_g=hp, groups=heads, a_group_cols=hp, b_group_rows=dw, unscale=w["wk"][1])
        c_hi, c_lo = attend(qf.view(n_rows * heads, dw), split_out=True)       # the kernel writes the (hi, lo) images of the context itself
        c = (c_hi.view(n_rows, heads * dw), c_lo.view(n_rows, heads * dw))
        _, att = ops.split_gemm(c, w["wv"][0], n_g=hp, k_g=dw, groups=heads, a_group_cols=dw, b_group_rows=hp, bias=w["bv"], unscale=w["wv"][1],
                                want_f32=False, want_split=True)
        out, _ = ops.split_gemm(att, w["wo"][0], n_g=dim, k_g=heads * hp, bias=w["bo"], unscale=w["wo"][1])
        if not next_split:
            return ops.residual_layernorm(rows, out, ln.weight, ln.bias, ln.eps)
        y, images = ops.residual_layernorm(rows, out, ln.weight, ln.bias, ln.eps, split_dp=dw)
        y._medtok_images = images              # read by the next layer's first product instead of a split_half pass over y
        return y

    @staticmethod
    def _folded_rows(layer, rows, attend, next_split=False):
        """_folded_layer for PACKED query rows [R, D] (no batch axis, nothing padded): the attention core is
        `attend(qf [R*heads, D]) -> ctx [R*heads, D]`, i.e. the ragged gfx950 kernel (ops.shared_kv_attention in eval;
        _RaggedAttentionFunction with the attention-probability dropout under autograd / in training)."""
        mha = layer.multihead_attn
        n_rows, dim = rows.shape
        heads, hd = mha.num_heads, mha.head_dim
        wq, wk, wv = mha.in_proj_weight.chunk(3)
        bq, _, bv = mha.in_proj_bias.chunk(3)
        plain = not torch.is_grad_enabled() and not torch.is_autocast_enabled() and rows.dtype == wk.dtype
        ln = layer.layer_norm
        if (plain and not layer.training and rows.is_cuda and rows.dtype == torch.float32 and SPLIT_PRODUCTS and dim % 4 == 0
                and ln.elementwise_affine and ln.bias is not None and mha.in_proj_bias is not None and n_rows >= SPLIT_MIN_ROWS
                and getattr(attend, "library_core", False)):
            return CrossAttention._folded_rows_split(layer, rows, attend, next_split)
        if plain and not layer.training and rows.is_cuda and 4.0 * n_rows * dim * dim * max(heads - 2, 0) <= COMBINE_MAX_EXTRA_FLOPS:
            # small widths (the reference's default e_dim = 64) are launch-bound: the query-side chain (in_proj -> fold) and the
            # value-side chain (Wv -> out_proj) each collapse into ONE GEMM against products of the layer's weights, formed once
            # per weight version in eval mode (12 launches per layer -> 4; twice the flops of the two-step form, which is why the
            # wide case keeps that)
            mq, cq, mo, co = CrossAttention._combined_weights(layer)
            qf = torch.addmm(cq, rows, mq)                                       # [R, heads * D] = the kernel's [R * heads, D]
            ctx = attend(qf.view(n_rows * heads, dim))
            out = torch.addmm(co, ctx.view(n_rows, heads * dim), mo)
            ln = layer.layer_norm
            if dim % 4 == 0 and ln.elementwise_affine and ln.bias is not None:
                return ops.residual_layernorm(rows, out, ln.weight, ln.bias, ln.eps)
            return ln(rows + out)
        if (not plain and rows.is_cuda and SPLIT_PRODUCTS and TRAIN_SPLIT_PRODUCTS and dim % 4 == 0 and hd % 4 == 0
                and mha.in_proj_bias is not None):
            # training / autograd: the four dense products and their backward on the library's own split-fp16 GEMMs
            # (_SplitLinearFunction).  The two per-head products become plain linears against BLOCK-DIAGONAL weights built with
            # differentiable torch ops (autograd extracts the per-head gradients itself): 4x the flops of the grouped form on two
            # of the four products -- immaterial at training sizes (a per-GPU batch of 256 codes is launch-bound), and one
            # autograd function instead of four.
            q = split_linear(rows, wq, bq)                                                             # [R, D]
            wk_bd = _HeadBlockDiag.apply(wk, heads, True)                                              # [H D, H hd]: blocks Wk_h^T
            qf = split_linear(q, wk_bd).view(n_rows * heads, dim)                                      # row r, head h: Wk_h^T q_{r,h}
            ctx = attend(qf.contiguous())
            wv_bd = _HeadBlockDiag.apply(wv, heads, False)                                             # [H hd, H D]: blocks Wv_h
            attended = split_linear(ctx.float().view(n_rows, heads * dim), wv_bd, bv)                   # [R, D]
            out = split_linear(attended, mha.out_proj.weight, mha.out_proj.bias)
            with torch.autocast(device_type="cuda", enabled=False):
                return layer.layer_norm(rows.float() + layer.dropout(out))
        q = torch.nn.functional.linear(rows, wq, bq)
        if plain:
            # one GEMM per head straight into / out of the [R, heads, D] layout the kernel works on (strided operands and
            # outputs, leading dimension heads*D or D): einsum's permute-and-copy of these R*heads*D tensors was ~5 % of the
            # forward.  `out=` has no autograd, so only where none is recorded.
            qf3 = rows.new_empty(n_rows, heads, dim)
            for h in range(heads):
                torch.mm(q[:, h * hd:(h + 1) * hd], wk[h * hd:(h + 1) * hd], out=qf3[:, h])
            ctx = attend(qf3.view(n_rows * heads, dim)).view(n_rows, heads, dim)
            attended = rows.new_empty(n_rows, dim)
            for h in range(heads):
                torch.addmm(bv[h * hd:(h + 1) * hd], ctx[:, h], wv[h * hd:(h + 1) * hd].t(), out=attended[:, h * hd:(h + 1) * hd])
        else:
            qf = torch.einsum("rhd,hdk->rhk", q.view(n_rows, heads, hd), wk.view(heads, hd, dim)).reshape(n_rows * heads, dim)
            ctx = attend(qf.contiguous()).view(n_rows, heads, dim)
            attended = torch.einsum("rhk,hdk->rhd", ctx.to(wv.dtype) if torch.is_autocast_enabled() else ctx, wv.view(heads, hd, dim)).reshape(n_rows, dim) + bv
        ln = layer.layer_norm
        if plain and not layer.training and rows.is_cuda and dim % 4 == 0 and dim <= 4096 and ln.elementwise_affine and ln.bias is not None:
            # residual + LayerNorm in one pass over the rows (dropout is the identity in eval)
            return ops.residual_layernorm(rows, mha.out_proj(attended), ln.weight, ln.bias, ln.eps)
        return ln(rows + layer.dropout(mha.out_proj(attended)))

    def _pooled_packed(self, text, valid_len, nodes_sorted, batch_sorted, slot, counts, starts, max_nodes, core, autograd=False, join=True,
                       lists=None, images=None):
        """`pooled` with no padding of rows anywhere: packed query rows, ragged attention core.
        `core(q, q_start, q_len, kv, kv_start, kv_len, max_q_len, scale)` is ops.shared_kv_attention at inference (or, from
        pooled_reference, the oracle's restatement); with `autograd` every call goes through _RaggedAttentionFunction instead
        (HIP forward with the layer's attention dropout in training mode + HIP backward).
        The gfx950 kernels take D = 64 or a multiple of 128 up to 768: any other width D <= 768 runs on the same kernels with
        zero COLUMNS appended to queries and keys (they change neither the scores nor the first D output columns); 768 < D <= 1024
        runs at width 1024, inference only."""
        bsz, seq_len, dim = text.shape
        mha = self.model[0].multihead_attn
        heads, scale = mha.num_heads, mha.head_dim ** -0.5
        dev = text.device
        pad = (ops.attention_width(dim) - dim) if (text.is_cuda or autograd) else 0
        widen = (lambda t: torch.nn.functional.pad(t, (0, pad))) if pad else (lambda t: t)
        key_sink = cls_rows = None
        if autograd and not pad:                           # training: the text rows behind one gradient buffer (KEY_GRADIENT_SINK)
            text = fan_out_text(text)
            fan = getattr(text, "_medtok_fan", None)
            if fan is not None:
                key_sink, cls_rows = fan.sink, fan.cls
        text_flat = text.reshape(bsz * seq_len, dim)
        if nodes_sorted.shape[0] == 0:                     # no graph node anywhere: one zero row no code points at stands in for the key set
            nodes_sorted = text.new_zeros(1, dim)
        half_keys = images is not None and images[0][1] is None      # fp16 text rows, read as they stand (pooled())
        kv_nodes, kv_text = widen(nodes_sorted), widen(text_flat)
        if not autograd and text.is_cuda:                  # the inference kernel is fp32 whatever autocast hands over
            kv_nodes = kv_nodes.float()
            if not half_keys:
                kv_text = kv_text.float()

        def attend(qf, q_start, q_len, kv, kv_start, kv_len, max_q_len, max_kv_len, kv_split=None, split_out=False, sink=None):
            wide_in = qf.shape[1] != dim                   # _folded_rows_split hands over (and takes back) rows at the kernel width
            q_in = qf if wide_in else widen(qf)
            if split_out:                                  # (inference on the library's core only: the kernel writes the (hi, lo) images itself)
                assert qf.shape[1] == dim + pad and not autograd and core is ops.shared_kv_attention
                if kv_split is not None:
                    return ops.shared_kv_attention_split(q_in, q_start, q_len, kv_split, kv_start, kv_len, max_q_len, scale, split_out=True, variant=ATTENTION_VARIANT)
                return ops.shared_kv_attention(q_in, q_start, q_len, kv, kv_start, kv_len, max_q_len, scale, split_out=True)
            if kv_split is not None:                       # wide inference batches: 64-row blocks, keys by LDS-DMA from their fp16 images
                out = ops.shared_kv_attention_split(q_in.float(), q_start, q_len, kv_split, kv_start, kv_len, max_q_len, scale, variant=ATTENTION_VARIANT)
            elif autograd:
                p = float(mha.dropout) if self.training else 0.0
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0.0 else 0      # host RNG: no device sync
                out = _RaggedAttentionFunction.apply(q_in, kv, q_start, q_len, kv_start, kv_len, max_q_len, max_kv_len, scale, p, seed, sink)
            else:
                out = core(q_in.float() if q_in.is_cuda else q_in, q_start, q_len, kv, kv_start, kv_len, max_q_len, scale)
            return out[:, :dim].contiguous() if (pad and not wide_in) else out
        if lists is None:                                  # (pooled_reference with an injected core; pooled() brings them from _pack)
            code = torch.arange(bsz, device=dev, dtype=torch.long)
            lists = dict(t_start=code * heads, t_len=torch.full((bsz,), heads, device=dev, dtype=torch.long),
                         g_start=starts * heads, g_len=counts * heads, tok_start=code * seq_len, g_kv_len=valid_len)
            if LPT_ORDER and not autograd and bsz > 1:
                order = torch.argsort(valid_len, descending=True)
                for key in ("g_start", "g_len", "tok_start", "g_kv_len"):
                    lists[key] = lists[key][order]
        # text side: the CLS row of every code queries that code's nodes
        t_start, t_len = lists["t_start"], lists["t_len"]
        cur = cls_rows if cls_rows is not None else (text[:, 0].float().contiguous() if half_keys else text[:, 0].contiguous())
        lib_core = (not autograd) and core is ops.shared_kv_attention and text.is_cuda     # the split-product layer form needs the library's own core

        def text_attend(qf, **kw):
            return attend(qf, t_start, t_len, kv_nodes, starts, counts, heads, max_nodes, **kw)
        text_attend.library_core = lib_core
        if lib_core:
            text_attend.core_args = lambda: dict(q_start=t_start, q_len=t_len, max_q_len=heads, kv=kv_nodes, kv_split=None, kv_start=starts,
                                                 kv_len=counts, scale=scale)
        side = None
        text_split = images_ready = None
        # (wider than 768 -- BERT-large features -- only the image form runs, whatever the batch size)
        want_images = (not autograd and core is ops.shared_kv_attention and text.is_cuda and kv_text.shape[1] in ops.ATTENTION_SPLIT_WIDTHS
                       and SPLIT_ATTENTION and max_nodes > 0
                       and (nodes_sorted.shape[0] * heads >= SPLIT_ATTENTION_MIN_ROWS or kv_text.shape[1] > ops.ATTENTION_MAX_TRAIN_WIDTH))
        def text_chain(cur):
            for i, layer in enumerate(self.model):
                cur = self._folded_rows(layer, cur, text_attend, next_split=i + 1 < len(self.model))
            return cur
        use_side = lib_core and max_nodes > 0 and 0 < SIDE_STREAM_MIN_CODES <= bsz and not torch.is_grad_enabled()
        if autograd and MERGE_SIDES_IN_TRAINING and max_nodes > 0 and nodes_sorted.dtype == cur.dtype:
            # training: both directions share the layers' weights (:83,86), so the graph side's rows and the text side's CLS rows go
            # through every dense product of a layer TOGETHER -- one launch per product (and per gradient) instead of two, one
            # accumulation into each weight's .grad instead of two; only the attention core runs per side, on its row range
            g_start, g_len, tok_start, g_kv_len = lists["g_start"], lists["g_len"], lists["tok_start"], lists["g_kv_len"]
            n_g = nodes_sorted.shape[0]

            def both_attend(qf, **kw):
                cut = n_g * heads
                if TWO_SIDED_ATTENTION_NODE and not pad and not kw and qf.is_cuda and qf.shape[1] == dim:
                    p = float(mha.dropout) if self.training else 0.0
                    seeds = [int(v) for v in torch.randint(0, 2 ** 31 - 1, (2,))] if p > 0.0 else [0, 0]      # host RNG: no device sync
                    return _TwoSidedAttentionFunction.apply(qf, kv_text, kv_nodes, (g_start, g_len, tok_start, g_kv_len, max_nodes * heads, seq_len),
                                                            (t_start, t_len, starts, counts, heads, max_nodes), cut, scale, p, seeds[0], seeds[1], key_sink)
                # (split, not two slices: its backward is ONE concatenation of the two gradients; a slice's is a zero tensor of the
                # whole [R heads, D] matrix with the gradient copied in, and autograd then adds the two: 97 -> 30 us per layer)
                q_graph, q_text = torch.split(qf, [cut, qf.shape[0] - cut], dim=0)
                return torch.cat([attend(q_graph, g_start, g_len, kv_text, tok_start, g_kv_len, max_nodes * heads, seq_len, sink=key_sink, **kw),
                                  attend(q_text, t_start, t_len, kv_nodes, starts, counts, heads, max_nodes, **kw)], dim=0)
            both_attend.library_core = False
            rows = torch.cat([nodes_sorted, cur], dim=0)
            for layer in self.model:
                rows = self._folded_rows(layer, rows, both_attend)
            g, cur = torch.split(rows, [n_g, rows.shape[0] - n_g], dim=0)
            if TRAIN_SEGMENT_MEAN and g.is_cuda and g.dtype == torch.float32 and dim % 4 == 0:
                gm = _SegmentMeanFunction.apply(g, starts, counts, batch_sorted)      # ordered sums per code (no atomics), one launch
                return (cur, gm) if join else (cur, gm, None)
            if slot is None:
                slot = torch.arange(batch_sorted.numel(), device=batch_sorted.device) - starts[batch_sorted]
            padded = g.new_zeros(bsz, max_nodes, dim)
            padded[batch_sorted, slot] = g                     # deterministic mean (no atomics): pad, sum, divide
            gm = padded.sum(1) / counts.clamp(min=1).unsqueeze(-1).to(g.dtype)
            return (cur, gm) if join else (cur, gm, None)
        if use_side:
            # the text side (one query row per code and head) is independent of the graph side until the shared searches: second
            # stream.  Its launches are issued behind the graph side's first layer (by then the device has a layer of work queued
            # and the host is ahead of it).
            side, main = _side_stream(text.device)
            # everything the text chain reads was born on the main stream and may be dropped by the host before the chain has run
            _lend(side, cur, text, kv_nodes, t_start, t_len, starts, counts)
            if TEXT_CHAIN_AFTER_LAYER < 0:
                with torch.cuda.stream(side):
                    cur = text_chain(cur)
        else:
            cur = text_chain(cur)
        if max_nodes == 0:                                 # nothing to attend from: the node mean of every code is zero
            return (cur, cur.new_zeros(bsz, dim)) if join else (cur, cur.new_zeros(bsz, dim), None)
        # graph side: every node queries the valid text tokens of its code
        g_start, g_len, tok_start, g_kv_len = lists["g_start"], lists["g_len"], lists["tok_start"], lists["g_kv_len"]
        g = nodes_sorted
        # inference on the library's own core: the text rows become (hi, lo) fp16 images ONCE per forward (valid tokens only) -- every
        # query tile of a code and both layers read them
        def make_images():
            nonlocal text_split, images_ready
            if not want_images or text_split is not None:
                return
            if images is not None:                         # (pooled() issued them before anything else)
                text_split, images_ready = images
            elif (KEYS_SPLIT_IN_KERNEL and ATTENTION_VARIANT == 2 and kv_text.dtype == torch.float32 and kv_text.is_contiguous()
                    and kv_text.shape[1] in ops.ATTENTION_HALF_KEY_WIDTHS):
                text_split = kv_text                       # the fp32 rows themselves: the kernel splits every chunk it copies
            else:
                text_split = ops.split_half(kv_text, seg_len=valid_len, seg_rows=seq_len)

        def images_for_use():
            nonlocal images_ready
            make_images()
            if images_ready is not None:                   # first use of the images made on the other stream
                torch.cuda.current_stream(text.device).wait_event(images_ready)
                for t in text_split:
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream(text.device))
                images_ready = None
            return text_split

        def graph_attend(qf, **kw):
            return attend(qf, g_start, g_len, kv_text, tok_start, g_kv_len, max_nodes * heads, seq_len, kv_split=images_for_use(), sink=key_sink, **kw)
        graph_attend.library_core = lib_core
        if lib_core:
            graph_attend.core_args = lambda: dict(q_start=g_start, q_len=g_len, max_q_len=max_nodes * heads, kv=kv_text, kv_split=images_for_use(),
                                                  kv_start=tok_start, kv_len=g_kv_len, scale=scale)
        for i, layer in enumerate(self.model):
            g = self._folded_rows(layer, g, graph_attend, next_split=i + 1 < len(self.model))
            if use_side and i == min(TEXT_CHAIN_AFTER_LAYER, len(self.model) - 1):
                with torch.cuda.stream(side):
                    cur = text_chain(cur)
        pending = None
        if side is not None:
            if join:
                _join_side(side, main, (cur,))
            else:
                pending = (side, main)                     # the caller keeps using the side stream (get_shared_info: the shared-text search)
        if not autograd and not torch.is_grad_enabled() and g.is_cuda and g.dtype == torch.float32 and dim % 4 == 0:
            gm = ops.segment_mean(g, starts, counts)               # rows of a code are adjacent: one ordered chain per column
            return (cur, gm) if join else (cur, gm, pending)
        if TRAIN_SEGMENT_MEAN and g.is_cuda and g.dtype == torch.float32 and dim % 4 == 0:
            gm = _SegmentMeanFunction.apply(g, starts, counts, batch_sorted)          # (the same under autograd)
            return (cur, gm) if join else (cur, gm, pending)
        if slot is None:                                   # (pooled() leaves the in-code position of a node to this fallback)
            slot = torch.arange(batch_sorted.numel(), device=batch_sorted.device) - starts[batch_sorted]
        padded = g.new_zeros(bsz, max_nodes, dim)
        padded[batch_sorted, slot] = g                     # deterministic mean (no atomics): pad, sum, divide
        gm = padded.sum(1) / counts.clamp(min=1).unsqueeze(-1).to(g.dtype)
        return (cur, gm) if join else (cur, gm, pending)

    @staticmethod
    def _pack(text, text_mask, nodes, batch, heads=None, lpt=False):
        """Shared prologue of pooled / pooled_reference: dtype alignment, the codes' node counts and offsets, nodes in code order.
        One host read (the largest node count sizes the launches; the same read validates `batch` and tells whether it is sorted).
        Every small device op that needs no host value is issued IN FRONT of that read -- behind it the device queue is empty and
        each launch would be exposed to the host's dispatch latency (measured at BASELINE sizes: 2 ms of mostly idle device between
        the read and the first dense product).  With `heads`, the (start, length) lists of both attention sides are part of that:
        returned as a dict (`lpt`: the graph side's lists longest key set first)."""
        bsz, seq_len = text.shape[0], text.shape[1]
        if nodes.dtype != text.dtype:                      # autocast hands over bf16 text features and fp32 node features
            common = torch.promote_types(nodes.dtype, text.dtype)
            nodes, text = nodes.to(common), text.to(common)
        valid = text_mask.to(torch.bool)
        valid_len = valid.sum(1)
        batch = batch.reshape(-1).to(torch.long)
        n_nodes = batch.numel()
        dev = batch.device if n_nodes else text.device
        position = torch.arange(n_nodes, device=dev)
        if n_nodes:
            # node counts per code without torch.bincount (which reads the id range back to the host, twice): ids are clamped
            # for the scatter and validated by the ONE host read of this call, which also brings the largest count (it sizes
            # the launch) and whether `batch` is already sorted (PyG batch vectors are: then there is nothing to rank)
            clamped = batch.clamp(0, max(bsz - 1, 0))
            counts = torch.zeros(bsz, dtype=torch.long, device=dev).scatter_add_(0, clamped, torch.ones_like(batch))
            unsorted = (batch[1:] < batch[:-1]).any() if n_nodes > 1 else torch.zeros((), dtype=torch.bool, device=dev)
            stats = torch.stack([counts.max(), batch.min(), batch.max(), unsorted.to(torch.long)])
            starts = torch.cumsum(counts, 0) - counts
            slot = position - starts[clamped]              # (the sorted case, speculatively)
        else:
            counts = torch.zeros(bsz, dtype=torch.long, device=dev)
            starts = torch.zeros(bsz, dtype=torch.long, device=dev)
            slot = position
            stats = None
        lists = None
        if heads is not None:
            code = torch.arange(bsz, device=dev, dtype=torch.long)
            lists = dict(t_start=code * heads, t_len=torch.full((bsz,), heads, device=dev, dtype=torch.long),
                         g_start=starts * heads, g_len=counts * heads, tok_start=code * seq_len, g_kv_len=valid_len)
            if lpt and bsz > 1:
                # longest blocks first: the kernel takes (code, tile) blocks in list order, and a block's time is its key count -- the
                # lists are permuted (the rows they point at are not), so results are the same and the launch has a short tail
                order_k = torch.argsort(valid_len, descending=True)
                for key in ("g_start", "g_len", "tok_start", "g_kv_len"):
                    lists[key] = lists[key][order_k]
        if stats is not None:
            max_nodes, id_lo, id_hi, unsorted = stats.tolist()       # <- the host read
            if id_lo < 0 or id_hi >= bsz:
                raise ValueError(f"pooled(): `batch` must hold code ids in [0, {bsz}); range seen: [{id_lo}, {id_hi}]")
        else:
            max_nodes, unsorted = 0, 0
        if unsorted:                        # nodes of one code need not be contiguous in `batch`: rank them with a stable sort
            order = torch.argsort(batch, stable=True)
            slot = position - starts[batch[order]]
            nodes, batch = nodes[order], batch[order]
        if heads is None:
            return text, valid, nodes, batch, slot, counts, starts, max_nodes
        return text, valid, valid_len, nodes, batch, slot, counts, starts, max_nodes, lists

    def _small_weights(self):
        """The layers' weights in the layout medtok_cross_attention_small_f32 reads (include/medtok_vq.h), [layers, 4 * 64 * 64 + 6 * 64]
        fp32: Wq^T | Wk | Wv^T | Wo^T | bq | bv | bo | ln gamma | ln beta | pad -- cached per (storage, version) of every tensor."""
        params = []
        for layer in self.model:
            mha, ln = layer.multihead_attn, layer.layer_norm
            params += [mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, ln.weight, ln.bias]
        key = tuple((t.data_ptr(), t._version, t.device) for t in params)

        def build():
            rows = []
            for layer in self.model:
                mha, ln = layer.multihead_attn, layer.layer_norm
                wq, wk, wv = (t.detach().float() for t in mha.in_proj_weight.chunk(3))
                bq, _, bv = (t.detach().float() for t in mha.in_proj_bias.chunk(3))
                wo, bo = mha.out_proj.weight.detach().float(), mha.out_proj.bias.detach().float()
                rows.append(torch.cat([wq.t().reshape(-1), wk.reshape(-1), wv.t().reshape(-1), wo.t().reshape(-1), bq, bv, bo,
                                       ln.weight.detach().float(), ln.bias.detach().float(), torch.zeros_like(bo)]))
            return torch.stack(rows).contiguous()
        return _cached(self, "_medtok_small_cache", key, build, params[0].device)

    def small_eligible(self, text, nodes) -> bool:
        """inference at e_dim = 64 with 4 heads on fp32 device tensors: the two-launch path (ops.cross_attention_small)"""
        mha, ln = self.model[0].multihead_attn, self.model[0].layer_norm
        return (SMALL_WIDTH_FUSED and not self.training and not torch.is_grad_enabled() and text.is_cuda and nodes.is_cuda
                and text.dim() == 3 and text.shape[-1] == 64 and mha.num_heads == 4 and mha.in_proj_bias is not None
                and mha._qkv_same_embed_dim and all(l.layer_norm.elementwise_affine and l.layer_norm.bias is not None for l in self.model)
                and all(abs(l.layer_norm.eps - ln.eps) == 0 for l in self.model)
                and text.dtype in (torch.float32, torch.float16, torch.bfloat16) and text.shape[0] > 0)

    def pooled_small(self, text, text_mask, nodes, batch):
        """pooled() for small_eligible() inputs: [B, 2, 64] fp32 = (attended CLS row, mean of the attended nodes) per code, in two
        launches, nothing read back -- the call captures into a HIP graph.  `batch` must be non-decreasing (PyG batch vectors are);
        the kernels OR what they see otherwise into self.small_status (int32 [4] on the device; bit 0: not sorted, bit 1: an id
        outside [0, B)), which check_small_status() turns into the ValueError pooled() raises -- called by VectorQuantizer.forward
        where it synchronises anyway, by the inference driver per batch, or by the caller."""
        bsz, seq_len, dim = text.shape
        dev = text.device
        st = self._status_word(dev)
        with torch.autocast(device_type="cuda", enabled=False):
            pooled2 = torch.empty((bsz, 2, dim), dtype=torch.float32, device=dev)
            heads = self.model[0].multihead_attn.num_heads
            ops.cross_attention_small(text.float().contiguous(), text_mask, nodes.float().contiguous(), batch.reshape(-1).to(torch.long),
                                      self._small_weights(), len(self.model), (dim // heads) ** -0.5, self.model[0].layer_norm.eps, pooled2, st)
        return pooled2

    # The paths without a host read (the two-launch small-width path; any width with max_nodes_bound) cannot sort an unsorted `batch`
    # vector themselves.  False (default): the reference's "any batch vector" semantics (`batch == idx`, :135) are kept -- the forward
    # verifies the device word where it reads the host anyway (or with one 4-byte read at its end) and, for an unsorted vector, runs
    # again on the stably sorted nodes; out-of-range ids raise.  True: the caller guarantees a non-decreasing (PyG-style) vector, nothing
    # is read back (what a HIP-graph capture needs; check_status() remains available).
    assume_sorted_batch = False

    def _status_word(self, dev):
        st = getattr(self, "small_status", None)
        if st is None or st.device != dev:
            if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                # (a zero-fill recorded into the graph would erase earlier flags on every replay)
                raise RuntimeError("CrossAttention: the status word must exist before a HIP-graph capture: run one warm-up forward first")
            st = self.small_status = torch.zeros(4, dtype=torch.int32, device=dev)
        return st

    def take_status(self, word=None):
        """Host side of the device checks: reads (unless `word` was read already) and clears small_status.  Returns True when the ONLY
        finding is an unsorted `batch` vector -- the caller then runs again on stably sorted nodes -- False when there is none; raises
        pooled()'s ValueError for ids outside [0, B) or a node count above max_nodes_bound."""
        st = getattr(self, "small_status", None)
        if st is None:
            return False
        if word is None:
            word = int(st[0].item())
        if not word:
            return False
        st.zero_()
        if word & 2:
            raise ValueError("pooled(): `batch` holds code ids outside [0, B)")
        if word & 4:
            raise ValueError(f"pooled(): a code has more nodes than max_nodes_bound = {getattr(self, 'max_nodes_bound', None)}; raise the bound, or set it to "
                             "None (one host read per call)")
        return True

    @staticmethod
    def sort_by_code(nodes, batch):
        """nodes / batch in code order, in-code order kept (stable): what `z_graph[batch == idx]` selects (:135)"""
        batch = batch.reshape(-1).to(torch.long)
        order = torch.argsort(batch, stable=True)
        return nodes[order], batch[order]

    def check_status(self):
        """Host read of small_status (a synchronisation): raises what pooled() raises for a batch vector the paths without a host
        read (the two-launch small-width path; any width with max_nodes_bound set) cannot take, and clears the word."""
        st = getattr(self, "small_status", None)
        if st is None:
            return
        word = int(st[0].item())
        if word:
            st.zero_()
            if word & 2:
                raise ValueError("pooled(): `batch` holds code ids outside [0, B)")
            if word & 4:
                raise ValueError(f"pooled(): a code has more nodes than max_nodes_bound = {getattr(self, 'max_nodes_bound', None)}; raise the bound, or set it to "
                                 "None (one host read per call)")
            raise ValueError(UNSORTED_BATCH_MESSAGE)

    check_small_status = check_status

    def prepack(self, text_mask, batch):
        """(not in the reference) Issue pooled()'s prologue -- the counts / offsets / launch lists of all codes -- NOW, and start the copy of
        its four host-read values behind an event of their own.  A caller that knows `text_mask` and `batch` before it queues other
        work (MultimodalTokenizer.forward: before the text mapping's 131 072-row product) calls this first: pooled() then waits for
        that event instead of draining the whole queue in front of its read (0.1 - 0.5 ms of a 9 ms training step, by the host's
        speed).  The next pooled() call on the same two tensors takes the result; any other call ignores it.  No-op off the GPU,
        with max_nodes_bound set (nothing is read then) or during a HIP-graph capture."""
        self._prepacked = None
        if not (PREPACK_CODES and torch.is_tensor(text_mask) and torch.is_tensor(batch) and text_mask.is_cuda and batch.is_cuda and text_mask.dim() == 2
                and getattr(self, "max_nodes_bound", None) is None and not torch.cuda.is_current_stream_capturing()):
            return
        mha = self.model[0].multihead_attn
        if SMALL_WIDTH_FUSED and not self.training and not torch.is_grad_enabled() and mha.embed_dim == 64 and mha.num_heads == 4:
            return                          # (the two-launch small-width path has no prologue and no host read)
        b = batch.reshape(-1).to(torch.long)
        lpt = LPT_ORDER and not (self.training or torch.is_grad_enabled())
        pk = ops.pack_codes(text_mask, b, self.model[0].multihead_attn.num_heads, lpt)
        host = getattr(self, "_prepack_host", None)
        if host is None:
            host = self._prepack_host = torch.empty(4, dtype=torch.int64, pin_memory=True)
        host.copy_(pk["stats"], non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(text_mask.device))
        self._prepacked = ((text_mask.data_ptr(), text_mask._version, tuple(text_mask.shape), b.data_ptr(), b._version, b.numel(), bool(lpt),
                            torch.cuda.current_stream(text_mask.device).cuda_stream), pk, host, done)

    def _take_prepacked(self, text_mask, batch, lpt):
        pre, self._prepacked = getattr(self, "_prepacked", None), None
        if pre is None or not (torch.is_tensor(text_mask) and text_mask.is_cuda):
            return None
        key = (text_mask.data_ptr(), text_mask._version, tuple(text_mask.shape), batch.data_ptr(), batch._version, batch.numel(), bool(lpt),
               torch.cuda.current_stream(text_mask.device).cuda_stream)          # (the same inputs, on the stream the prologue was queued on)
        return pre[1:] if pre[0] == key else None

    def pooled(self, text, text_mask, nodes, batch, join=True):
        """Batched equivalent of the reference's per-code loop (:133-142) -- the PRODUCT path: gfx950 kernels only.

        text [B, L, D] with a left-aligned mask [B, L]; nodes [sum n_i, D] with a PyG-style `batch` vector, all on an MI355X.
        Returns (CLS row of the attended text [B, D], mean of the attended graph nodes [B, D]).  Queries never interact, so the
        text side only evaluates its CLS query.  Everything runs packed (nothing padded to [B, max, D]) on the ragged attention
        kernels with the key/value projections folded into the queries: ops.shared_kv_attention at inference (fp32 whatever
        autocast says), _RaggedAttentionFunction (HIP forward with dropout + HIP backward) in training / under autograd.  Widths
        the kernels do not take natively (not 64 / a multiple of 128 / 1024) get zero columns appended; D > 1024 raises
        MedTokLibraryError (768 < D <= 1024: inference only, at most 4 heads) -- there is no eager-PyTorch fallback (pooled_reference below is the test-side comparator).
        One host sync per call.  A code with no nodes, or no valid token, attends to nothing: its context is zero (the
        reference's per-code loop would take a softmax over an empty row there).
        join=False (get_shared_info): returns a third value, (side stream, main stream) or None -- when the text side ran on the
        second stream, it is NOT joined yet: the pooled text rows may only be used on that stream until _join_side()."""
        if not (text.is_cuda and nodes.is_cuda):
            raise ops.MedTokLibraryError(f"CrossAttention.pooled: expected tensors on an MI355X (cuda/HIP) device, got {text.device} / "
                                         f"{nodes.device}; medtok_amd has no CPU path")
        width = ops.attention_width(text.shape[-1])        # raises for widths the kernels cannot take
        needs_grad = torch.is_grad_enabled() and (text.requires_grad or nodes.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        autograd = self.training or needs_grad
        if width > ops.ATTENTION_MAX_TRAIN_WIDTH:
            if autograd:
                raise ops.MedTokLibraryError(f"CrossAttention.pooled: width D = {text.shape[-1]} runs at inference only (eval mode under torch.no_grad()); "
                                             f"the training / autograd kernels take D <= {ops.ATTENTION_MAX_TRAIN_WIDTH}")
            if self.model[0].multihead_attn.num_heads > 4:
                raise ops.MedTokLibraryError(f"CrossAttention.pooled: width D = {text.shape[-1]} takes at most 4 heads (the text side's kernel keeps "
                                             f"one query row per head in registers)")
        heads = self.model[0].multihead_attn.num_heads
        bsz, seq_len, dim = text.shape
        if bsz == 0:
            z = text.new_zeros(0, dim)
            return (z, z) if join else (z, z, None)
        # fp16 text features (a caller under fp16 autocast, the reference's default mode: train_MedTok.py:212,394) ARE the hi image of
        # the graph side's keys and have no lo part: no image pass, half the key bytes, two matrix passes per product instead of three
        half_keys = (not autograd and not torch.is_grad_enabled() and text.dtype == torch.float16 and text.is_contiguous()
                     and SPLIT_ATTENTION and ATTENTION_VARIANT == 2 and dim in ops.ATTENTION_HALF_KEY_WIDTHS
                     and nodes.shape[0] * heads >= SPLIT_ATTENTION_MIN_ROWS)
        if half_keys:
            nodes = nodes.float()                          # (queries and the text side's keys: fp32)
        elif nodes.dtype != text.dtype:                    # autocast hands over bf16 text features and fp32 node features
            common = torch.promote_types(nodes.dtype, text.dtype)
            nodes, text = nodes.to(common), text.to(common)
        batch = batch.reshape(-1).to(torch.long)
        # counts / offsets / launch lists of all codes: three small launches (ops.pack_codes), nothing read back yet
        # (training too since round 6: the same device-side checks; a trainer reads them with the usage counts -- VectorQuantizer.forward
        # does -- or calls check_status().  What differs from inference: a flagged batch raises, nothing is repeated on sorted nodes.)
        bound = getattr(self, "max_nodes_bound", None)      # (getattr: a module pickled before the attribute existed)
        early = None
        if bound is not None:
            bound = int(bound)
            if bound <= 0:
                raise ValueError(f"max_nodes_bound = {bound} must be a positive node count (or None)")
            pk = ops.pack_codes(text_mask, batch, heads, LPT_ORDER and not autograd, count_bound=bound, status=self._status_word(text.device))
        else:
            early = self._take_prepacked(text_mask, batch, LPT_ORDER and not autograd)
            pk = early[0] if early is not None else ops.pack_codes(text_mask, batch, heads, LPT_ORDER and not autograd)
        images = None
        if half_keys:
            images = ((text.view(bsz * seq_len, dim), None), None)
        elif (not autograd and not torch.is_grad_enabled() and SPLIT_ATTENTION and 0 < SIDE_STREAM_MIN_CODES <= bsz
                and not (KEYS_SPLIT_IN_KERNEL and ATTENTION_VARIANT == 2 and dim in ops.ATTENTION_HALF_KEY_WIDTHS)
                and text.dtype == torch.float32 and nodes.dtype == torch.float32 and text.is_contiguous()
                and dim in ops.ATTENTION_SPLIT_WIDTHS and nodes.shape[0] * heads >= SPLIT_ATTENTION_MIN_ROWS and nodes.shape[0] > 0):
            # The (hi, lo) fp16 images of the valid text rows -- the keys of the graph side, an HBM-bound pass over the whole text
            # batch (1.3 ms at BASELINE sizes) that needs nothing but the token counts: on a stream of its own, under the host
            # read below and the first dense products.
            image_stream, _ = _side_stream(text.device, 2)
            _lend(image_stream, pk["valid_len"])
            with torch.cuda.stream(image_stream):
                text_split = ops.split_half(text.view(bsz * seq_len, dim), seg_len=pk["valid_len"], seg_rows=seq_len)
                ready = torch.cuda.Event()
                ready.record(image_stream)
            images = (text_split, ready)
        # ONE host read per call: the largest node count sizes the launches; the same read validates `batch` and tells whether it
        # is sorted (PyG batch vectors are)
        if bound is not None:               # ... or none: launches sized from the bound, the checks left to the device word (check_status)
            max_nodes, id_lo, id_hi, unsorted = min(bound, batch.numel()), 0, bsz - 1, 0
        elif early is not None:             # (prepack(): the counts reached the host behind an event of their own, not behind the queue)
            early[2].synchronize()
            max_nodes, id_lo, id_hi, unsorted = early[1].tolist()
        else:
            max_nodes, id_lo, id_hi, unsorted = pk["stats"].tolist()
        if batch.numel() == 0:
            max_nodes, unsorted = 0, 0
        elif id_lo < 0 or id_hi >= bsz:
            raise ValueError(f"pooled(): `batch` must hold code ids in [0, {bsz}); range seen: [{id_lo}, {id_hi}]")
        if unsorted:                        # nodes of one code need not be contiguous in `batch`: bring them together (stable)
            order = torch.argsort(batch, stable=True)
            nodes, batch = nodes[order], batch[order]
        lists = {k: pk[k] for k in ("t_start", "t_len", "g_start", "g_len", "tok_start", "g_kv_len")}
        if autograd:
            return self._pooled_packed(text.contiguous(), pk["valid_len"], nodes.contiguous(), batch, None, pk["counts"], pk["starts"],
                                       max_nodes, ops.shared_kv_attention, autograd=True, join=join, lists=lists, images=images)
        with torch.autocast(device_type="cuda", enabled=False):     # inference: fp32(-equivalent) arithmetic whatever autocast says
            return self._pooled_packed(text.contiguous(), pk["valid_len"], nodes.contiguous(), batch, None, pk["counts"], pk["starts"],
                                       max_nodes, ops.shared_kv_attention, autograd=False, join=join, lists=lists, images=images)

    def pooled_reference(self, text, text_mask, nodes, batch, fold=None, core=None):
        """TEST-SIDE COMPARATOR, never called by the product path: the same function as pooled() in plain torch ops on any device
        -- padded [B, max_nodes, D] batches with nn.MultiheadAttention (`fold=False`: projected keys) or the folded-projection
        algebra (`fold=True`), or the packed form around an injected attention core (`core`: the oracle's restatement of the
        ragged kernel; pins the packing logic on CPU).  `fold=None` picks the cheaper padded form from the shapes."""
        text, valid, nodes_in_order, batch_in_order, slot, counts, starts, max_nodes = self._pack(text, text_mask, nodes, batch)
        bsz, seq_len, dim = text.shape
        heads = self.model[0].multihead_attn.num_heads
        if core is not None:
            return self._pooled_packed(text.contiguous(), valid.sum(1), nodes_in_order.contiguous(), batch_in_order, slot, counts, starts,
                                       max_nodes, core, autograd=False)
        if fold is None:
            # flops: 4 n D^2 + 4 n H T D folded  vs  4 T D^2 + 4 n T D projected
            fold = max_nodes * (dim + (heads - 1) * seq_len) < seq_len * dim

        padded = text.new_zeros(bsz, max_nodes, text.shape[-1])
        padded[batch_in_order, slot] = nodes_in_order
        node_valid = torch.arange(max_nodes, device=batch_in_order.device)[None, :] < counts[:, None]

        # text side: one CLS query per code against that code's nodes -- always cheaper folded
        q_text = text[:, :1]
        for layer in self.model:
            q_text = self._folded_layer(layer, q_text, padded, node_valid)
        pooled_text = q_text[:, 0]

        # graph side: the nodes of a code query its text tokens
        if fold:
            q_graph = padded
            for layer in self.model:
                q_graph = self._folded_layer(layer, q_graph, text, valid)
            w = node_valid.unsqueeze(-1).to(q_graph.dtype)
            pooled_graph = (q_graph * w).sum(1) / counts.clamp(min=1).unsqueeze(-1).to(q_graph.dtype)
        else:
            t_kv = text.transpose(0, 1)                    # (L, B, D) keys/values for graph queries
            q_graph = padded.transpose(0, 1)               # (M, B, D)
            for layer in self.model:
                q_graph = layer(q_graph, t_kv, t_kv, key_padding_mask=~valid)
            w = node_valid.transpose(0, 1).unsqueeze(-1).to(q_graph.dtype)
            pooled_graph = (q_graph * w).sum(0) / counts.clamp(min=1).unsqueeze(-1).to(q_graph.dtype)
        return pooled_text, pooled_graph


class _SoftVQFunction(torch.autograd.Function):
    """One search in train mode: (zq_ste, vq, commit, xhat, idx, w) from projected rows x and a codebook region.

    Forward is the fused gfx950 path (rownorm -> search -> soft assign, medtok_soft_vq_forward_f32) plus the
    fixed-order loss reduction.  Backward is ONE sparse kernel (medtok_soft_vq_backward_f32): per row only the k
    selected codes carry gradient -- the dense N x K matrix the reference's autograd graph differentiates
    (:120-125,157-182,203-214) is exactly zero everywhere else.  The per-(row, slot) code gradients are summed per
    code in row order by the EMA-statistics kernels (no atomics: bit-reproducible), then taken through F.normalize.
    The upstream gradients of vq / commit stay on the device (0-dim tensors); nothing synchronises the host."""

    @staticmethod
    def forward(ctx, x, weight, what, wsq, topk, path, beta):
        ctx.set_materialize_grads(False)
        r = ops.soft_vq_forward(x.detach(), what, wsq, topk, path, want_sqerr=True)
        n, d = x.shape
        vq = ops.sum_scale(r["row_sqerr"], (1.0 / (n * d)) if n else float("nan"))      # mean of nothing: nan, like F.mse_loss
        commit = ops.sum_scale(r["row_sqerr"], beta / (n * d))
        ctx.save_for_backward(x, weight, r["xhat"], what, r["idx"], r["w"])
        ctx.beta = beta
        ctx.mark_non_differentiable(r["idx"], r["w"])
        return r["zq"], vq, commit, r["xhat"], r["idx"], r["w"]

    @staticmethod
    def backward(ctx, g_zq_ste, g_vq, g_commit, g_xhat, _gi, _gw):
        x, weight, xhat, what, idx, w = ctx.saved_tensors
        n, d = x.shape
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        as_f32 = lambda t: None if t is None else t.float()
        gx, g_code = ops.soft_vq_backward(x.detach(), xhat, what, idx, w, g_xhat=as_f32(g_xhat), g_out=as_f32(g_zq_ste),
                                          g_vq=as_f32(g_vq), g_commit=as_f32(g_commit),
                                          vq_scale=2.0 / (n * d), commit_scale=2.0 * ctx.beta / (n * d),
                                          want_gx=want_x, want_g_code=want_w)
        gw = None
        if want_w:
            bins, g_what = ops.ema_stats(g_code, idx.reshape(-1), what.shape[0])
            gw = ops.normalize_backward(g_what, what, weight.detach(), live=bins)
        return gx, gw, None, None, None, None, None


def _lib_multi_max():
    from . import _lib
    return _lib.MULTI_SEARCH_MAX


class _SoftVQMultiFunction(torch.autograd.Function):
    """All searches of a training forward under ONE autograd node: (zq_ste, vq, commit, xhat, idx, w) per search from its rows and its
    region of the codebook -- the per-search forward of _SoftVQFunction, search by search.  What changes is the BACKWARD of the codebook:
    the per-(row, slot) code gradients of all searches are summed per code in ONE segmented sum over global code ids and taken through
    F.normalize once, so the weight receives ONE dense gradient.  (Six _SoftVQFunction nodes on six slices of the weight hand autograd
    six dense [n_e, D] gradients -- a zero fill and a copy each for the slices -- that it then adds up: 1.9 ms of fills and adds per
    step at n_e = 49152, D = 768.)"""

    @staticmethod
    def forward(ctx, weight, what, wsq, topk, path, beta, regions, *xs):
        ctx.set_materialize_grads(False)
        outs, saved, nondiff = [], [], []
        # a per-GPU batch of searches (at most 4096 rows each, the exact path): ONE call of three launches for all of them
        # (ops.soft_vq_forward_multi: the per-search bits) instead of five launches per search
        batched = None
        if (TRAIN_BATCHED_SEARCHES and path in (ops.PATH_AUTO, ops.PATH_F32_MFMA) and 1 <= len(xs) <= _lib_multi_max() and topk <= 8
                and all(x.is_cuda and x.shape[0] > 0 and ops.multi_search_eligible(x.shape[0], hi - lo, x.shape[1], topk) for x, (lo, hi) in zip(xs, regions))):
            batched = ops.soft_vq_forward_multi([dict(x=x.detach().float(), what=what[lo:hi], wsq=wsq[lo:hi].contiguous()) for x, (lo, hi) in zip(xs, regions)],
                                                topk, want_sqerr=True)
        # the searches of ONE region (both shared ones; a modality's two views) as one call on their rows stacked: the region's codes are
        # streamed once for both and five launches serve two searches (any row count gives every row the same bits: the library's plans
        # only cut the code axis, and the per-split lists are joined in the (distance, index) total order)
        stacked = {}
        if batched is None and TRAIN_STACK_SEARCHES_OF_A_REGION and all(x.is_cuda and x.dim() == 2 for x in xs):
            by_region = {}
            for i, reg in enumerate(regions):
                by_region.setdefault((tuple(reg), xs[i].shape[1], xs[i].dtype), []).append(i)
            for (reg, _, _), members in by_region.items():
                if len(members) > 1 and all(xs[i].shape[0] > 0 for i in members):
                    lo, hi = reg
                    r = ops.soft_vq_forward(torch.cat([xs[i].detach() for i in members]), what[lo:hi], wsq[lo:hi].contiguous(), topk, path, want_sqerr=True)
                    a = 0
                    for i in members:
                        b = a + xs[i].shape[0]
                        stacked[i] = {k: (v[a:b] if torch.is_tensor(v) else v) for k, v in r.items()}
                        a = b
        for i, (x, (lo, hi)) in enumerate(zip(xs, regions)):
            r = (batched[i] if batched is not None else stacked[i] if i in stacked else
                 ops.soft_vq_forward(x.detach(), what[lo:hi], wsq[lo:hi].contiguous(), topk, path, want_sqerr=True))
            n, d = x.shape
            outs += [r["zq"], ops.sum_scale(r["row_sqerr"], (1.0 / (n * d)) if n else float("nan")), ops.sum_scale(r["row_sqerr"], beta / (n * d) if n else float("nan")),
                     r["xhat"], r["idx"], r["w"]]
            saved += [x, r["xhat"], r["idx"], r["w"]]
            nondiff += [r["idx"], r["w"]]
        ctx.save_for_backward(weight, what, *saved)
        ctx.regions, ctx.beta, ctx.m = tuple(regions), beta, len(xs)
        ctx.mark_non_differentiable(*nondiff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        weight, what, *saved = ctx.saved_tensors
        want_w = ctx.needs_input_grad[0]
        as_f32 = lambda t: None if t is None else t.float()
        gxs, g_codes, ids = [], [], []
        for i in range(ctx.m):
            x, xhat, idx, w = saved[4 * i: 4 * i + 4]
            lo, hi = ctx.regions[i]
            g_zq, g_vq, g_commit, g_xhat = grads[6 * i: 6 * i + 4]
            n, d = x.shape
            gx, g_code = ops.soft_vq_backward(x.detach(), xhat, what[lo:hi], idx, w, g_xhat=as_f32(g_xhat), g_out=as_f32(g_zq), g_vq=as_f32(g_vq),
                                              g_commit=as_f32(g_commit), vq_scale=2.0 / (n * d), commit_scale=2.0 * ctx.beta / (n * d),
                                              want_gx=ctx.needs_input_grad[7 + i], want_g_code=want_w)
            gxs.append(gx)
            if want_w and n:
                g_codes.append(g_code)
                ids.append(idx.reshape(-1) + lo if lo else idx.reshape(-1))
        gw = None
        if want_w:
            if g_codes:
                bins, g_what = ops.ema_stats(torch.cat(g_codes), torch.cat(ids), what.shape[0])
                # (codes no row selected: bins = 0, their gradient rows are zeros -- written without reading the codebook)
                gw = ops.normalize_backward(g_what, what, weight.detach(), live=bins)
            else:
                gw = torch.zeros_like(weight)
        return (gw, None, None, None, None, None, None, *gxs)


class _Norm(tuple):
    """(what, wsq) of the normalised codebook; `prepared`: ops.prepare_codebook's entry per region ("text", "graph", "shared") when a
    search of the caller takes the fp16 shortlist, else None.  Read-only once in the cache."""
    prepared = None


class VectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta, entropy_loss_ratio, l2_norm, show_usage, split, kmeans=False,
                 num_head=4, k=5):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.entropy_loss_ratio = entropy_loss_ratio
        self.l2_norm = l2_norm
        self.show_usage = show_usage
        self.split = split
        if not 1 <= int(k) <= ops.MAX_TOPK or int(k) > n_e // 3:
            raise ValueError(f"VectorQuantizer: k={k} unsupported -- the search kernels keep lists of up to {ops.MAX_TOPK} codes per row "
                             f"(reference default 5), and a modality region holds n_e // 3 = {n_e // 3} codes")
        self.k = k
        self.kmeans_init = kmeans
        self.initted = False
        self.search_path = ops.PATH_AUTO
        if not l2_norm:
            raise NotImplementedError("the reference only defines the l2_norm=True path (:147-151,194-200)")
        if split[0] != e_dim or split[1] != e_dim:
            raise ValueError("e_dim must equal both split sizes (the reference's projections map split[i] -> e_dim and search e_dim)")
        # The kernels read rows as 16-byte groups.  A width that is not a multiple of 4 (the reference takes any e_dim, :91) is searched
        # with zero COLUMNS appended to rows and codes -- they change no norm, no dot product, no distance -- and sliced off the results.
        self._pad = (-int(e_dim)) % 4

        self.cross_attn = CrossAttention(e_dim, num_head, dropout=0.1, layers=2)
        self.proj_text = nn.Linear(self.split[0], e_dim)
        self.proj_graph = nn.Linear(self.split[1], e_dim)
        if self.kmeans_init:
            self.codebook = EmbeddingEMA(self.n_e, self.split[0])
        else:
            self.codebook = nn.Embedding(self.n_e, self.e_dim)
        if self.show_usage:
            self.register_buffer("codebook_used", torch.zeros(USAGE_WINDOW))
        self._norm_cache = None

    # ------------------------------------------------------------------ codebook views
    def _region(self, types):
        region = self.codebook.weight.shape[0] // 3
        if types == "text":
            return 0, region
        if types == "graph":
            return self.n_e - region, self.n_e
        return 0, self.n_e

    def invalidate_codebook_cache(self):
        """Drop the cached normalised codebook.  Call after writing the weight in a way autograd's version counter
        does not see (`codebook.weight.data.copy_()/.uniform_()`, a master-weight copy-back): such writes leave
        `_version` unchanged, so an eval-mode cache cannot notice them by itself."""
        self._norm_cache = None
        for layer in self.cross_attn.model:           # the folded-weight products of the cross-attention layers
            layer._medtok_fold_cache = None
            layer._medtok_split_cache = None
        self.proj_text._medtok_split_cache = None
        self.proj_graph._medtok_split_cache = None
        self._medtok_proj_both_cache = None

    def train(self, mode: bool = True):
        if mode != self.training:
            self._norm_cache = None          # a mode switch is where weights typically change hands (an eval()/train(False) that
        return super().train(mode)           # changes nothing keeps the eval-mode cache: quantize_pooled / tokenize call it per batch)

    def _load_from_state_dict(self, *args, **kwargs):
        self._norm_cache = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _normalised_codebook(self, rebuild=None, prepare=False):
        """normalize(codebook.weight) and its row norms.  The reference re-normalises on every call (:148,198,200);
        so does this in training mode (one K x D pass, ~60 us at n_e = 49152, D = 768 -- and every optimizer step
        changes the weight anyway): forward() re-normalises once and hands the result to its 4-6 searches, a search
        called on its own re-normalises itself.  In eval mode the result is cached per (storage, version) and
        dropped on train()/eval() switches, load_state_dict and invalidate_codebook_cache()."""
        wt = self.codebook.weight
        key = (wt.data_ptr(), wt._version, wt.device, wt.shape)
        stale = self.training if rebuild is None else rebuild
        # prepare (inference, some search of the caller takes the fp16 shortlist): the three regions' fp16 image / start values /
        # largest norm come out of the same two launches, once per weight version, instead of three passes over its region in
        # every search (ops.prepare_codebook)
        prepare = (prepare and PREPARED_CODEBOOK and not self.training and not torch.is_grad_enabled() and wt.is_cuda
                   and wt.dtype == torch.float32 and self.e_dim % 4 == 0)
        regions = lambda: {t: self._region(t) for t in ("text", "graph", "shared")}

        def build():
            if not prepare:
                w_in = wt.detach()
                if getattr(self, "_pad", 0):
                    w_in = torch.nn.functional.pad(w_in.float(), (0, self._pad))
                return _Norm(ops.rownorm(w_in))
            what, wsq, prepared = ops.prepare_codebook(wt.detach(), regions())
            norm = _Norm((what, wsq))
            norm.prepared = prepared
            return norm
        norm = _cached(self, "_norm_cache", key, build, wt.device, rebuild=stale)
        if prepare and getattr(norm, "prepared", None) is None:
            def upgrade(old=norm):           # an entry built by a caller that had no use for the image: add it (a new entry: readers
                _, _, prepared = ops.prepare_codebook(None, regions(), normalised=(old[0], old[1]))     # on other streams wait for it)
                new = _Norm((old[0], old[1]))
                new.prepared = prepared
                return new
            norm = _cached(self, "_norm_cache", key, upgrade, wt.device, rebuild=True)
        return norm

    def project(self, x, types):
        """proj_text / proj_graph (reference :190,192: nn.Linear(split[i], e_dim)).  Inference on wide batches: the library's own
        split-fp16 product (medtok_split_gemm_f16: fp32-accurate, ~2.5x the library fp32 GEMM; weights split once per weight
        version) -- anything else (training, autograd, autocast, small batches, CPU): the nn.Linear as it stands."""
        lin = self.proj_text if types == "text" else self.proj_graph
        if (SPLIT_PRODUCTS and TRAIN_SPLIT_PRODUCTS and self.training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and lin.bias is not None
                and lin.in_features % 4 == 0 and lin.out_features % 4 == 0 and (x.requires_grad or lin.weight.requires_grad)):
            return split_linear(x, lin.weight, lin.bias)        # under autograd: forward and backward on the library's own product
        # (inference: fp32-accurate whatever autocast says -- an autocast caller's half-precision rows are widened, like fp16 tensors)
        if not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
            x = x.float()
        if (not SPLIT_PRODUCTS or self.training or torch.is_grad_enabled() or not x.is_cuda
                or x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] < SPLIT_MIN_ROWS or lin.bias is None
                or lin.in_features % 32 or lin.out_features % 4 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16):
            return lin(x)
        key = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version, lin.weight.device)

        def build():
            import math
            w = lin.weight.detach().float().contiguous()
            amax = float(w.abs().max())
            scale = 2.0 ** (11 - math.floor(math.log2(amax))) if amax > 0.0 and math.isfinite(amax) else 1.0
            return ops.split_half(w, dp=w.shape[1], scale=scale), 1.0 / scale, lin.bias.detach().float().contiguous()
        w_split, unscale, bias = _cached(lin, "_medtok_split_cache", key, build, lin.weight.device)
        out, _ = ops.split_gemm(ops.split_half(x), w_split, n_g=lin.out_features, k_g=lin.in_features, bias=bias, unscale=unscale)
        return out

    def project_both(self, z):
        """proj_text and proj_graph of z = [text | graph] ([B, 2 e]) as ONE grouped product (two launches: the images of z, the
        grouped GEMM) instead of two products of two launches each -- the same kernel on the same operands, group by group, as two
        project() calls.  Returns [B, 2 e] (text projection | graph projection), or None where project()'s split path does not
        apply (training, autograd, autocast, non-fp32, odd shapes)."""
        lt, lg = self.proj_text, self.proj_graph
        e = self.e_dim
        if (not SPLIT_PRODUCTS or self.training or torch.is_grad_enabled() or not z.is_cuda or z.dtype != torch.float32
                or z.dim() != 2 or z.shape[1] != 2 * e or lt.bias is None or lg.bias is None or e % 32 or not z.is_contiguous()
                or lt.in_features != e or lg.in_features != e or lt.out_features != e or lg.out_features != e):
            return None
        key = tuple((t.data_ptr(), t._version) for t in (lt.weight, lt.bias, lg.weight, lg.bias)) + (lt.weight.device,)

        def build():
            import math
            w = torch.cat([lt.weight.detach().float(), lg.weight.detach().float()], 0).contiguous()            # [2 e, e]
            amax = float(w.abs().max())
            scale = 2.0 ** (11 - math.floor(math.log2(amax))) if amax > 0.0 and math.isfinite(amax) else 1.0
            return ops.split_half(w, dp=e, scale=scale), 1.0 / scale, torch.cat([lt.bias.detach().float(), lg.bias.detach().float()]).contiguous()
        w_split, unscale, bias = _cached(self, "_medtok_proj_both_cache", key, build, lt.weight.device)
        out, _ = ops.split_gemm(ops.split_half(z, dp=2 * e), w_split, n_g=e, k_g=e, groups=2, a_group_cols=e, b_group_rows=e, bias=bias, unscale=unscale)
        return out

    def get_distance(self, x, y):
        """Dense distance matrix (reference :120-125); for inspection only -- the
        forward path never materialises it."""
        return torch.sum(x ** 2, dim=1, keepdim=True) + torch.sum(y ** 2, dim=1) - 2 * x @ y.t()

    # ------------------------------------------------------------------ one search
    def _search(self, x, types, training, out=None, norm=None):
        """one search: (zq, vq, commit, xhat, idx, w).  `norm`: the (normalised codebook, squared norms) the caller's forward shares."""
        lo, hi = self._region(types)
        n = x.shape[0]
        x = x.float()                   # autocast callers hand over fp16/bf16; the search is fp32
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.codebook.weight.requires_grad)
        pad = getattr(self, "_pad", 0)
        if pad:
            # e_dim % 4 != 0: rows and codes with zero columns appended (differentiably, where autograd is recording: the slices of
            # the gradients are autograd's), results sliced back.  The mean-squared losses divide by n * (e_dim + pad) inside: rescaled.
            e = self.e_dim
            fpad = torch.nn.functional.pad
            if norm is None:
                norm = self._normalised_codebook()
            what, wsq = norm
            xp = fpad(x, (0, pad))
            if training and needs_grad:
                wp = fpad(self.codebook.weight.float(), (0, pad))
                zq, vq, commit, xhat, idx, w = _SoftVQFunction.apply(xp, wp[lo:hi], what[lo:hi], wsq[lo:hi].contiguous(), self.k,
                                                                     self.search_path, float(self.beta))
                fix = float(e + pad) / float(e)
                return zq[:, :e], vq * fix, commit * fix, xhat[:, :e], idx, w
            r = ops.soft_vq_forward(xp.detach(), what[lo:hi], wsq[lo:hi].contiguous(), self.k, self.search_path, want_sqerr=training)
            if training:
                vq = ops.sum_scale(r["row_sqerr"], (1.0 / (n * e)) if n else float("nan"))
                commit = self.beta * vq
            else:
                vq, commit = torch.tensor(0.0), torch.tensor(0.0)
            zq = r["zq"][:, :e]
            if out is not None:
                out.copy_(zq)
                zq = out
            elif needs_grad and x.requires_grad:
                zq = zq + (x - x.detach())
            return zq, vq, commit, r["xhat"][:, :e], r["idx"], r["w"]
        if norm is None:
            norm = self._normalised_codebook(prepare=not training and ops.takes_filter_path(n, hi - lo, x.shape[1], self.k, self.search_path))
        what, wsq = norm
        prepared = getattr(norm, "prepared", None)
        prepared = None if prepared is None or training else prepared["text" if types == "text" else "graph" if types == "graph" else "shared"]
        if training and needs_grad:
            return _SoftVQFunction.apply(x, self.codebook.weight[lo:hi], what[lo:hi], wsq[lo:hi].contiguous(), self.k,
                                         self.search_path, float(self.beta))
        r = ops.soft_vq_forward(x.detach().float(), what[lo:hi], wsq[lo:hi].contiguous(), self.k, self.search_path,
                                want_sqerr=training, out=out, prepared=prepared)
        if training:
            vq = ops.sum_scale(r["row_sqerr"], (1.0 / (n * x.shape[1])) if n else float("nan"))
            commit = self.beta * vq
        else:
            vq = torch.tensor(0.0)
            commit = torch.tensor(0.0)
        zq = r["zq"]
        if needs_grad and x.requires_grad and out is None:
            zq = zq + (x - x.detach())      # eval under autograd: the straight-through estimator's identity gradient (:214)
        return zq, vq, commit, r["xhat"], r["idx"], r["w"]

    # Per-call state travels in arguments and return values, never on the module: two threads (or two graph captures) may run one
    # module.  What IS shared, by design and as in the reference: the weights, the usage window `codebook_used` (in-place state, part
    # of the state dict) and the caches keyed by weight version (_cached: built once, read-only afterwards).
    def _shared(self, z_text, z_graph, text_mask, batch, norm=None, usage_counts=None, verify_batch=True):
        """get_shared_info plus the token ids / weights of its two searches: (embedding, losses, usage, tokens)."""
        # inference: the two searches write their halves of the [B, 2 e_dim] result in place (no torch.cat).  The buffer is
        # allocated HERE, before pooled() forks its side stream: a block the allocator hands out on this stream may still be the
        # scratch of kernels queued on it, and only a stream that has waited for this one (the fork does) may write to it early.
        emb = None
        if not self.training and not torch.is_grad_enabled() and z_text.is_cuda and self.e_dim % 4 == 0:
            emb = torch.empty((z_text.shape[0], 2 * self.e_dim), dtype=torch.float32, device=z_text.device)
        small = emb is not None and self.cross_attn.small_eligible(z_text, z_graph)
        if small or (emb is not None and MERGE_SHARED_SEARCHES and z_text.shape[0] > 0):
            # inference: ONE search over the interleaved rows [text_0, graph_0, text_1, ...]: its [2 B, e] result is the [B, 2 e]
            # shared embedding, its ids [2 B, k] are both token lists
            bsz, e = z_text.shape[0], self.e_dim
            if small:
                both = self.cross_attn.pooled_small(z_text, text_mask, z_graph, batch)                  # [B, 2, e]
                if verify_batch and not self.cross_attn.assume_sorted_batch and not torch.cuda.is_current_stream_capturing():
                    # any batch vector, like the reference (`batch == idx`, :135): one 4-byte read of the device checks; an unsorted
                    # vector runs again on the stably sorted nodes (PyG vectors are sorted: never taken there)
                    if self.cross_attn.take_status():
                        z_graph, batch = CrossAttention.sort_by_code(z_graph, batch)
                        both = self.cross_attn.pooled_small(z_text, text_mask, z_graph, batch)
            else:
                pooled_text, pooled_graph, pending = self.cross_attn.pooled(z_text, text_mask, z_graph, batch, join=False)
                if pending is not None:
                    _join_side(*pending, (pooled_text,))
                both = torch.stack((pooled_text.float(), pooled_graph.float()), dim=1)
            zq, vq, cm, xhat, idx, w = self._search(both.view(2 * bsz, e), "shared", False, out=emb.view(2 * bsz, e), norm=norm)
            xhat, idx, w = xhat.view(bsz, 2, e), idx.view(bsz, 2, -1), w.view(bsz, 2, -1)
            usage = self.codebook_usage(idx.reshape(bsz, -1), types="shared", _counts=usage_counts) if self.show_usage else 0.0
            tokens = {"shared_text_tokens": idx[:, 0], "shared_text_tokens_weights": w[:, 0],
                      "shared_graph_tokens": idx[:, 1], "shared_graph_tokens_weights": w[:, 1]}
            return emb, (vq + vq, cm + cm, xhat[:, 0], xhat[:, 1], emb[:, :e], emb[:, e:]), usage, tokens
        pooled_text, pooled_graph, pending = self.cross_attn.pooled(z_text, text_mask, z_graph, batch, join=False)
        out_t, out_g = (emb[:, :self.e_dim], emb[:, self.e_dim:]) if emb is not None else (None, None)
        if pending is not None:
            # the text side ran on the second stream: its shared search follows it there, beside the graph side's tail and search
            side, main = pending
            if emb is not None:
                _lend(side, emb)
            with torch.cuda.stream(side):
                r_t = self._search(pooled_text, "shared", self.training, out=out_t, norm=norm)
            r_g = self._search(pooled_graph, "shared", self.training, out=out_g, norm=norm)
            _join_side(side, main, (pooled_text, *r_t))
        else:
            r_t = self._search(pooled_text, "shared", self.training, out=out_t, norm=norm)
            r_g = self._search(pooled_graph, "shared", self.training, out=out_g, norm=norm)
        zq_t, vq_t, cm_t, xhat_t, idx_t, w_t = r_t
        zq_g, vq_g, cm_g, xhat_g, idx_g, w_g = r_g
        usage = self.codebook_usage(torch.cat([idx_t, idx_g], dim=-1), types="shared", _counts=usage_counts) if self.show_usage else 0.0
        tokens = {"shared_text_tokens": idx_t, "shared_text_tokens_weights": w_t,
                  "shared_graph_tokens": idx_g, "shared_graph_tokens_weights": w_g}
        return (emb if emb is not None else torch.cat([zq_t, zq_g], dim=-1), (vq_t + vq_g, cm_t + cm_g, xhat_t, xhat_g, zq_t, zq_g), usage, tokens)

    def _specific(self, original_embedding, types, norm=None, usage_counts=None):
        """specific_embedding plus the ids / weights of its search: (zq, losses, usage, idx, w)."""
        if types in ("text", "graph"):
            original_embedding = self.project(original_embedding, types)
        zq, vq, commit, xhat, idx, w = self._search(original_embedding, types, self.training, norm=norm)
        usage = self.codebook_usage(idx, types=types + "-specific", _counts=usage_counts)
        return zq, (vq, commit, xhat, zq), usage, idx, w

    # ------------------------------------------------------------------ reference API
    def get_shared_info(self, z_text, z_graph, text_mask, batch):
        return self._shared(z_text, z_graph, text_mask, batch)[:3]

    def specific_embedding(self, original_embedding, types="text", return_tokens=False):
        """reference :187-217.  return_tokens (additive): also the region-local ids [N, k] and softmax weights the reference computes
        and drops (R3) -- (zq, losses, usage, idx, w)."""
        r = self._specific(original_embedding, types)
        return r if return_tokens else r[:3]

    def codebook_usage(self, min_encoding_indices, types="shared", _counts=None):
        """Fraction of codes seen in the sliding id window (reference :219-236).
        Returns a Python float, which costs one host sync per call as in the
        reference; forward() batches its three calls into one sync (`_counts`: the caller's list of device counts)."""
        if not self.show_usage:
            return 0.0
        count = ops.usage_update_(self.codebook_used, min_encoding_indices, self.n_e)
        if _counts is not None:
            _counts.append(count)
            return count
        return count.item() / self.n_e

    def _forward_small_batch(self, z, text_features, graph_node_features, text_attention_mask, batch, z_aug, norm):
        """forward() at inference for batches whose searches all take the exact path with few rows (a serving batch; the reference's
        own per-GPU batch of 256): cross-attention, then ALL searches of the forward in one call of three launches, then all updates
        of the usage window in one call of two.  Same values as the general form below; None when it does not apply."""
        bsz, e, k = z.shape[0], self.e_dim, self.k
        if not (BATCHED_SMALL_SEARCHES and not self.training and not torch.is_grad_enabled() and z.is_cuda and bsz > 0 and self.e_dim % 4 == 0
                and z.dtype in (torch.float32, torch.float16, torch.bfloat16)):
            return None
        z = z.float()
        z_aug = None if z_aug is None else z_aug.float()
        region = self.codebook.weight.shape[0] // 3
        if not (ops.multi_search_eligible(bsz, region, e, k) and ops.multi_search_eligible(2 * bsz, self.n_e, e, k)):
            return None
        if self.search_path not in (ops.PATH_AUTO, ops.PATH_F32_MFMA):
            return None                 # a forced search path or plan bits (bench --path, tests): the general form runs what the caller named
        z_text, z_graph = torch.split(z, self.split, dim=-1)
        aug = (None, None) if z_aug is None else torch.split(z_aug, self.split, dim=-1)
        emb = torch.empty((bsz, 2 * e), dtype=torch.float32, device=z.device)
        small_attn = self.cross_attn.small_eligible(text_features, graph_node_features)
        # the two-launch cross-attention needs a sorted `batch` vector and flags anything else on the device; unless the caller
        # vouches for it (assume_sorted_batch) or a HIP graph is being captured, the flag is read where this forward reads the host
        # anyway (behind the usage counts: the window kernel does not write the window when it is set) or with one 4-byte read at
        # the end, and an unsorted vector runs again on the stably sorted nodes -- the reference takes any batch vector (:135)
        verify = small_attn and not self.cross_attn.assume_sorted_batch and not torch.cuda.is_current_stream_capturing()
        if small_attn:
            both = self.cross_attn.pooled_small(text_features, text_attention_mask, graph_node_features, batch)
        else:
            pooled_text, pooled_graph = self.cross_attn.pooled(text_features, text_attention_mask, graph_node_features, batch)
            both = torch.stack((pooled_text.float(), pooled_graph.float()), dim=1)

        if callable(norm):                   # (forward() hands over a thunk: the normalisation is enqueued behind the cross-attention)
            norm = norm()
        what, wsq = norm

        def again():
            nodes_s, batch_s = CrossAttention.sort_by_code(graph_node_features, batch)
            return self._forward_small_batch(z, text_features, nodes_s, text_attention_mask, batch_s, z_aug, norm)
        searches = [dict(x=both.view(2 * bsz, e), what=what, wsq=wsq, out=emb.view(2 * bsz, e))]
        for zz, parts in ((z, (z_text, z_graph)),) + (((z_aug, aug),) if z_aug is not None else ()):
            proj = self.project_both(zz)
            for i, types in enumerate(("text", "graph")):
                lo, hi = self._region(types)
                x = proj[:, i * e:(i + 1) * e] if proj is not None else self.project(parts[i], types).float()
                searches.append(dict(x=x, what=what[lo:hi], wsq=wsq[lo:hi].contiguous()))
        res = ops.soft_vq_forward_multi(searches, k)
        sh = res[0]
        xhat_s, idx_s, w_s = sh["xhat"].view(bsz, 2, e), sh["idx"].view(bsz, 2, k), sh["w"].view(bsz, 2, k)
        zero = torch.tensor(0.0)
        u_shared = u_text = u_graph = 0.0
        if self.show_usage:
            st = getattr(self.cross_attn, "small_status", None)
            st = st if st is not None and st.device == z.device else None
            # one read: the usage counts and, behind them, the cross-attention's status word (copied there by the window kernel)
            cnt = ops.usage_update_multi_(self.codebook_used, [idx_s.reshape(bsz, 2 * k)] + [r["idx"] for r in res[1:]], self.n_e, extra_word=st)
            vals = cnt.cpu().tolist()
            if st is not None and vals[-1] and self.cross_attn.take_status(int(vals[-1])):
                if not verify:
                    raise ValueError(UNSORTED_BATCH_MESSAGE)
                return again()                       # (the window is as it was: ops.usage_update_multi_ skips its write under a set word)
            u_shared, u_text, u_graph = (v / self.n_e for v in vals[:3])
        elif verify and self.cross_attn.take_status():
            return again()
        out = {
            "graph_feature": z_graph,
            "text_feature": z_text,
            "shared_text_embedding": emb[:, :e],
            "shared_graph_embedding": emb[:, e:],
            "shared_embed_loss": (zero.clone(), zero.clone(), xhat_s[:, 0], xhat_s[:, 1], emb[:, :e], emb[:, e:]),
            "shared_codebook_usage": u_shared,
            "specific_embedding_text": res[1]["zq"],
            "text_specific_loss": (zero.clone(), zero.clone(), res[1]["xhat"], res[1]["zq"]),
            "text_specific_usage": u_text,
            "specific_embedding_graph": res[2]["zq"],
            "graph_specific_loss": (zero.clone(), zero.clone(), res[2]["xhat"], res[2]["zq"]),
            "graph_specific_usage": u_graph,
            "specific_embedding_text_aug": res[3]["zq"] if z_aug is not None else None,
            "specific_embedding_graph_aug": res[4]["zq"] if z_aug is not None else None,
            "shared_text_tokens": idx_s[:, 0], "shared_text_tokens_weights": w_s[:, 0],
            "shared_graph_tokens": idx_s[:, 1], "shared_graph_tokens_weights": w_s[:, 1],
            "text_tokens": res[1]["idx"], "text_tokens_weights": res[1]["w"],
            "graph_tokens": res[2]["idx"], "graph_tokens_weights": res[2]["w"],
        }
        return out

    def _forward_train(self, z, text_features, graph_node_features, text_attention_mask, batch, z_aug, norm):
        """forward() in training mode under autograd with a trainable codebook: the cross-attention, then ALL searches under one
        autograd node (_SoftVQMultiFunction: one dense codebook gradient instead of six), then all usage-window updates in one call.
        Same values as the general form below; None when it does not apply."""
        if not (TRAIN_SINGLE_CODEBOOK_GRADIENT and self.training and torch.is_grad_enabled() and z.is_cuda and z.shape[0] > 0
                and self.codebook.weight.requires_grad and self.e_dim % 4 == 0):
            return None
        what, wsq = norm
        z_text, z_graph = torch.split(z, self.split, dim=-1)
        aug = None if z_aug is None else torch.split(z_aug, self.split, dim=-1)
        pooled_text, pooled_graph = self.cross_attn.pooled(text_features, text_attention_mask, graph_node_features, batch)
        xs = [pooled_text, pooled_graph, self.project(z_text, "text"), self.project(z_graph, "graph")]
        regions = [self._region("shared"), self._region("shared"), self._region("text"), self._region("graph")]
        if aug is not None:
            xs += [self.project(aug[0], "text"), self.project(aug[1], "graph")]
            regions += [self._region("text"), self._region("graph")]
        res = _SoftVQMultiFunction.apply(self.codebook.weight, what, wsq, self.k, self.search_path, float(self.beta), regions, *[x.float() for x in xs])
        r = [res[6 * i: 6 * i + 6] for i in range(len(xs))]               # (zq, vq, commit, xhat, idx, w) per search
        u_shared = u_text = u_graph = 0.0
        if self.show_usage:
            # (with max_nodes_bound set the cross-attention validated `batch` on the device: its status word comes with the counts in the
            # one host read, and vetoes the window update when it is set)
            st = getattr(self.cross_attn, "small_status", None) if getattr(self.cross_attn, "max_nodes_bound", None) is not None else None
            if st is not None and st.device != self.codebook_used.device:
                st = None
            cnt = ops.usage_update_multi_(self.codebook_used, [torch.cat([r[0][4], r[1][4]], dim=-1)] + [q[4] for q in r[2:]], self.n_e, extra_word=st)
            vals = cnt.cpu()
            if st is not None and int(vals[-1]) and self.cross_attn.take_status(int(vals[-1])):
                raise ValueError(UNSORTED_BATCH_MESSAGE)
            u_shared, u_text, u_graph = (vals[:3].double() / self.n_e).tolist()
        out = {
            "graph_feature": z_graph,
            "text_feature": z_text,
            "shared_text_embedding": r[0][0],
            "shared_graph_embedding": r[1][0],
            "shared_embed_loss": (r[0][1] + r[1][1], r[0][2] + r[1][2], r[0][3], r[1][3], r[0][0], r[1][0]),
            "shared_codebook_usage": u_shared,
            "specific_embedding_text": r[2][0],
            "text_specific_loss": (r[2][1], r[2][2], r[2][3], r[2][0]),
            "text_specific_usage": u_text,
            "specific_embedding_graph": r[3][0],
            "graph_specific_loss": (r[3][1], r[3][2], r[3][3], r[3][0]),
            "graph_specific_usage": u_graph,
            "specific_embedding_text_aug": r[4][0] if aug is not None else None,
            "specific_embedding_graph_aug": r[5][0] if aug is not None else None,
            "shared_text_tokens": r[0][4], "shared_text_tokens_weights": r[0][5],
            "shared_graph_tokens": r[1][4], "shared_graph_tokens_weights": r[1][5],
            "text_tokens": r[2][4], "text_tokens_weights": r[2][5],
            "graph_tokens": r[3][4], "graph_tokens_weights": r[3][5],
        }
        return out

    def forward(self, z, text_features, graph_node_features, text_attention_mask, batch, z_aug=None):
        counts = []                     # the usage counts stay on the device: one sync at the end
        # one normalisation of the codebook per forward (training: re-normalised every forward, like the reference's every call;
        # eval: the cache per weight version), shared by its 4-6 searches
        bsz_, region_ = z.shape[0], self.codebook.weight.shape[0] // 3

        def make_norm():
            return self._normalised_codebook(rebuild=self.training,
                                             prepare=not self.training and z.is_cuda and bsz_ > 0 and self.e_dim % 4 == 0
                                             and (ops.takes_filter_path(2 * bsz_, self.n_e, self.e_dim, self.k, self.search_path)
                                                  or ops.takes_filter_path(bsz_, region_, self.e_dim, self.k, self.search_path)))
        # (small batches are latency-bound: their cross-attention -- which needs nothing of the codebook -- is launched first, the
        # codebook is normalised under it)
        small = self._forward_small_batch(z, text_features, graph_node_features, text_attention_mask, batch, z_aug, make_norm)
        if small is not None:
            return small
        norm = make_norm()
        train = self._forward_train(z, text_features, graph_node_features, text_attention_mask, batch, z_aug, norm)
        if train is not None:
            return train
        z_text_embedding, z_graph_embedding = torch.split(z, self.split, dim=-1)
        aug = (None, None) if z_aug is None else torch.split(z_aug, self.split, dim=-1)
        early = None
        if (not self.training and not torch.is_grad_enabled() and z.is_cuda
                and 0 < SIDE_STREAM_MIN_CODES <= z.shape[0]):
            # inference: the modality-specific searches depend on nothing the cross-attention produces: second stream, joined
            # below; their usage-window updates stay in the reference's order (shared, text, graph, aug text, aug graph: :241-250)
            side, main = _side_stream(z.device, 1)
            _lend(side, norm)
            with torch.cuda.stream(side):
                early = [self._search(self.project(x, types), types, False, norm=norm) for x, types in
                         ((z_text_embedding, "text"), (z_graph_embedding, "graph"), (aug[0], "text"), (aug[1], "graph")) if x is not None]
        shared_embedding, shared_embed_loss, u_shared, tokens = self._shared(
            text_features, graph_node_features, text_attention_mask, batch, norm=norm, usage_counts=counts, verify_batch=True)
        shared_text_embedding, shared_graph_embedding = torch.split(shared_embedding, self.split, dim=-1)
        if early is not None:
            _join_side(side, main, [t for r in early for t in r])
            results = []
            for (zq, vq, commit, xhat, idx, w), types in zip(early, ("text", "graph", "text", "graph")):
                usage = self.codebook_usage(idx, types=types + "-specific", _counts=counts)
                results.append((zq, (vq, commit, xhat, zq), usage, idx, w))
        else:
            results = [self._specific(z_text_embedding, "text", norm, counts), self._specific(z_graph_embedding, "graph", norm, counts)]
            if z_aug is not None:
                # the reference discards these two usage values but its window still slides (:249-250)
                results += [self._specific(aug[0], "text", norm, counts), self._specific(aug[1], "graph", norm, counts)]
        spec_text, text_specific_loss, u_text, tokens["text_tokens"], tokens["text_tokens_weights"] = results[0]
        spec_graph, graph_specific_loss, u_graph, tokens["graph_tokens"], tokens["graph_tokens_weights"] = results[1]
        spec_text_aug = results[2][0] if z_aug is not None else None
        spec_graph_aug = results[3][0] if z_aug is not None else None
        if self.show_usage and counts:
            st = getattr(self.cross_attn, "small_status", None)
            if st is not None and st.device == counts[0].device:          # one read for the usage counts and the small path's status word
                vals = torch.cat([torch.stack(counts[:3]), st[:1]]).cpu()
                if int(vals[3]) and self.cross_attn.take_status(int(vals[3])):      # (assume_sorted_batch / max_nodes_bound callers only)
                    raise ValueError(UNSORTED_BATCH_MESSAGE)
                u_shared, u_text, u_graph = (vals[:3].double() / self.n_e).tolist()
            else:
                u_shared, u_text, u_graph = (torch.stack(counts[:3]).cpu().double() / self.n_e).tolist()
        out = {
            "graph_feature": z_graph_embedding,
            "text_feature": z_text_embedding,
            "shared_text_embedding": shared_text_embedding,
            "shared_graph_embedding": shared_graph_embedding,
            "shared_embed_loss": shared_embed_loss,
            "shared_codebook_usage": u_shared,
            "specific_embedding_text": spec_text,
            "text_specific_loss": text_specific_loss,
            "text_specific_usage": u_text,
            "specific_embedding_graph": spec_graph,
            "graph_specific_loss": graph_specific_loss,
            "graph_specific_usage": u_graph,
            "specific_embedding_text_aug": spec_text_aug,
            "specific_embedding_graph_aug": spec_graph_aug,
        }
        out.update(tokens)   # region-local ids, as torch.topk returns them in the reference
        return out

    def global_token_ids(self, local_ids, types):
        """Map region-local ids (what the searches return) to rows of codebook.weight."""
        lo, _ = self._region(types)
        return local_ids + lo


def compute_entropy_loss(affinity, loss_type="softmax", temperature=0.01):
    """Entropy regulariser over a dense [N, K] affinity matrix (reference :273-287).  Kept for API parity only: nothing in the
    reference calls it (entropy_loss_ratio is stored and never read, :96), and the search kernels never materialise the N x K
    matrix it wants -- a caller must bring its own (e.g. -VectorQuantizer.get_distance(x, y)).  Plain torch ops, not in place
    (the reference divides its argument in place)."""
    flat_affinity = affinity.reshape(-1, affinity.shape[-1]) / temperature
    probs = torch.softmax(flat_affinity, dim=-1)
    log_probs = torch.log_softmax(flat_affinity + 1e-5, dim=-1)
    if loss_type != "softmax":
        raise ValueError("Entropy loss {} not supported".format(loss_type))
    avg_probs = torch.mean(probs, dim=0)
    avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + 1e-5))
    sample_entropy = -torch.mean(torch.sum(probs * log_probs, dim=-1))
    return sample_entropy - avg_entropy
