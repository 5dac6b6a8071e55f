// attention_small.h -- CrossAttention.pooled at the reference's own width (e_dim = 64, 4 heads: train_MedTok.py:363-368) in TWO launches
// (the layers, then the node mean): both layers of both directions, from the raw text / node features to the pooled rows the shared
// searches read.
// Included by medtok_vq.hip after attention_pp.h; gfx950 only.
//
//     for every code b (vector_quantization_soft_one_new.py:133-142, 17-88):
//         pooled_text[b]  = CrossAttention(text[b, :valid], nodes_b)[0][0]          (the CLS row of the attended text)
//         pooled_graph[b] = CrossAttention(text[b, :valid], nodes_b)[1].mean(0)     (the mean of the attended nodes)
//
// Why a kernel of its own.  At this width the layer-by-layer path (seven launches per layer and side: images, four dense products,
// attention core, residual + LayerNorm) is 28 launches of 5-40 us for a few MFLOP each -- a B = 256 forward spent 0.32 ms of its
// 0.45 ms of kernel time (and most of its 0.9 ms of wall time) there.  But cross-attention couples nothing across query rows: a
// node's two layers depend on that node and on the ORIGINAL text rows of its code only (:83,86).  So one block takes a tile of 8
// consecutive nodes through both layers without leaving the CU:
//   * query tile = 8 nodes x 4 heads = 32 rows.  Per layer: q' = Wq x + bq, the per-head fold qf_h = Wk_h^T q'_h (the key bias is
//     the same for every key: softmax drops it), the attention core softmax(scale qf K^T) K over the raw rows of the other modality,
//     att_h = Wv_h ctx_h + bv_h, out_proj, residual + LayerNorm.  The five small dense steps are fp32 FMAs with lane = output column
//     (64 columns = one wavefront; the weights arrive transposed where that makes the loads coalesced: one 256-byte line per wave
//     and k, L2-resident); cross-wave partial sums meet in LDS.
//   * attention core with keys as the A rows: S^T = K qf^T leaves every lane with 16 keys of ONE query row, so the online softmax
//     is lane-local (its partner half-wave by v_permlane32_swap), and the probabilities it ends with ARE the B operand of the second
//     product ctx^T += K^T P^T -- the contraction runs over the keys in the order the accumulator registers hold them.
//     SPLIT = true (the product; round 6): both products on the fp16 matrix pipe as three passes over (hi, lo) pairs
//     (v_mfma_f32_32x32x16_f16: a_hi b_hi + a_hi b_lo + a_lo b_hi, ~2^-22 relative -- the arithmetic of the wide kernels,
//     attention_dma.h; 24 MFMAs of 32 cycles per 32-key chunk where the fp32 pipe needs 64 of 64 cycles).  A wave parks its chunk as
//     (hi, lo) fp16 images [column block of 32][key][64 B] with the 16-byte pieces XOR-swizzled by f(key >> 2) (attention_dma.h's
//     image: conflict-free for both operand shapes); the score product reads key rows in the order "key bits 2 and 3 swapped", so
//     that the 8 probabilities a lane holds per 16-key step belong to 8 CONSECUTIVE keys -- exactly the 8 keys ds_read_b64_tr_b16
//     delivers per column for the value product's A operand, and the probabilities go from the accumulator registers to the B
//     operand in the order they stand.  SPLIT = false: the round-5 core on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fmaf
//     chains), kept as medtok_debug_cross_attention_small_exact_f32 -- the tests' second opinion.  The four waves of
//     a block walk the key chunks (32 keys) round-robin, each with its own running maximum / sum / context, and merge at the end of
//     the layer; no block-wide barrier inside the key loop.  A wave prefetches its next chunk into registers under the MFMAs of
//     the current one and parks it in its own LDS slice (rows padded to 68 floats: conflict-free for both operand shapes).
//   * no prologue launches and no host read: a tile is 8 consecutive rows of the node array (grid = ceil(N / 8) + B from the shapes
//     alone); it may span several codes of the (sorted, PyG-style) batch vector and runs one key pass per code it touches.  A code's
//     token count is counted from its mask row by the block that needs it; its node range comes from a binary search of the batch
//     vector.  Blocks ceil(N / 8) .. + B - 1 are the text side (the CLS row of code b against b's nodes).
//   * the node mean needs all tiles of a code: a second, tiny launch (cross_attention64_mean_kernel: rows added in node order).
// `status` (int32 [4], zeroed by the caller once; OR-ed): bit 0 = the batch vector is not sorted, bit 1 = an id outside [0, B).
// Results are then wrong for the codes involved -- the host-side wrapper checks the word where it synchronises anyway.
#pragma once

constexpr int XS_D = 64, XS_H = 4, XS_HD = 16, XS_G = 8, XS_ROWS = XS_G * XS_H;      // 32 query rows per tile
constexpr int XS_LD = 68;                         // padded row length (floats) of the LDS tiles
constexpr int XS_LAYER_FLOATS = 4 * XS_D * XS_D + 6 * XS_D;       // WqT | Wk | WvT | WoT | bq | bv | bo | gamma | beta | (pad)

struct XSmallArgs {
    const float *text;            // [B, L, 64]
    const void *mask;             // [B, L], mask_bytes per element (1 / 4 / 8); non-zero = valid, left-aligned
    const float *nodes;           // [N, 64], rows of one code adjacent (sorted batch vector)
    const int64_t *batch;         // [N]
    const float *weights;         // [layers][XS_LAYER_FLOATS]
    float *y_nodes;               // [N, 64]: the attended nodes (read by the mean kernel)
    float *pooled;                // text row of code b at pooled + b * pooled_stride, graph row at pooled + b * pooled_stride + graph_off
    int *status;
    long n_codes, seq_len, n_nodes, pooled_stride, graph_off;
    int mask_bytes, layers, n_graph_tiles;
    float scale, ln_eps;
};

__device__ __forceinline__ float xs_other_half(float v, int lh)
{
    const unsigned b = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(lh ? r[0] : r[1]);
}

__device__ __forceinline__ float xs_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// first index i in [0, n) with batch[i] >= v (batch non-decreasing), by a whole wavefront: every round the 64 lanes probe 64 evenly
// spaced positions of the remaining range at once (one memory round trip narrows it 65-fold: 5 000 nodes take two rounds, where a
// binary search makes thirteen dependent loads).  All lanes return the same value.
__device__ __forceinline__ long xs_lower_bound(const int64_t *__restrict__ batch, long n, long v)
{
    const int lane = threadIdx.x & 63;
    long lo = 0, hi = n;                       // answer in [lo, hi]
    while (hi - lo > 0) {
        const long span = hi - lo, step = (span + 63) / 64;
        const long pos = lo + (long)lane * step;                        // lane's probe (positions >= hi count as "not less")
        const bool less = pos < hi && batch[pos] < v;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(less);
        const int cnt = __builtin_popcountll(m);                      // probes 0 .. cnt-1 are < v (monotone)
        if (cnt == 0) { hi = lo; break; }
        const long last_less = lo + (long)(cnt - 1) * step;
        lo = last_less + 1;
        hi = min(hi, last_less + step);
    }
    return lo;
}

typedef __fp16 xs_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ void xs_split4(const float4 v, half4v &h, half4v &l)
{
    h = (half4v){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    l = (half4v){(_Float16)(v.x - (float)h[0]), (_Float16)(v.y - (float)h[1]), (_Float16)(v.z - (float)h[2]), (_Float16)(v.w - (float)h[3])};
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void cross_attention64_kernel(XSmallArgs a)
{
    // ---- LDS
    __shared__ __attribute__((aligned(16))) float s_x[XS_G][XS_D];            // the tile's rows: layer input, then its output
    __shared__ __attribute__((aligned(16))) float s_q[XS_G][XS_D];            // q' = Wq x + bq; later att = Wv ctx + bv
    // folded queries [4 n + h][c]; once every wave holds them as MFMA operands the same bytes receive the contexts [4 n + h][c]
    // (first written behind the block-wide barrier of the first merge; read by the W_v step, which ends before the next fold)
    __shared__ __attribute__((aligned(16))) float s_qf[XS_ROWS][XS_LD];
    float (*s_ctx)[XS_LD] = s_qf;
    __shared__ __attribute__((aligned(16))) float s_kv[4][32][XS_LD];         // one key chunk per wave; after the key loop: its partial contexts
    __shared__ __attribute__((aligned(16))) float s_part[4][XS_G][XS_D];      // cross-wave partial sums of the dense steps
    __shared__ float s_ml[4][2][XS_ROWS];                                     // per wave: running maximum, running sum
    __shared__ long s_seg[XS_G + 1][5];                                        // segments of the tile: code, first local row, rows, keys, first key row
    __shared__ int s_nseg;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const bool text_side = (int)blockIdx.x >= a.n_graph_tiles;
    const long B = a.n_codes, L = a.seq_len, N = a.n_nodes;

    // ---- the tile's query rows and its segments (runs of one code)
    int n_live;                 // query "nodes" in the tile (1 on the text side)
    long row0 = 0;              // graph side: first node row of the tile
    if (text_side) {
        const long b = (long)blockIdx.x - a.n_graph_tiles;
        n_live = 1;
        if (tid < XS_D) s_x[0][tid] = a.text[b * L * XS_D + tid];             // the CLS row
        for (int i = tid; i < (XS_G - 1) * XS_D; i += 256) s_x[1 + i / XS_D][i % XS_D] = 0.f;
        if (tid == 0) { s_seg[0][0] = b; s_seg[0][1] = 0; s_seg[0][2] = 1; s_nseg = 1; }
    } else {
        row0 = (long)blockIdx.x * XS_G;
        n_live = (int)min((long)XS_G, N - row0);
        for (int i = tid; i < XS_G * XS_D; i += 256) {
            const int n = i / XS_D, c = i % XS_D;
            s_x[n][c] = n < n_live ? a.nodes[(row0 + n) * XS_D + c] : 0.f;
        }
        if (wv == 0) {
            // lane n < 8: node n of the tile, lane 8: the node in front of the tile (one load each); a segment starts where the id changes
            const long node = lane < XS_G ? row0 + lane : row0 - 1;
            const bool have = lane < XS_G ? lane < n_live : row0 > 0;
            const long id = have ? a.batch[node] : -0x7fffffffffffffffL;
            const long prev = __shfl(id, lane == 0 ? XS_G : (lane - 1) & 63, 64);       // (lane 0 looks at the node in front of the tile)
            const bool live = lane < n_live;
            int bad = 0;
            if (live && id < prev) bad |= 1;
            if (live && (id < 0 || id >= B)) bad |= 2;
            const long code = id < 0 ? 0 : (id >= B ? B - 1 : id);
            const long pcode = prev < 0 ? 0 : (prev >= B ? B - 1 : prev);
            const bool start = live && (lane == 0 || code != pcode);
            const unsigned long long starts = __builtin_amdgcn_ballot_w64(start);
            const int sidx = __builtin_popcountll(starts & ((2ull << lane) - 1)) - 1;         // segment of this node
            if (start) {
                const unsigned long long later = starts & ~((2ull << lane) - 1);
                const int next = later ? __builtin_ctzll(later) : n_live;
                s_seg[sidx][0] = code; s_seg[sidx][1] = lane; s_seg[sidx][2] = next - lane;
            }
            if (lane == 0) s_nseg = __builtin_popcountll(starts);
            const unsigned long long anybad = __builtin_amdgcn_ballot_w64(bad != 0);
            if (anybad) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off, 64);
                if (lane == 0) atomicOr(a.status, bad);
            }
        }
    }
    __syncthreads();
    const int nseg = s_nseg;

    // ---- the key set of every segment, once for both layers: graph side -> the code's valid text rows (its token count = the non-zero
    // entries of its mask row: wave w counts segments w, w + 4, ...); text side -> the code's nodes (their range in the batch vector)
    if (text_side) {
        if (wv == 0) {
            const long code = s_seg[0][0];
            const long lo = xs_lower_bound(a.batch, N, code), hi = xs_lower_bound(a.batch, N, code + 1);
            if (lane == 0) { s_seg[0][3] = hi - lo; s_seg[0][4] = lo; }
        }
    } else {
        for (int sg = wv; sg < nseg; sg += 4) {
            const char *mrow = reinterpret_cast<const char *>(a.mask) + s_seg[sg][0] * L * a.mask_bytes;
            int cnt = 0;
            for (long i = lane; i < L; i += 64) {
                const bool nz = a.mask_bytes == 1 ? mrow[i] != 0 : (a.mask_bytes == 4 ? reinterpret_cast<const int *>(mrow)[i] != 0
                                                                                       : reinterpret_cast<const long *>(mrow)[i] != 0);
                cnt += nz ? 1 : 0;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (lane == 0) { s_seg[sg][3] = cnt; s_seg[sg][4] = 0; }
        }
    }
    __syncthreads();

    // ---- MFMA helpers (v_mfma_f32_32x32x2_f32: A [32 x 2], B [2 x 32]; lane (i, k) holds A[i][k] / B[k][i])
    const float *kvw = &s_kv[wv][0][0];
    // SPLIT: the wave's key chunk as fp16 images in the same bytes: [plane hi | lo : 4 KB][column block of 32 : 2 KB][key : 64 B],
    // 16-byte piece p of a key row at position p ^ f(key >> 2), f(x) = -x & 3
    char *kimg = reinterpret_cast<char *>(&s_kv[wv][0][0]);
    const int pk = (li & 19) | ((li & 8) >> 1) | ((li & 4) << 1);           // score product: A row li = key "li with bits 2 and 3 swapped"
    const int a1_off = pk * 64 + ((lh ^ ((0 - (pk >> 2)) & 3)) << 4);       // its 16-byte piece of k step 0 (k step 1: position ^ 2; steps 2, 3: next block)
    int v_off[2];                                                           // value product: this lane's piece of the two transposed reads
    {
        const int g = lane >> 4, tt = lane & 15;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int key = 8 * (g >> 1) + 4 * rd + (tt >> 2);
            const int pos = (2 * (g & 1) + ((tt & 3) >> 1)) ^ ((0 - (key >> 2)) & 3);
            v_off[rd] = key * 64 + pos * 16 + 8 * (tt & 1);
        }
    }
    const int park_blk = ((lane & 15) >> 3) * 2048 + 8 * (lane & 1), park_piece = (lane & 7) >> 1, park_row = lane >> 4;

    for (int layer = 0; layer < a.layers; ++layer) {
        const float *W = a.weights + (long)layer * XS_LAYER_FLOATS;
        const float *WqT = W, *Wk = W + 4096, *WvT = W + 8192, *WoT = W + 12288;
        const float *bq = W + 16384, *bv = bq + 64, *bo = bq + 128, *gamma = bq + 192, *beta = bq + 256;

        // ---- q'[n][o] = sum_i x[n][i] WqT[i][o] + bq[o]: wave = quarter of i, lane = o
        {
            float acc[XS_G];
#pragma unroll
            for (int n = 0; n < XS_G; ++n) acc[n] = 0.f;
            float w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = WqT[(16 * wv + i) * XS_D + lane];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int n = 0; n < XS_G; ++n) {
                    const float4 xv = *reinterpret_cast<const float4 *>(&s_x[n][16 * wv + 4 * i4]);
                    acc[n] = fmaf(xv.x, w[4 * i4], acc[n]); acc[n] = fmaf(xv.y, w[4 * i4 + 1], acc[n]);
                    acc[n] = fmaf(xv.z, w[4 * i4 + 2], acc[n]); acc[n] = fmaf(xv.w, w[4 * i4 + 3], acc[n]);
                }
#pragma unroll
            for (int n = 0; n < XS_G; ++n) s_part[wv][n][lane] = acc[n];
        }
        __syncthreads();
        for (int i = tid; i < XS_G * XS_D; i += 256) {
            const int n = i >> 6, o = i & 63;
            s_q[n][o] = ((s_part[0][n][o] + s_part[1][n][o]) + (s_part[2][n][o] + s_part[3][n][o])) + bq[o];
        }
        __syncthreads();
        // ---- the fold: qf[4 n + h][c] = sum_j q'[n][16 h + j] Wk[16 h + j][c]: wave = head, lane = c
        {
            float acc[XS_G];
#pragma unroll
            for (int n = 0; n < XS_G; ++n) acc[n] = 0.f;
            float w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = Wk[(16 * wv + j) * XS_D + lane];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int n = 0; n < XS_G; ++n) {
                    const float4 qv = *reinterpret_cast<const float4 *>(&s_q[n][16 * wv + 4 * j4]);
                    acc[n] = fmaf(qv.x, w[4 * j4], acc[n]); acc[n] = fmaf(qv.y, w[4 * j4 + 1], acc[n]);
                    acc[n] = fmaf(qv.z, w[4 * j4 + 2], acc[n]); acc[n] = fmaf(qv.w, w[4 * j4 + 3], acc[n]);
                }
#pragma unroll
            for (int n = 0; n < XS_G; ++n) s_qf[4 * n + wv][lane] = acc[n] * a.scale;        // (the score scale rides on the queries)
        }
        __syncthreads();

        // ---- the attention core, one pass per code the tile touches
        // the queries as B operands: lane (row li, half lh) holds qf[li][8 g + 4 lh + e], g = 0..7, e = 0..3
        float4 qb[8];
        half8 qbh[4], qbl[4];           // SPLIT: (hi, lo) of qf[li][16 ks + 8 lh + e], ks = 0..3
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                half4v h0, l0, h1, l1;
                xs_split4(*reinterpret_cast<const float4 *>(&s_qf[li][16 * ks + 8 * lh]), h0, l0);
                xs_split4(*reinterpret_cast<const float4 *>(&s_qf[li][16 * ks + 8 * lh + 4]), h1, l1);
                qbh[ks] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                qbl[ks] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) qb[g] = *reinterpret_cast<const float4 *>(&s_qf[li][8 * g + 4 * lh]);
        }

        for (int sg = 0; sg < nseg; ++sg) {
            const long code = s_seg[sg][0];
            const long klen = s_seg[sg][3];
            const float *kbase = text_side ? a.nodes + s_seg[sg][4] * XS_D : a.text + code * L * XS_D;
            const int nchunk = (int)((klen + 31) >> 5);

            f32x16 ctx0, ctx1;                   // ctx^T: lane = query row li; registers = columns (r & 3) + 8 (r >> 2) + 4 lh (+ 32 for ctx1)
#pragma unroll
            for (int r = 0; r < 16; ++r) { ctx0[r] = 0.f; ctx1[r] = 0.f; }
            float m_run = -INFINITY, l_run = 0.f;

            // this wave's chunks: wv, wv + 4, ...; a chunk is 32 rows of 256 B, fetched as 8 x (64 lanes x 16 B): lane -> row 4 q + (lane >> 4)
            // (eight named registers and a macro: hipcc leaves an array that is filled under a condition and lives across iterations --
            // or one captured by reference in a lambda -- in scratch memory)
            float4 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7;
            pf0 = pf1 = pf2 = pf3 = pf4 = pf5 = pf6 = pf7 = make_float4(0.f, 0.f, 0.f, 0.f);
#define XS_FETCH1(cc, q, dst) dst = *reinterpret_cast<const float4 *>(kbase + min((long)32 * (cc) + 4 * (q) + (lane >> 4), klen - 1) * XS_D + 4 * (lane & 15));
            // (past the end of the key set: the last key again -- its score is masked below)
#define XS_FETCH(cc) XS_FETCH1(cc, 0, pf0) XS_FETCH1(cc, 1, pf1) XS_FETCH1(cc, 2, pf2) XS_FETCH1(cc, 3, pf3) XS_FETCH1(cc, 4, pf4) XS_FETCH1(cc, 5, pf5) XS_FETCH1(cc, 6, pf6) XS_FETCH1(cc, 7, pf7)
            if (wv < nchunk) { XS_FETCH(wv) }
            for (int c = wv; c < nchunk; c += 4) {
                // park the fetched chunk in this wave's LDS slice (the previous chunk's readers -- this wave -- are done)
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                if constexpr (SPLIT) {
                    // fetch register q holds key row 4 q + (lane >> 4), columns 4 (lane & 15) ..+3: 8 bytes of each image
#define XS_PARK(q, src) { half4v h_, l_; xs_split4(src, h_, l_);                                                                           \
                          char *p_ = kimg + park_blk + (4 * (q) + park_row) * 64 + ((park_piece ^ ((0 - (q)) & 3)) << 4);                  \
                          *reinterpret_cast<half4v *>(p_) = h_; *reinterpret_cast<half4v *>(p_ + 4096) = l_; }
                    XS_PARK(0, pf0) XS_PARK(1, pf1) XS_PARK(2, pf2) XS_PARK(3, pf3) XS_PARK(4, pf4) XS_PARK(5, pf5) XS_PARK(6, pf6) XS_PARK(7, pf7)
#undef XS_PARK
                    if (c + 4 < nchunk) { XS_FETCH(c + 4) }
                    // ---- S^T [32 keys x 32 rows] = K qf^T on 32x32x16: A = key rows in the order pk(li), 8 consecutive columns per lane
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const char *ap = kimg + (ks >> 1) * 2048 + ((ks & 1) ? (a1_off ^ 32) : a1_off);
                        const half8 ah = *reinterpret_cast<const half8 *>(ap), al = *reinterpret_cast<const half8 *>(ap + 4096);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qbh[ks], sacc, 0, 0, 0);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qbl[ks], sacc, 0, 0, 0);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qbh[ks], sacc, 0, 0, 0);
                    }
                } else {
#define XS_PARK(q, src) *reinterpret_cast<float4 *>(&s_kv[wv][4 * (q) + (lane >> 4)][4 * (lane & 15)]) = src;
                    XS_PARK(0, pf0) XS_PARK(1, pf1) XS_PARK(2, pf2) XS_PARK(3, pf3) XS_PARK(4, pf4) XS_PARK(5, pf5) XS_PARK(6, pf6) XS_PARK(7, pf7)
#undef XS_PARK
                    if (c + 4 < nchunk) { XS_FETCH(c + 4) }
                    // ---- S^T [32 keys x 32 rows] = K qf^T: A = keys (lane: key li, columns 8 g + 4 lh + e)
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const float4 ka = *reinterpret_cast<const float4 *>(kvw + li * XS_LD + 8 * g + 4 * lh);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qb[g].x, sacc, 0, 0, 0);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qb[g].y, sacc, 0, 0, 0);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qb[g].z, sacc, 0, 0, 0);
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qb[g].w, sacc, 0, 0, 0);
                    }
                }
                // ---- online softmax of row li over the chunk's keys: this lane holds keys (r & 3) + 8 (r >> 2) + 4 lh (SPLIT: with the
                // permuted key rows, (r & 7) + 16 (r >> 3) + 8 lh), its partner half-wave the rest
                const long kfirst = (long)32 * c + (SPLIT ? 8 : 4) * lh;
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool valid = kfirst + (SPLIT ? (r & 7) + 16 * (r >> 3) : (r & 3) + 8 * (r >> 2)) < klen;
                    sacc[r] = valid ? sacc[r] : -INFINITY;
                    mx = fmaxf(mx, sacc[r]);
                }
                mx = fmaxf(mx, xs_other_half(mx, lh));
                const float m_new = fmaxf(m_run, mx);              // finite: every chunk holds at least one valid key
                const float alpha = xs_exp(m_run - m_new);         // 0 on the wave's first chunk
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = xs_exp(sacc[r] - m_new);
                    psum += sacc[r];
                }
                psum += xs_other_half(psum, lh);
                l_run = fmaf(l_run, alpha, psum);
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ctx0[r] *= alpha; ctx1[r] *= alpha; }
                // ---- ctx^T [64 columns x 32 rows] += K^T P^T: contraction index of step s = the key register s holds
                if constexpr (SPLIT) {
                    // two k steps of 16 keys: B = the probabilities of keys 16 s + 8 lh + e = registers 8 s + e, as (hi, lo);
                    // A = K^T: 8 consecutive keys of column li (+ 32 per tile), two transposed reads of 4 keys per image
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        half8 ph_, pl_;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            ph_[e] = (_Float16)sacc[8 * ss + e];
                            pl_[e] = (_Float16)(sacc[8 * ss + e] - (float)ph_[e]);
                        }
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const char *vb = kimg + t * 2048 + ss * 1024;
                            typedef __attribute__((address_space(3))) xs_fp16x4 *lptr;
                            const half4v h0 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(vb + v_off[0])));
                            const half4v h1 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(vb + v_off[1])));
                            const half4v l0 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(vb + 4096 + v_off[0])));
                            const half4v l1 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(vb + 4096 + v_off[1])));
                            const half8 kh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), kl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                            f32x16 &cx = t ? ctx1 : ctx0;
                            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, ph_, cx, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, pl_, cx, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ph_, cx, 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int key = (s & 3) + 8 * (s >> 2) + 4 * lh;
                        const float a0 = kvw[key * XS_LD + li], a1 = kvw[key * XS_LD + 32 + li];
                        ctx0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, sacc[s], ctx0, 0, 0, 0);
                        ctx1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, sacc[s], ctx1, 0, 0, 0);
                    }
                }
            }
#undef XS_FETCH
#undef XS_FETCH1
            // ---- merge the four waves' partial (maximum, sum, context): through each wave's own LDS slice
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * lh;
                s_kv[wv][li][col] = ctx0[r];
                s_kv[wv][li][32 + col] = ctx1[r];
            }
            if (lh == 0) { s_ml[wv][0][li] = m_run; s_ml[wv][1][li] = l_run; }
            __syncthreads();
            {
                const int row = tid >> 3, c0 = 8 * (tid & 7);
                const int r_lo = 4 * (int)s_seg[sg][1], r_hi = r_lo + 4 * (int)s_seg[sg][2];
                if (row >= r_lo && row < r_hi) {
                    float mm = fmaxf(fmaxf(s_ml[0][0][row], s_ml[1][0][row]), fmaxf(s_ml[2][0][row], s_ml[3][0][row]));
                    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    float lsum = 0.f;
                    if (mm > -INFINITY) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float f = xs_exp(s_ml[w][0][row] - mm);            // 0 for a wave that saw no chunk (its maximum is -inf)
                            lsum = fmaf(s_ml[w][1][row], f, lsum);
                            const float4 u = *reinterpret_cast<const float4 *>(&s_kv[w][row][c0]), v = *reinterpret_cast<const float4 *>(&s_kv[w][row][c0 + 4]);
                            o[0] = fmaf(u.x, f, o[0]); o[1] = fmaf(u.y, f, o[1]); o[2] = fmaf(u.z, f, o[2]); o[3] = fmaf(u.w, f, o[3]);
                            o[4] = fmaf(v.x, f, o[4]); o[5] = fmaf(v.y, f, o[5]); o[6] = fmaf(v.z, f, o[6]); o[7] = fmaf(v.w, f, o[7]);
                        }
                    }
                    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;                   // no key at all: the context is zero
#pragma unroll
                    for (int e = 0; e < 8; ++e) s_ctx[row][c0 + e] = o[e] * inv;
                }
            }
            __syncthreads();
        }

        // ---- att[n][o] = sum_c ctx[4 n + (o >> 4)][c] WvT[c][o] + bv[o]: wave = quarter of c, lane = o
        {
            float acc[XS_G];
#pragma unroll
            for (int n = 0; n < XS_G; ++n) acc[n] = 0.f;
            float w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = WvT[(16 * wv + i) * XS_D + lane];
            const int hh = lane >> 4;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int n = 0; n < XS_G; ++n) {
                    const float4 cv = *reinterpret_cast<const float4 *>(&s_ctx[4 * n + hh][16 * wv + 4 * i4]);
                    acc[n] = fmaf(cv.x, w[4 * i4], acc[n]); acc[n] = fmaf(cv.y, w[4 * i4 + 1], acc[n]);
                    acc[n] = fmaf(cv.z, w[4 * i4 + 2], acc[n]); acc[n] = fmaf(cv.w, w[4 * i4 + 3], acc[n]);
                }
#pragma unroll
            for (int n = 0; n < XS_G; ++n) s_part[wv][n][lane] = acc[n];
        }
        __syncthreads();
        for (int i = tid; i < XS_G * XS_D; i += 256) {
            const int n = i >> 6, o = i & 63;
            s_q[n][o] = ((s_part[0][n][o] + s_part[1][n][o]) + (s_part[2][n][o] + s_part[3][n][o])) + bv[o];
        }
        __syncthreads();
        // ---- out[n][o] = sum_i att[n][i] WoT[i][o] + bo[o]
        {
            float acc[XS_G];
#pragma unroll
            for (int n = 0; n < XS_G; ++n) acc[n] = 0.f;
            float w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = WoT[(16 * wv + i) * XS_D + lane];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int n = 0; n < XS_G; ++n) {
                    const float4 av = *reinterpret_cast<const float4 *>(&s_q[n][16 * wv + 4 * i4]);
                    acc[n] = fmaf(av.x, w[4 * i4], acc[n]); acc[n] = fmaf(av.y, w[4 * i4 + 1], acc[n]);
                    acc[n] = fmaf(av.z, w[4 * i4 + 2], acc[n]); acc[n] = fmaf(av.w, w[4 * i4 + 3], acc[n]);
                }
#pragma unroll
            for (int n = 0; n < XS_G; ++n) s_part[wv][n][lane] = acc[n];
        }
        __syncthreads();
        // ---- residual + LayerNorm (:47-50): wave = rows wv, wv + 4; lane = column
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = wv + 4 * k;
            const float v = s_x[n][lane] + (((s_part[0][n][lane] + s_part[1][n][lane]) + (s_part[2][n][lane] + s_part[3][n][lane])) + bo[lane]);
            const float mean = wave_butterfly_sum(v) * (1.0f / XS_D);
            const float dlt = v - mean;
            const float var = wave_butterfly_sum(dlt * dlt) * (1.0f / XS_D);
            const float y = dlt * rsqrtf(var + a.ln_eps) * gamma[lane] + beta[lane];
            s_x[n][lane] = n < n_live ? y : 0.f;
        }
        __syncthreads();
    }

    // ---- results
    if (text_side) {
        const long b = (long)blockIdx.x - a.n_graph_tiles;
        if (tid < XS_D) a.pooled[b * a.pooled_stride + tid] = s_x[0][tid];
    } else {
        for (int i = tid; i < n_live * XS_D; i += 256) a.y_nodes[(row0 + (i >> 6)) * XS_D + (i & 63)] = s_x[i >> 6][i & 63];
    }
}

// pooled_graph[b] = mean of the attended rows of code b, rows added in node order (one ordered chain per column; a code without
// nodes gives zeros) -- the `.mean(dim=0)` of :140-141.  One wavefront per code; its node range by binary search of the batch vector.
__global__ __launch_bounds__(256) void cross_attention64_mean_kernel(const float *__restrict__ y_nodes, const int64_t *__restrict__ batch, long n_nodes,
                                                                     long n_codes, float *__restrict__ pooled, long pooled_stride, long graph_off)
{
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_codes) return;
    const int lane = threadIdx.x & 63;
    const long lo = xs_lower_bound(batch, n_nodes, b), hi = xs_lower_bound(batch, n_nodes, b + 1);
    // (eight loads in flight, added in node order: one ordered chain per column whatever the batching)
    float s = 0.f;
    long r = lo;
    for (; r + 8 <= hi; r += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = y_nodes[(r + i) * XS_D + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; r < hi; ++r) s += y_nodes[r * XS_D + lane];
    pooled[b * pooled_stride + graph_off + lane] = hi > lo ? s / (float)(hi - lo) : 0.f;
}
