// attention_kernels.h -- variable-length "shared key/value" attention core for get_shared_info's cross-attention
// (vector_quantization_soft_one_new.py:17-88,133-142).  Included by medtok_vq.hip; gfx950 only.
//
// The host folds nn.MultiheadAttention's key/value projections into the query side (see
// CrossAttention._folded_layer), which leaves, per medical code b,
//     out[r, :] = softmax_j( scale * <q[r, :], kv[j, :]> ) . kv            r in the code's query rows, j in its key rows
// with the RAW rows of the other modality as both keys and values, shared by all heads (a head is just another
// query row).  Rows are ragged: code b owns q rows [q_start[b], +q_len[b]) and kv rows [kv_start[b], +kv_len[b]);
// nothing is padded, nothing of size rows x keys reaches memory.
//
// Block = (32 query rows of one code) x all of its keys, W = 8 waves (D % 256 == 0), 4 waves (other multiples of 128) or 2 waves (D = 64).  Both products run on the exact
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TF peak) -- fp16/bf16 inputs would break the 1e-5 parity bar:
//   keys   : each wave owns D / W columns of everything -- queries, keys, outputs -- so it fetches, parks and reads only its
//            own column slice of a chunk of 32 key rows: full 128-byte lines per load (a thread's MFMA operands would be
//            16 B out of every line, re-fetching each line four times through a thrashing L1), parked in the wave's own LDS
//            slice (row stride 32 NT + 4 floats: conflict-free for both operand shapes below) where both products read it.
//            No barrier guards the chunk buffer (a wave's LDS operations complete in order); the two barriers per chunk are
//            those around the softmax step (LDS-only: s_waitcnt lgkmcnt(0) + s_barrier, never a vmcnt drain).
//            The next chunk's fetch is woven into the score MFMAs of this one (one load per four MFMAs) and stays in flight
//            under the softmax step and the second product;
//   scores : each wave owns D / W columns, keeps its slice of the 32 query rows in registers for the block's lifetime
//            and accumulates a partial 32 x 32 score tile per chunk; the W partials meet in LDS;
//   softmax: online (running max / sum per row), 8 threads per row, exact expf;
//   output : each wave owns D / W output columns = NT tiles of 32, accumulators rescaled per chunk by exp(m_old - m_new).
#pragma once

// Dropout on the attention probabilities (nn.MultiheadAttention(dropout=0.1) in training, reference :21,30): element (query row r
// of the packed query matrix, key j of its code) is kept when a 32-bit hash of (seed, r, j) clears the threshold -- stateless, so
// the forward and both backward kernels regenerate the same mask; kept probabilities are scaled by 1 / (1 - p).
__device__ __forceinline__ bool att_keep(unsigned seed, long qrow, int key, unsigned thresh)
{
    unsigned h = seed ^ ((unsigned)qrow * 0x9E3779B1u) ^ ((unsigned)(qrow >> 32) * 0x7F4A7C15u) ^ ((unsigned)key * 0x85EBCA77u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h >= thresh;                         // thresh = p * 2^32: P(keep) = 1 - p
}

// Reductions over the TPR (4, 8 or 16) consecutive lanes that share a score row, on the VALU's DPP path: quad permutes, then the
// mirror of 8 and of 16 lanes.  Same tree as an xor butterfly (same bits, every lane ends with the result) without its LDS
// round trips (__shfl_xor is ds_bpermute_b32: eight dependent ones per chunk here; 9.87 -> 9.65 us per 32-key chunk at D = 768;
// operands of the second product two steps ahead instead of one: no change).
template <int CTRL>
__device__ __forceinline__ float att_dpp(float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int TPR>
__device__ __forceinline__ float att_group_max(float v)
{
    v = fmaxf(v, att_dpp<0xB1>(v));                       // quad_perm [1,0,3,2]
    v = fmaxf(v, att_dpp<0x4E>(v));                       // quad_perm [2,3,0,1]
    if (TPR >= 8) v = fmaxf(v, att_dpp<0x141>(v));        // row_half_mirror
    if (TPR >= 16) v = fmaxf(v, att_dpp<0x140>(v));       // row_mirror
    return v;
}
template <int TPR>
__device__ __forceinline__ float att_group_sum(float v)
{
    v += att_dpp<0xB1>(v);
    v += att_dpp<0x4E>(v);
    if (TPR >= 8) v += att_dpp<0x141>(v);
    if (TPR >= 16) v += att_dpp<0x140>(v);
    return v;
}

// Eight waves keep the per-lane state (query slice + output tiles + key fetch = 12 NT registers each) inside the 256
// architectural VGPRs; with four waves at D = 768 hipcc parks the query slice in AGPRs and serialises the fetch.
template <int W, int NT>      // waves per block, output column tiles per wave; D = 32 * W * NT
__global__ __launch_bounds__(64 * W) void shared_kv_attention_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const float *__restrict__ kv, const int64_t *__restrict__ kv_start, const int64_t *__restrict__ kv_len,
    float scale, float *__restrict__ out, int q_tiles, float *__restrict__ lse = nullptr, unsigned drop_thresh = 0, unsigned seed = 0,
    float keep_scale = 1.f)
{
    constexpr int D = 32 * W * NT;
    constexpr int SL = 32 * NT + 4;                // LDS row stride of a wave's key slice in floats (conflict-free for both operand shapes)
    constexpr int KV_FLOATS = W * 32 * SL;         // W private slices of 32 key rows
    constexpr int THREADS = 64 * W;
    constexpr int EPT = 1024 / THREADS;            // score elements per thread in the softmax step (4 or 2)
    constexpr int TPR = 32 / EPT;                  // threads per score row (8 or 16)
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    float *kvs = att_sm;                                                              // [W][32][SL] current key chunk, one column slice per wave
    float (*part)[32][33] = reinterpret_cast<float (*)[32][33]>(kvs + KV_FLOATS);       // [W][32][33] per-wave partial scores [row][key]
    float (*pt)[33] = reinterpret_cast<float (*)[33]>(kvs + KV_FLOATS + W * 32 * 33);   // [32][33] probabilities, transposed [key][row]
    float *alpha_s = kvs + KV_FLOATS + (W + 1) * 32 * 33, *l_s = alpha_s + 32;
    // 1-D grid: block id = code * q_tiles + query tile (the tiles of one code stay adjacent: they share its keys in the L2;
    // grid.y would cap a call at 65535 codes)
    const int b = (int)(blockIdx.x / (unsigned)q_tiles), qt = (int)(blockIdx.x % (unsigned)q_tiles);
    const int ql = (int)q_len[b];
    if (qt * 32 >= ql) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int slice = wave * 32 * NT;      // this wave's D columns [slice, slice + 32 NT)

    // Key chunk: every wave fetches, parks and reads ONLY its own column slice of the 32 key rows -- both products need nothing
    // else -- so the chunk buffer carries no dependency between waves: no barrier after the park, none before the next one,
    // and a wave that is done with a chunk parks the next while the others still multiply.  Lane (lr, lc) = (lane / 8, lane % 8)
    // loads float4 #(8 ic + lc) of slice row 8 ir + lr for load i = (ir, ic) = (i / NT, i % NT): eight full 128-byte lines per
    // wave instruction (the slice starts on a line: 128 NT bytes per wave).  Rows past the code's last key are clamped (their
    // probabilities are zero); every LDS / global offset is a compile-time constant off a per-lane base.
    constexpr int NF = 4 * NT;
    const int lr = lane >> 3, lc = lane & 7;
    float *kw = kvs + wave * 32 * SL;              // this wave's slice [32][SL]
    float4 kf[NF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) {
            const float *src = kv + (ks + min(k0 + 8 * ir + lr, kl - 1)) * (long)D + slice + 4 * lc;
#pragma unroll
            for (int ic = 0; ic < NT; ++ic) kf[ir * NT + ic] = ld4(src + 32 * ic);
        }
    };
    // rows 8 ir .. 8 ir + 7 of the fetched chunk: registers -> the wave's slice
    auto park_rows = [&](int ir) __attribute__((always_inline)) {
        float *dst = kw + (8 * ir + lr) * SL + 4 * lc;
#pragma unroll
        for (int ic = 0; ic < NT; ++ic) *reinterpret_cast<float4 *>(dst + 32 * ic) = kf[ir * NT + ic];
    };
    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the key fetch that is meant to stay in
    // flight under the second product
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (kl > 0) fetch(0);
    bool parked0 = false;

    // query slice of row (qt*32 + li): element 8g + 4 lh + j feeds MFMA step 4g + j (k index lh); keys use the same map
    float4 qf[4 * NT];
    {
        const float *qrow = q + (qs + min(qt * 32 + li, ql - 1)) * (long)D + slice + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4 * NT; ++g) qf[g] = ld4(qrow + 8 * g);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float m_run = -INFINITY, l_run = 0.f;          // online-softmax state of row tid / TPR, replicated in its TPR threads
    for (int k0 = 0; k0 < kl; k0 += 32) {
        // The first chunk is parked here; every later one was parked during the second product of the chunk before it (below).
        if (!parked0) {
#pragma unroll
            for (int ir = 0; ir < 4; ++ir) park_rows(ir);
            parked0 = true;
        }
        // next chunk: one 16-byte load per group of four score MFMAs below (NF = 4 NT loads, 4 NT groups), so that their issue
        // rides in the matrix pipe's shadow -- issued as one burst (12 wave-instructions x 8 waves through the CU's one address
        // path) they held every wave for ~1 us per chunk: 9.36 -> 8.30 us per chunk at D = 768.  (The loads stay unconditional;
        // past the last key they re-read one row.)
        const float *fsrc[4];
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) fsrc[ir] = kv + (ks + min(k0 + 32 + 8 * ir + lr, kl - 1)) * (long)D + slice + 4 * lc;

        // ---- partial scores over this wave's D / W columns
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            // operands four groups (16 MFMAs ~ 1000 cycles of pipe) ahead of their use; the empty asm keeps hipcc from
            // hoisting every ds_read of the fully unrolled loop to the top
            const float *krow = kw + li * SL + 4 * lh;
            float4 cur[4], nxt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = *reinterpret_cast<const float4 *>(krow + 8 * j);
#pragma unroll
            for (int gb = 0; gb < NT; ++gb) {
                if (gb + 1 < NT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const float4 *>(krow + 8 * (4 * (gb + 1) + j));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g = 4 * gb + j;
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g].x, cur[j].x, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g].y, cur[j].y, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g].z, cur[j].z, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g].w, cur[j].w, s, 0, 0, 0);
                    kf[g] = ld4(fsrc[g / NT] + 32 * (g % NT));
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
        lds_barrier();

        // ---- join the W partials, online softmax: thread -> row tid / TPR, keys EPT (tid % TPR) .. + EPT - 1
        {
            const int row = tid / TPR, kq = (tid % TPR) * EPT;
            float v[EPT], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) sc += part[w2][row][kq + j];
                sc *= scale;
                v[j] = (k0 + kq + j < kl) ? sc : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
            static_assert(TPR == 4 || TPR == 8 || TPR == 16, "score-row groups of 4, 8 or 16 lanes");
            mx = att_group_max<TPR>(mx);
            const float m_new = fmaxf(m_run, mx);           // finite: every chunk holds at least one valid key
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float p = expf(v[j] - m_new);         // exp(-inf) = 0 for masked keys
                // training: dropped probabilities do not reach the value product; the normaliser is that of the full softmax
                pt[kq + j][row] = (drop_thresh && !att_keep(seed, qs + qt * 32 + row, k0 + kq + j, drop_thresh)) ? 0.f : p;
                ps += p;
            }
            ps = att_group_sum<TPR>(ps);
            const float a = expf(m_run - m_new);            // 0 on the first chunk (m_run = -inf)
            l_run = fmaf(l_run, a, ps);
            m_run = m_new;
            if (tid % TPR == 0) alpha_s[row] = a;
        }
        lds_barrier();

        // ---- out = alpha * out + P . KV  on this wave's NT column tiles
        {
            float a16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a16[r] = alpha_s[(r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= a16[r];
            const float *kcol = kw + lh * SL + li;
            float pc, pn = 0.f, kc[NT], kn[NT];
            pc = pt[lh][li];
#pragma unroll
            for (int t = 0; t < NT; ++t) { kc[t] = kcol[32 * t]; kn[t] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                if (s2 + 1 < 16) {
                    pn = pt[2 * (s2 + 1) + lh][li];
#pragma unroll
                    for (int t = 0; t < NT; ++t) kn[t] = kcol[2 * (s2 + 1) * SL + 32 * t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pc, kc[t], acc[t], 0, 0, 0);
                asm volatile("" ::: "memory");
                // Step s2 was the last reader of key rows 2 s2, 2 s2 + 1 (the operands of step s2 + 1 are already in registers):
                // once eight rows are dead the NEXT chunk's rows go into their place -- the park rides between the MFMAs instead
                // of standing in front of the next chunk's (a wave's LDS operations complete in order; nobody else reads this slice).
                if (s2 % 4 == 3 && s2 < 15) park_rows(s2 / 4);
                pc = pn;
#pragma unroll
                for (int t = 0; t < NT; ++t) kc[t] = kn[t];
            }
            park_rows(3);
        }
        // (no barrier here: the key slice is this wave's own, and the score / probability tiles are next written behind the
        // two barriers of the next chunk)
    }
    if (tid % TPR == 0) {
        l_s[tid / TPR] = l_run;
        // log-sum-exp of the scaled scores (the backward kernels rebuild the probabilities from it); -inf for an empty key set
        if (lse && qt * 32 + tid / TPR < ql) lse[qs + qt * 32 + tid / TPR] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (qt * 32 + row < ql) {
            const float lsum = l_s[row];
            const float inv = lsum > 0.f ? keep_scale / lsum : 0.f;     // a code with no key rows (empty graph / empty text) attends to nothing: context 0, not 0/0
            float *o = out + (qs + qt * 32 + row) * (long)D + slice + li;
#pragma unroll
            for (int t = 0; t < NT; ++t) o[32 * t] = acc[t][r] * inv;
        }
    }
}

// Output of one 32-row tile for the inference kernels: the accumulators (one column slice per wave, MFMA C layout) go through LDS
// so that every row leaves as full 16-byte pieces -- fp32 (out) and/or the (hi, lo) fp16 images (oh, ol) that the next dense
// product reads directly (split_gemm.h): no conversion pass over the [rows x heads, D] context in between.  `stage` = LDS the
// main loop no longer needs, 32 x (D / HALVES + 4) floats: the tile goes out in HALVES column passes (the waves whose slices
// lie in the pass write, everybody stores).  All threads of the block call it (two barriers per pass).
template <int W, int NT, int HALVES = 1>
__device__ __forceinline__ void att_store_tile(float *stage, const f32x16 (&acc)[NT], const float (&l16)[16], int rows_valid, long row0,
                                               float *__restrict__ out, _Float16 *__restrict__ oh, _Float16 *__restrict__ ol,
                                               int slice, int li, int lh, int tid)
{
    constexpr int D = 32 * W * NT, DH = D / HALVES, SD = DH + 4, THREADS = 64 * W;
    static_assert(W % HALVES == 0, "a wave's slice lies in one pass");
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
        if (slice / DH == h) {                                          // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;        // l16[r] = the softmax denominator of that row
                const float inv = l16[r] > 0.f ? 1.f / l16[r] : 0.f;    // a code with no key rows attends to nothing: context 0, not 0/0
#pragma unroll
                for (int t = 0; t < NT; ++t) stage[row * SD + slice - h * DH + li + 32 * t] = acc[t][r] * inv;
            }
        }
        __syncthreads();
        for (int i = tid; i < 32 * (DH / 8); i += THREADS) {
            const int row = i / (DH / 8), c8 = (i % (DH / 8)) * 8;
            if (row >= rows_valid) continue;
            const float4 a = *reinterpret_cast<const float4 *>(stage + row * SD + c8), b = *reinterpret_cast<const float4 *>(stage + row * SD + c8 + 4);
            const long o = (row0 + row) * (long)D + h * DH + c8;
            if (out) { st4(out + o, a); st4(out + o + 4, b); }
            if (oh) {
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                half8 hh, ll;
#pragma unroll
                for (int e = 0; e < 8; ++e) { hh[e] = (_Float16)v[e]; ll[e] = (_Float16)(v[e] - (float)hh[e]); }
                *reinterpret_cast<half8 *>(oh + o) = hh;
                *reinterpret_cast<half8 *>(ol + o) = ll;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- the same attention core on the fp16 matrix pipe (inference)
// out[r] = softmax_j(scale <q[r], kv[j]>) . kv with both products as THREE v_mfma_f32_32x32x16_f16 each on (hi, lo) fp16 pairs
// (x = hi + lo, hi = fp16(x), lo = fp16(x - hi): hi hi + hi lo + lo hi reproduces the fp32 product to ~2^-22 relative, see
// split_gemm.h) -- 16/3 of the fp32 pipe's rate, with the fp32 softmax between them unchanged.  Attention outputs are tolerance
// items (1e-5), checked against the oracle like the fp32 kernel above, which stays the training forward (dropout, log-sum-exp).
//
// Same decomposition as the fp32 kernel: block = 32 query rows of one code x all its keys, W waves that each own D / W columns
// of queries, keys and outputs, 32-key chunks, wave-private key slices in LDS (no barrier guards them), two LDS-only barriers
// per chunk around the softmax.  What differs:
//   keys    : a wave converts the fp32 rows it fetched to (hi, lo) on the way into LDS -- two fp16 planes per slice, row stride
//             32 NT + 8 halves (conflict-free ds_read_b128 of 8 consecutive columns of 32 different keys);
//   scores  : the query slice lives in registers as (hi, lo) MFMA A operands; per 16 columns one b128 read of each plane and
//             three MFMAs; the next chunk's global loads ride between them, two per k step;
//   values  : the second product contracts over KEYS, so its B operand is 8 consecutive keys of one column: sixteen 16-bit LDS
//             reads per operand pair straight into register halves (ds_read_u16_d16 / _d16_hi) -- the probabilities are
//             written by the softmax step as (hi, lo) rows [query][key] and read with one b128 per plane and k step.
template <int W, int NT>
__global__ __launch_bounds__(64 * W) void shared_kv_attention_f16s_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const float *__restrict__ kv, const int64_t *__restrict__ kv_start, const int64_t *__restrict__ kv_len,
    float scale, float *__restrict__ out, _Float16 *__restrict__ out_h, _Float16 *__restrict__ out_l, int q_tiles)
{
    constexpr int D = 32 * W * NT;
    constexpr int SLH = 32 * NT + 8;               // halves per key row of a wave's plane
    constexpr int PLANE = 32 * SLH;                // halves per plane (32 keys)
    constexpr int KV_HALVES = W * 2 * PLANE;       // W slices x (hi, lo)
    constexpr int THREADS = 64 * W;
    constexpr int EPT = 1024 / THREADS;            // score elements per thread in the softmax step (2, 4 or 8)
    constexpr int TPR = 32 / EPT;                  // threads per score row
    constexpr int PSL = 40;                        // halves per probability row: 32 keys + 8 (conflict-free b128 reads)
    typedef _Float16 halfE __attribute__((ext_vector_type(EPT)));
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    _Float16 *kvs = reinterpret_cast<_Float16 *>(att_sm);                                     // [W][2][32][SLH]
    float (*part)[32][33] = reinterpret_cast<float (*)[32][33]>(kvs + KV_HALVES);               // [W][32][33] partial scores [row][key]
    _Float16 *ph = reinterpret_cast<_Float16 *>(&part[W][0][0]), *pl = ph + 32 * PSL;          // probabilities [row][key], hi and lo
    float *alpha_s = reinterpret_cast<float *>(pl + 32 * PSL), *l_s = alpha_s + 32;
    const int b = (int)(blockIdx.x / (unsigned)q_tiles), qt = (int)(blockIdx.x % (unsigned)q_tiles);
    const int nq = (int)q_len[b];
    if (qt * 32 >= nq) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int slice = wave * 32 * NT;
    _Float16 *kw_h = kvs + wave * 2 * PLANE, *kw_l = kw_h + PLANE;

    constexpr int NF = 4 * NT;
    const int lr = lane >> 3, lc = lane & 7;
    float4 kf[NF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) {
            const float *src = kv + (ks + min(k0 + 8 * ir + lr, kl - 1)) * (long)D + slice + 4 * lc;
#pragma unroll
            for (int ic = 0; ic < NT; ++ic) kf[ir * NT + ic] = ld4(src + 32 * ic);
        }
    };
    // rows 8 ir .. 8 ir + 7 of the fetched chunk: fp32 registers -> (hi, lo) planes of the wave's slice
    auto park_rows = [&](int ir) __attribute__((always_inline)) {
        _Float16 *dh = kw_h + (8 * ir + lr) * SLH + 4 * lc, *dl = kw_l + (8 * ir + lr) * SLH + 4 * lc;
#pragma unroll
        for (int ic = 0; ic < NT; ++ic) {
            const float4 x = kf[ir * NT + ic];
            half4v h, l;
            h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
            l[0] = (_Float16)(x.x - (float)h[0]); l[1] = (_Float16)(x.y - (float)h[1]);
            l[2] = (_Float16)(x.z - (float)h[2]); l[3] = (_Float16)(x.w - (float)h[3]);
            *reinterpret_cast<half4v *>(dh + 32 * ic) = h;
            *reinterpret_cast<half4v *>(dl + 32 * ic) = l;
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (kl > 0) fetch(0);
    bool parked0 = false;

    // query slice as MFMA A operands: k step s covers columns slice + 16 s .. + 15; lane (li, lh) holds row li, columns + 8 lh .. + 7
    half8 qh[2 * NT], qlo[2 * NT];
    {
        const float *qrow = q + (qs + min(qt * 32 + li, nq - 1)) * (long)D + slice + 8 * lh;
#pragma unroll
        for (int s = 0; s < 2 * NT; ++s) {
            const float4 a = ld4(qrow + 16 * s), c = ld4(qrow + 16 * s + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qh[s][e] = (_Float16)v[e];
                qlo[s][e] = (_Float16)(v[e] - (float)qh[s][e]);
            }
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float m_run = -INFINITY, l_run = 0.f;
    for (int k0 = 0; k0 < kl; k0 += 32) {
        if (!parked0) {
#pragma unroll
            for (int ir = 0; ir < 4; ++ir) park_rows(ir);
            parked0 = true;
        }
        const float *fsrc[4];
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) fsrc[ir] = kv + (ks + min(k0 + 32 + 8 * ir + lr, kl - 1)) * (long)D + slice + 4 * lc;

        // ---- partial scores over this wave's D / W columns: hi hi into one accumulator, the two cross terms into another
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        {
            const _Float16 *krow_h = kw_h + li * SLH + 8 * lh, *krow_l = kw_l + li * SLH + 8 * lh;
#pragma unroll
            for (int s = 0; s < 2 * NT; ++s) {
                const half8 kh = *reinterpret_cast<const half8 *>(krow_h + 16 * s);
                const half8 kq = *reinterpret_cast<const half8 *>(krow_l + 16 * s);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], kh, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], kq, s1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qlo[s], kh, s1, 0, 0, 0);
                // next chunk: two of its 4 NT loads per k step (unconditional; past the last key they re-read one row)
                kf[2 * s] = ld4(fsrc[(2 * s) / NT] + 32 * ((2 * s) % NT));
                kf[2 * s + 1] = ld4(fsrc[(2 * s + 1) / NT] + 32 * ((2 * s + 1) % NT));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s0[r] + s1[r];
        lds_barrier();

        // ---- join the W partials, online softmax: thread -> row tid / TPR, keys EPT (tid % TPR) .. + EPT - 1
        {
            const int row = tid / TPR, kq0 = (tid % TPR) * EPT;
            float v[EPT], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) sc += part[w2][row][kq0 + j];
                sc *= scale;
                v[j] = (k0 + kq0 + j < kl) ? sc : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
            static_assert(TPR == 4 || TPR == 8 || TPR == 16, "score-row groups of 4, 8 or 16 lanes");
            mx = att_group_max<TPR>(mx);
            const float m_new = fmaxf(m_run, mx);
            float psum = 0.f;
            halfE hv, lv;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float pr = expf(v[j] - m_new);
                hv[j] = (_Float16)pr;
                lv[j] = (_Float16)(pr - (float)hv[j]);
                psum += pr;
            }
            *reinterpret_cast<halfE *>(ph + row * PSL + kq0) = hv;
            *reinterpret_cast<halfE *>(pl + row * PSL + kq0) = lv;
            psum = att_group_sum<TPR>(psum);
            const float a = expf(m_run - m_new);
            l_run = fmaf(l_run, a, psum);
            m_run = m_new;
            if (tid % TPR == 0) alpha_s[row] = a;
        }
        lds_barrier();

        // ---- out = alpha * out + P . KV on this wave's NT column tiles; the contraction runs over the chunk's 32 keys (2 k steps)
        {
            float a16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a16[r] = alpha_s[(r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= a16[r];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 pa_h = *reinterpret_cast<const half8 *>(ph + li * PSL + 16 * s + 8 * lh);
                const half8 pa_l = *reinterpret_cast<const half8 *>(pl + li * PSL + 16 * s + 8 * lh);
                const _Float16 *vcol_h = kw_h + (16 * s + 8 * lh) * SLH + li, *vcol_l = kw_l + (16 * s + 8 * lh) * SLH + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    half8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { vh[e] = vcol_h[e * SLH + 32 * t]; vl[e] = vcol_l[e * SLH + 32 * t]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, vh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, vl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_l, vh, acc[t], 0, 0, 0);
                }
                asm volatile("" ::: "memory");
                // keys 16 s .. 16 s + 15 have had their last reader (a wave's LDS operations complete in order; nobody else reads
                // this slice): the next chunk's rows take their place
                park_rows(2 * s);
                park_rows(2 * s + 1);
            }
        }
    }
    // the row sums move to registers first: the staging tile overwrites the LDS they live in
    __syncthreads();
    if (tid % TPR == 0) l_s[tid / TPR] = l_run;
    __syncthreads();
    float l16[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) l16[r] = l_s[(r & 3) + 8 * (r >> 2) + 4 * lh];
    __syncthreads();
    att_store_tile<W, NT>(att_sm, acc, l16, nq - qt * 32, qs + qt * 32, out, out_h, out_l, slice, li, lh, tid);
}

// ---------------------------------------------------------------- the same core for a FEW query rows per code
// The text side of get_shared_info (vector_quantization_soft_one_new.py:133-139; only the CLS row of the attended text is used,
// :139): one query row per code and head against that code's graph nodes -- 4 rows x <= a few dozen keys.  On the 32-row matrix
// kernels that is one eighth of a tile per 512-thread block with 68 KB of LDS: 0.2 ms per launch alone and, beside the graph
// side's launches, up to 2.4 ms of waiting for LDS and wave slots.  Here ONE WAVEFRONT owns a code: lane l holds columns
// 4 l + 256 t of its <= NQ query rows and output rows in registers, the keys stream through once (the next key's loads in
// flight under the current key's arithmetic), scores are plain fp32 dot products (per lane in increasing column order, then the
// row-of-16 DPP butterfly and the four row sums in a fixed order), online softmax per key.  No LDS, no barrier, no matrix pipe:
// fp32 FMA arithmetic (more accurate than the three-pass fp16 form; the same tolerance class).  d % 4 == 0, d <= 256 CT.
__device__ __forceinline__ float fewq_wave_sum(float v)
{
    v += att_dpp<0xB1>(v);
    v += att_dpp<0x4E>(v);
    v += att_dpp<0x141>(v);
    v += att_dpp<0x140>(v);                                   // every lane: the sum of its row of 16
    const int i = (int)__float_as_uint(v);
    return (__uint_as_float((unsigned)__builtin_amdgcn_readlane(i, 0)) + __uint_as_float((unsigned)__builtin_amdgcn_readlane(i, 16))) +
           (__uint_as_float((unsigned)__builtin_amdgcn_readlane(i, 32)) + __uint_as_float((unsigned)__builtin_amdgcn_readlane(i, 48)));
}

template <int NQ, int CT = 3>                                 // CT column chunks of 256 per lane: 3 (d <= 768), 4 (d <= 1024)
__global__ __launch_bounds__(256) void shared_kv_attention_fewq_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const float *__restrict__ kv, const int64_t *__restrict__ kv_start, const int64_t *__restrict__ kv_len,
    long n_codes, int d, float scale, float *__restrict__ out, _Float16 *__restrict__ out_h, _Float16 *__restrict__ out_l)
{
    const int lane = threadIdx.x & 63;
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_codes) return;
    const int nq = min((int)q_len[b], NQ);
    if (nq <= 0) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    bool on[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) on[t] = 4 * lane + 256 * t < d;
    float4 qv[NQ][CT], acc[NQ][CT];
    float m[NQ], l[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            acc[r][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[r][t] = (r < nq && on[t]) ? ld4(q + (qs + r) * (long)d + 4 * lane + 256 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 kc[CT], kn[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        kc[t] = (kl > 0 && on[t]) ? ld4(kv + ks * (long)d + 4 * lane + 256 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        kn[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int j = 0; j < kl; ++j) {
        if (j + 1 < kl) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
                if (on[t]) kn[t] = ld4(kv + (ks + j + 1) * (long)d + 4 * lane + 256 * t);
        }
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            if (r >= nq) break;                               // (wave-uniform)
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                p = fmaf(qv[r][t].x, kc[t].x, p); p = fmaf(qv[r][t].y, kc[t].y, p);
                p = fmaf(qv[r][t].z, kc[t].z, p); p = fmaf(qv[r][t].w, kc[t].w, p);
            }
            const float s = fewq_wave_sum(p) * scale;
            const float m_new = fmaxf(m[r], s);
            const float a = expf(m[r] - m_new);               // 0 at the first key (m = -inf)
            const float pr = expf(s - m_new);
            l[r] = fmaf(l[r], a, pr);
            m[r] = m_new;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                acc[r][t].x = fmaf(pr, kc[t].x, acc[r][t].x * a); acc[r][t].y = fmaf(pr, kc[t].y, acc[r][t].y * a);
                acc[r][t].z = fmaf(pr, kc[t].z, acc[r][t].z * a); acc[r][t].w = fmaf(pr, kc[t].w, acc[r][t].w * a);
            }
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) kc[t] = kn[t];
    }
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
        if (r >= nq) break;
        const float inv = l[r] > 0.f ? 1.f / l[r] : 0.f;      // a code with no key rows attends to nothing: context 0, not 0/0
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if (!on[t]) continue;
            const float4 v = make_float4(acc[r][t].x * inv, acc[r][t].y * inv, acc[r][t].z * inv, acc[r][t].w * inv);
            const long o = (qs + r) * (long)d + 4 * lane + 256 * t;
            if (out) st4(out + o, v);
            if (out_h) {
                half4v hh, ll;
                hh[0] = (_Float16)v.x; hh[1] = (_Float16)v.y; hh[2] = (_Float16)v.z; hh[3] = (_Float16)v.w;
                ll[0] = (_Float16)(v.x - (float)hh[0]); ll[1] = (_Float16)(v.y - (float)hh[1]);
                ll[2] = (_Float16)(v.z - (float)hh[2]); ll[3] = (_Float16)(v.w - (float)hh[3]);
                *reinterpret_cast<half4v *>(out_h + o) = hh;
                *reinterpret_cast<half4v *>(out_l + o) = ll;
            }
        }
    }
}

// ---------------------------------------------------------------- around the core: residual + LayerNorm, node mean
// CrossAttentionLayer's tail (vector_quantization_soft_one_new.py:47-50):  y[r] = LayerNorm(a[r] + b[r]) * gamma + beta  (biased
// variance, eps inside the square root).  One wavefront per row; the row a + b stays in registers between the three passes
// (mean, centred sum of squares, output), so HBM sees each operand once: 12 D bytes per row instead of the 20 D of an add kernel
// followed by a LayerNorm kernel.  Lane l owns float4 #(l + 64 t): the per-lane sums run in increasing element order and meet in
// the xor butterfly -- the order oracle_residual_layernorm_f32 restates.  d % 4 == 0, d <= 4096.
// y_hi / y_lo (optional): the (hi, lo) fp16 images [n, dp] of y as medtok_split_half_f32 would make them (dp >= d, zero columns
// appended) -- the next layer's first dense product reads them, and a separate pass over y is saved.
constexpr int LN_MAXV = 16;
template <bool SPLIT>
__global__ __launch_bounds__(256) void residual_layernorm_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 long n, int d, float eps, float *__restrict__ y,
                                                                 _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int dp)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int cpr = d >> 2;                                  // float4 chunks per row
    const float *pa = a + row * d, *pb = b + row * d;
    float4 v[LN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int c = lane + 64 * t;
        if (c < cpr) {
            const float4 x = ld4(pa + 4 * c), z = ld4(pb + 4 * c);
            v[t] = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w);
            sum += v[t].x; sum += v[t].y; sum += v[t].z; sum += v[t].w;
        }
    }
    const float mean = wave_butterfly_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        if (lane + 64 * t < cpr) {
            v[t].x -= mean; v[t].y -= mean; v[t].z -= mean; v[t].w -= mean;
            sq = fmaf(v[t].x, v[t].x, sq); sq = fmaf(v[t].y, v[t].y, sq); sq = fmaf(v[t].z, v[t].z, sq); sq = fmaf(v[t].w, v[t].w, sq);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_butterfly_sum(sq) / (float)d + eps);
    float *py = y + row * d;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int c = lane + 64 * t;
        if (c < cpr) {
            const float4 g = ld4(gamma + 4 * c), h = ld4(beta + 4 * c);
            const float4 o = make_float4(fmaf(v[t].x * rstd, g.x, h.x), fmaf(v[t].y * rstd, g.y, h.y), fmaf(v[t].z * rstd, g.z, h.z),
                                         fmaf(v[t].w * rstd, g.w, h.w));
            st4(py + 4 * c, o);
            if (SPLIT) {
                half4v hh, ll;
                hh[0] = (_Float16)o.x; hh[1] = (_Float16)o.y; hh[2] = (_Float16)o.z; hh[3] = (_Float16)o.w;
                ll[0] = (_Float16)(o.x - (float)hh[0]); ll[1] = (_Float16)(o.y - (float)hh[1]);
                ll[2] = (_Float16)(o.z - (float)hh[2]); ll[3] = (_Float16)(o.w - (float)hh[3]);
                *reinterpret_cast<half4v *>(y_hi + row * dp + 4 * c) = hh;
                *reinterpret_cast<half4v *>(y_lo + row * dp + 4 * c) = ll;
            }
        } else if (SPLIT && 4 * c < dp) {
            const half4v z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<half4v *>(y_hi + row * dp + 4 * c) = z;
            *reinterpret_cast<half4v *>(y_lo + row * dp + 4 * c) = z;
        }
    }
}

// Mean of each code's attended graph nodes (:140-141): out[b] = (sum of rows [start[b], start[b] + len[b]) of x) / max(len[b], 1),
// rows added in order (one fp32 chain per column: deterministic, no atomics, nothing padded).  Thread = one float4 column.
__global__ __launch_bounds__(256) void segment_mean_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg_start,
                                                           const int64_t *__restrict__ seg_len, int d, float *__restrict__ out)
{
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (4 * c >= d) return;
    const long b = blockIdx.x, r0 = seg_start[b];
    const int len = (int)seg_len[b];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *p = x + r0 * d + 4 * c;
    int r = 0;
    for (; r + 8 <= len; r += 8) {            // eight rows' loads in flight, added in row order (one load per L2 round trip: 47 us at B = 256)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(p + (long)(r + u) * d);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; r < len; ++r) {
        const float4 v = ld4(p + (long)r * d);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float den = (float)(len > 1 ? len : 1);
    st4(out + b * d + 4 * c, make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den));
}
