// attention_backward.h -- backward of the ragged shared-key/value attention core (attention_kernels.h), so that training
// runs get_shared_info's cross-attention (vector_quantization_soft_one_new.py:17-88,133-142) on the same packed, unpadded
// representation as inference.  Included by medtok_vq.hip; gfx950 only.
//
// Per code, with S = scale Q KV^T, P = softmax_rows(S), M the dropout mask scaled by 1/(1-p), O = (P o M) KV:
//     dPm = dO KV^T          dP = dPm o M          delta_r = <dO_r, O_r>  (= sum_j P_rj dP_rj)
//     dS  = P o (dP - delta)
//     dQ  = scale dS KV                                              shared_kv_attention_dq_kernel   (block = 32 query rows)
//     dKV = (P o M)^T dO + scale dS^T Q                              shared_kv_attention_dkv_kernel  (block = 32 key rows)
// P is rebuilt from the log-sum-exp the forward stored (exp(S - lse)), the mask from the stateless hash (att_keep): nothing of
// size rows x keys is ever stored.  Same tiling and exact fp32 MFMA as the forward: W waves each own D / W columns; a 32-row
// chunk of the OTHER operand is parked in LDS (row stride D + 4), the W partial 32 x 32 products meet in LDS.
#pragma once

template <int W, int NT>
struct AttShape {
    static constexpr int D = 32 * W * NT, LD = D + 4, THREADS = 64 * W;
    static constexpr int EPT = 1024 / THREADS, TPR = 32 / EPT;
    static constexpr int FT = D / 4 < 32 ? D / 4 : 32, CI = D / 4 / FT, RP = THREADS / FT, RI = 32 / RP, NF = RI * CI;
    // LDS floats: parked chunk (fp32 [32][D + 4], or -- half-precision form -- TWO 16-bit images [32][D + 32]) + per-wave partial
    // products + two 32 x 33 operand tiles + two 32-entry row-statistic arrays
    static constexpr int LDH = D + 32;                    // halves per row of a 16-bit chunk image (row stride = 64 bytes mod 256)
    static constexpr int CHUNK = 32 * LDH;                // floats: two such images, or one fp32 chunk
    static constexpr size_t LDS_FLOATS = (size_t)CHUNK + (size_t)W * 32 * 33 + 2 * 32 * 33 + 64;
};

// partial 32 x 32 product over this wave's D / W columns: A rows from registers (reg[g] = elements 8g + 4 lh .. + 3 of the slice of
// row li), B rows = rows of the parked chunk.  Result r <-> [A row (r & 3) + 8 (r >> 2) + 4 lh][B row li].
struct AttNoHook { __device__ __forceinline__ void operator()(int) const {} };
// `hook(g)` runs behind the four MFMAs of group g (g = 0 .. 4 NT - 1): the callers weave the NF = 4 NT loads of their next chunk
// there, one per group, instead of issuing them as a burst (see shared_kv_attention_kernel).
template <int W, int NT, typename Hook = AttNoHook>
__device__ __forceinline__ f32x16 att_partial(const float4 (&reg)[4 * NT], const float *kvs, int slice, int li, int lh, Hook hook = Hook())
{
    constexpr int LD = AttShape<W, NT>::LD;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float *krow = kvs + li * LD + slice + 4 * lh;
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
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(reg[g].x, cur[j].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(reg[g].y, cur[j].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(reg[g].z, cur[j].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(reg[g].w, cur[j].w, s, 0, 0, 0);
            hook(g);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    return s;
}

// acc[t] += X^T-shaped product: A[i][k] = x[k][i] (a 32 x 33 LDS tile, contraction index first), B[k][col] = parked chunk row k
// `hook(s2)` runs behind the MFMAs of step s2 (0 .. 15).
template <int W, int NT, typename Hook = AttNoHook>
__device__ __forceinline__ void att_accumulate(f32x16 (&acc)[NT], const float (*x)[33], const float *kvs, int slice, int li, int lh, Hook hook = Hook())
{
    constexpr int LD = AttShape<W, NT>::LD;
    const float *kcol = kvs + lh * LD + slice + li;
    float pc = x[lh][li], pn = 0.f, kc[NT], kn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { kc[t] = kcol[32 * t]; kn[t] = 0.f; }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        if (s2 + 1 < 16) {
            pn = x[2 * (s2 + 1) + lh][li];
#pragma unroll
            for (int t = 0; t < NT; ++t) kn[t] = kcol[2 * (s2 + 1) * LD + 32 * t];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pc, kc[t], acc[t], 0, 0, 0);
        hook(s2);
        asm volatile("" ::: "memory");
        pc = pn;
#pragma unroll
        for (int t = 0; t < NT; ++t) kc[t] = kn[t];
    }
    // (a chunk is 4 NT loads per thread and this product has 16 steps: NT = 5 -- D = 640 -- has four loads left over.  Until round 6
    // they were never issued: the re-parked Q chunk of the fp32 dKV kernel held stale rows and the key gradient was wrong beyond
    // column 128 at that width; found when the half-precision backward, whose products differ, stopped agreeing with it.)
#pragma unroll
    for (int g = 16; g < 4 * NT; ++g) hook(g);
}

// ---- the same two primitives in ONE half-precision pass (fp16, or bf16 with BF) with fp32 accumulation: the precision class
// torch.autocast gives the reference's attention products (nn.MultiheadAttention under autocast computes Q K^T and P V in half
// precision, train_MedTok.py:212,394).  The chunk stays parked as fp32 (same LDS traffic: eight floats per lane and step either way);
// operands are rounded on the way into v_mfma_f32_32x32x16_{f16,bf16} -- 1/16 of the matrix time of the fp32 form.  A lane of a
// 32 x 32 x 16 operand holds 8 CONSECUTIVE k (columns 16 s + 8 lh ..) where the 32 x 32 x 2 form holds 4 + 4 interleaved: the
// register operands are loaded in that order (att_load_rows_h).
typedef unsigned short att_u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ half8 att_pack8(const float (&v)[8])
{
    att_u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if constexpr (BF) { const __bf16 b = (__bf16)v[e]; r[e] = __builtin_bit_cast(unsigned short, b); }
        else { const _Float16 h = (_Float16)v[e]; r[e] = __builtin_bit_cast(unsigned short, h); }
    }
    return __builtin_bit_cast(half8, r);
}
template <bool BF>
__device__ __forceinline__ f32x16 att_mfma16(half8 a, half8 b, f32x16 c)
{
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8, a), __builtin_bit_cast(att_bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// row `src` (pointing at the wave's column slice of the row): the lane's 2 NT operands of 8 consecutive columns
template <int NT, bool BF>
__device__ __forceinline__ void att_load_rows_h(half8 (&reg)[2 * NT], const float *src, int lh)
{
#pragma unroll
    for (int st = 0; st < 2 * NT; ++st) {
        const float4 a = ld4(src + 16 * st + 8 * lh), b = ld4(src + 16 * st + 8 * lh + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        reg[st] = att_pack8<BF>(v);
    }
}
template <int W, int NT, bool BF, typename Hook = AttNoHook>
__device__ __forceinline__ f32x16 att_partial_h(const half8 (&reg)[2 * NT], const float *kvs, int slice, int li, int lh, Hook hook = Hook())
{
    constexpr int LD = AttShape<W, NT>::LD;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float *krow = kvs + li * LD + slice + 8 * lh;
#pragma unroll
    for (int st = 0; st < 2 * NT; ++st) {
        const float4 a = *reinterpret_cast<const float4 *>(krow + 16 * st), b = *reinterpret_cast<const float4 *>(krow + 16 * st + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        s = att_mfma16<BF>(reg[st], att_pack8<BF>(v), s);
        hook(2 * st);
        hook(2 * st + 1);
        asm volatile("" ::: "memory");
    }
    return s;
}
template <int W, int NT, bool BF, typename Hook = AttNoHook>
__device__ __forceinline__ void att_accumulate_h(f32x16 (&acc)[NT], const float (*x)[33], const float *kvs, int slice, int li, int lh, Hook hook = Hook())
{
    constexpr int LD = AttShape<W, NT>::LD;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int k0 = 16 * st + 8 * lh;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = x[k0 + e][li];
        const half8 a = att_pack8<BF>(av);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = kvs[(k0 + e) * LD + slice + 32 * t + li];
            acc[t] = att_mfma16<BF>(a, att_pack8<BF>(bv), acc[t]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) hook(8 * st + e);
        asm volatile("" ::: "memory");
    }
}

// ---- the half-precision form with the chunk PARKED AS ITS 16-BIT IMAGE (round 6).  The fp32 chunk was read back and rounded by every
// product that used it -- per 32 x 32 x 16 MFMA two 128-bit reads + 12 conversion / pack instructions (first product) or eight 32-bit
// reads + 12 (second): the kernels ran at the LDS's and the VALU's rate, a fifth of the matrix pipe's.  Rounded ONCE on the way in (the
// same roundings of the same values: the same bits), a first-product operand is ONE 128-bit read and a second-product operand -- eight
// consecutive rows of one column -- TWO transposed reads (ds_read_b64_tr_b16: the 16 lanes of a group fetch a [4 rows][16 columns]
// block, 8 bytes each, and every lane receives one column of it), neither with a conversion.
// Image: [32 rows][D + 32] halves.  The row stride is 64 bytes mod 256, so the four rows of a transposed read's blocks (two groups =
// 64 bytes per row) fall into disjoint banks; the 16-byte slots of a row are XOR-swizzled by (row / 4) % 4, so the 128-bit reads of
// consecutive rows at one column fall into different slots too (the rows of a transposed read share row / 4: the swizzle only permutes
// the four slots of their aligned 64-byte group).
template <bool BF>
__device__ __forceinline__ unsigned short att_cvt16(float v)
{
    if constexpr (BF) { const __bf16 b = (__bf16)v; return __builtin_bit_cast(unsigned short, b); }
    else { const _Float16 h = (_Float16)v; return __builtin_bit_cast(unsigned short, h); }
}
typedef unsigned short att_u16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 att_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
// byte offset of the 4 halves at (row, col), col a multiple of 4
template <int LDH>
__device__ __forceinline__ int att_img_off(int row, int col)
{
    return row * (LDH * 2) + ((((col >> 3) ^ ((row >> 2) & 3)) << 4) | ((col & 4) << 1));
}
template <int W, int NT, bool BF, typename Hook = AttNoHook>
__device__ __forceinline__ f32x16 att_partial_hh(const half8 (&reg)[2 * NT], const char *img, int slice, int li, int lh, Hook hook = Hook())
{
    constexpr int LDH = AttShape<W, NT>::LDH;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const char *krow = img + li * (LDH * 2);
    const int sw = (li >> 2) & 3, c0 = (slice >> 3) + lh;           // 16-byte slot of step 0; step st: + 2 st
#pragma unroll
    for (int st = 0; st < 2 * NT; ++st) {
        const half8 b = *reinterpret_cast<const half8 *>(krow + (((c0 + 2 * st) ^ sw) << 4));
        s = att_mfma16<BF>(reg[st], b, s);
        hook(2 * st);
        hook(2 * st + 1);
        asm volatile("" ::: "memory");
    }
    return s;
}
template <int W, int NT, bool BF, typename Hook = AttNoHook>
__device__ __forceinline__ void att_accumulate_hh(f32x16 (&acc)[NT], const float (*x)[33], const char *img, int slice, int lane, Hook hook = Hook())
{
    constexpr int LDH = AttShape<W, NT>::LDH;
    typedef __attribute__((address_space(3))) att_fp16x4 *lptr;
    const int li = lane & 31, lh = lane >> 5, grp = lane >> 4, i = lane & 15;
    // this lane's share of the transposed reads: row 8 lh + 4 rd + i / 4 of the k step, columns 16 (grp & 1) + 4 (i & 3) .. + 3 of the tile
    int tr_off[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) tr_off[rd] = att_img_off<LDH>(8 * lh + 4 * rd + (i >> 2), slice + 16 * (grp & 1) + 4 * (i & 3));
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int k0 = 16 * st + 8 * lh;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = x[k0 + e][li];
        const half8 a = att_pack8<BF>(av);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // (rows + 16 st: the same (row / 4) % 4; column tile t: four slots further, above the swizzled bits)
            const half4v b0 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(img + tr_off[0] + st * 16 * (LDH * 2) + t * 64)));
            const half4v b1 = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lptr)(img + tr_off[1] + st * 16 * (LDH * 2) + t * 64)));
            acc[t] = att_mfma16<BF>(a, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7), acc[t]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) hook(8 * st + e);
        asm volatile("" ::: "memory");
    }
}

#define ATT_LDS_BARRIER()                                                   \
    do {                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
        __builtin_amdgcn_s_barrier();                                       \
        asm volatile("" ::: "memory");                                      \
    } while (0)

// ---------------------------------------------------------------- dQ
// HM: 0 = exact fp32 MFMA; 1 / 2 = one fp16 / bf16 pass (autocast callers)
template <int W, int NT, int HM = 0>
__global__ __launch_bounds__(64 * W) void shared_kv_attention_dq_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const float *__restrict__ kv, const int64_t *__restrict__ kv_start, const int64_t *__restrict__ kv_len,
    const float *__restrict__ d_out, const float *__restrict__ lse, const float *__restrict__ delta, float scale,
    float *__restrict__ dq, int q_tiles, unsigned drop_thresh, unsigned seed, float keep_scale)
{
    using G = AttShape<W, NT>;
    constexpr int D = G::D, LD = G::LD, EPT = G::EPT, TPR = G::TPR, FT = G::FT, CI = G::CI, RP = G::RP, RI = G::RI, NF = G::NF;
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    float *kvs = att_sm;
    float (*part)[32][33] = reinterpret_cast<float (*)[32][33]>(kvs + G::CHUNK);
    float (*pt)[33] = reinterpret_cast<float (*)[33]>(kvs + G::CHUNK + W * 32 * 33);
    float *lse_s = kvs + G::CHUNK + W * 32 * 33 + 2 * 32 * 33, *del_s = lse_s + 32;
    const int b = (int)(blockIdx.x / (unsigned)q_tiles), qt = (int)(blockIdx.x % (unsigned)q_tiles);
    const int ql = (int)q_len[b];
    if (qt * 32 >= ql) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int slice = wave * 32 * NT;
    const int f_r0 = tid / FT, f_c = (tid % FT) * 4;
    float4 kf[NF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
            const float *src = kv + (ks + min(k0 + f_r0 + RP * ri, kl - 1)) * (long)D + f_c;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) kf[ri * CI + ci] = ld4(src + 4 * FT * ci);
        }
    };
    char *kvh = reinterpret_cast<char *>(kvs);                           // HM: the chunk as its 16-bit image (att_img_off)
    auto park = [&]() {
        if constexpr (HM) {
#pragma unroll
            for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float4 v = kf[ri * CI + ci];
                    const att_u16x4 h = {att_cvt16<HM == 2>(v.x), att_cvt16<HM == 2>(v.y), att_cvt16<HM == 2>(v.z), att_cvt16<HM == 2>(v.w)};
                    *reinterpret_cast<att_u16x4 *>(kvh + att_img_off<G::LDH>(f_r0 + RP * ri, f_c + 4 * FT * ci)) = h;
                }
            return;
        }
        float *dst = kvs + f_r0 * LD + f_c;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) *reinterpret_cast<float4 *>(dst + RP * ri * LD + 4 * FT * ci) = kf[ri * CI + ci];
    };
    if (kl > 0) fetch(0);
    float4 qf[HM ? 1 : 4 * NT], dof[HM ? 1 : 4 * NT];
    half8 qh[HM ? 2 * NT : 1], doh[HM ? 2 * NT : 1];
    {
        const long r = qs + min(qt * 32 + li, ql - 1);
        if constexpr (HM) {
            att_load_rows_h<NT, HM == 2>(qh, q + r * (long)D + slice, lh);
            att_load_rows_h<NT, HM == 2>(doh, d_out + r * (long)D + slice, lh);
        } else {
            const float *qrow = q + r * (long)D + slice + 4 * lh, *drow = d_out + r * (long)D + slice + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4 * NT; ++g) { qf[g] = ld4(qrow + 8 * g); dof[g] = ld4(drow + 8 * g); }
        }
    }
    if (tid < 32) {
        const long r = qs + min(qt * 32 + tid, ql - 1);
        lse_s[tid] = lse[r];
        del_s[tid] = delta[r];
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();
    const int row = tid / TPR, kq = (tid % TPR) * EPT;
    for (int k0 = 0; k0 < kl; k0 += 32) {
        park();
        // the next chunk's rows: one 16-byte load behind every four MFMAs of the first product (rows past the last key clamped)
        const float *fsrc[RI];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) fsrc[ri] = kv + (ks + min(k0 + 32 + f_r0 + RP * ri, kl - 1)) * (long)D + f_c;
        ATT_LDS_BARRIER();
        auto weave = [&](int g) { kf[g] = ld4(fsrc[g / CI] + 4 * FT * (g % CI)); };
        f32x16 s;                                                                  // S partial: [query row][key]
        if constexpr (HM) s = att_partial_hh<W, NT, HM == 2>(qh, kvh, slice, li, lh, weave);
        else s = att_partial<W, NT>(qf, kvs, slice, li, lh, weave);
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
        ATT_LDS_BARRIER();
        float sc[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) a += part[w2][row][kq + j];
            sc[j] = a * scale;
        }
        ATT_LDS_BARRIER();                                                         // everyone has read the S partials
        if constexpr (HM) s = att_partial_hh<W, NT, HM == 2>(doh, kvh, slice, li, lh);     // dPm partial = dO . KV^T
        else s = att_partial<W, NT>(dof, kvs, slice, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
        ATT_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            float dp = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) dp += part[w2][row][kq + j];
            const int key = k0 + kq + j;
            const float p = key < kl ? expf(sc[j] - lse_s[row]) : 0.f;
            if (drop_thresh) dp = att_keep(seed, qs + qt * 32 + row, key, drop_thresh) ? dp * keep_scale : 0.f;
            pt[kq + j][row] = p * (dp - del_s[row]) * scale;                       // scale dS, [key][row]
        }
        ATT_LDS_BARRIER();
        if constexpr (HM) att_accumulate_hh<W, NT, HM == 2>(acc, pt, kvh, slice, lane);    // dQ += (scale dS) . KV
        else att_accumulate<W, NT>(acc, pt, kvs, slice, li, lh);
        ATT_LDS_BARRIER();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (qt * 32 + orow < ql) {
            float *o = dq + (qs + qt * 32 + orow) * (long)D + slice + li;
#pragma unroll
            for (int t = 0; t < NT; ++t) o[32 * t] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------- dKV
// One launch takes the key gradient of up to DKV_SRC_MAX "sources" -- attention calls over the SAME keys (kv, kv_start, kv_len) with their
// own queries, upstream gradients, row statistics and dropout masks: the layers of CrossAttention all attend to the original other
// modality, so their key gradients are one sum; a block walks the sources' query chunks one after the other into one accumulator and
// stores its 32 key rows once (one launch per layer: a second pass over the [kv_rows, D] gradient to add into it).
constexpr int DKV_SRC_MAX = 4;
struct DkvSource {
    const float *q, *d_out, *lse, *delta;
    const int64_t *q_start, *q_len;
    float scale, keep_scale;
    unsigned thresh, seed;
};
struct DkvSources { DkvSource s[DKV_SRC_MAX]; int count; };

template <int W, int NT, int HM = 0>
__global__ __launch_bounds__(64 * W) void shared_kv_attention_dkv_kernel(
    const DkvSources S, const float *__restrict__ kv, const int64_t *__restrict__ kv_start, const int64_t *__restrict__ kv_len,
    float *__restrict__ dkv, int kv_tiles, int accumulate)
{
    using G = AttShape<W, NT>;
    constexpr int D = G::D, LD = G::LD, EPT = G::EPT, TPR = G::TPR, FT = G::FT, CI = G::CI, RP = G::RP, RI = G::RI, NF = G::NF;
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    float *kvs = att_sm;                                                                // parked chunk: 32 query rows of Q or of dO
    float (*part)[32][33] = reinterpret_cast<float (*)[32][33]>(kvs + G::CHUNK);
    float (*pm)[33] = reinterpret_cast<float (*)[33]>(kvs + G::CHUNK + W * 32 * 33);     // (P o M)   [query][key]
    float (*ds)[33] = pm + 32;                                                          // scale dS  [query][key]
    float *lse_s = kvs + G::CHUNK + W * 32 * 33 + 2 * 32 * 33, *del_s = lse_s + 32;
    const int b = (int)(blockIdx.x / (unsigned)kv_tiles), kt = (int)(blockIdx.x % (unsigned)kv_tiles);
    const int kl = (int)kv_len[b];
    if (kt * 32 >= kl) return;
    const long ks = kv_start[b];
    // the source being walked (set at the top of the source loop below)
    long qs = 0;
    int ql = 0;
    const float *q = nullptr, *d_out = nullptr, *lse = nullptr, *delta = nullptr;
    float scale = 0.f, keep_scale = 1.f;
    unsigned drop_thresh = 0, seed = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int slice = wave * 32 * NT;
    const int f_r0 = tid / FT, f_c = (tid % FT) * 4;
    float4 kf[NF];
    auto fetch = [&](const float *src_base, int c0) {
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
            const float *src = src_base + (qs + min(c0 + f_r0 + RP * ri, ql - 1)) * (long)D + f_c;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) kf[ri * CI + ci] = ld4(src + 4 * FT * ci);
        }
    };
    char *kvh16 = reinterpret_cast<char *>(kvs);                         // HM: the parked chunks as 16-bit images (att_img_off)
    auto park = [&]() {
        if constexpr (HM) return;                                        // (the half-precision loop below parks with park_to)
        float *dst = kvs + f_r0 * LD + f_c;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) *reinterpret_cast<float4 *>(dst + RP * ri * LD + 4 * FT * ci) = kf[ri * CI + ci];
    };
    float4 kvf[HM ? 1 : 4 * NT];                          // this block's 32 key rows, the wave's column slice
    half8 kvh[HM ? 2 * NT : 1];
    {
        if constexpr (HM) {
            att_load_rows_h<NT, HM == 2>(kvh, kv + (ks + min(kt * 32 + li, kl - 1)) * (long)D + slice, lh);
        } else {
            const float *krow = kv + (ks + min(kt * 32 + li, kl - 1)) * (long)D + slice + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4 * NT; ++g) kvf[g] = ld4(krow + 8 * g);
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int krow_t = tid / TPR, qc = (tid % TPR) * EPT;        // elementwise steps: this thread's key row and query columns
    // The three chunk fetches of an iteration (Q, dO, Q again) ride between MFMAs, one 16-byte load per group / step, instead of
    // going out as bursts: dO behind the groups of the first product, Q-again behind the steps of the first accumulation, and the
    // NEXT iteration's Q behind the steps of the last accumulation (the first one is fetched here).  Rows past the code's last
    // query are clamped.
    auto woven = [&](const float *src_base, int c0) __attribute__((always_inline)) {
        return [=, &kf](int g) __attribute__((always_inline)) {
            if (g < NF) {
                const int ri = g / CI, ci = g % CI;
                kf[g] = ld4(src_base + (qs + min(c0 + f_r0 + RP * ri, ql - 1)) * (long)D + f_c + 4 * FT * ci);
            }
        };
    };
    for (int si = 0; si < S.count; ++si) {
    {
        const DkvSource &src = S.s[si];
        qs = src.q_start[b]; ql = (int)src.q_len[b];
        q = src.q; d_out = src.d_out; lse = src.lse; delta = src.delta;
        scale = src.scale; keep_scale = src.keep_scale; drop_thresh = src.thresh; seed = src.seed;
    }
    if constexpr (HM) {
        // Half-precision form: the 16-bit images of the Q chunk AND of the dO chunk are resident together (two [32][D + 8] images in the
        // space of one fp32 chunk), so Q is parked once per iteration, not twice: 6 block-wide synchronisations per chunk instead of 9,
        // two fetches instead of three -- the next iteration's Q rides behind the first product's MFMAs, its dO behind the second's.
        char *bq = kvh16, *bo = kvh16 + 32 * G::LDH * 2;
        float4 ko[NF];
        auto park_to = [&](char *buf, const float4 (&regs)[NF]) __attribute__((always_inline)) {
#pragma unroll
            for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float4 v = regs[ri * CI + ci];
                    const att_u16x4 h = {att_cvt16<HM == 2>(v.x), att_cvt16<HM == 2>(v.y), att_cvt16<HM == 2>(v.z), att_cvt16<HM == 2>(v.w)};
                    *reinterpret_cast<att_u16x4 *>(buf + att_img_off<G::LDH>(f_r0 + RP * ri, f_c + 4 * FT * ci)) = h;
                }
        };
        auto woven_into = [&](float4 (&regs)[NF], const float *src_base, int c0) __attribute__((always_inline)) {
            return [=, &regs](int g) __attribute__((always_inline)) {
                if (g < NF) {
                    const int ri = g / CI, ci = g % CI;
                    regs[g] = ld4(src_base + (qs + min(c0 + f_r0 + RP * ri, ql - 1)) * (long)D + f_c + 4 * FT * ci);
                }
            };
        };
        if (ql > 0) {
            fetch(q, 0);
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) {
                const float *src = d_out + (qs + min(f_r0 + RP * ri, ql - 1)) * (long)D + f_c;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) ko[ri * CI + ci] = ld4(src + 4 * FT * ci);
            }
        }
        for (int c0 = 0; c0 < ql; c0 += 32) {
            __syncthreads();                                     // the previous chunk's readers of the images / pm / ds / the statistics are done
            if (tid < 32) {
                const long r = qs + min(c0 + tid, ql - 1);
                lse_s[tid] = lse[r];
                del_s[tid] = delta[r];
            }
            park_to(bq, kf);
            park_to(bo, ko);
            ATT_LDS_BARRIER();
            f32x16 s = att_partial_hh<W, NT, HM == 2>(kvh, bq, slice, li, lh, woven_into(kf, q, c0 + 32));      // S^T partial: [key][query]
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
            ATT_LDS_BARRIER();
            float p[EPT];
            bool keep[EPT];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float a = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) a += part[w2][krow_t][qc + j];
                const int key = kt * 32 + krow_t, qr = c0 + qc + j;
                p[j] = (key < kl && qr < ql) ? expf(a * scale - lse_s[qc + j]) : 0.f;
                keep[j] = !drop_thresh || att_keep(seed, qs + qr, key, drop_thresh);
                pm[qc + j][krow_t] = keep[j] ? p[j] * keep_scale : 0.f;
            }
            ATT_LDS_BARRIER();                                                     // the S partials are consumed, pm is complete
            s = att_partial_hh<W, NT, HM == 2>(kvh, bo, slice, li, lh, woven_into(ko, d_out, c0 + 32));         // dPm^T partial = KV . dO^T
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
            ATT_LDS_BARRIER();
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float dp = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) dp += part[w2][krow_t][qc + j];
                dp = keep[j] ? dp * keep_scale : 0.f;
                ds[qc + j][krow_t] = p[j] * (dp - del_s[qc + j]) * scale;
            }
            att_accumulate_hh<W, NT, HM == 2>(acc, pm, bo, slice, lane);           // dKV += (P o M)^T . dO
            ATT_LDS_BARRIER();                                                     // ds complete
            att_accumulate_hh<W, NT, HM == 2>(acc, ds, bq, slice, lane);           // dKV += (scale dS)^T . Q
        }
    } else {
    if (ql > 0) fetch(q, 0);
    for (int c0 = 0; c0 < ql; c0 += 32) {
        __syncthreads();                                         // the previous chunk's readers of kvs / pm / ds / the statistics are done
        if (tid < 32) {
            const long r = qs + min(c0 + tid, ql - 1);
            lse_s[tid] = lse[r];
            del_s[tid] = delta[r];
        }
        park();                                                                    // Q chunk (fetched during the previous iteration)
        ATT_LDS_BARRIER();
        f32x16 s;                                                                  // S^T partial: [key][query]; dO chunk on its way
        s = att_partial<W, NT>(kvf, kvs, slice, li, lh, woven(d_out, c0));
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
        ATT_LDS_BARRIER();
        float p[EPT];
        bool keep[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) a += part[w2][krow_t][qc + j];
            const int key = kt * 32 + krow_t, qr = c0 + qc + j;
            p[j] = (key < kl && qr < ql) ? expf(a * scale - lse_s[qc + j]) : 0.f;
            keep[j] = !drop_thresh || att_keep(seed, qs + qr, key, drop_thresh);
            pm[qc + j][krow_t] = keep[j] ? p[j] * keep_scale : 0.f;
        }
        ATT_LDS_BARRIER();                                                         // S partials and the Q chunk are consumed
        park();                                                                    // dO chunk
        ATT_LDS_BARRIER();
        s = att_partial<W, NT>(kvf, kvs, slice, li, lh);                         // dPm^T partial = KV . dO^T
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = s[r];
        ATT_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            float dp = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) dp += part[w2][krow_t][qc + j];
            dp = keep[j] ? dp * keep_scale : 0.f;
            ds[qc + j][krow_t] = p[j] * (dp - del_s[qc + j]) * scale;
        }
        // dKV += (P o M)^T . dO   (pm was complete two barriers ago); Q again on its way
        att_accumulate<W, NT>(acc, pm, kvs, slice, li, lh, woven(q, c0));
        ATT_LDS_BARRIER();                                                         // dO chunk consumed, ds complete
        park();                                                                    // Q chunk again
        ATT_LDS_BARRIER();
        // dKV += (scale dS)^T . Q; the next iteration's Q on its way
        att_accumulate<W, NT>(acc, ds, kvs, slice, li, lh, woven(q, c0 + 32));
    }
    }
    }                                                             // (the next source)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kt * 32 + orow < kl) {
            float *o = dkv + (ks + kt * 32 + orow) * (long)D + slice + li;
            // accumulate: dkv already holds a gradient of these keys (another layer's, over the same rows: a block owns its rows, so
            // the sum is ordered: held + this launch's)
            if (accumulate) {
#pragma unroll
                for (int t = 0; t < NT; ++t) o[32 * t] += acc[t][r];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) o[32 * t] = acc[t][r];
            }
        }
    }
}
