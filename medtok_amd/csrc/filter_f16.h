// filter_f16.h -- fp16-MFMA shortlist + exact fp32 re-score: the fast search path (MEDTOK_PATH_F16_FILTER).
// Included by medtok_vq.hip; gfx950 only.
//
// Idea.  The exact search is bound by the fp32 matrix pipe (157 TFLOP/s).  v_mfma_f32_32x32x16_f16
// runs 16x faster, but its scores s~ only approximate the contract's fp32 fmaf-chain score s.  With a
// PROVEN bound |d~ - d| <= eps per (row, code) the approximate pass is used only to discard codes that
// cannot be in the exact top-k; everything that might be is re-scored with the exact chain.
//
//   exact:   d  = (xsq + wsq[c]) - 2 s      s  = canonical fp32 chain (oracle/medtok_oracle.c)
//   filter:  d~ = (xsq + wsq[c]) - 2 s~     s~ = 2^-16 * MFMA_f16(2^8 xhat, 2^8 what)
//
// Error bound.  fp16 has an 11-bit significand: after the exact power-of-two prescale every element
// is rounded with relative error <= 2^-11 (elements below 2^-22 may flush: absolute error <= 2^-22
// each), so |sum x~e~ - sum xe| <= (2^-10 + 2^-22) sum|x_i e_i| + flush <= (2^-10 + 2^-22)|x||e| + flush.
// Products of two fp16 are exact in fp32; for the MFMA's fp32 accumulation and for the exact chain's
// own rounding we budget D * 2^-21 relative (a 1-ulp-per-add model is D * 2^-24; measured on MI355X in
// tests/test_gpu_filter.py::test_filter_score_error_bound: <= 0.04 of the budget), hence
//   |s~ - s| <= gamma * sqrt(xsq * wsq_max),   gamma = 2^-10 + 2^-20 + D * 2^-21
//   |d~ - d| <= eps := 2 gamma sqrt(xsq wsq_max) + slack       (slack: flush + 4 ulp of d)
// Shortlist rule.  Let t~ be the k-th smallest d~ of a row.  k codes have d <= t~ + eps, so the exact
// k-th smallest d is <= t~ + eps, so every exact top-k member has d~ <= t~ + 2 eps.  Keeping
// {c : d~(c) <= T + 2 eps} for ANY T >= t~ therefore keeps them all; the kernel uses each lane's running
// k-th best (which only ever over-estimates t~) so one pass suffices.  The re-score kernel then tightens
// to the row's true t~, evaluates the exact chain for the survivors (typically 6-8 of K) and selects the
// top-k by (d, index) -- the same bits the fp32 path produces.  Rows whose candidate buffers overflow,
// or whose norms leave the range the bound assumes, are redone by the exact fp32 kernel (INDIRECT mode).
#pragma once

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

constexpr int F_BM = 128, F_BN = 128, F_BK = 64;              // codes x rows x k (fp16 elements) per stage
constexpr int F_ROWB = F_BK * 2;                              // bytes per staged tile row (128)
constexpr int F_TILEB = F_BM * F_ROWB;                        // 16 KB per operand tile
constexpr int F_STAGEB = 2 * F_TILEB;                         // A + B
constexpr size_t F_LDS_BYTES = 2 * F_STAGEB;                  // double buffered: 64 KB -> 2 blocks / CU
constexpr int F_CAP = 64;                                     // candidate slots per (row, owner)
constexpr int F_OWN_PER_SPLIT = 4;                            // 2 half-waves x 2 code-side waves
constexpr float F_PRESCALE = 256.0f;                          // 2^8 on both operands
constexpr float F_UNSCALE = 1.0f / 65536.0f;
constexpr float F_NORM_LIMIT = 4.0f;                          // |x|^2, |e|^2 above this -> exact path

__host__ __device__ inline float filter_gamma(int d) { return 0x1p-10f + 0x1p-20f + (float)d * 0x1p-21f; }

// eps = bound on |d~ - d| for a row with squared norm xn against codes with squared norm <= en_max
__device__ __forceinline__ float filter_eps(float xn, float en_max, int d)
{
    const float mag = sqrtf(xn * en_max);
    const float flush = 0x1p-21f * sqrtf((float)d) * sqrtf(fmaxf(xn, en_max));
    const float ulps = 0x1p-21f * (xn + en_max + 2.0f * mag);
    return 2.0f * filter_gamma(d) * mag * 1.0001f + 2.0f * flush + ulps;
}

// ---------------------------------------------------------------- operand preparation
// fp32 rows -> prescaled fp16 rows in a zero-padded [n_pad, dp] image (dp % 64 == 0, n_pad % 128 == 0).
__global__ __launch_bounds__(256) void to_half_kernel(const float *__restrict__ src, long n, int d, long n_pad, int dp,
                                                      _Float16 *__restrict__ dst)
{
    const int cpr = dp / 8;                                   // 16-byte chunks per row
    const long total = n_pad * cpr;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / cpr;
        const int c = (int)(t - r * cpr) * 8;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (r < n) {
            if (c < d) a = ld4(src + r * d + c);
            if (c + 4 < d) b = ld4(src + r * d + c + 4);
        }
        half8 h;
        h[0] = (_Float16)(a.x * F_PRESCALE); h[1] = (_Float16)(a.y * F_PRESCALE);
        h[2] = (_Float16)(a.z * F_PRESCALE); h[3] = (_Float16)(a.w * F_PRESCALE);
        h[4] = (_Float16)(b.x * F_PRESCALE); h[5] = (_Float16)(b.y * F_PRESCALE);
        h[6] = (_Float16)(b.z * F_PRESCALE); h[7] = (_Float16)(b.w * F_PRESCALE);
        *reinterpret_cast<half8 *>(dst + r * dp + c) = h;
    }
}

// max of wsq[0..k) -> out[0] (single block; NaN propagates as +inf so the filter bails out)
__global__ __launch_bounds__(1024) void wsq_max_kernel(const float *__restrict__ wsq, int k, float *__restrict__ out)
{
    __shared__ float sh[1024];
    float m = 0.f;
    for (int i = threadIdx.x; i < k; i += 1024) {
        const float v = wsq[i];
        m = (v > m || !(v == v)) ? (v == v ? v : INFINITY) : m;
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ---------------------------------------------------------------- value-only top-k (thresholds)
template <int T>
__device__ __forceinline__ void thr_insert(float (&tv)[T], float v)
{
    if (v < tv[T - 1]) {
#pragma unroll
        for (int j = T - 1; j >= 1; --j) {
            const bool lt_prev = v < tv[j - 1];
            const bool lt = v < tv[j];
            tv[j] = lt_prev ? tv[j - 1] : (lt ? v : tv[j]);
        }
        tv[0] = v < tv[0] ? v : tv[0];
    }
}

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    // async global -> LDS, 16 B per lane; the LDS address is wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------- the filter kernel
// Block = 4 waves (2 code-side x 2 row-side), tile 128 codes x 128 rows, each wave 64 x 64 = 2 x 2 MFMA
// tiles of 32x32x16.  Operands arrive by LDS-DMA (global_load_lds, 16 B/lane) into a double buffer; a
// tile row is 128 B = 8 chunks, stored at chunk position c ^ ((row >> 1) & 7) so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-byte bank slots (the permutation is applied
// to the per-lane SOURCE address; the LDS image itself is lane-linear as the DMA requires).
// As in the fp32 kernel, codes are the A rows: a lane ends up with 16 codes of one input row per tile,
// so thresholds and candidate appends are lane-local.
template <int TOPK, bool DUMP>
__global__ __launch_bounds__(256, 2) void filter_f16_kernel(
    const _Float16 *__restrict__ xh, const _Float16 *__restrict__ wh, const float *__restrict__ xsq,
    const float *__restrict__ wsq, const float *__restrict__ en_max_ptr, long n, int k_codes, int dp, int d,
    int codes_per_split, int own_total, uint2 *__restrict__ cand, int *__restrict__ cand_cnt,
    float *__restrict__ dump)
{
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const long row0 = (long)blockIdx.x * F_BN;
    const int split = blockIdx.y;
    const int code_lo = split * codes_per_split;
    const int code_hi = min(k_codes, code_lo + codes_per_split);
    const int nct = (code_hi - code_lo + F_BM - 1) / F_BM;
    const int nkb = dp / F_BK;
    const int nstage = nct * nkb;

    // ---- staging: wave w DMA-copies tile rows [32w, 32w+32) of A and of B, 8 rows per instruction
    const int s_r = lane >> 3, s_c = lane & 7;
    int pct = 0, pkb = 0;
    auto stage = [&](int buf) {
        char *base = fsm + buf * F_STAGEB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 32 + q * 8 + s_r;                       // tile row this lane feeds
            const int c = s_c ^ ((r >> 1) & 7);                          // source chunk for LDS chunk s_c
            const long koff = (long)pkb * F_BK + c * 8;
            const _Float16 *ga = wh + (long)(code_lo + pct * F_BM + r) * dp + koff;
            const _Float16 *gb = xh + (row0 + r) * dp + koff;
            glds16(ga, base + (wave * 32 + q * 8) * F_ROWB);
            glds16(gb, base + F_TILEB + (wave * 32 + q * 8) * F_ROWB);
        }
        if (++pkb == nkb) { pkb = 0; ++pct; }
    };

    // ---- per-lane state: for each of the wave's two 32-row column tiles, the k smallest d~ so far
    float tv[2][TOPK];
    int cnt[2] = {0, 0};
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int j = 0; j < TOPK; ++j) tv[nn][j] = INFINITY;
    long xrow[2];
    float xn[2], win[2];
    const float en_max = en_max_ptr[0];
    bool sane = en_max <= F_NORM_LIMIT;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        xrow[nn] = row0 + wn * 64 + nn * 32 + li;
        xn[nn] = xsq[min(xrow[nn], n - 1)];
        win[nn] = 2.0f * filter_eps(xn[nn], en_max, d);
    }
    const int owner = split * F_OWN_PER_SPLIT + wm * 2 + lh;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // fragment addresses: row i of the tile, chunk (2t + lh) ^ ((i >> 1) & 7)
    int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ia = wm * 64 + m * 32 + li, ib = wn * 64 + m * 32 + li;
        a_off[m] = ia * F_ROWB; a_sw[m] = (ia >> 1) & 7;
        b_off[m] = F_TILEB + ib * F_ROWB; b_sw[m] = (ib >> 1) & 7;
    }

    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ct = 0, kb = 0;
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) stage(buf ^ 1);
        const char *base = fsm + buf * F_STAGEB;
#pragma unroll
        for (int t = 0; t < F_BK / 16; ++t) {
            half8 af[2], bf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                af[m] = *reinterpret_cast<const half8 *>(base + a_off[m] + (((2 * t + lh) ^ a_sw[m]) << 4));
                bf[m] = *reinterpret_cast<const half8 *>(base + b_off[m] + (((2 * t + lh) ^ b_sw[m]) << 4));
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bf[nn], acc[m][nn], 0, 0, 0);
        }
        if (++kb == nkb) {
            const int cbase = code_lo + ct * F_BM + wm * 64 + 4 * lh;
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                float dv[2][16];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int code = cbase + 32 * m + (r & 3) + 8 * (r >> 2);
                        const float en = wsq[min(code, k_codes - 1)];
                        const float sum = xn[nn] + en;
                        const float two = 2.0f * (acc[m][nn][r] * F_UNSCALE);
                        float v = sum - two;
                        if (code >= code_hi) v = INFINITY;
                        if (DUMP) {
                            if (code < code_hi && xrow[nn] < n) dump[xrow[nn] * k_codes + code] = acc[m][nn][r] * F_UNSCALE;
                        }
                        dv[m][r] = v;
                        thr_insert<TOPK>(tv[nn], v);
                        acc[m][nn][r] = 0.f;
                    }
                if (!DUMP) {
                    const float lim = tv[nn][TOPK - 1] + win[nn];
                    uint2 *slot = cand + ((long)xrow[nn] * own_total + owner) * F_CAP;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int code = cbase + 32 * m + (r & 3) + 8 * (r >> 2);
                            if (dv[m][r] <= lim && code < code_hi) {
                                if (cnt[nn] < F_CAP && xrow[nn] < n) slot[cnt[nn]] = make_uint2(__float_as_uint(dv[m][r]), (unsigned)code);
                                ++cnt[nn];
                            }
                        }
                }
            }
            kb = 0;
            ++ct;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!DUMP) {
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            // rows outside the range the bound assumes are forced onto the exact path
            const bool ok = sane && xn[nn] <= F_NORM_LIMIT;
            if (xrow[nn] < n) cand_cnt[xrow[nn] * own_total + owner] = ok ? cnt[nn] : F_CAP + 1;
        }
    }
}

// ---------------------------------------------------------------- exact re-score
// 16 lanes per row.  Phase 1: the row's true t~ (k-th smallest d~ over all owners' candidates).
// Phase 2: every candidate with d~ <= t~ + 2 eps is re-scored with the canonical fp32 chain and merged
// into an exact (d, index) top-k.  Overflowed rows are appended to the fallback list.
template <int TOPK>
__global__ __launch_bounds__(256) void rescore_kernel(
    const uint2 *__restrict__ cand, const int *__restrict__ cand_cnt, int own_total,
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, const float *__restrict__ en_max_ptr, long n, int k_codes, int d, int topk_out,
    int64_t *__restrict__ out_idx, float *__restrict__ out_dist, int *__restrict__ fb_count, int *__restrict__ fb_rows)
{
    const int l16 = threadIdx.x & 15;
    const long pos = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long row = min(pos, n - 1);
    const float xn = xsq[row];
    const float win = 2.0f * filter_eps(xn, en_max_ptr[0], d);
    const uint2 *rc = cand + row * own_total * F_CAP;
    const int *cc = cand_cnt + row * own_total;

    // ---- phase 1
    float tv[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) tv[j] = INFINITY;
    bool overflow = false;
    for (int o = 0; o < own_total; ++o) overflow |= cc[o] > F_CAP;
    if (overflow) {
        // shortlist incomplete (buffer full, or the filter refused the row): the exact kernel redoes it.
        // The whole 16-lane group takes this branch together, so the width-16 shuffles below stay safe.
        if (l16 == 0 && pos < n) fb_rows[atomicAdd(fb_count, 1)] = (int)row;
        return;
    }
    for (int o = 0; o < own_total; ++o) {
        const int m = cc[o];
        for (int sidx = l16; sidx < m; sidx += 16) thr_insert<TOPK>(tv, __uint_as_float(rc[o * F_CAP + sidx].x));
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        float pv[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) pv[j] = __shfl_xor(tv[j], off, 16);
#pragma unroll
        for (int j = 0; j < TOPK; ++j) thr_insert<TOPK>(tv, pv[j]);
    }
    float kth = tv[0];
#pragma unroll
    for (int j = 1; j < TOPK; ++j) kth = (j < topk_out) ? tv[j] : kth;
    const float lim = kth + win;

    // ---- phase 2
    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffff; }
    const float *xr = xhat + row * d;
    for (int o = 0; o < own_total; ++o) {
        const int m = cc[o];
        for (int sidx = l16; sidx < m; sidx += 16) {
            const uint2 e = rc[o * F_CAP + sidx];
            if (__uint_as_float(e.x) <= lim && e.y < (unsigned)k_codes) {
                const int code = (int)e.y;
                const float *wr = what + (long)code * d;
                float accv = 0.f;
                int i = 0;
                for (; i + 8 <= d; i += 8) {                 // canonical order within a group: 0,4,1,5,2,6,3,7
                    const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4);
                    const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4);
                    accv = fmaf(x0.x, w0.x, accv); accv = fmaf(x1.x, w1.x, accv);
                    accv = fmaf(x0.y, w0.y, accv); accv = fmaf(x1.y, w1.y, accv);
                    accv = fmaf(x0.z, w0.z, accv); accv = fmaf(x1.z, w1.z, accv);
                    accv = fmaf(x0.w, w0.w, accv); accv = fmaf(x1.w, w1.w, accv);
                }
                if (i < d) {                                  // D % 8 == 4: the last half group
                    const float4 x0 = ld4(xr + i), w0 = ld4(wr + i);
                    accv = fmaf(x0.x, w0.x, accv); accv = fmaf(x0.y, w0.y, accv);
                    accv = fmaf(x0.z, w0.z, accv); accv = fmaf(x0.w, w0.w, accv);
                }
                const float sum = xn + wsq[code];
                const float two = 2.0f * accv;
                topk_insert_lex<TOPK>(bv, bi, sum - two, code);
            }
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        float pv[TOPK];
        int pi[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], off, 16); pi[j] = __shfl_xor(bi[j], off, 16); }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);
    }
    if (l16 == 0 && pos < n) {
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) { out_idx[row * topk_out + j] = bi[j]; out_dist[row * topk_out + j] = bv[j]; }
    }
}
