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

constexpr int F_BM = 256, F_BN = 256, F_BK = 32;              // codes x rows x k (fp16 elements) per stage
constexpr int F_THREADS = 512;                                // 8 waves: 2 code-side x 4 row-side, wave tile 128 x 64
constexpr int F_ROWB = F_BK * 2;                              // bytes per staged tile row (64)
constexpr int F_TILEB = F_BM * F_ROWB;                        // 16 KB per operand tile
constexpr int F_STAGEB = 2 * F_TILEB;                         // A + B = 32 KB
constexpr int F_RING = 4;                                     // stages in the LDS ring: 3 in flight while 1 is computed on
constexpr size_t F_LDS_BYTES = (size_t)F_RING * F_STAGEB;     // 128 KB -> 1 block (8 waves) / CU
constexpr size_t F_THR_BYTES = 256 * 2 * 5 * sizeof(float);  // per row: the k <= 5 smallest d~ of each code-side wave pair
constexpr size_t F_SMEM_BYTES = F_LDS_BYTES + F_THR_BYTES + 512 * 32;   // + thresholds + candidate parking (2 x 2 x 8 B per lane)
constexpr int F_GLDS_PER_STAGE = 4;                           // LDS-DMA instructions each wave issues per stage
constexpr int F_CAP = 96;                                     // candidate slots per (row, owner): ~25-30 used on random data
#ifndef MEDTOK_FILTER_WM
#define MEDTOK_FILTER_WM 2
#endif
constexpr int F_WM = MEDTOK_FILTER_WM;                        // code-side waves (2: wave tile 128 x 64; 1: wave tile 256 x 32)
constexpr int F_WN = 8 / F_WM;                                // row-side waves
constexpr int F_MT = 8 / F_WM;                                // 32-code MFMA tiles per wave
constexpr int F_NT = 8 / F_WN;                                // 32-row MFMA tiles per wave
constexpr int F_OWN_PER_SPLIT = 2 * F_WM;                     // candidate lists per row and split: 2 half-waves x code-side waves
constexpr float F_PRESCALE = 256.0f;                          // 2^8 on both operands
constexpr float F_UNSCALE = 1.0f / 65536.0f;
constexpr float F_NORM_LIMIT = 4.0f;                          // |x|^2, |e|^2 above this -> exact path
constexpr int R_ROWS = 32;                                    // rows per re-score block (8 lanes each)
constexpr int R_SURV = 64;                                    // survivors per row the re-score kernel can hold

__host__ __device__ inline float filter_gamma(int d) { return 0x1p-10f + 0x1p-20f + (float)d * 0x1p-21f; }

// eps = bound on |d~ - d| for a row with squared norm xn against codes with squared norm <= en_max
__device__ __forceinline__ float filter_eps(float xn, float en_max, int d)
{
    const float mag = sqrtf(xn * en_max);
    const float flush = 0x1p-21f * sqrtf((float)d) * sqrtf(fmaxf(xn, en_max));
    const float ulps = 0x1p-20f * (xn + en_max + 2.0f * mag);      // a handful of fp32 roundings of d-sized values
    return 2.0f * filter_gamma(d) * mag * 1.0001f + 2.0f * flush + ulps;
}

// ---------------------------------------------------------------- operand preparation
// fp32 rows -> prescaled fp16 rows in a zero-padded [n_pad, dp] image (dp % 64 == 0, n_pad % 128 == 0).
// (A k-blocked image [n/16][dp/32][16][32], which makes every LDS-DMA instruction read 8 full 128-byte lines instead of
// 16 half-used ones, measured 5-9 % SLOWER: with row-major rows the second stage that touches a line finds it in the L2.)
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

// ---------------------------------------------------------------- the filter kernel
// (Rejected variant, measured 14-24 % slower: 4-wave blocks of 256 codes x 128 rows, two per CU, hoping that co-resident
// blocks drifting apart overlap one block's epilogue / DMA issue with the other's MFMAs -- it moves 1.5 x the L2 -> LDS
// bytes per flop, and operand delivery is what binds this kernel.  The ring could also be 3 stages deep instead of 4:
// the slot of stage s is free after the mid-iteration barrier of s.)
// Block = 8 waves (2 code-side x 4 row-side), tile 256 codes x 256 rows, wave tile 128 x 64 = 4 x 2 MFMA
// tiles of 32x32x16 (L2->LDS traffic per flop halves against a 128^2 tile; at fp16 rates that is what
// binds).  Operands arrive by LDS-DMA (global_load_lds, 16 B/lane) into a double buffer; a tile row is
// 128 B = 8 chunks, stored at chunk position c ^ ((row >> 1) & 7) so that the 16 rows a ds_read_b128
// lane group touches land on 16 distinct 16-byte bank slots (the permutation is applied to the
// per-lane SOURCE address; the LDS image itself is lane-linear as the DMA requires).
// As in the fp32 kernel, codes are the A rows: a lane ends up with 16 codes of one input row per tile,
// so thresholds and candidate appends are lane-local.  wsqp is a 16-byte aligned copy of wsq padded
// with +inf to a multiple of 256, so the epilogue reads it as float4 with no bounds checks.
template <int TOPK, bool DUMP>
__global__ __launch_bounds__(F_THREADS, 2) void filter_f16_kernel(
    const _Float16 *__restrict__ xh, const _Float16 *__restrict__ wh, const float *__restrict__ xsq,
    const float *__restrict__ wsqp, const float *__restrict__ en_max_ptr, long n, int k_codes, int dp, int d,
    int codes_per_split, int own_total, uint2 *__restrict__ cand, int *__restrict__ cand_cnt,
    float *__restrict__ dump, int xcd_rows, int n_splits, int row_tile_base, int row_tile_end)
{
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / F_WN, wn = wave % F_WN;
    const int li = lane & 31, lh = lane >> 5;
    // Block -> (row tile, code split).  Plain: grid (row tiles, splits).  XCD-aware (xcd_rows > 0): consecutive block ids
    // go round-robin to the 8 XCDs and each XCD runs 32 of its blocks at a time, so block j of XCD x is made
    // (row tile (32/S of them per chunk), split j % S): the CUs of one XCD then share 32/S x tiles (L2-resident)
    // and S code streams instead of streaming 32 different x tiles through a 4 MB L2.  (A non-temporal hint on the
    // code-side loads, meant to protect the x tiles further, measured 25 % slower.)
    // A launch covers row tiles [row_tile_base, row_tile_end): the tail of a large search (the last, partly filled round of
    // blocks) is launched separately with more, shorter code splits.
    long row_tile = blockIdx.x;
    int split = blockIdx.y;
    if (xcd_rows > 0) {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3, c = j >> 5, i = j & 31;
        split = i % n_splits;
        row_tile = (long)(c * 8 + xcd) * xcd_rows + i / n_splits;
    }
    row_tile += row_tile_base;
    if (row_tile >= row_tile_end) return;
    const long row0 = row_tile * F_BN;
    const int code_lo = split * codes_per_split;
    const int code_hi = min(k_codes, code_lo + codes_per_split);
    const int nct = (code_hi - code_lo + F_BM - 1) / F_BM;
    const int nkb = dp / F_BK;
    const int nstage = nct * nkb;

    // ---- staging: wave w DMA-copies tile rows [32w, 32w+32) of A and of B, 16 rows (of 64 B) per instruction.
    // A tile row holds 4 chunks of 16 B, stored at chunk position c ^ ((row >> 2) & 3).  The per-lane part of
    // the source address is a loop-invariant 32-bit offset; everything that moves (code tile, k block, the
    // block's row base) is wave-uniform and stays in SGPRs, so a stage costs no per-lane address arithmetic.
    const int s_r = lane >> 2, s_c = lane & 3;
    unsigned lane_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = wave * 32 + q * 16 + s_r;                          // tile row this lane feeds
        const int c = s_c ^ ((r >> 2) & 3);                              // source chunk for LDS chunk s_c
        lane_off[q] = (unsigned)(r * dp + c * 8) * 2u;
    }
    const char *wbase = reinterpret_cast<const char *>(wh) + (long)code_lo * dp * 2;
    const char *xbase = reinterpret_cast<const char *>(xh) + row0 * dp * 2;
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 32 * F_ROWB);
    int pct = 0, pkb = 0, pidx = 0;         // next stage to issue: (code tile, k block, linear index)
    // Issues stage `pidx` into ring slot pidx % 4 and advances -- except past the end, where it re-issues the
    // LAST stage into the slot that already holds it (same bytes, harmless) so the steady-state loop body has
    // no branch around its DMA and one instruction schedule fits every iteration.
    // buffer-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds): SGPR descriptor + loop-invariant 32-bit lane offset +
    // SGPR stage offset -- no per-lane 64-bit address arithmetic per instruction (+3 % over global_load_lds here)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)xbase, 0, -1, 0x00020000);
    auto stage = [&]() {
        char *base = fsm + (pidx & (F_RING - 1)) * F_STAGEB + wave_lds;
        // readfirstlane: hipcc otherwise keeps the stage counters in VGPRs and wraps every buffer load in a waterfall loop
        const int ua = __builtin_amdgcn_readfirstlane((pct * F_BM * dp + pkb * F_BK) * 2);    // a code split's fp16 image is < 2 GB
        const int ub = __builtin_amdgcn_readfirstlane(pkb * F_BK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(base + q * 16 * F_ROWB), 16,
                                                     (int)lane_off[q], ua, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(base + F_TILEB + q * 16 * F_ROWB), 16,
                                                     (int)lane_off[q], ub, 0, 0);
        }
        const bool more = pidx + 1 < nstage;
        const bool wrap = pkb + 1 == nkb;
        pidx += more ? 1 : 0;
        pct += (more && wrap) ? 1 : 0;
        pkb = more ? (wrap ? 0 : pkb + 1) : pkb;
    };

    // ---- per-lane state: for each of the wave's two 32-row column tiles, the k smallest d~ so far
    float tv[F_NT][TOPK], lim[F_NT], xn[F_NT], win[F_NT];
    // Candidates found during a code tile are parked in a lane-private LDS slot pair (value, code) and written out
    // once per tile: scattered global stores inside the value loop stall the wave, and registers are scarce here.
    int np[F_NT] = {}, cnt[F_NT] = {};
    float *thr_share = reinterpret_cast<float *>(fsm + F_LDS_BYTES);    // [F_BN][2 code-side waves][5]: sorted k-smallest lists
    uint2 *park = reinterpret_cast<uint2 *>(fsm + F_LDS_BYTES + F_THR_BYTES) + tid * 4;   // [nn][slot]
    auto xrow_of = [&](int nn) -> long { return row0 + wn * (32 * F_NT) + nn * 32 + li; };
    const float en_max = en_max_ptr[0];
    const bool sane = en_max <= F_NORM_LIMIT;
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn) {
#pragma unroll
        for (int j = 0; j < TOPK; ++j) tv[nn][j] = INFINITY;
        lim[nn] = -INFINITY;                 // nothing is appended before the warm-up pass has set a finite limit
        xn[nn] = xsq[min(xrow_of(nn), n - 1)];
        win[nn] = 2.0f * filter_eps(xn[nn], en_max, d);
    }
    bool live[F_NT];                                         // padding rows never append (their limit stays -inf)
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn) live[nn] = xrow_of(nn) < n;
    const int owner = split * F_OWN_PER_SPLIT + wm * 2 + lh;
    auto put = [&](int nn, float u, int code) {       // append (d~, code) to this lane's candidate list (live rows only)
        if (cnt[nn] < F_CAP)
            cand[(xrow_of(nn) * own_total + owner) * F_CAP + cnt[nn]] = make_uint2(__float_as_uint(u + xn[nn]), (unsigned)code);
        ++cnt[nn];
    };
    constexpr int TL = TOPK < 5 ? TOPK : 5;       // list length shared per (row, code-side wave)
    for (int i = tid; i < F_BN * 2 * 5; i += F_THREADS) thr_share[i] = INFINITY;

    f32x16 acc[F_MT][F_NT];
#pragma unroll
    for (int m = 0; m < F_MT; ++m)
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // fragment addresses within a stage: row i of the tile, chunk (2t + lh) ^ ((i >> 2) & 3).  Rows 32 apart
    // share the swizzle term, so one address per (operand, t) plus compile-time row offsets covers all tiles.
    int a_adr[2], b_adr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ia = wm * (32 * F_MT) + li, ib = wn * (32 * F_NT) + li;
        a_adr[t] = ia * F_ROWB + (((2 * t + lh) ^ ((ia >> 2) & 3)) << 4);
        b_adr[t] = F_TILEB + ib * F_ROWB + (((2 * t + lh) ^ ((ib >> 2) & 3)) << 4);
    }

    // ---- software pipeline.  LDS ring of 4 stages; MFMA operands double-buffered in registers so the
    // ds_reads of the NEXT k16-step are in flight while the MFMAs of the current one issue:
    //   iteration s:  read frags(s, t1)  | MFMA(s, t0)
    //                 vmcnt (own part of stage s+1 landed) -> raw barrier (everyone's has; slot s-1 is free)
    //                 LDS-DMA stage s+3  | read frags(s+1, t0) | MFMA(s, t1)
    // __syncthreads() would drain vmcnt(0) here (an LDS-DMA is a pending LDS write), hence the raw barrier.
    auto load_frags = [&](half8 (&fa)[F_MT], half8 (&fb)[F_NT], int slot, int t) {
#ifdef MEDTOK_FILTER_NOLDS        // dev experiment: operands stay whatever they were
        if (slot >= 0) return;
#endif
        const char *pa = fsm + slot * F_STAGEB + a_adr[t];
        const char *pb = fsm + slot * F_STAGEB + b_adr[t];
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn) fb[nn] = *reinterpret_cast<const half8 *>(pb + nn * 32 * F_ROWB);
#pragma unroll
        for (int m = 0; m < F_MT; ++m) fa[m] = *reinterpret_cast<const half8 *>(pa + m * 32 * F_ROWB);
    };
    auto mfma_group = [&](const half8 (&fa)[F_MT], const half8 (&fb)[F_NT]) {
#ifdef MEDTOK_FILTER_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#ifdef MEDTOK_FILTER_NOMFMA      // dev experiment: keep the operand reads alive, skip the matrix work
#pragma unroll
        for (int m = 0; m < F_MT; ++m) asm volatile("" ::"v"(fa[m]));
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn) asm volatile("" ::"v"(fb[nn]));
        return;
#endif
#pragma unroll
        for (int m = 0; m < F_MT; ++m)
#pragma unroll
            for (int nn = 0; nn < F_NT; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb[nn], acc[m][nn], 0, 0, 0);
#ifdef MEDTOK_FILTER_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    half8 fa0[F_MT], fb0[F_NT], fa1[F_MT], fb1[F_NT];
    constexpr int LGKM0 = 0xC07F;           // s_waitcnt lgkmcnt(0) only (vmcnt / expcnt fields at their maxima)
    stage(); stage(); stage();              // stages 0..2 (clamped when the block has fewer)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(fa0, fb0, 0, 0);
    int ct = 0, kb = 0;
    // The two waves of a SIMD (w and w + 4) issue their LDS-DMA at different points of the stage: an issuing wave is
    // held for ~100 cycles per instruction, and in lockstep both would leave the matrix pipe idle at the same time
    // (+2.5-3 % measured).  Either way a wave has issued all of stage s+3 between the waits of iterations s and s+1,
    // so the counted vmcnt below is the same for both halves.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
    // The second-dispatched half of the block loses issue arbitration to the older half at every segment start
    // (priority, then age): one static priority bump for it, no flips inside the loop (+2 %, levels 1 and 3 alike;
    // -1.5 % when given to the older half instead; flips around the MFMA groups measured -1 %).
    if (late) __builtin_amdgcn_s_setprio(3);
    for (int s = 0; s < nstage; ++s) {
        if (late && s > 0) stage();         // waves 4-7: stage s+2 (slot s-2, free since the barrier of iteration s-1)
        constexpr int NRD = F_MT + F_NT;     // operand reads per k16-step (8 MFMAs)
        load_frags(fa1, fb1, s & (F_RING - 1), 1);
        mfma_group(fa0, fb0);
        // interleave: the 6 operand reads of the next k16-step ride between the first MFMAs
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, NRD - 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        // fa1/fb1 have landed (free: 8 MFMAs went by); own part of stage s+1 has landed; then everyone's has
        __builtin_amdgcn_s_waitcnt(LGKM0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifndef MEDTOK_FILTER_NOBAR       // dev experiment
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#ifndef MEDTOK_FILTER_NODMA       // dev experiment: without it the ring keeps its prologue contents
        if (!late) stage();                 // waves 0-3: stage s+3 (slot s-1: everyone is past reading it)
#endif
        load_frags(fa0, fb0, (s + 1) & (F_RING - 1), 0);      // (past the last stage this reads stale LDS, never used)
        mfma_group(fa1, fb1);
        // after the barrier the matrix pipe restarts at once; DMA issue and operand reads ride between MFMAs
        // (other interleave patterns for this half measured the same; two reads per MFMA in the first half: -5 %)
#pragma unroll
        for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, NRD / 3, 0); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifdef MEDTOK_FILTER_NOEPI      // dev experiment: main loop only (results are garbage)
        if (++kb == nkb) {
#pragma unroll
            for (int m = 0; m < F_MT; ++m)
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn) {
                    asm volatile("" ::"v"(acc[m][nn]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
                }
            kb = 0; ++ct;
        }
        continue;
#endif
        if (++kb == nkb) {
            // ---- epilogue.  Everything is kept relative to the lane's own |x|^2:  u = en - 2 s~ (one fmaf per
            // value, shared by nothing else), thresholds and limits in the same u scale; d~ = u + xn is formed only
            // for the few values that are stored.
            const int cbase = code_lo + ct * F_BM + wm * (32 * F_MT) + 4 * lh;
            const bool warm = (ct == 0);
            if (!DUMP && warm) {
                // First code tile: learn the thresholds from all 128 codes BEFORE appending anything, so the
                // candidate lists do not fill up with the loose early threshold (appends only ever need T >= t~).
#pragma unroll
                for (int m = 0; m < F_MT; ++m) {
                    float4 wen[4];          // vector loads, one wait per 16 values (a load per value drains the DMA ring each time)
#pragma unroll
                    for (int g = 0; g < 4; ++g) wen[g] = ld4(wsqp + cbase + 32 * m + 8 * g);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float4 e4 = wen[r >> 2];
                        const float en = (r & 3) == 0 ? e4.x : (r & 3) == 1 ? e4.y : (r & 3) == 2 ? e4.z : e4.w;
#pragma unroll
                        for (int nn = 0; nn < F_NT; ++nn) thr_insert<TOPK>(tv[nn], fmaf(acc[m][nn][r], -0x1p-15f, en));
                    }
                }
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn) lim[nn] = live[nn] ? fminf(tv[nn][TOPK - 1] + win[nn], 3.0e38f) : -INFINITY;
            }
#pragma unroll
            for (int m = 0; m < F_MT; ++m) {
                f32x16 env;                  // |e|^2 of this lane's 16 codes of the tile, in accumulator-register order
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 e4 = ld4(wsqp + cbase + 32 * m + 8 * g);
                    env[4 * g] = e4.x; env[4 * g + 1] = e4.y; env[4 * g + 2] = e4.z; env[4 * g + 3] = e4.w;
                }
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn) {
#ifndef MEDTOK_FILTER_EPI_SCALAR
                    // Four values per test: two packed fmas (bit-identical to fmaf per element), a 4-way min, ONE compare
                    // and branch; only a quad that holds a passing value in some lane is scanned value by value.  A hit
                    // is rare per lane but not per wave (64 lanes x 4 values at p ~ 2e-3: a third of the quads).  (+1 % over
                    // the per-value form below; clearing the accumulators by a C = 0 first MFMA instead of 128 v_movs was
                    // tried too: the per-stage branch it needs costs more than the moves.)
                    if (!DUMP) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 cc = {-0x1p-15f, -0x1p-15f};
                            const f32x2 a01 = {acc[m][nn][4 * g], acc[m][nn][4 * g + 1]}, a23 = {acc[m][nn][4 * g + 2], acc[m][nn][4 * g + 3]};
                            const f32x2 e01 = {env[4 * g], env[4 * g + 1]}, e23 = {env[4 * g + 2], env[4 * g + 3]};
                            const f32x2 u01 = __builtin_elementwise_fma(a01, cc, e01), u23 = __builtin_elementwise_fma(a23, cc, e23);
                            const float uq[4] = {u01.x, u01.y, u23.x, u23.y};
                            if (fminf(fminf(uq[0], uq[1]), fminf(uq[2], uq[3])) <= lim[nn]) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (uq[j] <= lim[nn]) {
                                        const int code = cbase + 32 * m + j + 8 * g;
                                        if (np[nn] < 2) { park[nn * 2 + np[nn]] = make_uint2(__float_as_uint(uq[j]), (unsigned)code); ++np[nn]; }
                                        else put(nn, uq[j], code);
                                    }
                            }
                        }
                    } else
#endif
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float u = fmaf(acc[m][nn][r], -0x1p-15f, env[r]);      // padded codes carry en = +inf
                        if (DUMP) {
                            const int code = cbase + 32 * m + (r & 3) + 8 * (r >> 2);
                            if (code < code_hi && xrow_of(nn) < n) dump[xrow_of(nn) * k_codes + code] = acc[m][nn][r] * F_UNSCALE;
                        } else if (u <= lim[nn]) {                                 // lim is finite, so +inf never passes
                            // rare per lane (about 6/m after m codes) but not per wave: keep this body minimal -- park the
                            // candidate in LDS; only a third hit within one code tile pays for a global store right here
                            // (that one skips the k-smallest list: T stays valid, a touch looser).
                            const int code = cbase + 32 * m + (r & 3) + 8 * (r >> 2);
                            if (np[nn] < 2) { park[nn * 2 + np[nn]] = make_uint2(__float_as_uint(u), (unsigned)code); ++np[nn]; }
                            else put(nn, u, code);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
                }
            }
            if (!DUMP) {
                // per tile: flush, update the lane's k-smallest list, then combine the row's four owners into one threshold
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn) {
                    // write out the parked candidates (at most two store instructions per tile) and fold them into the
                    // k-smallest list; the warm-up pass already counted the first tile's values
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < np[nn]) {
                            const uint2 e = park[nn * 2 + q];
                            put(nn, __uint_as_float(e.x), (int)e.y);
                            if (!warm) thr_insert<TOPK>(tv[nn], __uint_as_float(e.x));
                        }
                    }
                    np[nn] = 0;
                    // The row's k-th best over ALL codes seen so far, exactly: merge the sorted k-lists of the row's four
                    // owners.  (The minimum of the owners' own k-th bests -- the old rule -- is only about the 4k-th best
                    // of the union: 2-3 x the candidates.)  Two sorted lists a, b: {min(a_i, b_{k-1-i})} are the k smallest
                    // of their union.  The lh partner comes by shuffle; the other code-side wave publishes its merged list
                    // in LDS -- possibly one tile old, which is still a list of values of real codes, so T stays valid.
                    // (Extending the union across code splits through agent-scope global lists was measured 10 % SLOWER:
                    // the extra global loads/stores in the epilogue cost more than the candidates they save.)
                    float t;
                    if (TOPK <= 5) {
                        float c[TL];
#pragma unroll
                        for (int i = 0; i < TL; ++i) c[i] = fminf(tv[nn][i], __shfl_xor(tv[nn][TL - 1 - i], 32, 64));
#pragma unroll
                        for (int pass = 0; pass < TL; ++pass)              // odd-even transposition: c ascending
#pragma unroll
                            for (int i = pass & 1; i + 1 < TL; i += 2) {
                                const float lo = fminf(c[i], c[i + 1]), hi = fmaxf(c[i], c[i + 1]);
                                c[i] = lo; c[i + 1] = hi;
                            }
                        float *lst = thr_share + (size_t)(wn * (32 * F_NT) + nn * 32 + li) * 10;
                        if (lh == 0) {
#pragma unroll
                            for (int i = 0; i < TL; ++i) lst[wm * 5 + i] = c[i];
                        }
                        t = -INFINITY;
#pragma unroll
                        for (int i = 0; i < TL; ++i) t = fmaxf(t, fminf(c[i], lst[(wm ^ 1) * 5 + TL - 1 - i]));
                    } else {                 // k = 8: the lists would not fit beside the ring; the pair's looser min rule
                        t = tv[nn][TOPK - 1];
                        t = fminf(t, __shfl_xor(t, 32, 64));
                    }
                    lim[nn] = live[nn] ? fminf(t + win[nn], 3.0e38f) : -INFINITY;
                }
            }
            kb = 0;
            ++ct;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail re-issues may still be in flight
    if (!DUMP) {
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn) {
            // rows outside the range the bound assumes are forced onto the exact path
            const bool ok = sane && xn[nn] <= F_NORM_LIMIT;
            if (live[nn]) cand_cnt[xrow_of(nn) * own_total + owner] = ok ? cnt[nn] : F_CAP + 1;
        }
    }
}

// ---------------------------------------------------------------- exact re-score
// Block = 32 rows x 8 lanes.  Phase 1 (per row, 8 lanes): the row's true t~ = k-th smallest d~ over all
// owners' candidates.  Phase 2a: candidates with d~ <= t~ + 2 eps are compacted into an LDS list.
// Phase 2b: the block's survivors (about 7 per row, so ~220 for 256 threads) are spread densely over the
// threads and each is re-scored with the canonical fp32 chain; x and code rows stream from L1/L2 (the
// survivors of a row run side by side, so its x row is fetched once).  Phase 3 (per row): exact
// (d, index) top-k of the row's survivors.
template <int TOPK>
__global__ __launch_bounds__(256) void rescore_kernel(
    const uint2 *__restrict__ cand, const int *__restrict__ cand_cnt, int own_total,
    const uint2 *__restrict__ cand_tail, const int *__restrict__ cnt_tail, int own_tail, long tail_start,
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, const float *__restrict__ en_max_ptr, long n, int k_codes, int d, int topk_out,
    int64_t *__restrict__ out_idx, float *__restrict__ out_dist, int *__restrict__ fb_count, int *__restrict__ fb_rows,
    const float *__restrict__ xref, float *__restrict__ w_out, float *zq_out, long zq_stride)
{
    __shared__ int s_code[R_ROWS * R_SURV];
    __shared__ float s_d[R_ROWS * R_SURV];
    __shared__ int s_cnt[2 * R_ROWS + 1];                 // [R_ROWS] survivors, [R_ROWS] offsets, total
    const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const long pos = (long)blockIdx.x * R_ROWS + g;
    const long row = min(pos, n - 1);
    if (threadIdx.x < R_ROWS) s_cnt[threadIdx.x] = 0;
    const float xn = xsq[row];
    const float win = 2.0f * filter_eps(xn, en_max_ptr[0], d);
    // rows from tail_start on were filtered by the tail launch: its own lists, own_tail of them per row
    const bool in_tail = row >= tail_start;
    if (in_tail) own_total = own_tail;
    const uint2 *rc = in_tail ? cand_tail + (row - tail_start) * own_total * F_CAP : cand + row * own_total * F_CAP;
    const int *cc = in_tail ? cnt_tail + (row - tail_start) * own_total : cand_cnt + row * own_total;

    bool overflow = false;
    for (int o = 0; o < own_total; ++o) overflow |= cc[o] > F_CAP;
    __syncthreads();
    // ---- phase 1 + 2a (skipped for rows the filter gave up on; the 8-lane group branches together)
    if (!overflow) {
        float tv[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) tv[j] = INFINITY;
        for (int o = 0; o < own_total; ++o) {
            const int m = cc[o];
            for (int sidx = l8; sidx < m; sidx += 8) thr_insert<TOPK>(tv, __uint_as_float(rc[o * F_CAP + sidx].x));
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            float pv[TOPK];
#pragma unroll
            for (int j = 0; j < TOPK; ++j) pv[j] = __shfl_xor(tv[j], off, 8);
#pragma unroll
            for (int j = 0; j < TOPK; ++j) thr_insert<TOPK>(tv, pv[j]);
        }
        float kth = tv[0];
#pragma unroll
        for (int j = 1; j < TOPK; ++j) kth = (j < topk_out) ? tv[j] : kth;
        const float lim = kth + win;
        for (int o = 0; o < own_total; ++o) {
            const int m = cc[o];
            for (int sidx = l8; sidx < m; sidx += 8) {
                const uint2 e = rc[o * F_CAP + sidx];
                if (__uint_as_float(e.x) <= lim && e.y < (unsigned)k_codes) {
                    const int p = atomicAdd(&s_cnt[g], 1);
                    if (p < R_SURV) s_code[g * R_SURV + p] = (int)e.y;
                }
            }
        }
    }
    __syncthreads();
    // rows with more survivors than the list holds also go to the exact path
    if (threadIdx.x < R_ROWS && s_cnt[threadIdx.x] > R_SURV) s_cnt[threadIdx.x] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int r = 0; r < R_ROWS; ++r) { s_cnt[R_ROWS + r] = run; run += max(s_cnt[r], 0); }
        s_cnt[2 * R_ROWS] = run;
    }
    __syncthreads();
    // ---- phase 2b: dense exact chains
    const int total = s_cnt[2 * R_ROWS];
    for (int wi = threadIdx.x; wi < total; wi += 256) {
        int rr = 0;
#pragma unroll
        for (int r = 1; r < R_ROWS; ++r) rr += (wi >= s_cnt[R_ROWS + r]) ? 1 : 0;
        const int j = wi - s_cnt[R_ROWS + rr];
        const int code = s_code[rr * R_SURV + j];
        const long arow = min((long)blockIdx.x * R_ROWS + rr, n - 1);
        const float *xr = xhat + arow * d;
        const float *wr = what + (long)code * d;
        float accv = 0.f;
        int i = 0;
        for (; i + 16 <= d; i += 16) {               // canonical order within a group of 8: 0,4,1,5,2,6,3,7
            const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4), x2 = ld4(xr + i + 8), x3 = ld4(xr + i + 12);
            const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4), w2 = ld4(wr + i + 8), w3 = ld4(wr + i + 12);
            accv = fmaf(x0.x, w0.x, accv); accv = fmaf(x1.x, w1.x, accv);
            accv = fmaf(x0.y, w0.y, accv); accv = fmaf(x1.y, w1.y, accv);
            accv = fmaf(x0.z, w0.z, accv); accv = fmaf(x1.z, w1.z, accv);
            accv = fmaf(x0.w, w0.w, accv); accv = fmaf(x1.w, w1.w, accv);
            accv = fmaf(x2.x, w2.x, accv); accv = fmaf(x3.x, w3.x, accv);
            accv = fmaf(x2.y, w2.y, accv); accv = fmaf(x3.y, w3.y, accv);
            accv = fmaf(x2.z, w2.z, accv); accv = fmaf(x3.z, w3.z, accv);
            accv = fmaf(x2.w, w2.w, accv); accv = fmaf(x3.w, w3.w, accv);
        }
        for (; i + 8 <= d; i += 8) {
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
        const float sum = xsq[arow] + wsq[code];
        const float two = 2.0f * accv;
        s_d[rr * R_SURV + j] = sum - two;
    }
    __syncthreads();
    // ---- phase 3: exact top-k of the row's survivors
    const int mine = s_cnt[g];
    if (overflow || mine < 0) {
        if (l8 == 0 && pos < n) fb_rows[atomicAdd(fb_count, 1)] = (int)row;
        return;
    }
    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffff; }
    for (int j = l8; j < mine; j += 8) topk_insert_lex<TOPK>(bv, bi, s_d[g * R_SURV + j], s_code[g * R_SURV + j]);
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
        float pv[TOPK];
        int pi[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], off, 8); pi[j] = __shfl_xor(bi[j], off, 8); }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);
    }
    if (l8 == 0 && pos < n) {
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) { out_idx[row * topk_out + j] = valid_code(bi[j], j, k_codes); out_dist[row * topk_out + j] = bv[j]; }
    }
#pragma unroll
    for (int j = 0; j < TOPK; ++j) bi[j] = valid_code(bi[j], j, k_codes);       // the fused assignment below gathers with these
    // ---- phase 4 (one-call forward only): the soft assignment of soft_assign_kernel, same arithmetic, while the top-k
    // code rows this block has just re-scored are still hot in the L2 (a separate launch re-gathers them from the
    // Infinity Cache: 9 GB per 600k-row search).  The row's 8 lanes all hold the merged (bv, bi) lists.
    if (zq_out && pos < n) {
        float wj[TOPK];
        const float m = -bv[0];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) { wj[j] = expf(-bv[j] - m); sum += wj[j]; }
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) wj[j] = wj[j] / sum;
        if (w_out && l8 == 0) {
#pragma unroll
            for (int j = 0; j < TOPK; ++j)
                if (j < topk_out) w_out[row * topk_out + j] = wj[j];
        }
        const float *xr = xref + row * d;
        float *o = zq_out + row * zq_stride;
        for (int i = l8 * 4; i < d; i += 32) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < TOPK; ++j)
                if (j < topk_out) {
                    const float4 e = ld4(what + (long)bi[j] * d + i);
                    a.x = fmaf(wj[j], e.x, a.x); a.y = fmaf(wj[j], e.y, a.y);
                    a.z = fmaf(wj[j], e.z, a.z); a.w = fmaf(wj[j], e.w, a.w);
                }
            const float4 x = ld4(xr + i);
            st4(o + i, make_float4(x.x + (a.x - x.x), x.y + (a.y - x.y), x.z + (a.z - x.z), x.w + (a.w - x.w)));
        }
    }
}

// padded, aligned copy of wsq: [k_pad] with +inf beyond k_codes
__global__ __launch_bounds__(256) void pad_wsq_kernel(const float *__restrict__ wsq, int k, int k_pad, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k_pad) out[i] = i < k ? wsq[i] : INFINITY;
}
