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
// own rounding we budget D * 2^-22 relative: D additions that each round to nearest err by D * 2^-24, D additions that each
// TRUNCATE by D * 2^-23; both sums together (the MFMA's under the worse model, the exact chain's under its own) are 3 D 2^-24, the
// budget is 4 D 2^-24 (rounds 1-4 carried twice that; measured on MI355X in
// tests/test_gpu_filter.py::test_filter_score_error_bound: a small fraction of the budget: the matrix pipe sums the 16 products
// of an instruction before it rounds), hence
//   |s~ - s| <= gamma * sqrt(xsq * wsq_max),   gamma = 2^-10 + 2^-20 + D * 2^-22
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
typedef unsigned fu32x4 __attribute__((ext_vector_type(4)));

constexpr int F_BM = 256, F_BN = 256, F_BK = 32;              // codes x rows x k (fp16 elements) per stage
constexpr int F_THREADS = 512;                                // 8 waves: 2 code-side x 4 row-side, wave tile 128 x 64
constexpr int F_ROWB = F_BK * 2;                              // bytes per staged tile row (64)
constexpr int F_TILEB = F_BM * F_ROWB;                        // 16 KB per operand tile
constexpr int F_STAGEB = 2 * F_TILEB;                         // A + B = 32 KB
constexpr int F_RING = 4;                                     // stages in the LDS ring: 3 in flight while 1 is computed on
constexpr size_t F_LDS_BYTES = (size_t)F_RING * F_STAGEB;     // 128 KB -> 1 block (8 waves) / CU
constexpr size_t F_THR_BYTES = 256 * 64;                      // per row: the k <= 5 smallest d~ of each code-side wave pair (2 x 32 B)
constexpr size_t F_INIT_BYTES = 2 * 256 * 4;                  // the accumulator start values (-2^15 |e|^2) of two code tiles
constexpr size_t F_SMEM_BYTES = F_LDS_BYTES + F_THR_BYTES + F_INIT_BYTES;    // 146 KB
constexpr int F_GLDS_PER_STAGE = 4;                           // LDS-DMA instructions each wave issues per stage
// Candidate slots per (row, owner); a list that overflows sends its row to the exact kernel.  Measured at N = 600k, D = 768
// (tools/cap_experiment.py, random rows): 32 slots overflow for 49 (K = 16384) / 506 (K = 49152) rows and cost 5-8 % through the
// fallback; 48 and 96 overflow for none and time the same -- 48 halves the candidate region (1.8 GB instead of 3.7 GB per
// 600k-row search).  Inputs that defeat the filter (thousands of near-identical codes) overflow any capacity and are simply
// searched exactly.
constexpr int F_CAP = 48;
constexpr int F_WM = 2;                                       // code-side waves (wave tile 128 x 64)
constexpr int F_WN = 8 / F_WM;                                // row-side waves
constexpr int F_MT = 8 / F_WM;                                // 32-code MFMA tiles per wave
constexpr int F_NT = 8 / F_WN;                                // 32-row MFMA tiles per wave
constexpr int F_OWN_PER_SPLIT = 2 * F_WM;                     // candidate lists per row and split: 2 half-waves x code-side waves
constexpr float F_PRESCALE = 256.0f;                          // 2^8 on both operands
constexpr float F_UNSCALE = 1.0f / 65536.0f;
constexpr float F_NORM_LIMIT = 4.0f;                          // |x|^2, |e|^2 above this -> exact path
constexpr int R_ROWS = 32;                                    // rows per re-score block (8 lanes each)
// survivors per row the re-score kernel can hold: 28.7 KB of LDS per 32-row block = five blocks per CU, what its 90 VGPRs allow anyway
// (64 until round 6: a row next to a cluster of ~100 near-identical codes keeps them all -- bench.py --data clustered_codebook)
constexpr int R_SURV = 112;

// gamma: operand rounding (2^-10 + 2^-20: 11-bit significands on both sides) + the accumulation budget.  The accumulation part must cover,
// in eps = 2 gamma mag, the d + 63 additions of the approximate score (each may truncate: 2^-23 of a magnitude <= mag) plus the exact
// chain's own d roundings (2^-24 each): (1.5 d + 63) 2^-22 mag.  2 d 2^-22 covers that from d = 126 up; below it the term is
// (0.75 d + 32) 2^-22 (at the default e_dim = 64 the round-5 budget rested on measured slack, not on the stated proof).
__host__ __device__ inline float filter_gamma(int d)
{
    const float acc = (float)d > 0.75f * (float)d + 32.0f ? (float)d : 0.75f * (float)d + 32.0f;
    return 0x1p-10f + 0x1p-20f + acc * 0x1p-22f;
}

// eps = bound on |d~ - d| for a row with squared norm xn against codes with squared norm <= en_max
__device__ __forceinline__ float filter_eps(float xn, float en_max, int d)
{
    const float mag = sqrtf(xn * en_max);
    const float flush = 0x1p-21f * sqrtf((float)d) * sqrtf(fmaxf(xn, en_max));
    const float ulps = 0x1p-20f * (xn + en_max + 2.0f * mag);      // a handful of fp32 roundings of d-sized values
    // the accumulators start at -2^15 |e|^2 instead of 0: every one of the <= d + 63 additions of the chain rounds a value of
    // magnitude <= 2^15 (|e|^2 + 2 mag), i.e. 2^-24 (|e|^2 + 2 mag) in d units; the 2 mag part is inside gamma's D 2^-22,
    // the |e|^2 part is budgeted here with the same factor 4 over the one-ulp-per-addition model (2 over truncating additions)
    const float start = (float)(d + 64) * 0x1p-22f * en_max;
    return 2.0f * filter_gamma(d) * mag * 1.0001f + 2.0f * flush + ulps + start;
}

// ---------------------------------------------------------------- operand preparation
// fp32 rows -> prescaled fp16 rows in a zero-padded [n_pad, dp] image (dp % 64 == 0, n_pad % 128 == 0).
// (A k-blocked image [n/16][dp/32][16][32], which makes every LDS-DMA instruction read 8 full 128-byte lines instead of
// 16 half-used ones, measured 5-9 % SLOWER: with row-major rows the second stage that touches a line finds it in the L2.)
__global__ __launch_bounds__(256) void to_half_kernel(const float *__restrict__ src, long n, int d, long n_pad, int dp,
                                                      _Float16 *__restrict__ dst, int *__restrict__ zero_word = nullptr)
{
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;      // (see rownorm_kernel)
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

// max of wsq[0..k) -> out[0] (single block; NaN propagates as +inf so the filter bails out).  The search's other two bits of
// per-codebook preparation ride along (they were a launch and a memset of their own: ~10 us per search at serving sizes): wsqp (if
// given) = the accumulator start values [k_pad], -2^15 |e|^2 (exact: a power-of-two scale), -inf beyond k so that padded codes never
// pass; *zero_me (if given) = 0 (the count of rows handed to the exact kernel).
__device__ __forceinline__ void wsq_max_body(const float *__restrict__ wsq, int k, float *__restrict__ out,
                                             float *__restrict__ wsqp, int k_pad, int *__restrict__ zero_me)
{
    __shared__ float sh[1024];
    float m = 0.f;
    for (int i = threadIdx.x; i < k; i += 1024) {
        const float v = wsq[i];
        m = (v > m || !(v == v)) ? (v == v ? v : INFINITY) : m;
        if (wsqp) wsqp[i] = v * -32768.0f;
    }
    if (wsqp)
        for (int i = k + threadIdx.x; i < k_pad; i += 1024) wsqp[i] = -INFINITY;
    if (zero_me && threadIdx.x == 0) *zero_me = 0;
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}
__global__ __launch_bounds__(1024) void wsq_max_kernel(const float *__restrict__ wsq, int k, float *__restrict__ out,
                                                       float *__restrict__ wsqp = nullptr, int k_pad = 0, int *__restrict__ zero_me = nullptr)
{
    wsq_max_body(wsq, k, out, wsqp, k_pad, zero_me);
}
// The same for several code ranges of one codebook in one launch (block = range): what a search's preparation computes per call,
// once per weight version for every region the caller searches (medtok_codebook_prepare_f32).
constexpr int PREP_MAX_REGIONS = 4;
struct RegionPrep { const float *wsq; int k; float *en_max; float *wsqp; int k_pad; };
struct RegionPrepArgs { RegionPrep r[PREP_MAX_REGIONS]; };
__global__ __launch_bounds__(1024) void wsq_max_regions_kernel(const RegionPrepArgs a)
{
    const RegionPrep q = a.r[blockIdx.x];
    wsq_max_body(q.wsq, q.k, q.en_max, q.wsqp, q.k_pad, nullptr);
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

// thr_insert without branches: one min/max pair per slot carries the displaced value down the list.  (In the scan every
// instruction of a dependent, branchy insert is exposed latency; NaN never gets here -- a NaN fails the >= test -- and the
// warm-up pass maps it to +inf first.)
template <int T>
__device__ __forceinline__ void thr_insert_bf(float (&tv)[T], float v)
{
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const float lo = fminf(tv[j], v);
        v = fmaxf(tv[j], v);
        tv[j] = lo;
    }
}

// Per-lane state of one 32-row column tile of the filter kernel: the k smallest u = d~ - |x|^2 seen so far, the limit in
// accumulator scale, the row's constants and its candidate list.  Plain structs and force-inlined functions with explicit
// references, not closures: hipcc leaves by-reference lambda captures that are reached through other lambdas in scratch
// memory (a flat load plus a vmcnt wait in the middle of the scan).
template <int TOPK>
struct FilterRow {
    float tv[TOPK];       // sorted ascending
    float L;              // a value passes when acc >= L  (<=> u <= t + win)
    float xn, win;        // |x|^2; 2 eps (-inf for padding rows: their limit stays +inf and nothing is ever appended)
    // The lane's candidate list of the row, as byte offsets from a wave-uniform base (SGPR base + 32-bit offset stores, no 64-bit
    // address arithmetic per hit).  pos counts every hit; a hit beyond the capacity overwrites the LAST slot (store offset =
    // min(pos, endm8)), which is harmless: a list with more than F_CAP hits sends its row to the exact kernel anyway.
    unsigned pos, endm8;  // next slot; last slot
    unsigned lst;         // LDS byte address of the row's shared k-lists: [2 code-side waves][8 floats, 5 used]
};
constexpr int F_LST_ROWB = 64;        // bytes per row of the shared k-lists (two 32-byte halves: 16-byte aligned b128 reads)

// LDS accesses of the epilogue go through asm.  hipcc orders ds_writes -- and, outside the main loop, ds_reads -- behind ALL
// pending LDS-DMA ("s_waitcnt vmcnt(0)": a DMA is a pending LDS write that might alias), i.e. every one of them drained the
// operand ring: for the waves that have just issued their DMA, a round trip to the L2 or beyond, several times per scan.
// These bytes are never DMA targets.  (Same-wave LDS operations complete in order; the untracked lgkmcnt ticks only ever
// make the compiler's own counted waits more conservative.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_store_list(unsigned addr, const float (&c)[5])
{
    f32x4 q = {c[0], c[1], c[2], c[3]};
    // (s_nop: a VALU write to the data registers of a 128-bit LDS store needs two wait states on gfx950; hipcc counts them for its own
    // stores only)
    // (leading s_nop: the same hazard in the other direction -- the instruction in front may be the asm v_min / v_max that wrote q)
    asm volatile("s_nop 0\n\tds_write_b128 %0, %1\n\tds_write_b32 %0, %2 offset:16\n\ts_nop 0" ::"v"(addr), "v"(q), "v"(c[4]) : "memory");
}
// The other code-side wave's lists of both rows: four reads issued early (they may be a few stages stale anyway), waited for
// where the values are needed.  Between the two statements the destination registers belong to the hardware: the wait
// statement names them as read-write so that hipcc neither uses nor moves them before it.
struct OtherLists { f32x4 q0, q1; float e0, e1; };
__device__ __forceinline__ void lds_load_lists_issue(unsigned addr0, unsigned addr1, OtherLists &o)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b32 %1, %4 offset:16\n\tds_read_b128 %2, %5\n\tds_read_b32 %3, %5 offset:16"
                 : "=&v"(o.q0), "=&v"(o.e0), "=&v"(o.q1), "=&v"(o.e1) : "v"(addr0), "v"(addr1) : "memory");
}
__device__ __forceinline__ void lds_load_lists_wait(OtherLists &o)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.q0), "+v"(o.e0), "+v"(o.q1), "+v"(o.e1)::"memory");
}
// value of the lane 32 away: v_permlane32_swap exchanges the upper half of one operand with the lower half of the other, so with
// both operands = v the results hold v's lower half twice and its upper half twice (one VALU instruction; ds_bpermute is an
// LDS round trip)
__device__ __forceinline__ float other_half(float v, int lh)
{
    const unsigned b = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(lh ? r[0] : r[1]);
}

// Single-instruction helpers for the scan: fmaxf / fminf on MFMA results make hipcc put a canonicalising v_max in front of each
// of them, and plain -O3 packs neighbouring f32 operations into v_pk_* (slower beside MFMAs).
__device__ __forceinline__ float v_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__device__ __forceinline__ float v_med3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// sorted insert without a dependent chain: t_j' = med3(t_{j-1}, t_j, v) (t_{j-1} <= t_j), t_0' = min(t_0, v); v must not be NaN
template <int T>
__device__ __forceinline__ void thr_insert_med3(float (&tv)[T], float v)
{
#pragma unroll
    for (int j = T - 1; j >= 1; --j) tv[j] = v_med3(tv[j - 1], tv[j], v);
    tv[0] = v_min(tv[0], v);
}

// One candidate: the score in ACCUMULATOR scale (acc = -2^15 u; the re-score kernel converts: d~ = fma(acc, -2^-15, |x|^2), the
// same bits as rounding u + |x|^2) and the code.
__device__ __forceinline__ void filter_append(unsigned &pos, unsigned endm8, const char *cbase, float v, unsigned code)
{
    *reinterpret_cast<uint2 *>(const_cast<char *>(cbase) + min(pos, endm8)) = make_uint2(__float_as_uint(v), code);
    pos += 8;
}

// The hit path of the scan as ONE straight-line, exec-masked instruction sequence (about 28 instructions; the compiler's
// version of the same logic is 60-70 with four nested branches, and every instruction of it is exposed: while a wave scans,
// its SIMD has no matrix work).  For the lanes whose quad maximum passes: find its position (three compares), append
// (acc, code) with two SGPR-base stores, fold u = -2^-15 acc into the sorted k-list with independent v_med3
// (t_j' = med3(t_{j-1}, t_j, u): no dependent chain), and note in `multi` whether the quad's SECOND largest value passes as well
// (p ~ 4e-5 per lane and quad on random data): the tile is then revisited by filter_scan_rest.  L is not moved here: the
// merge at the end of the code tile recomputes it from the lists.
// (UCALC: how u = -2^-15 acc is formed -- a multiply, or, for filter_rows64_kernel, an fma with a wave-uniform bias that is 0 or
// +inf: with +inf the med3 chain leaves the list as it is, i.e. the same instructions append without inserting.)
#define F_HIT_HEAD(UCALC)                                                                \
    "v_cmp_ge_f32 vcc, %[mx], %[L]\n\t"                                                  \
    "s_and_saveexec_b64 %[sv], vcc\n\t"                                                  \
    "v_cmp_eq_f32 vcc, %[a2], %[mx]\n\t"                                                 \
    "v_cndmask_b32_e64 %[j], 3, 2, vcc\n\t"                                              \
    "v_cmp_eq_f32 vcc, %[a1], %[mx]\n\t"                                                 \
    "v_cndmask_b32_e64 %[j], %[j], 1, vcc\n\t"                                           \
    "v_cmp_eq_f32 vcc, %[a0], %[mx]\n\t"                                                 \
    "v_cndmask_b32_e64 %[j], %[j], 0, vcc\n\t"                                           \
    "v_max_f32 %[p], %[a0], %[a1]\n\t"                                                   \
    "v_max_f32 %[q], %[a2], %[a3]\n\t"                                                   \
    "v_min_f32 %[p], %[p], %[q]\n\t"                                                     \
    "v_min_f32 %[q], %[a0], %[a1]\n\t"                                                   \
    "v_min_f32 %[u], %[a2], %[a3]\n\t"                                                   \
    "v_max3_f32 %[p], %[p], %[q], %[u]\n\t"                                              \
    "v_cmp_ge_f32 vcc, %[p], %[L]\n\t"                                                   \
    "s_or_b64 %[multi], %[multi], vcc\n\t"                                               \
    UCALC                                                                                \
    "v_add_u32 %[j], %[j], %[cbg]\n\t"                                                   \
    "v_min_u32 %[q], %[pos], %[endm8]\n\t"                                               \
    "global_store_dword %[q], %[mx], %[base]\n\t"                                        \
    "global_store_dword %[q], %[j], %[base] offset:4\n\t"                                \
    "v_add_u32 %[pos], 8, %[pos]\n\t"
#define F_HIT_TAIL "s_mov_b64 exec, %[sv]"
#define F_HIT_OUT(r) [pos] "+v"(r.pos), [multi] "+s"(multi), [sv] "=&s"(sv), [j] "=&v"(j), [p] "=&v"(p), [q] "=&v"(q), [u] "=&v"(u)
#define F_HIT_IN(r) [mx] "v"(mx), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [L] "v"(r.L), [cbg] "v"(cbg), [endm8] "v"(r.endm8), [base] "s"(cbase)
#define F_HIT_ASM(UCALC, ...)                                                                                                                   \
    if constexpr (TOPK == 1) {                                                                                                                  \
        asm volatile(F_HIT_HEAD(UCALC) "v_min_f32 %[t0], %[t0], %[u]\n\t" F_HIT_TAIL                                                            \
                     : F_HIT_OUT(r), [t0] "+v"(r.tv[0]) : F_HIT_IN(r) __VA_ARGS__ : "vcc");                                                     \
    } else if constexpr (TOPK == 5) {                                                                                                           \
        asm volatile(F_HIT_HEAD(UCALC)                                                                                                          \
                     "v_med3_f32 %[t4], %[t3], %[t4], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t3], %[t2], %[t3], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t2], %[t1], %[t2], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t1], %[t0], %[t1], %[u]\n\t"                                                                                 \
                     "v_min_f32 %[t0], %[t0], %[u]\n\t" F_HIT_TAIL                                                                              \
                     : F_HIT_OUT(r), [t0] "+v"(r.tv[0]), [t1] "+v"(r.tv[1]), [t2] "+v"(r.tv[2]), [t3] "+v"(r.tv[3]), [t4] "+v"(r.tv[4])         \
                     : F_HIT_IN(r) __VA_ARGS__ : "vcc");                                                                                        \
    } else {                                                                                                                                    \
        static_assert(TOPK == 8, "k-lists of 1, 5 or 8");                                                                                       \
        asm volatile(F_HIT_HEAD(UCALC)                                                                                                          \
                     "v_med3_f32 %[t7], %[t6], %[t7], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t6], %[t5], %[t6], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t5], %[t4], %[t5], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t4], %[t3], %[t4], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t3], %[t2], %[t3], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t2], %[t1], %[t2], %[u]\n\t"                                                                                 \
                     "v_med3_f32 %[t1], %[t0], %[t1], %[u]\n\t"                                                                                 \
                     "v_min_f32 %[t0], %[t0], %[u]\n\t" F_HIT_TAIL                                                                              \
                     : F_HIT_OUT(r), [t0] "+v"(r.tv[0]), [t1] "+v"(r.tv[1]), [t2] "+v"(r.tv[2]), [t3] "+v"(r.tv[3]), [t4] "+v"(r.tv[4]),        \
                       [t5] "+v"(r.tv[5]), [t6] "+v"(r.tv[6]), [t7] "+v"(r.tv[7])                                                               \
                     : F_HIT_IN(r) __VA_ARGS__ : "vcc");                                                                                        \
    }
template <int TOPK, bool BIASED = false>
__device__ __forceinline__ void filter_hit(FilterRow<TOPK> &r, float a0, float a1, float a2, float a3, float mx, int cbg, const char *cbase,
                                           unsigned long &multi, float cc = 0.f, float bias = 0.f)
{
    unsigned long sv;
    unsigned j;
    float p, q, u;
    if constexpr (BIASED) {
        F_HIT_ASM("v_fma_f32 %[u], %[mx], %[cc], %[bias]\n\t", , [cc] "v"(cc), [bias] "s"(bias))
    } else {
        F_HIT_ASM("v_mul_f32 %[u], 0xb8000000, %[mx]\n\t")
    }
}
#undef F_HIT_ASM
#undef F_HIT_HEAD
#undef F_HIT_TAIL
#undef F_HIT_OUT
#undef F_HIT_IN

// The scan of one finished accumulator tile: four values per test (a 4-way max and ONE compare, wave-uniform branch); only a
// quad that holds a passing value in some lane -- rare per lane, but 64 lanes x 4 values at p ~ 3e-3 is half the quads --
// runs the hit sequence above.  cb = the lane's first code of this 32-code group.
// (Round 1 parked hits in LDS and flushed them once per 128 values; with 16 values per scan and row the parking only added
// LDS round trips -- each behind a drain of the DMA ring, see above.)
template <int TOPK, bool COUNT = false, bool BIASED = false>
__device__ __forceinline__ void filter_scan(FilterRow<TOPK> &r, const f32x16 &a, int cb, const char *cbase, unsigned long &multi, unsigned *n_hit = nullptr,
                                            float cc = 0.f, float bias = 0.f)
{
    // all four quad maxima and their tests first (the limit does not move inside a tile: filter_hit leaves L alone), then scalar
    // branches on masks that are long since in SGPRs: a compare followed at once by the branch on it costs the VALU's latency and,
    // with the common no-hit case as the taken branch, a refetch per quad
    float mx[4];
    unsigned long h[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) mx[g] = v_max(v_max3(a[4 * g], a[4 * g + 1], a[4 * g + 2]), a[4 * g + 3]);
#pragma unroll
    for (int g = 0; g < 4; ++g) h[g] = __builtin_amdgcn_ballot_w64(mx[g] >= r.L);
    if (__builtin_expect((h[0] | h[1] | h[2] | h[3]) != 0, 0)) {
        unsigned long long t0 = 0;
        if (COUNT) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (h[g]) {
                filter_hit<TOPK, BIASED>(r, a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3], mx[g], cb + 8 * g, cbase, multi, cc, bias);
                if (COUNT) ++n_hit[0];    // (dev probe: how many of the wave's quad tests run the hit sequence, and for how long)
            }
        if (COUNT) n_hit[1] += (unsigned)(__builtin_amdgcn_s_memtime() - t0);
    }
}

// Everything of a tile that passes and is NOT its quad's maximum (first position holding it) -- the revisit after `multi`.
// With INSERT = false: ALL passing values, appended but not folded into the k-list -- the warm-up tile, whose values the
// lists have seen already.  Plain divergent code: rare (multi) or once per block (warm-up).
template <int TOPK, bool INSERT, bool BIASED = false>
__device__ __forceinline__ void filter_scan_rest(FilterRow<TOPK> &r, const f32x16 &a, int cb, const char *cbase, float cc = 0.f, float bias = 0.f)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float a0 = a[4 * g], a1 = a[4 * g + 1], a2 = a[4 * g + 2], a3 = a[4 * g + 3];
        const float mx = v_max(v_max3(a0, a1, a2), a3);
        if (mx >= r.L) {
            const int jm = !INSERT ? -1 : (a0 == mx ? 0 : a1 == mx ? 1 : a2 == mx ? 2 : 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
                if (j != jm && v >= r.L) {
                    filter_append(r.pos, r.endm8, cbase, v, (unsigned)(cb + j + 8 * g));
                    if (INSERT) thr_insert_bf<TOPK>(r.tv, BIASED ? fmaf(v, cc, bias) : v * -0x1p-15f);
                }
            }
        }
    }
}

// acc >= L  <=>  u <= t + win.  t = +inf (fewer than k real codes seen): everything finite passes, but not the -inf of padded
// codes.  win = -inf marks a padding row: its limit stays +inf whatever t is (inf - inf would be NaN -> everything passes).
__device__ __forceinline__ float filter_limit(float t, float win)
{
    return win == -INFINITY ? INFINITY : fmaxf(-32768.0f * (t + win), -3.0e38f);
}

// After the scans: combine the four owners of each of the wave's two rows into one limit per row.
// The row's k-th best over ALL codes seen so far, exactly: merge the sorted k-lists of the row's four owners.  (The minimum of
// the owners' own k-th bests is only about the 4k-th best of the union: 2-3 x the candidates.)  Two sorted lists a, b:
// {min(a_i, b_{k-1-i})} are the k smallest of their union.  The lh partner comes by shuffle; the other code-side wave
// publishes its merged list in LDS -- possibly a few stages old, which is still a list of values of real codes, so T stays
// valid.  (Extending the union across code splits through agent-scope global lists was measured 10 % SLOWER.)
// Single-instruction min / max: the lists never hold NaN (the warm-up pass maps it to +inf), so the canonicalising v_max that
// fminf / fmaxf put in front of every operand is not needed.
template <int TOPK>
__device__ __forceinline__ void filter_merge_pair(FilterRow<TOPK> &r0, FilterRow<TOPK> &r1, int wm, int lh)
{
    constexpr int TL = TOPK < 5 ? TOPK : 5;       // list length shared per (row, code-side wave)
    float t0, t1;
    if (TOPK <= 5) {
        OtherLists ot;
        lds_load_lists_issue(r0.lst + (wm ^ 1) * 32, r1.lst + (wm ^ 1) * 32, ot);
        float c0[5], c1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { c0[i] = INFINITY; c1[i] = INFINITY; }
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            c0[i] = v_min(r0.tv[i], other_half(r0.tv[TL - 1 - i], lh));
            c1[i] = v_min(r1.tv[i], other_half(r1.tv[TL - 1 - i], lh));
        }
#pragma unroll
        for (int pass = 0; pass < TL; ++pass)              // odd-even transposition: ascending
#pragma unroll
            for (int i = pass & 1; i + 1 < TL; i += 2) {
                const float lo0 = v_min(c0[i], c0[i + 1]), hi0 = v_max(c0[i], c0[i + 1]);
                c0[i] = lo0; c0[i + 1] = hi0;
                const float lo1 = v_min(c1[i], c1[i + 1]), hi1 = v_max(c1[i], c1[i + 1]);
                c1[i] = lo1; c1[i + 1] = hi1;
            }
        if (lh == 0) {
            lds_store_list(r0.lst + wm * 32, c0);
            lds_store_list(r1.lst + wm * 32, c1);
        }
        lds_load_lists_wait(ot);
        const float o0[5] = {ot.q0[0], ot.q0[1], ot.q0[2], ot.q0[3], ot.e0}, o1[5] = {ot.q1[0], ot.q1[1], ot.q1[2], ot.q1[3], ot.e1};
        t0 = -INFINITY; t1 = -INFINITY;
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            t0 = v_max(t0, v_min(c0[i], o0[TL - 1 - i]));
            t1 = v_max(t1, v_min(c1[i], o1[TL - 1 - i]));
        }
    } else {                 // k = 8: the lists would not fit beside the ring; the pair's looser min rule
        t0 = v_min(r0.tv[TOPK - 1], other_half(r0.tv[TOPK - 1], lh));
        t1 = v_min(r1.tv[TOPK - 1], other_half(r1.tv[TOPK - 1], lh));
    }
    r0.L = filter_limit(t0, r0.win);
    r1.L = filter_limit(t1, r1.win);
}

// The same for ONE row (filter_rows64_kernel merges its two rows one after the other: half the temporaries live at a time, in a
// kernel that has no register to spare), on single-instruction min / max: the lists never hold NaN (the learning pass maps it to
// +inf), so the canonicalising v_max that fminf / fmaxf put in front of every operand is not needed.
template <int TOPK>
__device__ __forceinline__ void filter_merge_one(FilterRow<TOPK> &r, unsigned lst, int wm, int lh)
{
    constexpr int TL = TOPK < 5 ? TOPK : 5;
    float t;
    if (TOPK <= 5) {
        f32x4 oq;
        float oe;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %2 offset:16" : "=&v"(oq), "=&v"(oe) : "v"(lst + (unsigned)((wm ^ 1) * 32)) : "memory");
        float c[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) c[i] = INFINITY;
#pragma unroll
        for (int i = 0; i < TL; ++i) c[i] = v_min(r.tv[i], other_half(r.tv[TL - 1 - i], lh));
#pragma unroll
        for (int pass = 0; pass < TL; ++pass)              // odd-even transposition: ascending
#pragma unroll
            for (int i = pass & 1; i + 1 < TL; i += 2) {
                const float lo = v_min(c[i], c[i + 1]), hi = v_max(c[i], c[i + 1]);
                c[i] = lo; c[i + 1] = hi;
            }
        if (lh == 0) lds_store_list(lst + (unsigned)(wm * 32), c);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(oq), "+v"(oe)::"memory");
        const float o[5] = {oq[0], oq[1], oq[2], oq[3], oe};
        t = -INFINITY;
#pragma unroll
        for (int i = 0; i < TL; ++i) t = v_max(t, v_min(c[i], o[TL - 1 - i]));
    } else {
        t = v_min(r.tv[TOPK - 1], other_half(r.tv[TOPK - 1], lh));
    }
    r.L = filter_limit(t, r.win);
}

// accumulator start values of one 32-code group: -2^15 |e|^2 of this lane's 16 codes, both row tiles.  sp is wave-uniform ->
// scalar loads (lgkmcnt, not vmcnt): the ring's DMA queue is not drained.
__device__ __forceinline__ void filter_init_group(f32x16 &a0, f32x16 &a1, const float *__restrict__ sp, int lh)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        const float lo = sp[c], hi = sp[c + 4];
        const float v = lh ? hi : lo;
        a0[r] = v; a1[r] = v;
    }
}

// The same from LDS (the block keeps the start values of the next code tile there, see the kernel): four ds_read_b128 straight
// into the accumulator registers, no VALU.  asm for the reason given above lds_store_list; the caller waits (lds_init_wait).
__device__ __forceinline__ f32x16 lds_init_issue(unsigned addr)
{
    f32x4 q0, q1, q2, q3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr));
    const auto lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
// (only the first row tile's registers are loaded: see `step` in the kernel)
__device__ __forceinline__ void lds_init_wait(f32x16 (&a)[4][2])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[2][0]), "+v"(a[3][0]));
}

// ---------------------------------------------------------------- the filter kernel
// (Rejected variant, measured 14-24 % slower: 4-wave blocks of 256 codes x 128 rows, two per CU, hoping that co-resident
// blocks drifting apart overlap one block's epilogue / DMA issue with the other's MFMAs -- it moves 1.5 x the L2 -> LDS
// bytes per flop, and operand delivery is what binds this kernel.)
// Block = 8 waves (2 code-side x 4 row-side), tile 256 codes x 256 rows, wave tile 128 x 64 = 4 x 2 MFMA
// tiles of 32x32x16 (L2->LDS traffic per flop halves against a 128^2 tile; at fp16 rates that is what
// binds).  Operands arrive by LDS-DMA (buffer_load ... lds, 16 B/lane) into a 4-stage ring; a staged tile row is
// 64 B = 4 chunks, stored at chunk position c ^ ((row >> 2) & 3) so that the rows a ds_read_b128 lane group
// touches land on distinct 16-byte bank slots (the permutation is applied to the per-lane SOURCE address; the
// LDS image itself is lane-linear as the DMA requires).
// As in the fp32 kernel, codes are the A rows: a lane ends up with 16 codes of one input row per tile,
// so thresholds and candidate appends are lane-local.
//
// Scores in accumulator scale.  wsqs[c] = -2^15 |e_c|^2 (padded with -inf to a multiple of 256).  A code group's
// accumulators START at that value instead of 0, so that after the k loop   acc = 2^16 s~ - 2^15 |e|^2 = -2^15 u,
// u = |e|^2 - 2 s~ = d~ - |x|^2:  the epilogue tests  acc >= L  (L = -2^15 * limit, per lane) with no arithmetic per
// value.  The start values reach the lanes through LDS: wave 0 DMA-copies the 256 values of code tile t+2 (1 KB, one
// instruction) into one of two 1 KB buffers during the scan of tile t, and a group restarts with eight ds_read_b128 straight
// into its accumulator registers.  (A vector load here would wait for vmcnt(0), i.e. drain the whole LDS-DMA ring, every
// time; round 2's first form read them through the scalar cache and selected per half-wave -- 2 scalar loads with their
// latency plus 56 VALU instructions per group, measured 1.1 ms of a 15 ms launch, 40 % of the whole scan phase.)
//
// Staggered epilogues -- built twice, measured, removed.  The scan of a finished code tile is VALU work, and while both waves
// of a SIMD scan the matrix pipe idles (about 20 % of the kernel).  Because the k16 half steps of a dot product may be visited
// in any order and the x tile's k blocks recur, each 32-code group can switch code tiles in the middle of a different stage
// (the DMA then feeds the two halves of that stage's code rows from two code tiles), so that a SIMD's two waves never scan at
// the same time.  (a) Round 2, first form: the group's scan (the branchy one) in the middle of its switch stage: bit-identical,
// 11-13 % SLOWER (K = 16384: 17.5 vs 15.5 ms).  (b) Second form: the hand-off through the MFMA destination (last MFMAs of the
// tile write D != C, first MFMAs of the next read the start values as C: no copies) and a fully branch-free, exec-masked scan
// of 8 quads woven between the next two stages' MFMAs (vmcnt(12) for the stage wait): 17.95 ms.  Both for the same reason: the
// ring has ONE block-wide barrier per stage, so whatever one wave does beyond its MFMAs in a stage is added to that stage for
// all eight -- about +990 cycles per scanning stage for ~140 extra instructions, in 16 of 24 stages, against one exposed scan
// per tile.  What pays instead is making that one scan short (filter_hit above).
// TIMED (dev probe, tools/r04/filter_probe.py; the product instantiates TIMED = false): every wave accumulates, in shader-clock
// cycles (s_memtime), how long the five segments of its stage loop take -- MFMA group 1 (+ operand reads), the wait for its own
// and the block's copies (s_waitcnt, then s_barrier), MFMA group 2 (+ DMA issue), the tile epilogue -- and writes the sums to
// `dump` (as uint64[blocks][8 waves][8]).
template <int TOPK, bool DUMP, bool TIMED = false>
__global__ __launch_bounds__(F_THREADS, 2) void filter_f16_kernel(
    const _Float16 *__restrict__ xh, const _Float16 *__restrict__ wh, const float *__restrict__ xsq,
    const float *__restrict__ wsqs, const float *__restrict__ en_max_ptr, long n, int k_codes, int dp, int d,
    int codes_per_split, int own_total, uint2 *__restrict__ cand, int *__restrict__ cand_cnt,
    float *__restrict__ dump, int xcd_rows, int n_splits, int row_tile_base, int row_tile_end)
{
    static_assert(F_WM == 2 && F_MT == 4 && F_NT == 2, "the wave tile is 128 codes x 64 rows");
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);                   // provably wave-uniform copies
    const int wm = wave / F_WN, wn = wave % F_WN, wm_s = wave_s / F_WN;
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

    // ---- staging: wave w DMA-copies tile rows [32w, 32w+32) of A (= code group w of the tile) and of B, 16 rows (of
    // 64 B) per instruction.  The per-lane part of the source address is a loop-invariant 32-bit offset; everything that
    // moves (code tile, k block, the block's row base) is wave-uniform and stays in SGPRs.
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
    const int wave_lds = wave_s * 32 * F_ROWB;
    const int tile_bytes = F_BM * dp * 2;
    int pkb = 0, pidx = 0, ptile = 0;                   // next stage to issue: k block, linear index, code tile
    // Issues stage `pidx` into ring slot pidx % 4 and advances -- except past the end, where it re-issues the
    // LAST stage into the slot that already holds it (same bytes, harmless) so the steady-state loop body has
    // no branch around its DMA and one instruction schedule fits every iteration.
    // buffer-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds): SGPR descriptor + loop-invariant 32-bit lane offset +
    // SGPR stage offset -- no per-lane 64-bit address arithmetic per instruction (+3 % over global_load_lds here)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)xbase, 0, -1, 0x00020000);
    auto stage = [&]() __attribute__((always_inline)) {
        char *base = fsm + (pidx & (F_RING - 1)) * F_STAGEB + wave_lds;
        // readfirstlane: hipcc otherwise keeps the stage counters in VGPRs and wraps every buffer load in a waterfall loop
        const int ua = __builtin_amdgcn_readfirstlane(ptile * tile_bytes + pkb * F_BK * 2);    // a code split's fp16 image is < 2 GB
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
        pkb = more ? (wrap ? 0 : pkb + 1) : pkb;
        ptile += (more && wrap) ? 1 : 0;
    };

    // start values of code tile `tile` -> LDS buffer tile & 1 (wave 0; an ordinary member of its in-order DMA stream)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(wsqs + code_lo), 0, -1, 0x00020000);
    auto stage_init = [&](int tile) __attribute__((always_inline)) {
        if (wave_s == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void *)(fsm + F_LDS_BYTES + F_THR_BYTES + (tile & 1) * (F_BM * 4)),
                                                     16, lane * 16, __builtin_amdgcn_readfirstlane(min(tile, nct - 1) * (F_BM * 4)), 0, 0);
    };
    // the lane's 16 start values of group M: codes wm 128 + M 32 + 8 g + 4 lh + {0..3}, g = 0..3
    const unsigned init_adr = (unsigned)(size_t)(fsm + F_LDS_BYTES + F_THR_BYTES) + (unsigned)(wm * (32 * F_MT) + 4 * lh) * 4u;
#define F_INIT_LDS(M, tile) \
    do { acc[M][0] = lds_init_issue(init_adr + (unsigned)(((tile) & 1) * (F_BM * 4) + (M) * 128)); } while (0)

    // ---- per-lane state: one FilterRow for each of the wave's two 32-row column tiles
    float *thr_share = reinterpret_cast<float *>(fsm + F_LDS_BYTES);    // [F_BN][2 code-side waves][8]: sorted k-smallest lists (5 used)
    const float en_max = en_max_ptr[0];
    const bool sane = en_max <= F_NORM_LIMIT;
    const int owner = split * F_OWN_PER_SPLIT + wm * 2 + lh;
    // the block's candidate lists: wave-uniform base + per-lane 32-bit byte offsets (256 rows x own_total lists of F_CAP slots)
    const char *cbase = reinterpret_cast<const char *>(cand + row0 * own_total * F_CAP);
    FilterRow<TOPK> row[F_NT];
    unsigned list_start[F_NT];
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn) {
        const int rl = wn * (32 * F_NT) + nn * 32 + li;
        const long xr = row0 + rl;
#pragma unroll
        for (int j = 0; j < TOPK; ++j) row[nn].tv[j] = INFINITY;
        row[nn].L = INFINITY;                // nothing is appended before the warm-up pass has set a finite limit
        row[nn].xn = xsq[min(xr, n - 1)];
        row[nn].win = xr < n ? 2.0f * filter_eps(row[nn].xn, en_max, d) : -INFINITY;
        list_start[nn] = (unsigned)((rl * own_total + owner) * F_CAP) * 8u;
        row[nn].pos = list_start[nn];
        row[nn].endm8 = list_start[nn] + (F_CAP - 1) * 8u;
        row[nn].lst = (unsigned)(size_t)(fsm + F_LDS_BYTES) + (unsigned)rl * F_LST_ROWB;
    }
    for (int i = tid; i < (int)(F_THR_BYTES / 4); i += F_THREADS) thr_share[i] = INFINITY;

    f32x16 acc[F_MT][F_NT];
    // start values of group M for code tile `tile` (wave-uniform address: see filter_init_group)
#define F_INIT_GROUP(M, tile) \
    filter_init_group(acc[M][0], acc[M][1], wsqs + __builtin_amdgcn_readfirstlane(code_lo + (tile) * F_BM + wm_s * (32 * F_MT) + (M) * 32), lh)
    // the lane's first code of group M in code tile `tile`
#define F_LANE_CB(M, tile) (code_lo + (tile) * F_BM + wm * (32 * F_MT) + (M) * 32 + 4 * lh)

    // fragment addresses within a stage: row i of the tile, chunk (2t + lh) ^ ((i >> 2) & 3).  Rows 32 apart
    // share the swizzle term, so one address per (operand, t) plus compile-time row offsets covers all tiles.
    int a_adr[2], b_adr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ia = wm * (32 * F_MT) + li, ib = wn * (32 * F_NT) + li;
        a_adr[t] = ia * F_ROWB + (((2 * t + lh) ^ ((ia >> 2) & 3)) << 4);
        b_adr[t] = F_TILEB + ib * F_ROWB + (((2 * t + lh) ^ ((ib >> 2) & 3)) << 4);
    }

    // ---- software pipeline.  LDS ring of 4 stages; the ds_reads of the NEXT k16-step are in flight while the MFMAs of the
    // current one issue:
    //   iteration s:  MFMA(s, t0)  | read frags(s, t1)
    //                 vmcnt (own part of stage s+1 landed) -> raw barrier (everyone's has; slot s-1 is free)
    //                 LDS-DMA stage s+3  | MFMA(s, t1) | read frags(s+1, t0)
    // __syncthreads() would drain vmcnt(0) here (an LDS-DMA is a pending LDS write), hence the raw barrier.
    // Registers: the code-side fragment of group m is dead once its two MFMAs are issued, so the next step's fragment is read
    // into the SAME registers right behind them (6 MFMAs = 190+ cycles ahead of its first use); only the two row-side
    // fragments, which every MFMA of the step reads, are double-buffered: 32 operand registers instead of 48.
    half8 fa[F_MT], fbA[F_NT], fbB[F_NT];
    auto read_a = [&](int m, int slot, int t) __attribute__((always_inline)) {
        fa[m] = *reinterpret_cast<const half8 *>(fsm + slot * F_STAGEB + a_adr[t] + m * 32 * F_ROWB);
    };
    auto read_b = [&](half8 (&fb)[F_NT], int slot, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn) fb[nn] = *reinterpret_cast<const half8 *>(fsm + slot * F_STAGEB + b_adr[t] + nn * 32 * F_ROWB);
    };
    // one k16 step on (fa, fb_cur); meanwhile the operands of step (slot, t) arrive in fa / fb_nxt.
    // FIRST = the first k16 step of a code tile: only the FIRST row tile's accumulators hold the start values; the second row
    // tile's MFMA reads them from there as its C operand (D != C), then the first row tile's MFMA runs in place.  A restart
    // therefore loads 16 registers per group, not 32, and copies nothing.
    auto step = [&](const half8 (&fb_cur)[F_NT], half8 (&fb_nxt)[F_NT], int slot, int t, auto first) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        read_b(fb_nxt, slot, t);
#pragma unroll
        for (int m = 0; m < F_MT; ++m) {
            if (FIRST) {
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb_cur[1], acc[m][0], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb_cur[0], acc[m][0], 0, 0, 0);
            } else {
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb_cur[nn], acc[m][nn], 0, 0, 0);
            }
            read_a(m, slot, t);
        }
    };

    // The scan of a finished code tile: all four groups of the wave.
    unsigned long multi = 0;             // wave-uniform: some lane had a second passing value in one quad (filter_hit)
    auto tile_epilogue = [&](int tile, bool warm) __attribute__((always_inline)) {
        // (the scan reads the accumulators with inline-asm v_max3, which hipcc's hazard recognizer does not protect: the 18 wait
        // states a 16-pass XDL result needs before a VALU read are held here, once per code tile, by a statement that owns all
        // eight tiles -- see filter_rows64_kernel, where a reordered asm consumer did read half-finished sums)
        asm volatile("s_nop 15\n\ts_nop 1"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        if (DUMP) {
#pragma unroll
            for (int m = 0; m < F_MT; ++m)
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int code = F_LANE_CB(m, tile) + (r & 3) + 8 * (r >> 2);
                        const long xr = row0 + wn * (32 * F_NT) + nn * 32 + li;
                        // s~ as the search sees it: (|e|^2 - u) / 2 with u = -2^-15 acc  (includes the rounding of the non-zero start)
                        if (code < code_hi && xr < n) dump[xr * k_codes + code] = fmaf(acc[m][nn][r], F_UNSCALE, wsqs[code] * -F_UNSCALE);
                    }
            if (tile + 1 < nct) { F_INIT_GROUP(0, tile + 1); F_INIT_GROUP(1, tile + 1); F_INIT_GROUP(2, tile + 1); F_INIT_GROUP(3, tile + 1); }
            return;
        }
        if (warm) {
            // First code tile: learn the thresholds from all 128 codes BEFORE appending anything, so the
            // candidate lists do not fill up with the loose early threshold (appends only ever need T >= t~).
#pragma unroll
            for (int m = 0; m < F_MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int nn = 0; nn < F_NT; ++nn) {
                        const float u = acc[m][nn][r] * -0x1p-15f;
                        thr_insert_med3<TOPK>(row[nn].tv, u == u ? u : INFINITY);
                    }
#pragma unroll
            for (int nn = 0; nn < F_NT; ++nn) row[nn].L = filter_limit(row[nn].tv[TOPK - 1], row[nn].win);
        }
#define F_ONE(M)                                                                                                   \
        do {                                                                                                       \
            const int cb_ = F_LANE_CB(M, tile);                                                                    \
            _Pragma("unroll") for (int nn = 0; nn < F_NT; ++nn) {                                                  \
                if (warm) filter_scan_rest<TOPK, false>(row[nn], acc[M][nn], cb_, cbase);                          \
                else {                                                                                             \
                    filter_scan<TOPK>(row[nn], acc[M][nn], cb_, cbase, multi);                                     \
                    if (multi) { filter_scan_rest<TOPK, true>(row[nn], acc[M][nn], cb_, cbase); multi = 0; }       \
                }                                                                                                  \
            }                                                                                                      \
            F_INIT_LDS(M, tile + 1);        /* (past the last tile: values of a clamped tile, never scanned) */     \
        } while (0)
        F_ONE(0); F_ONE(1); F_ONE(2); F_ONE(3);
#undef F_ONE
        stage_init(tile + 2);               // into the buffer this tile's start values came from (last read a whole tile ago)
        filter_merge_pair<TOPK>(row[0], row[1], wm, lh);
        lds_init_wait(acc);
    };

    constexpr int LGKM0 = 0xC07F;           // s_waitcnt lgkmcnt(0) only (vmcnt / expcnt fields at their maxima)
    {   // start values of the first code tile by vector loads (nothing is in flight yet, so the wait they need is harmless;
        // four groups through the scalar cache at once would cost 128 SGPRs)
#pragma unroll
        for (int m = 0; m < F_MT; ++m) {
            float4 e4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e4[g] = ld4(wsqs + F_LANE_CB(m, 0) + 8 * g);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = e4[r >> 2];
                const float v = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
                acc[m][0][r] = v;
            }
        }
    }
    stage_init(1);                          // (older than everything the wait below leaves in flight)
    stage(); stage(); stage();              // stages 0..2 (clamped when the block has fewer)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(fbA, 0, 0);
#pragma unroll
    for (int m = 0; m < F_MT; ++m) read_a(m, 0, 0);
    // The two waves of a SIMD (w and w + 4) issue their LDS-DMA at different points of the stage: an issuing wave is
    // held for ~100 cycles per instruction, and in lockstep both would leave the matrix pipe idle at the same time
    // (+2.5-3 % measured).  Either way a wave has issued all of stage s+3 between the waits of iterations s and s+1,
    // so the counted vmcnt below is the same for both halves.
    const bool late = wave_s >= 4;
    // The second-dispatched half of the block loses issue arbitration to the older half at every segment start
    // (priority, then age): one static priority bump for it, no flips inside the loop (+2 %, levels 1 and 3 alike;
    // -1.5 % when given to the older half instead; flips around the MFMA groups measured -1 %).
    if (late) __builtin_amdgcn_s_setprio(3);

    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tq = 0;       // TIMED: mfma 1, copy wait, barrier, mfma 2, epilogue, stages
    auto tick = [&](int k) __attribute__((always_inline)) {
        if (TIMED) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm[k] += now - tq;
            tq = now;
        }
    };
    // first half of an iteration: MFMA(s, t0) with the operand reads of (s, t1) between them, then the stage barrier
    auto first_half = [&](int s, auto first) __attribute__((always_inline)) {
        if (late && s > 0) stage();         // waves 4-7: stage s+2 (slot s-2, free since the barrier of iteration s-1)
        step(fbA, fbB, s & (F_RING - 1), 1, first);
        __builtin_amdgcn_sched_group_barrier(0x100, F_NT, 0);
#pragma unroll
        for (int i = 0; i < F_MT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, F_NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        // the next step's operands have landed; own part of stage s+1 has landed; then everyone's has
        __builtin_amdgcn_s_waitcnt(LGKM0);
        tick(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        tick(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tick(2);
    };
    // second half: (DMA of stage s+3,) MFMA(s, t1) with the operand reads of (s+1, t0) between them
    // (past the last stage the reads fetch stale LDS, never used)
    auto second_half = [&](int s) __attribute__((always_inline)) {
        step(fbB, fbA, (s + 1) & (F_RING - 1), 0, std::false_type{});
        // after the barrier the matrix pipe restarts at once; DMA issue and operand reads ride between MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, F_NT, 0);
#pragma unroll
        for (int i = 0; i < F_MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, F_NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };

    // every group switches code tiles together: the scan of all four runs at the end of the tile's last stage
    auto iteration = [&](int s, auto first) __attribute__((always_inline)) {
        first_half(s, first);
        if (!late) stage();             // waves 0-3: stage s+3 (slot s-1: everyone is past reading it)
        second_half(s);
    };
    int s = 0;
    if (TIMED) tq = __builtin_amdgcn_s_memtime();
    for (int ct = 0; ct < nct; ++ct) {
        iteration(s, std::true_type{});
        tick(3);
        ++s;
        for (int kb = 1; kb < nkb; ++kb, ++s) { iteration(s, std::false_type{}); tick(3); }
        tile_epilogue(ct, ct == 0);
        tick(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail re-issues may still be in flight
    if (TIMED) {
        if (lane == 0 && dump) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(dump) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
            o[0] = tm[0]; o[1] = tm[1]; o[2] = tm[2]; o[3] = tm[3]; o[4] = tm[4]; o[5] = (unsigned long long)nstage; o[6] = (unsigned long long)nct;
            o[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
    if (!DUMP) {
#pragma unroll
        for (int nn = 0; nn < F_NT; ++nn) {
            // rows outside the range the bound assumes are forced onto the exact path
            const long xr = row0 + wn * (32 * F_NT) + nn * 32 + li;
            const bool ok = sane && row[nn].xn <= F_NORM_LIMIT;
            if (xr < n) cand_cnt[xr * own_total + owner] = ok ? (int)((row[nn].pos - list_start[nn]) >> 3) : F_CAP + 1;
        }
    }
#undef F_INIT_GROUP
#undef F_INIT_LDS
#undef F_LANE_CB
}

// ---------------------------------------------------------------- the filter kernel for rows of at most 64 elements
// The reference's own shape is e_dim = 64 (train_MedTok.py: --embed_dim 64, n_e = 21 000).  There the kernel above spends its time
// around the matrix work, not in it: a code tile is two 32-deep stages, so the x tile's k blocks are copied again for every code
// tile although the whole 256 x 64 tile is 32 KB, every stage pays a block-wide barrier, and the scan of a finished tile -- VALU
// work of about the length of two tiles' MFMAs per wave -- stops all eight waves of the CU's only block at the same time
// (profiles/r04_kernel_stats_refdefault.csv: 0.155 of the f16 peak).  This form:
//   * a row is ONE 128-byte line.  The block's x rows live in REGISTERS for its whole life (a wave's 64 rows x 64 k = 8 B
//     fragments = 32 registers); only code tiles move: 256 codes = 32 KB per tile, LDS-DMA of whole lines (8 rows per
//     instruction), two ring slots, ONE barrier per code tile (tile t + 1 is copied during tile t's MFMAs and scan);
//   * blocks of FOUR waves (2 code-side x 2 row-side, tile 256 codes x 128 rows, the same 128 x 64 wave tile and the same
//     per-lane state as above), 74 KB of LDS: two blocks per CU that drift apart, so that one block's scan (VALU) runs beside
//     the other's MFMAs on every SIMD -- the overlap the staggered-epilogue experiments above could not get inside one block.
//     (At D = 768 this block shape lost to the 8-wave one because it moves 1.5 x the L2 -> LDS bytes per flop; with the rows
//     in registers it moves the same.)
//   * the scan is what this kernel does most of the time (a tile's 32 MFMAs are ~850 cycles, its scan several thousand: every
//     instruction of a wave, scalar ones and no-ops included, takes a 4-cycle issue slot), and the first tiles are the dear ones:
//     with limits learnt from 256 codes, tile t still has ~160 / t lanes per wave passing, each one a 30-instruction hit
//     sequence.  So the block LEARNS first: for its first R64_LEARN tiles it only keeps, per lane and 16-code accumulator tile,
//     the best score (8 x v_max3) and folds that into the lane's k-list -- the k smallest of those per-tile bests are scores
//     of k DIFFERENT codes, so their largest is a valid limit, and it is within one rank of the k-th best of all 2048 codes seen
//     -- appends nothing, and comes back to those tiles at the END (32 MFMAs each, again) to collect their candidates with the
//     final limits, appending without inserting (the lists have seen these codes: a second insertion of the same code would
//     make the k-th entry smaller than the k-th best).  Tiles in between: the scan of filter_f16_kernel.
// Scores, thresholds, candidate lists and their owners (split x 4 + code-side wave x 2 + half-wave) are those of
// filter_f16_kernel: the re-score kernels do not know which one ran.
constexpr int R64_BM = 256, R64_BN = 128, R64_THREADS = 256;
constexpr int R64_ROWB = 128;                                   // bytes of a staged code row: all 64 halves
constexpr int R64_TILEB = R64_BM * R64_ROWB;                        // 32 KB per code tile
constexpr size_t R64_RING_BYTES = 2 * (size_t)R64_TILEB;
constexpr size_t R64_THR_BYTES = R64_BN * 64;                     // per row: the k-lists of the two code-side waves
constexpr size_t R64_INIT_BYTES = 2 * R64_BM * 4;                 // accumulator start values of two code tiles
constexpr size_t R64_SMEM_BYTES = R64_RING_BYTES + R64_THR_BYTES + R64_INIT_BYTES;      // 74 KB -> two blocks per CU
constexpr int R64_LEARN = 8;                                  // code tiles a block only learns its limits from (and revisits last)

// TIMED (dev probe, tools/r04/filter_probe.py): per-wave s_memtime sums of the loop's segments and the number of hit sequences run,
// written to `probe` (uint64 [blocks][4 waves][8]).
template <int TOPK, bool TIMED = false>
__global__ __launch_bounds__(R64_THREADS, 2) void filter_rows64_kernel(
    const _Float16 *__restrict__ xh, const _Float16 *__restrict__ wh, const float *__restrict__ xsq,
    const float *__restrict__ wsqs, const float *__restrict__ en_max_ptr, long n, int k_codes, int d,
    int codes_per_split, int own_total, uint2 *__restrict__ cand, int *__restrict__ cand_cnt, unsigned long long *__restrict__ probe = nullptr)
{
    static_assert(F_MT == 4 && F_NT == 2, "the wave tile is 128 codes x 64 rows");
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1, wm_s = wave_s >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const long row0 = (long)blockIdx.x * R64_BN;
    const int split = blockIdx.y;
    const int code_lo = split * codes_per_split;
    const int code_hi = min(k_codes, code_lo + codes_per_split);
    const int nct = (code_hi - code_lo + R64_BM - 1) / R64_BM;
    (void)wm_s;

    // ---- the wave's x rows as B operands of all four k16 steps (the fp16 image is padded to a multiple of 256 rows)
    half8 xf[F_NT][4];
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            xf[nn][t] = *reinterpret_cast<const half8 *>(xh + (row0 + wn * 64 + nn * 32 + li) * 64 + 16 * t + 8 * lh);

    // ---- code tiles by LDS-DMA: wave w copies tile rows [64 w, 64 w + 64), 8 rows (of 128 B) per instruction; LDS chunk position
    // p of row r holds source chunk p ^ (r & 7) (the permutation is on the per-lane source address: the image is lane-linear)
    const unsigned lane_off = (unsigned)((wave * 64 + (lane >> 3)) * 64 + (((lane & 7) ^ (lane >> 3)) << 3)) * 2u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(wh + (long)code_lo * 64), 0, -1, 0x00020000);
    // step s of the block works on code tile tile_of(s): the learning tiles 0 .. W-1, the rest, tiles 0 .. W-1 again
    const int W = min(R64_LEARN, nct), nsteps = nct + W;
    auto tile_of = [&](int st) __attribute__((always_inline)) -> int {
        st = min(st, nsteps - 1);
        return __builtin_amdgcn_readfirstlane(st < nct ? st : st - nct);
    };
    // (piece q of a tile's copy: the first tile's pieces go out together, later ones ride between the MFMAs of the tile before)
    auto stage_piece = [&](int st, int so, int q) __attribute__((always_inline)) {
        char *base = fsm + (st & 1) * R64_TILEB + wave_s * (64 * R64_ROWB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(base + q * 8 * R64_ROWB), 16, (int)lane_off,
                                                 so + q * 8 * R64_ROWB, 0, 0);
    };
    auto stage = [&](int st) __attribute__((always_inline)) {
        const int so = tile_of(st) * R64_TILEB;
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_piece(st, so, q);
    };
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(wsqs + code_lo), 0, -1, 0x00020000);
    auto stage_init = [&](int st) __attribute__((always_inline)) {
        const int so = tile_of(st) * (R64_BM * 4);      // (a local: hipcc's host pass rejects the lambda call inside the builtin's argument list)
        if (wave_s == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void *)(fsm + R64_RING_BYTES + R64_THR_BYTES + (st & 1) * (R64_BM * 4)),
                                                     16, lane * 16, so, 0, 0);
    };
    const unsigned init_adr = (unsigned)(size_t)(fsm + R64_RING_BYTES + R64_THR_BYTES) + (unsigned)(wm * (32 * F_MT) + 4 * lh) * 4u;
#define R64_INIT_LDS(M, st) \
    do { acc[M][0] = lds_init_issue(init_adr + (unsigned)(((st) & 1) * (R64_BM * 4) + (M) * 128)); } while (0)
#define R64_LANE_CB(M, tile) (code_lo + (tile) * R64_BM + wm * (32 * F_MT) + (M) * 32 + 4 * lh)

    // ---- per-lane state (as filter_f16_kernel)
    float *thr_share = reinterpret_cast<float *>(fsm + R64_RING_BYTES);
    const float en_max = en_max_ptr[0];
    const bool sane = en_max <= F_NORM_LIMIT;
    const int owner = split * F_OWN_PER_SPLIT + wm * 2 + lh;
    const char *cbase = reinterpret_cast<const char *>(cand + row0 * own_total * F_CAP);
    FilterRow<TOPK> row[F_NT];
    const unsigned lst0 = (unsigned)(size_t)(fsm + R64_RING_BYTES) + (unsigned)(wn * (32 * F_NT) + li) * F_LST_ROWB;     // (the second row's: + 32 rows)
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn) {
        const int rl = wn * (32 * F_NT) + nn * 32 + li;
        const long xr = row0 + rl;
#pragma unroll
        for (int j = 0; j < TOPK; ++j) row[nn].tv[j] = INFINITY;
        row[nn].L = INFINITY;
        // (per-lane values that are only needed once more -- the row's norm, the start of its list -- are formed again at the end)
        row[nn].xn = 0.f;
        row[nn].win = xr < n ? 2.0f * filter_eps(xsq[min(xr, n - 1)], en_max, d) : -INFINITY;
        row[nn].pos = (unsigned)((rl * own_total + owner) * F_CAP) * 8u;
        row[nn].endm8 = row[nn].pos + (F_CAP - 1) * 8u;
        row[nn].lst = 0;
    }
    for (int i = tid; i < (int)(R64_THR_BYTES / 4); i += R64_THREADS) thr_share[i] = INFINITY;

    f32x16 acc[F_MT][F_NT];
    {   // start values of the first code tile by vector loads (nothing is in flight yet)
#pragma unroll
        for (int m = 0; m < F_MT; ++m) {
            float4 e4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e4[g] = ld4(wsqs + R64_LANE_CB(m, 0) + 8 * g);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = e4[r >> 2];
                acc[m][0][r] = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (x fragments, norms and start values in registers before the first DMA)
    stage_init(1);
    stage(0);

    // A fragments: row i = wm 128 + m 32 + li of the tile, source chunk 2 t + lh at position (2 t + lh) ^ (i & 7); rows 32 apart
    // share the swizzle term
    unsigned a_adr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        a_adr[t] = (unsigned)(size_t)fsm + (unsigned)((wm * 128 + li) * R64_ROWB + (((2 * t + lh) ^ (li & 7)) << 4));

    // the 32 MFMAs of a code tile: 16 (k16 step, code group) pairs; the A operand of pair i + 2 is read while pair i's two MFMAs
    // issue -- three rotating 4-register fragments, not two sets of four: the registers are what this kernel is short of (asm reads
    // with counted waits: LDS reads return in order, and a C++ ds_read would be ordered behind the pending LDS-DMA of the next tile)
    auto mfma_tile = [&](int st) __attribute__((always_inline)) {
        const unsigned so = (unsigned)((st & 1) * R64_TILEB);
        const int dma_so = tile_of(st + 1) * R64_TILEB;
        unsigned adr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) adr[t] = a_adr[t] + so;
        fu32x4 fa[3];
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(fa[0]), "=&v"(fa[1]) : "v"(adr[0]) : "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = i >> 2, m = i & 3;
            fu32x4 &cur = fa[i % 3];
            if (i + 2 < 16) {
                fu32x4 &nxt = fa[(i + 2) % 3];
                const int t2 = (i + 2) >> 2, m2 = (i + 2) & 3;
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\ts_waitcnt lgkmcnt(2)" : "=&v"(nxt), "+v"(cur) : "v"(adr[t2]), "i"(m2 * 4096) : "memory");
            } else if (i + 1 < 16) {
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur) : : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur) : : "memory");
            }
            const half8 av = __builtin_bit_cast(half8, cur);
            if (t == 0) {       // only the first row tile's registers hold the start values (D != C for the second)
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, xf[1][0], acc[m][0], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, xf[0][0], acc[m][0], 0, 0, 0);
            } else {
#pragma unroll
                for (int nn = 0; nn < F_NT; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, xf[nn][t], acc[m][nn], 0, 0, 0);
            }
            // the next tile's copy, one piece behind every second MFMA pair: an LDS-DMA instruction holds its wave for ~50 cycles,
            // which here pass under the two MFMAs just issued (64 cycles of the pipe) instead of in front of the tile's first one
            if (i & 1) stage_piece(st + 1, dma_so, i >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    unsigned long multi = 0;
    unsigned n_hit[2] = {0, 0};
    float cc = -0x1p-15f;                   // (in a VGPR: the hit sequence's fma takes its one scalar operand for the bias)
    asm volatile("" : "+v"(cc));
    // learning step: the best of each accumulator tile into the lane's list; no appends.  The limits are set by the merges of
    // the last two learning steps (the second one sees the other code-side wave's list of the first).
    auto learn_epilogue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < F_MT; ++m) {
#pragma unroll
            for (int nn = 0; nn < F_NT; ++nn) {
                const f32x16 &a = acc[m][nn];
                float mx = v_max3(a[0], a[1], a[2]);
#pragma unroll
                for (int r = 3; r + 1 < 16; r += 2) mx = v_max3(mx, a[r], a[r + 1]);
                mx = v_max(mx, a[15]);
                const float u = mx * -0x1p-15f;
                thr_insert_med3<TOPK>(row[nn].tv, u == u ? u : INFINITY);
            }
            R64_INIT_LDS(m, st + 1);
        }
        if (st + 2 >= W) {
            filter_merge_one<TOPK>(row[0], lst0, wm, lh);
            filter_merge_one<TOPK>(row[1], lst0 + 32 * F_LST_ROWB, wm, lh);
        }
        lds_init_wait(acc);
    };
    // scanning step: bias = 0 -> append and insert (tiles W .. nct-1); bias = +inf -> append only (the learning tiles, revisited)
    auto scan_epilogue = [&](int st, float bias) __attribute__((always_inline)) {
        const int tile = tile_of(st);
#define R64_ONE(M)                                                                                                           \
        do {                                                                                                                 \
            const int cb_ = R64_LANE_CB(M, tile);                                                                            \
            _Pragma("unroll") for (int nn = 0; nn < F_NT; ++nn) {                                                            \
                filter_scan<TOPK, TIMED, true>(row[nn], acc[M][nn], cb_, cbase, multi, n_hit, cc, bias);                      \
                if (multi) { filter_scan_rest<TOPK, true, true>(row[nn], acc[M][nn], cb_, cbase, cc, bias); multi = 0; }     \
            }                                                                                                                \
            R64_INIT_LDS(M, st + 1);                                                                                         \
        } while (0)
        R64_ONE(0); R64_ONE(1); R64_ONE(2); R64_ONE(3);
#undef R64_ONE
        // the limits are recomputed from the lists after each of the first four scanned tiles, then after every fourth and after
        // the last one: a limit that is a few tiles old is still the k-th best of real codes (valid, slightly looser)
        const int since = st - W;
                if (st < nct && (since < 4 || (since & 3) == 3 || st == nct - 1)) {
            filter_merge_one<TOPK>(row[0], lst0, wm, lh);
            filter_merge_one<TOPK>(row[1], lst0 + 32 * F_LST_ROWB, wm, lh);
        }
        lds_init_wait(acc);
    };

    unsigned long long tm[5] = {0, 0, 0, 0, 0}, tq = 0;        // TIMED: copy wait, barrier, DMA issue, MFMAs, epilogue
    auto tick = [&](int k) __attribute__((always_inline)) {
        if (TIMED) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm[k] += now - tq;
            tq = now;
        }
    };
    if (TIMED) tq = __builtin_amdgcn_s_memtime();
    // one step up to its epilogue: this wave's share of step t's tile (and wave 0's start values of step t + 1) has landed; then
    // everyone's has, and everyone is past the MFMAs of step t - 1 (the other ring slot) and past the start-value reads of step t
    // (buffer t & 1); the next tile's copy goes out, then the MFMAs.  (Two loops, not one with a branch on the phase: with both
    // epilogues behind one loop head hipcc spilled the x fragments.)
    auto step_head = [&](int t) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tick(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tick(1);
        stage_init(t + 2);
        tick(2);
        mfma_tile(t);                   // (with the copy of step t + 1's tile woven in; past the end: the last tile again, into the slot nobody reads any more)
        // The epilogues read the accumulators with inline-asm v_max3: hipcc's hazard recognizer does not count the wait states an
        // XDL result needs before a VALU read for asm consumers (18 for a 16-pass MFMA), and its scheduler is free to move such a
        // statement up to right behind the MFMA that feeds it.  Measured: with the learning epilogue, 335 of 600 000 rows lost
        // their k-th code to limits computed from half-finished sums.  One statement that owns all eight tiles and holds the
        // wait states: nothing reads an accumulator before it, nothing after it is early.
        asm volatile("s_nop 15\n\ts_nop 1"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        tick(3);
    };
    int t = 0;
    for (; t < W; ++t) {
        step_head(t);
        learn_epilogue(t);
        tick(4);
    }
    for (; t < nsteps; ++t) {
        step_head(t);
        scan_epilogue(t, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(t < nct ? 0 : 0x7f800000)));
        tick(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TIMED && probe && lane == 0) {
        unsigned long long *o = probe + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        o[0] = tm[0]; o[1] = tm[1]; o[2] = tm[2]; o[3] = tm[3]; o[4] = tm[4]; o[5] = (unsigned long long)nsteps; o[6] = n_hit[0] | ((unsigned long long)n_hit[1] << 32);
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#pragma unroll
    for (int nn = 0; nn < F_NT; ++nn) {
        const long xr = row0 + wn * (32 * F_NT) + nn * 32 + li;
        const bool ok = sane && xsq[min(xr, n - 1)] <= F_NORM_LIMIT;
        const unsigned first = row[nn].endm8 - (F_CAP - 1) * 8u;
        if (xr < n) cand_cnt[xr * own_total + owner] = ok ? (int)((row[nn].pos - first) >> 3) : F_CAP + 1;
    }
#undef R64_INIT_LDS
#undef R64_LANE_CB
}

// ---------------------------------------------------------------- the same with 32-row wave tiles: three blocks per CU
// Where filter_rows64_kernel stands (profiles/r05_valu_refdefault.txt, counters per launch): the VALU is busy 45 % of the SIMD
// cycles, the matrix pipe 33 %, and a wave has an instruction in flight during 42 % of its cycles -- neither pipe binds; what binds
// is that a SIMD has TWO waves to cover the serial pieces of each other's tile (32 dependent-issue MFMAs, then a scan full of
// compare -> scalar branch chains, then a block-wide barrier), and the 128-register accumulator tile is what limits it to two.
// This form halves the wave tile instead: 128 codes x 32 rows = 4 accumulator tiles (64 registers), ONE row per lane, 16 registers
// of x fragments -- under 168 registers, three waves per SIMD.  Blocks of four waves, all row-side (tile 128 codes x 128 rows: the
// same L2 -> LDS bytes per flop as the 256 x 128 tile of two code-side waves), code tiles of 128 codes = 16 KB, LDS 33 KB -> three
// blocks per CU, each with its own barrier.  A wave reads the whole code tile as A operands (one ds_read_b128 per MFMA, twice the
// LDS traffic per flop: 1 KB per 32 matrix-pipe cycles and SIMD, half the LDS's rate if all four SIMDs are in their MFMA phase
// at once).  With one code-side wave a row's lists are complete inside the wave: the limit comes from the two half-waves' k-lists
// by one permlane exchange (no LDS lists, no sort), and a row has two candidate lists per split, not four.
// Scores, thresholds and list format are those of the kernels above; the learning phase covers 1024 codes.
// K = 21 000 / 7 000, 600 000 rows: 1.94 / 0.80 ms against 2.16 / 0.97 (0.33 / 0.27 of the f16 peak against 0.30 / 0.22).  Counters of
// this form: VALU 51 %, matrix pipe 45 %, both at once 17 % of the SIMD cycles, 3.4 waves per SIMD.  Two further forms, both correct,
// both slower (tools/r04/ab_rows64.py, one box): (a) 64 codes x 64 rows per wave in 2 x 2 blocks (an A fragment feeds two MFMAs
// again, 167 registers, three waves per SIMD, list exchange through LDS): 2.00 / 0.86 ms -- less LDS traffic, but the fourth wave
// hides more than that saves; (b) this kernel with the scan of code group m - 1 issued between the MFMAs of group m (code-group-
// major MFMA order, counted waits that allow for the start-value reads in between): 2.12 / 0.87 ms -- as in filter_f16_kernel,
// work woven into a wave's own MFMA phase loses to what the SIMD's other waves do with those issue slots by themselves.
constexpr int R64N_BM = 128, R64N_BN = 128, R64N_THREADS = 256;
constexpr int R64N_TILEB = R64N_BM * R64_ROWB;                      // 16 KB per code tile
constexpr size_t R64N_RING_BYTES = 2 * (size_t)R64N_TILEB;
constexpr size_t R64N_SMEM_BYTES = R64N_RING_BYTES + 2 * R64N_BM * 4;      // + start values of two code tiles: 33 KB
constexpr int R64N_LEARN = 8;                                      // (1024 codes; 16 / 12 / 4 tiles measured 2-4 % slower)
constexpr int R64N_OWN_PER_SPLIT = 2;                              // candidate lists per row and split: the two half-waves

// the k-th best of a row over the codes both half-waves have seen: {min(a_i, b_{k-1-i})} are the k smallest of the union of two
// sorted lists, their maximum is the k-th
template <int TOPK>
__device__ __forceinline__ void filter_merge_halves(FilterRow<TOPK> &r, int lh)
{
    float t = -INFINITY;
#pragma unroll
    for (int i = 0; i < TOPK; ++i) t = v_max(t, v_min(r.tv[i], other_half(r.tv[TOPK - 1 - i], lh)));
    r.L = filter_limit(t, r.win);
}

template <int TOPK>
__global__ __launch_bounds__(R64N_THREADS, 3) void filter_rows64n_kernel(
    const _Float16 *__restrict__ xh, const _Float16 *__restrict__ wh, const float *__restrict__ xsq,
    const float *__restrict__ wsqs, const float *__restrict__ en_max_ptr, long n, int k_codes, int d,
    int codes_per_split, int own_total, uint2 *__restrict__ cand, int *__restrict__ cand_cnt)
{
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 31, lh = lane >> 5;
    const long row0 = (long)blockIdx.x * R64N_BN;
    const int split = blockIdx.y;
    const int code_lo = split * codes_per_split;
    const int code_hi = min(k_codes, code_lo + codes_per_split);
    const int nct = (code_hi - code_lo + R64N_BM - 1) / R64N_BM;

    // ---- the wave's 32 x rows as B operands of the four k16 steps (the fp16 image is padded to a multiple of 256 rows)
    half8 xf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const half8 *>(xh + (row0 + wave * 32 + li) * 64 + 16 * t + 8 * lh);

    // ---- code tiles by LDS-DMA: wave w copies tile rows [32 w, 32 w + 32), 8 rows per instruction, chunk swizzle as above
    const unsigned lane_off = (unsigned)((wave * 32 + (lane >> 3)) * 64 + (((lane & 7) ^ (lane >> 3)) << 3)) * 2u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(wh + (long)code_lo * 64), 0, -1, 0x00020000);
    const int W = min(R64N_LEARN, nct), nsteps = nct + W;
    auto tile_of = [&](int st) __attribute__((always_inline)) -> int {
        st = min(st, nsteps - 1);
        return __builtin_amdgcn_readfirstlane(st < nct ? st : st - nct);
    };
    auto stage_piece = [&](int st, int so, int q) __attribute__((always_inline)) {
        char *base = fsm + (st & 1) * R64N_TILEB + wave_s * (32 * R64_ROWB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(base + q * 8 * R64_ROWB), 16, (int)lane_off,
                                                 so + q * 8 * R64_ROWB, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(wsqs + code_lo), 0, -1, 0x00020000);
    auto stage_init = [&](int st) __attribute__((always_inline)) {
        const int so = tile_of(st) * (R64N_BM * 4);
        if (wave_s == 0 && lane < 32)          // 128 start values = 512 bytes: half a wavefront of 16-byte lanes
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void *)(fsm + R64N_RING_BYTES + (st & 1) * (R64N_BM * 4)),
                                                     16, lane * 16, so, 0, 0);
    };
    const unsigned init_adr = (unsigned)(size_t)(fsm + R64N_RING_BYTES) + (unsigned)(4 * lh) * 4u;
#define R64N_INIT_LDS(M, st) \
    do { acc[M] = lds_init_issue(init_adr + (unsigned)(((st) & 1) * (R64N_BM * 4) + (M) * 128)); } while (0)
#define R64N_LANE_CB(M, tile) (code_lo + (tile) * R64N_BM + (M) * 32 + 4 * lh)

    // ---- per-lane state
    const float en_max = en_max_ptr[0];
    const bool sane = en_max <= F_NORM_LIMIT;
    const int owner = split * R64N_OWN_PER_SPLIT + lh;
    const char *cbase = reinterpret_cast<const char *>(cand + row0 * own_total * F_CAP);
    FilterRow<TOPK> row;
    const int rl = wave * 32 + li;
    const long xr = row0 + rl;
#pragma unroll
    for (int j = 0; j < TOPK; ++j) row.tv[j] = INFINITY;
    row.L = INFINITY;
    row.xn = 0.f;
    row.win = xr < n ? 2.0f * filter_eps(xsq[min(xr, n - 1)], en_max, d) : -INFINITY;
    row.pos = (unsigned)((rl * own_total + owner) * F_CAP) * 8u;
    row.endm8 = row.pos + (F_CAP - 1) * 8u;
    row.lst = 0;

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {       // start values of the first code tile by vector loads (nothing is in flight yet)
        float4 e4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) e4[g] = ld4(wsqs + R64N_LANE_CB(m, 0) + 8 * g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = e4[r >> 2];
            acc[m][r] = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_init(1);
    {
        const int so = tile_of(0) * R64N_TILEB;
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_piece(0, so, q);
    }

    // A fragments: row i = m 32 + li of the tile, source chunk 2 t + lh at position (2 t + lh) ^ (i & 7)
    unsigned a_adr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_adr[t] = (unsigned)(size_t)fsm + (unsigned)(li * R64_ROWB + (((2 * t + lh) ^ (li & 7)) << 4));

    // the 16 MFMAs of a code tile, (k16 step, code group) by (k16 step, code group): the A operand of MFMA i + 2 is read while
    // MFMA i issues (three rotating fragments, counted waits); one piece of the next tile's copy behind every fourth
    auto mfma_tile = [&](int st) __attribute__((always_inline)) {
        const unsigned so = (unsigned)((st & 1) * R64N_TILEB);
        const int dma_so = tile_of(st + 1) * R64N_TILEB;
        unsigned adr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) adr[t] = a_adr[t] + so;
        fu32x4 fa[3];
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(fa[0]), "=&v"(fa[1]) : "v"(adr[0]) : "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = i >> 2, m = i & 3;
            fu32x4 &cur = fa[i % 3];
            if (i + 2 < 16) {
                fu32x4 &nxt = fa[(i + 2) % 3];
                const int t2 = (i + 2) >> 2, m2 = (i + 2) & 3;
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\ts_waitcnt lgkmcnt(2)" : "=&v"(nxt), "+v"(cur) : "v"(adr[t2]), "i"(m2 * 4096) : "memory");
            } else if (i + 1 < 16) {
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur) : : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur) : : "memory");
            }
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, cur), xf[t], acc[m], 0, 0, 0);
            if ((i & 3) == 3) stage_piece(st + 1, dma_so, i >> 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    unsigned long multi = 0;
    float cc = -0x1p-15f;
    asm volatile("" : "+v"(cc));
    auto init_wait = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    };
    // learning step: the best of each accumulator tile into the lane's list, no appends; the limit is set after the last one
    auto learn_epilogue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x16 &a = acc[m];
            float mx = v_max3(a[0], a[1], a[2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) mx = v_max3(mx, a[r], a[r + 1]);
            mx = v_max(mx, a[15]);
            const float u = mx * -0x1p-15f;
            thr_insert_med3<TOPK>(row.tv, u == u ? u : INFINITY);
            R64N_INIT_LDS(m, st + 1);
        }
        if (st + 1 >= W) filter_merge_halves<TOPK>(row, lh);
        init_wait();
    };
    // scanning step: bias = 0 -> append and insert (tiles W .. nct-1); bias = +inf -> append only (the learning tiles, revisited)
    auto scan_epilogue = [&](int st, float bias) __attribute__((always_inline)) {
        const int tile = tile_of(st);
#define R64N_ONE(M)                                                                                                      \
        do {                                                                                                             \
            const int cb_ = R64N_LANE_CB(M, tile);                                                                       \
            filter_scan<TOPK, false, true>(row, acc[M], cb_, cbase, multi, nullptr, cc, bias);                            \
            if (multi) { filter_scan_rest<TOPK, true, true>(row, acc[M], cb_, cbase, cc, bias); multi = 0; }             \
            R64N_INIT_LDS(M, st + 1);                                                                                    \
        } while (0)
        R64N_ONE(0); R64N_ONE(1); R64N_ONE(2); R64N_ONE(3);
#undef R64N_ONE
        // the limit moves after each of the first four scanned tiles, then after every fourth and after the last one (a limit that is
        // a few tiles old is still the k-th best of real codes: valid, slightly looser; after every tile: 3-5 % slower)
        const int since = st - W;
        if (st < nct && (since < 4 || (since & 3) == 3 || st == nct - 1)) filter_merge_halves<TOPK>(row, lh);
        init_wait();
    };
    // one step up to its epilogue (see filter_rows64_kernel: the same protocol on a ring of 16 KB slots)
    auto step_head = [&](int t) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage_init(t + 2);
        mfma_tile(t);
        // (the wait states an asm reader of an MFMA result needs: see filter_rows64_kernel)
        asm volatile("s_nop 15\n\ts_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    };
    int t = 0;
    for (; t < W; ++t) {
        step_head(t);
        learn_epilogue(t);
    }
    for (; t < nsteps; ++t) {
        step_head(t);
        scan_epilogue(t, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(t < nct ? 0 : 0x7f800000)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const bool ok = sane && xsq[min(xr, n - 1)] <= F_NORM_LIMIT;
        const unsigned first = row.endm8 - (F_CAP - 1) * 8u;
        if (xr < n) cand_cnt[xr * own_total + owner] = ok ? (int)((row.pos - first) >> 3) : F_CAP + 1;
    }
#undef R64N_INIT_LDS
#undef R64N_LANE_CB
}

// ---------------------------------------------------------------- exact re-score
// Block = RR rows x 8 lanes (RR = 32; 8 for searches of a few thousand rows, where 32-row blocks would leave most CUs without one
// and each block would crawl through its ~220 chains alone: 4096 rows: 0.24 -> see DESIGN).  Phase 1 (per row, 8 lanes): the row's true t~ = k-th smallest d~ over all
// owners' candidates.  Phase 2a: candidates with d~ <= t~ + 2 eps are compacted into an LDS list.
// Phase 2b: the block's survivors (about 7 per row, so ~220 for 256 threads) are spread densely over the
// threads and each is re-scored with the canonical fp32 chain; x and code rows stream from L1/L2 (the
// survivors of a row run side by side, so its x row is fetched once).  Phase 3 (per row): exact
// (d, index) top-k of the row's survivors.
template <int TOPK, int RR = R_ROWS>
__global__ __launch_bounds__(8 * RR) void rescore_kernel(
    const uint2 *__restrict__ cand, const int *__restrict__ cand_cnt, int own_total,
    const uint2 *__restrict__ cand_tail, const int *__restrict__ cnt_tail, int own_tail, long tail_start,
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, const float *__restrict__ en_max_ptr, long n, int k_codes, int d, int topk_out,
    int64_t *__restrict__ out_idx, float *__restrict__ out_dist, int *__restrict__ fb_count, int *__restrict__ fb_rows,
    const float *__restrict__ xref, float *__restrict__ w_out, float *zq_out, long zq_stride)
{
    __shared__ int s_code[RR * R_SURV];
    __shared__ float s_d[RR * R_SURV];
    __shared__ int s_cnt[2 * RR + 1];                 // [RR] survivors, [RR] offsets, total
    const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const long pos = (long)blockIdx.x * RR + g;
    const long row = min(pos, n - 1);
    if (threadIdx.x < RR) s_cnt[threadIdx.x] = 0;
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
        // (from 16 owners up a lane takes owners l8, l8 + 8, ... instead of every 8th entry of every owner: searches of a few
        // thousand rows run with dozens of code splits, i.e. dozens of owners with a handful of entries each, and the walk over
        // them -- a dependent count load per owner -- is the latency of the whole launch.  The k-th smallest value does not depend
        // on who looked at which entry.)
        const bool by_owner = own_total >= 16;
        for (int o = by_owner ? l8 : 0; o < own_total; o += by_owner ? 8 : 1) {
            const int m = cc[o];
            for (int sidx = by_owner ? 0 : l8; sidx < m; sidx += by_owner ? 1 : 8)
                thr_insert<TOPK>(tv, fmaf(__uint_as_float(rc[o * F_CAP + sidx].x), -0x1p-15f, xn));
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
        for (int o = by_owner ? l8 : 0; o < own_total; o += by_owner ? 8 : 1) {
            const int m = cc[o];
            for (int sidx = by_owner ? 0 : l8; sidx < m; sidx += by_owner ? 1 : 8) {
                const uint2 e = rc[o * F_CAP + sidx];
                if (fmaf(__uint_as_float(e.x), -0x1p-15f, xn) <= lim && e.y < (unsigned)k_codes) {
                    const int p = atomicAdd(&s_cnt[g], 1);
                    if (p < R_SURV) s_code[g * R_SURV + p] = (int)e.y;
                }
            }
        }
    }
    __syncthreads();
    // rows with more survivors than the list holds also go to the exact path
    if (threadIdx.x < RR && s_cnt[threadIdx.x] > R_SURV) s_cnt[threadIdx.x] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int r = 0; r < RR; ++r) { s_cnt[RR + r] = run; run += max(s_cnt[r], 0); }
        s_cnt[2 * RR] = run;
    }
    __syncthreads();
    // ---- phase 2b: dense exact chains
    const int total = s_cnt[2 * RR];
    for (int wi = threadIdx.x; wi < total; wi += 8 * RR) {
        int rr = 0;
#pragma unroll
        for (int r = 1; r < RR; ++r) rr += (wi >= s_cnt[RR + r]) ? 1 : 0;
        const int j = wi - s_cnt[RR + rr];
        const int code = s_code[rr * R_SURV + j];
        const long arow = min((long)blockIdx.x * RR + rr, n - 1);
        const float *xr = xhat + arow * d;
        const float *wr = what + (long)code * d;
        float accv = 0.f;
        int i = 0;
        // canonical order within a group of 8: 0,4,1,5,2,6,3,7
#define R_CHAIN8(xa, xb, wa, wb)                                            \
        accv = fmaf(xa.x, wa.x, accv); accv = fmaf(xb.x, wb.x, accv);       \
        accv = fmaf(xa.y, wa.y, accv); accv = fmaf(xb.y, wb.y, accv);       \
        accv = fmaf(xa.z, wa.z, accv); accv = fmaf(xb.z, wb.z, accv);       \
        accv = fmaf(xa.w, wa.w, accv); accv = fmaf(xb.w, wb.w, accv)
        for (; i + 16 <= d; i += 16) {
            const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4), x2 = ld4(xr + i + 8), x3 = ld4(xr + i + 12);
            const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4), w2 = ld4(wr + i + 8), w3 = ld4(wr + i + 12);
            R_CHAIN8(x0, x1, w0, w1); R_CHAIN8(x2, x3, w2, w3);
        }
        for (; i + 8 <= d; i += 8) {
            const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4);
            const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4);
            R_CHAIN8(x0, x1, w0, w1);
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

// ---------------------------------------------------------------- exact re-score, few rows: one WAVEFRONT per row
// Searches of a few thousand rows (a serving batch; the four 4096-row searches of a forward) run the filter with up to 16 code
// splits, i.e. up to 64 candidate lists per row with a handful of entries each, and the launch's time is one block's LATENCY: the
// 8-lanes-per-row walk above visits its owners one dependent load after the other (count, then entry by entry: ~80 round trips
// per lane and pass), and its chains miss on every step.  Here a wave owns a row: the 64 counts come with one load, the entries
// with lane = owner, four entry indices in flight at a time; the lines of the survivors' code rows are all requested before the
// first chain starts.  Same candidates, same exact chains, same (d, index) selection and fused assignment as
// rescore_kernel: the same bits.  own_total, own_tail <= 64.
constexpr int RW_SURV = 64;                                  // survivors of a row in rescore_wave_kernel: one lane each
constexpr int RW_XMAX = 1032;                                // widest row whose x the wave keeps in LDS (the library's widest: 1028)
template <int TOPK>
__global__ __launch_bounds__(256) void rescore_wave_kernel(
    const uint2 *__restrict__ cand, const int *__restrict__ cand_cnt, int own_total,
    const uint2 *__restrict__ cand_tail, const int *__restrict__ cnt_tail, int own_tail, long tail_start,
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, const float *__restrict__ en_max_ptr, long n, int k_codes, int d, int topk_out,
    int64_t *__restrict__ out_idx, float *__restrict__ out_dist, int *__restrict__ fb_count, int *__restrict__ fb_rows,
    const float *__restrict__ xref, float *__restrict__ w_out, float *zq_out, long zq_stride)
{
    __shared__ int s_code[4][RW_SURV];
    __shared__ __attribute__((aligned(16))) float s_x[4][RW_XMAX];      // the wave's x row (the chains' common operand), d <= RW_XMAX
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wv;
    if (row >= n) return;                              // (no block-wide barrier below: a wave works alone)
    const float xn = xsq[row];
    const float win = 2.0f * filter_eps(xn, en_max_ptr[0], d);
    const bool in_tail = row >= tail_start;
    if (in_tail) own_total = own_tail;
    const uint2 *rc = in_tail ? cand_tail + (row - tail_start) * own_total * F_CAP : cand + row * own_total * F_CAP;
    const int *cc = in_tail ? cnt_tail + (row - tail_start) * own_total : cand_cnt + row * own_total;
    const int m_mine = lane < own_total ? cc[lane] : 0;
    if (__builtin_amdgcn_ballot_w64(m_mine > F_CAP)) {   // the filter gave this row up
        if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = (int)row;
        return;
    }
    // ---- phase 1: t~ = the k-th smallest d~ over all owners' entries
    float tv[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) tv[j] = INFINITY;
    // lane = owner, four entry indices per step: an owner's list is a handful of entries (a few dozen per row over all owners), so
    // the walk is max(count) / 4 steps of four loads with most lanes busy -- not eight loads per eight owners with one or two
    // lanes busy each (-10 us of ~120 at 4096 rows; which lane sees which entry changes neither t~ nor the (d, index) selection)
    int m_max = m_mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m_max = max(m_max, __shfl_xor(m_max, off, 64));
    for (int e0 = 0; e0 < m_max; e0 += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = e0 + u < m_mine ? fmaf(__uint_as_float(rc[lane * F_CAP + e0 + u].x), -0x1p-15f, xn) : INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) thr_insert<TOPK>(tv, v[u]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float pv[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) pv[j] = __shfl_xor(tv[j], off, 64);
#pragma unroll
        for (int j = 0; j < TOPK; ++j) thr_insert<TOPK>(tv, pv[j]);
    }
    float kth = tv[0];
#pragma unroll
    for (int j = 1; j < TOPK; ++j) kth = (j < topk_out) ? tv[j] : kth;
    const float lim = kth + win;
    // ---- phase 2a: entries with d~ <= t~ + 2 eps, compacted (ballot + prefix count: no atomics)
    int nsurv = 0;
    for (int e0 = 0; e0 < m_max; e0 += 4) {
        uint2 e[4];
        bool pass[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool have = e0 + u < m_mine;
            e[u] = have ? rc[lane * F_CAP + e0 + u] : make_uint2(0u, 0xffffffffu);
            pass[u] = have && fmaf(__uint_as_float(e[u].x), -0x1p-15f, xn) <= lim && e[u].y < (unsigned)k_codes;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass[u]);
            const int p = nsurv + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if (pass[u] && p < RW_SURV) s_code[wv][p] = (int)e[u].y;
            nsurv += __builtin_popcountll(bal);
        }
    }
    if (nsurv > RW_SURV) {                              // more survivors than the list holds: exact path
        if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = (int)row;
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, visible to all its lanes
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2b: the survivors' exact chains, one lane each; first every line they will walk is requested (codebook rows nobody
    // has touched come from the Infinity Cache / HBM: one round trip for all of them instead of one per chain step)
    {
        const int lines = (d * 4 + 127) >> 7;
        float sink = 0.f;
        for (int ti = lane; ti < (nsurv + 1) * lines; ti += 64) {
            const int wi = ti / lines, ln = ti - wi * lines;
            const float *src = wi < nsurv ? what + (long)s_code[wv][wi] * d : xhat + row * d;
            sink += src[min(ln * 32, d - 1)];
        }
        asm volatile("" ::"v"(sink));                  // (keeps the touches; nothing reads them)
    }
    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffff; }
    // the x row into the wave's LDS slice: every chain reads it (a broadcast per step), so the chain's global loads are the code
    // row's alone -- 64 elements of it per step in the registers the 32 + 32 form used (a chain is as many dependent L2 round
    // trips as it has steps: 24 -> 12 at d = 768)
    const bool x_lds = d <= RW_XMAX;
    if (x_lds) {
        for (int i = lane * 4; i < d; i += 256) *reinterpret_cast<float4 *>(&s_x[wv][i]) = ld4(xhat + row * d + i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < nsurv) {
        const int code = s_code[wv][lane];
        const float *xr = xhat + row * d;
        const float *wr = what + (long)code * d;
        float accv = 0.f;
        int i = 0;
        if (x_lds) {
            const float *xs = s_x[wv];
            for (; i + 64 <= d; i += 64) {
                float4 wq[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) wq[j] = ld4(wr + i + 4 * j);
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const float4 xa = *reinterpret_cast<const float4 *>(xs + i + 4 * j), xb = *reinterpret_cast<const float4 *>(xs + i + 4 * j + 4);
                    R_CHAIN8(xa, xb, wq[j], wq[j + 1]);
                }
            }
        }
        for (; i + 32 <= d; i += 32) {                 // a whole 128-byte line of both rows per step, 16 loads in flight
            const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4), x2 = ld4(xr + i + 8), x3 = ld4(xr + i + 12);
            const float4 x4 = ld4(xr + i + 16), x5 = ld4(xr + i + 20), x6 = ld4(xr + i + 24), x7 = ld4(xr + i + 28);
            const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4), w2 = ld4(wr + i + 8), w3 = ld4(wr + i + 12);
            const float4 w4 = ld4(wr + i + 16), w5 = ld4(wr + i + 20), w6 = ld4(wr + i + 24), w7 = ld4(wr + i + 28);
            R_CHAIN8(x0, x1, w0, w1); R_CHAIN8(x2, x3, w2, w3); R_CHAIN8(x4, x5, w4, w5); R_CHAIN8(x6, x7, w6, w7);
        }
        for (; i + 8 <= d; i += 8) {
            const float4 x0 = ld4(xr + i), x1 = ld4(xr + i + 4);
            const float4 w0 = ld4(wr + i), w1 = ld4(wr + i + 4);
            R_CHAIN8(x0, x1, w0, w1);
        }
        if (i < d) {                                   // D % 8 == 4: the last half group
            const float4 x0 = ld4(xr + i), w0 = ld4(wr + i);
            accv = fmaf(x0.x, w0.x, accv); accv = fmaf(x0.y, w0.y, accv);
            accv = fmaf(x0.z, w0.z, accv); accv = fmaf(x0.w, w0.w, accv);
        }
        const float sum = xn + wsq[code];
        const float two = 2.0f * accv;
        bv[0] = sum - two;
        bi[0] = code;
    }
    // ---- phase 3: exact (d, index) top-k of the survivors: every lane ends with the merged list
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float pv[TOPK];
        int pi[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], off, 64); pi[j] = __shfl_xor(bi[j], off, 64); }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) { out_idx[row * topk_out + j] = valid_code(bi[j], j, k_codes); out_dist[row * topk_out + j] = bv[j]; }
    }
#pragma unroll
    for (int j = 0; j < TOPK; ++j) bi[j] = valid_code(bi[j], j, k_codes);
    // ---- phase 4 (one-call forward only): the soft assignment, as in rescore_kernel (element-wise the same arithmetic)
    if (zq_out) {
        float wj[TOPK];
        const float m = -bv[0];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) { wj[j] = expf(-bv[j] - m); sum += wj[j]; }
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk_out) wj[j] = wj[j] / sum;
        if (w_out && lane == 0) {
#pragma unroll
            for (int j = 0; j < TOPK; ++j)
                if (j < topk_out) w_out[row * topk_out + j] = wj[j];
        }
        const float *xr = xref + row * d;
        float *o = zq_out + row * zq_stride;
        for (int i = lane * 4; i < d; i += 256) {
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
#undef R_CHAIN8

// the accumulator start values: [k_pad] -2^15 |e|^2 (exact: a power-of-two scale), -inf beyond k_codes so that padded codes never pass
__global__ __launch_bounds__(256) void pad_wsq_kernel(const float *__restrict__ wsq, int k, int k_pad, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k_pad) out[i] = i < k ? wsq[i] * -32768.0f : -INFINITY;
}
