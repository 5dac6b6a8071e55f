// medtok_vq.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of the MedTok VQ hot path.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (see medtok_amd/csrc/build.py).  Wave size is 64 everywhere; no other target.
//
// Kernels (reference call sites in include/medtok_vq.h):
//   rownorm_kernel        l2-normalise rows + |row|^2            HBM-bound
//   search_f32_kernel     N x K nearest-code search, top-k       MFMA-bound (fp32 matrix pipe)
//   merge_topk_kernel     joins per-code-split partial lists     tiny
//   soft_assign_kernel    softmax(-d), code mix, STE, sq. error  HBM-bound
//   sum_scale_kernel      fixed-order fp64 reduction             tiny
//   ema_*                 histogram, stable radix sort by code, segmented row sum, apply
//   usage_*               sliding id window + distinct count
//
// Arithmetic order is the one oracle/medtok_oracle.c documents; tests compare bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/medtok_vq.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";

static int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}

static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" int medtok_abi_version(void) { return MEDTOK_VQ_ABI_VERSION; }
extern "C" const char *medtok_last_error(void) { return g_err; }

// ---------------------------------------------------------------- optional self-profiling (bench.py)
// Between medtok_profile_begin() and medtok_profile_end() every launch of a search kernel is
// bracketed by HIP events recorded on its own launch stream; nothing synchronises until _end().
struct ProfRec { hipEvent_t a, b; double flops; int kind; };   // kind 0 = filter_f16_kernel, 1 = search_f32_kernel, 2 = attention forward (any kernel), 3 = attention backward (dQ + dKV), 4 = split_gemm_kernel
// Process-wide (the backward kernels are launched from autograd's own thread): a flag read on every launch, the records and the
// event pool behind one mutex that is only ever taken while profiling is on.
static std::atomic<bool> g_prof_on{false};
static std::atomic<unsigned> g_prof_kinds{~0u};      // bit k: launches of kind k are bracketed (medtok_profile_begin_kinds)
static inline bool prof_wanted(int kind) { return g_prof_on && ((g_prof_kinds >> kind) & 1u); }
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;

// Events come from a pool that medtok_profile_begin() fills BEFORE the timed region: recording is all a launch pays.
static std::vector<hipEvent_t> g_event_pool;
constexpr size_t PROF_POOL = 4096;

static hipEvent_t prof_mark(hipStream_t s)
{
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_event_pool.empty()) { e = g_event_pool.back(); g_event_pool.pop_back(); }
    }
    if (!e && hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, s);
    return e;
}
static void prof_push(hipEvent_t a, hipEvent_t b, double flops, int kind)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({a, b, flops, kind});
}

static int profile_begin_impl(unsigned kinds);
extern "C" int medtok_profile_begin(void) { return profile_begin_impl(~0u); }
// ... bracketing only the launches of the kinds in `kinds` (bit k = kind k of medtok_profile_end's arrays): a training step is ~180
// library launches, and two event records per launch are 0.5 ms of an 11 ms step -- the timed region then carries the events of
// its dominant kernel only (bench.py), the other kinds come from a pass behind it
extern "C" int medtok_profile_begin_kinds(unsigned kinds) { return profile_begin_impl(kinds); }
static int profile_begin_impl(unsigned kinds)
{
    g_prof_kinds = kinds;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) { if (r.a) g_event_pool.push_back(r.a); if (r.b) g_event_pool.push_back(r.b); }
    g_prof.clear();
    while (g_event_pool.size() < PROF_POOL) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return fail("profile_begin: hipEventCreate failed");
        g_event_pool.push_back(e);
    }
    g_prof_on = true;
    return 0;
}

extern "C" int medtok_profile_end(double *ms, double *flops, int *launches)
{
    g_prof_on = false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int k = 0; k < MEDTOK_PROFILE_KINDS; ++k) { ms[k] = 0.0; flops[k] = 0.0; launches[k] = 0; }
    for (auto &r : g_prof) {
        float t = 0.f;
        if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.kind] += t; flops[r.kind] += r.flops; launches[r.kind] += 1;
        }
        if (r.a) g_event_pool.push_back(r.a);
        if (r.b) g_event_pool.push_back(r.b);
    }
    g_prof.clear();
    return 0;
}

// ---------------------------------------------------------------- shader-clock probe (bench.py)
// The chip clocks to its power budget: a fp16-MFMA-bound kernel sags to ~1.7 of 2.4 GHz, and how far differs from box to box by a few
// per cent -- as much as a round's kernel work moves the headline.  One idle wavefront per XCD (blocks 0..7 go round-robin to the 8
// XCDs) reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it starts and when the host
// raises `stop` (a word of pinned host memory, polled every ~20 us between s_sleeps; `max_ticks` of the 100 MHz counter bound the
// wait whatever happens to the flag), so that a bench line can state the clock its timed region ran at.  Launch it on a stream of its
// own, NOT one the measured work uses (streams that share a hardware queue serialise).  out: uint64 [8][4] = shader cycles, 100 MHz
// ticks, XCC id, polls.
__global__ __launch_bounds__(64) void clock_probe_kernel(const int *stop, unsigned long long max_ticks, unsigned long long *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long polls = 0, c1 = c0, r1 = r0;
    for (;;) {
        for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127);
        c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        ++polls;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || r1 - r0 >= max_ticks) break;
    }
    unsigned long long *o = out + (size_t)blockIdx.x * 4;
    o[0] = c1 - c0; o[1] = r1 - r0;
    o[2] = (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xf);       // HW_REG_XCC_ID[3:0]
    o[3] = polls;
}

extern "C" int medtok_debug_clock_probe(const int *stop_flag, uint64_t max_ticks_100mhz, uint64_t *out, void *stream)
{
    if (!stop_flag || !out) return fail("clock_probe: NULL argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(8), dim3(64), 0, (hipStream_t)stream, stop_flag, (unsigned long long)max_ticks_100mhz,
                       (unsigned long long *)out);
    return check_launch("clock_probe");
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of the launch: set it once per instantiation
// (and device), not before every launch.
template <auto Kernel>
static bool set_lds_once(size_t bytes)
{
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev == dev) return true;
    if (hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done_dev = dev;
    return true;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline long lmin(long a, long b) { return a < b ? a : b; }
static inline long lmax(long a, long b) { return a > b ? a : b; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float wave_butterfly_sum(float p)
{
    // xor butterfly, offsets 32..1: every lane ends with the same bits (a+b == b+a).
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ================================================================= rownorm
// One wavefront per row.  Lane l owns float4 #(l + 64 t): element i lands in chain
// (i/4)%64 and each chain accumulates in increasing i -- the oracle's canon_sumsq.
typedef _Float16 rn_half4 __attribute__((ext_vector_type(4)));
// xh (optional, NORMALIZE only): the fp16 filter's operand image of the normalised rows, [*, dp] prescaled by 2^8 exactly as
// to_half_kernel writes it -- the one-call forward saves that kernel's pass over xhat.
// n_img (with xh): the image has that many rows; those from n on are written as zeros (the filter reads whole row tiles).
// zero_word: an int the launch clears (the filter's count of rows handed to the exact kernel, when nothing else of the search's
// preparation runs: a prepared codebook).
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void rownorm_kernel(const float *__restrict__ x, long n, int d,
                                                      float *xhat, float *__restrict__ sqn, _Float16 *__restrict__ xh = nullptr, int dp = 0,
                                                      long n_img = 0, int *__restrict__ zero_word = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
    if (row >= n) {
        if (NORMALIZE && xh && row < n_img) {
            rn_half4 z;
            z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
            for (int i = lane * 4; i < dp; i += 256) *reinterpret_cast<rn_half4 *>(xh + row * dp + i) = z;
        }
        return;
    }
    const float *src = x + row * d;
    float *dst = xhat ? xhat + row * d : nullptr;
    float p = 0.f;
    if (NORMALIZE) {
        for (int i = lane * 4; i < d; i += 256) {
            float4 v = ld4(src + i);
            p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
        }
        const float nrm = sqrtf(wave_butterfly_sum(p));
        const float den = fmaxf(nrm, 1e-12f);
        p = 0.f;
        for (int i = lane * 4; i < d; i += 256) {
            float4 v = ld4(src + i);
            v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
            st4(dst + i, v);
            if (xh) {
                rn_half4 h;
                h[0] = (_Float16)(v.x * 256.0f); h[1] = (_Float16)(v.y * 256.0f); h[2] = (_Float16)(v.z * 256.0f); h[3] = (_Float16)(v.w * 256.0f);
                *reinterpret_cast<rn_half4 *>(xh + row * dp + i) = h;
            }
            p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
        }
        if (xh) {
            rn_half4 z;
            z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
            for (int i = d + lane * 4; i < dp; i += 256) *reinterpret_cast<rn_half4 *>(xh + row * dp + i) = z;
        }
    } else {
        for (int i = lane * 4; i < d; i += 256) {
            float4 v = ld4(src + i);
            if (dst && dst != src) st4(dst + i, v);
            p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
        }
    }
    p = wave_butterfly_sum(p);
    if (sqn && lane == 0) sqn[row] = p;
}

// Rows of at most 64 floats (the reference's e_dim): rownorm_kernel leaves 48 of a wavefront's 64 lanes without an element.  Here a
// row has 16 lanes (lane l of its group owns float4 #l, as there), four rows share a wavefront, and the butterfly runs over the
// offsets 8..1 only -- the offsets 32 and 16 of the one-row kernel add the zeros of idle lanes, so the bits are the same.
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void rownorm16_kernel(const float *__restrict__ x, long n, int d,
                                                        float *xhat, float *__restrict__ sqn, _Float16 *__restrict__ xh = nullptr, int dp = 0,
                                                        long n_img = 0, int *__restrict__ zero_word = nullptr)
{
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
    if (NORMALIZE && xh && row >= n && row < n_img && sub * 4 < dp) {      // image rows past the last input row: zeros
        rn_half4 z;
        z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
        *reinterpret_cast<rn_half4 *>(xh + row * dp + sub * 4) = z;
    }
    const bool live = row < n, mine = live && sub * 4 < d;
    const long r = live ? row : 0;
    const float *src = x + r * d + sub * 4;
    float4 v = mine ? ld4(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    float p = 0.f;
    p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    if (NORMALIZE) {
        const float den = fmaxf(sqrtf(p), 1e-12f);
        v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
        if (mine) {
            st4(xhat + r * d + sub * 4, v);
            if (xh) {
                rn_half4 h;
                h[0] = (_Float16)(v.x * 256.0f); h[1] = (_Float16)(v.y * 256.0f); h[2] = (_Float16)(v.z * 256.0f); h[3] = (_Float16)(v.w * 256.0f);
                *reinterpret_cast<rn_half4 *>(xh + r * dp + sub * 4) = h;
            }
        } else if (live && xh && sub * 4 < dp) {
            rn_half4 z;
            z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
            *reinterpret_cast<rn_half4 *>(xh + r * dp + sub * 4) = z;
        }
        p = 0.f;
        p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    } else if (mine && xhat && xhat != x) {
        st4(xhat + r * d + sub * 4, v);
    }
    if (sqn && live && sub == 0) sqn[row] = p;
}

// the launch every caller of the row-norm kernels goes through: the 16-lanes-per-row form where a row fits it
template <bool NORMALIZE>
static inline void launch_rownorm(hipStream_t s, const float *x, long n, int d, float *xhat, float *sqn, _Float16 *xh = nullptr, int dp = 0,
                                  long n_img = 0, int *zero_word = nullptr)
{
    const long rows = xh && n_img > n ? n_img : n;
    if (d <= 64 && (!xh || dp <= 64))
        hipLaunchKernelGGL(rownorm16_kernel<NORMALIZE>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, x, n, d, xhat, sqn, xh, dp, n_img, zero_word);
    else
        hipLaunchKernelGGL(rownorm_kernel<NORMALIZE>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, n, d, xhat, sqn, xh, dp, n_img, zero_word);
}

extern "C" int medtok_rownorm_f32(const float *x, int64_t n, int d, int normalize, float *xhat,
                                  float *sqn, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3)) return fail("rownorm: need n >= 0, d > 0, d %% 4 == 0 (n=%ld d=%d)", (long)n, d);
    if (n == 0) return 0;                      // (an empty torch tensor has a null data pointer: sizes first, pointers after)
    if (normalize && !xhat) return fail("rownorm: xhat required when normalize != 0");
    hipStream_t s = (hipStream_t)stream;
    if (normalize) launch_rownorm<true>(s, x, (long)n, d, xhat, sqn);
    else launch_rownorm<false>(s, x, (long)n, d, xhat, sqn);
    return check_launch("rownorm");
}

// single VALU instructions (fminf on MFMA results makes hipcc put a canonicalising v_max in front of each operand)
__device__ __forceinline__ float vs_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vs_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ================================================================= top-k list helpers
// Sorted ascending by (value, index).  A lane meets its codes in increasing index order,
// so a strict '<' on the value alone implements "ties -> lowest index" during the scan.
template <int T>
__device__ __forceinline__ void topk_insert(float (&bv)[T], int (&bi)[T], float v, int c)
{
    if (v < bv[T - 1]) {
#pragma unroll
        for (int j = T - 1; j >= 1; --j) {
            const bool lt_prev = v < bv[j - 1];
            const bool lt = v < bv[j];
            bv[j] = lt_prev ? bv[j - 1] : (lt ? v : bv[j]);
            bi[j] = lt_prev ? bi[j - 1] : (lt ? c : bi[j]);
        }
        const bool lt0 = v < bv[0];
        bv[0] = lt0 ? v : bv[0];
        bi[0] = lt0 ? c : bi[0];
    }
}

__device__ __forceinline__ bool lex_lt(float v, int c, float bv, int bc)
{
    return v < bv || (v == bv && c < bc);
}

// A row whose distances are NaN never inserts anything and its list keeps the sentinel index.  Token ids leave the library
// in range regardless -- slot j falls back to code j (torch.topk / argmin also return valid indices for such rows, so the
// reference's AMP loop survives an overflowed step; the gathers downstream index the codebook with these ids).
__device__ __forceinline__ int valid_code(int c, int j, int k_codes) { return (unsigned)c < (unsigned)k_codes ? c : j % k_codes; }

// Same, but for merging lists whose codes are not met in order: compare (value, index).
template <int T>
__device__ __forceinline__ void topk_insert_lex(float (&bv)[T], int (&bi)[T], float v, int c)
{
    if (lex_lt(v, c, bv[T - 1], bi[T - 1])) {
#pragma unroll
        for (int j = T - 1; j >= 1; --j) {
            const bool lt_prev = lex_lt(v, c, bv[j - 1], bi[j - 1]);
            const bool lt = lex_lt(v, c, bv[j], bi[j]);
            const float nv = lt_prev ? bv[j - 1] : (lt ? v : bv[j]);
            const int ni = lt_prev ? bi[j - 1] : (lt ? c : bi[j]);
            bv[j] = nv; bi[j] = ni;
        }
        const bool lt0 = lex_lt(v, c, bv[0], bi[0]);
        bv[0] = lt0 ? v : bv[0];
        bi[0] = lt0 ? c : bi[0];
    }
}

// ================================================================= fp32 MFMA search
// Block = 4 waves, tile = 128 codes x 128 rows, BK = 32.  Codes are the MFMA "A" rows and
// input rows the "B" columns, so after v_mfma_f32_32x32x2_f32 every lane holds 16 codes of
// ONE input row: the running top-k is lane-local (no cross-lane traffic until the end).
// Wave w owns input rows [32w, 32w+32) against all 128 codes (4 accumulator tiles).
//
// LDS keeps rows as they are in memory.  Lane (i, h) reads one float4 of each 8-wide k-group at
// +4h; register c then carries element 8g + 4h + c, and MFMA #c consumes (half 0 -> 8g+c,
// half 1 -> 8g+4+c).  The accumulation therefore visits each group as 0,4,1,5,2,6,3,7 -- the
// canonical chain order of the arithmetic contract (oracle/medtok_oracle.c) -- with no data
// permutation anywhere.  Rows are padded to 36 floats: conflict-free ds_read_b128 (16 lanes x
// stride 36 dwords hit 16 distinct 4-bank slots) and ds_write_b128.
constexpr int S_BM = 128, S_BN = 128;
constexpr int S_BK = 32;                                  // 2 blocks/CU (74 KB LDS); 16 with 3 blocks/CU measured the same
constexpr int S_LD = S_BK + 4;                            // row stride in floats (36 and 20 are both conflict-free)
constexpr int S_TILE = S_BM * S_LD;                       // floats per staged operand tile
constexpr size_t S_LDS_BYTES = (size_t)4 * S_TILE * sizeof(float);   // A[2] + B[2]
constexpr int S_TPR = S_BK / 8;                           // staging threads per tile row (8 floats each)
constexpr int S_RPP = 256 / S_TPR;                        // tile rows staged per pass
constexpr int S_PASSES = S_BM / S_RPP;
constexpr int S_WPS = S_BK == 32 ? 2 : 3;                 // waves per SIMD the register budget is sized for

// INDIRECT: the block's rows are row_list[row0 .. row0+128) (count read from *row_count on the
// device) -- the exact fallback for rows the fp16 filter could not shortlist.
// (the body is a force-inlined function of the block's coordinates: search_f32_kernel takes them from blockIdx, the batched kernel
// of the small-batch forward -- several searches in one launch -- from its descriptor table)
// EXCL (the second pass of a search for more than 8 codes per row): codes at or below the row's (distance, index) pair
// (excl_d[row], excl_i[row]) -- the last entry of the first pass's list -- are skipped, so the pass returns the NEXT best codes in the
// same total order (distance, then index).
template <int TOPK, bool FINAL, bool KTAIL, bool INDIRECT, bool EXCL = false>
__device__ __forceinline__ void search_f32_body(
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, long n, int k_codes, int d, int codes_per_split, int topk_out,
    float *__restrict__ pval, int *__restrict__ pidx, int64_t *__restrict__ out_idx,
    float *__restrict__ out_dist, const int *__restrict__ row_list, const int *__restrict__ row_count,
    int list_begin, int list_end, const unsigned block_x, const unsigned block_y,
    const float *__restrict__ excl_d = nullptr, const int64_t *__restrict__ excl_i = nullptr, int excl_stride = 0)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // direct form: this launch covers rows from list_begin on; its partial-result buffers span list_end rows (0 = all n)
    long row0 = (long)block_x * S_BN + (INDIRECT ? 0 : list_begin);
    const long part_rows = INDIRECT ? (long)(list_end - list_begin) : (list_end > 0 ? (long)list_end : n);     // row extent of the partial-result buffers
    if (INDIRECT) {
        // this launch covers list positions [list_begin, min(*row_count, list_end)); uniform for the whole grid
        n = min((long)*row_count, (long)list_end);
        row0 += list_begin;
        if (row0 >= n) return;
    }
    auto actual_row = [&](long pos) -> long {
        const long c = min(pos, n - 1);
        return INDIRECT ? (long)row_list[c] : c;
    };
    const int split = (int)block_y;
    const int code_lo = split * codes_per_split;
    const int code_hi = min(k_codes, code_lo + codes_per_split);
    const int nct = (code_hi - code_lo + S_BM - 1) / S_BM;
    const int nkb = (d + S_BK - 1) / S_BK;
    const int nstage = nct * nkb;

    const int srow = tid / S_TPR, sg = tid % S_TPR;
    float4 ra[S_PASSES][2], rb[S_PASSES][2];
    int kvalid = 0;                 // bit0/bit1: which float4 of the staged k-group lies inside D
    int pct = 0, pkb = 0;           // (code tile, k block) of the next stage to prefetch

    // Loads are unconditional: branching around them makes hipcc drain vmcnt(0) per load.  KTAIL
    // (D % 32 != 0) clamps the addresses into the row and zeroes the out-of-range float4 when it
    // is written to LDS (zeros leave the fmaf chain untouched).
    auto gload = [&]() {
        const int kofs = pkb * S_BK + sg * 8;
        int k0 = kofs, k1 = kofs + 4;
        if (KTAIL) {
            kvalid = (kofs < d ? 1 : 0) | (kofs + 4 < d ? 2 : 0);
            k0 = min(k0, d - 4);
            k1 = min(k1, d - 4);
        }
#pragma unroll
        for (int j = 0; j < S_PASSES; ++j) {
            const int crow = min(code_lo + pct * S_BM + srow + S_RPP * j, k_codes - 1);
            const float *p = what + (long)crow * d;
            const long xr = actual_row(row0 + srow + S_RPP * j);
            const float *q = xhat + xr * d;
            ra[j][0] = ld4(p + k0);
            ra[j][1] = ld4(p + k1);
            rb[j][0] = ld4(q + k0);
            rb[j][1] = ld4(q + k1);
        }
        if (++pkb == nkb) { pkb = 0; ++pct; }
    };
    auto lstore = [&](int buf) {
        float *A = smem + buf * S_TILE;
        float *B = smem + 2 * S_TILE + buf * S_TILE;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < S_PASSES; ++j) {
            float *pa = A + (srow + S_RPP * j) * S_LD + sg * 8;
            float *pb = B + (srow + S_RPP * j) * S_LD + sg * 8;
            if (KTAIL) {
                st4(pa, (kvalid & 1) ? ra[j][0] : z);
                st4(pa + 4, (kvalid & 2) ? ra[j][1] : z);
                st4(pb, (kvalid & 1) ? rb[j][0] : z);
                st4(pb + 4, (kvalid & 2) ? rb[j][1] : z);
            } else {
                st4(pa, ra[j][0]);
                st4(pa + 4, ra[j][1]);
                st4(pb, rb[j][0]);
                st4(pb + 4, rb[j][1]);
            }
        }
    };

    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0; }

    const long mypos = row0 + wave * 32 + li;
    const long myrow = actual_row(mypos);
    const float xn = xsq[myrow];
    float ex_d = -INFINITY;
    int ex_i = -1;
    if (EXCL) { ex_d = excl_d[myrow * excl_stride]; ex_i = (int)excl_i[myrow * excl_stride]; }

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // ---- epilogue of one code tile: d = (|x|^2 + |e|^2) - 2 x.e for this lane's 64 codes, fold into the list
    auto tile_epilogue = [&](int ct) __attribute__((always_inline)) {
        // ---- epilogue: d = (|x|^2 + |e|^2) - 2 x.e for this lane's 64 codes, fold into the list
        const int cbase = code_lo + ct * S_BM;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // |e|^2 of this lane's 16 codes of the tile: four consecutive codes per register group, so four 16-byte loads
            // (+2.3 % at K = 16384, k = 5); a group that straddles K or sits on an unaligned slice takes the scalar form
            // codes at or beyond the split's end get |e|^2 = +inf, i.e. d = +inf: never inserted, and no range test per value
            float en[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cbase + 32 * m + 8 * g + 4 * lh;
                // (argmin, TOPK = 1: the 16 scalar loads measured 4 % faster than the vector form -- its epilogue is nothing else)
                if (TOPK > 1 && c0 + 3 < code_hi && ((reinterpret_cast<uintptr_t>(wsq + c0) & 15) == 0)) {
                    const float4 e4 = ld4(wsq + c0);
                    en[4 * g] = e4.x; en[4 * g + 1] = e4.y; en[4 * g + 2] = e4.z; en[4 * g + 3] = e4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = wsq[min(c0 + j, k_codes - 1)];        // (unconditional load: a branch around it drains vmcnt)
                        en[4 * g + j] = c0 + j < code_hi ? e : INFINITY;
                    }
                }
            }
            // four codes per test: the smallest of their distances against the list's last entry, ONE wave-uniform branch; a
            // quad in which some lane has a better code is then folded in value by value, in code order as before
            // (ties -> lowest index).  (+0.6 % at k = 5; the argmin kernel keeps the per-value form: -0.5 % there.)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sum = xn + en[4 * g + j];
                    const float two = 2.0f * acc[m][4 * g + j];
                    dv[j] = sum - two;
                    acc[m][4 * g + j] = 0.f;
                    if (EXCL) {
                        const int c = cbase + 32 * m + j + 8 * g + 4 * lh;
                        if (dv[j] < ex_d || (dv[j] == ex_d && c <= ex_i)) dv[j] = INFINITY;
                    }
                }
                bool any = true;
                if (TOPK > 1) {
                    const float mn = vs_min(vs_min3(dv[0], dv[1], dv[2]), dv[3]);      // (NaN never wins a v_min: a NaN distance is never inserted)
                    any = __builtin_amdgcn_ballot_w64(mn < bv[TOPK - 1]) != 0;
                }
                if (any) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) topk_insert<TOPK>(bv, bi, dv[j], cbase + 32 * m + j + 8 * g + 4 * lh);
                }
            }
        }
    };

    if constexpr (!KTAIL && !INDIRECT) {
        // ---- operand staging by LDS-DMA (D % 32 == 0, rows addressed directly).  A stage = 32 floats of 128 code rows and 128 input rows, each row 128 B =
        // eight 16-byte chunks; wave w copies rows [32w, 32w+32) of both tiles, eight rows per global_load_lds_dwordx4 (lane l:
        // row l >> 3, chunk slot l & 7), straight from L2 into one of TWO 32 KB buffers -- no staging registers, no ds_write pass,
        // nothing to wait for before the MFMAs of a stage but the barrier.  The rows are unpadded (the DMA's LDS image is
        // lane-linear), so chunk c of row r is stored in slot c ^ ((r >> 1) & 7): the 16 lanes a ds_read_b128 services together
        // then hit 16 distinct 16-byte bank groups (rows of equal parity in such a group differ in bits 1..3 of r).  The swizzle is
        // applied to the per-lane SOURCE address.  Staging through registers + ds_write (the KTAIL form below, which needs it to
        // zero the columns past D) measured 130.8 TFLOP/s in the main loop against 152 with the staging removed; this form 143.7
        // (N = 600k, K = 16 384, D = 768; whole kernel 123.2 -> 137.4 TFLOP/s at k = 5, 126.8 -> 134.9 for the argmin at 100k x 8192).
        constexpr int ROWB = S_BK * 4, TILEB = S_BM * ROWB;            // 128 B per staged row, 16 KB per tile
        char *lds = reinterpret_cast<char *>(smem);                      // [2 buffers][A tile | B tile]
        const int d_r = lane >> 3, d_p = lane & 7;                       // row within the instruction, chunk slot
        // buffer-addressed DMA (SGPR descriptor + loop-invariant 32-bit lane offset + SGPR stage offset; hipcc drains
        // vmcnt(0) before every ds_read that follows a global_load_lds, but not after the raw-buffer form).  The descriptors
        // carry the valid byte range: rows past K or past n read as zeros instead of touching memory (such codes get
        // |e|^2 = +inf in the epilogue, such rows are never written back).
        // The code-side descriptor is rebuilt per stage for the stage's code tile (base = the tile's first row, range = its rows
        // inside K: a handful of SALU operations), so every offset stays far below 2^31 whatever K * D is.
        const float *abase_p = what + (long)code_lo * d, *bbase_p = xhat + row0 * d;
        const long rows_left = n - row0;
        const int b_bytes = (int)(rows_left < S_BN ? rows_left : S_BN) * d * 4;
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)bbase_p, 0, b_bytes, 0x00020000);
        unsigned lane_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + 8 * i + d_r;
            lane_off[i] = (unsigned)(r * d + 4 * (d_p ^ ((r >> 1) & 7))) * 4u;
        }
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        int pct = 0, pkb = 0;
        // One stage = eight DMA instructions per wave.  Issued as a burst they hold the wave (and, through the CU's one address
        // path, its neighbours) at the head of the stage; instead the descriptor is prepared once per stage and the eight pieces
        // go out one per group of four MFMAs of the stage's first two k steps.  Past the last stage the same stage is issued
        // again into the buffer nobody reads (same bytes, harmless): no branch around a DMA.
        __amdgpu_buffer_rsrc_t ars = brs;
        int ub = 0;
        char *abase = lds, *bbase = lds;
        auto dma_prepare = [&](int buf) __attribute__((always_inline)) {
            abase = lds + buf * 2 * TILEB + (wave_s * 32) * ROWB; bbase = abase + TILEB;
            ub = __builtin_amdgcn_readfirstlane(pkb * S_BK * 4);
            const int tile_codes = __builtin_amdgcn_readfirstlane(min(S_BM, k_codes - code_lo - pct * S_BM));
            ars = __builtin_amdgcn_make_buffer_rsrc((void *)(abase_p + (long)__builtin_amdgcn_readfirstlane(pct) * S_BM * d), 0, tile_codes * d * 4, 0x00020000);
            const bool wrap = pkb + 1 == nkb, more = !(wrap && pct + 1 == nct);
            pkb = more ? (wrap ? 0 : pkb + 1) : pkb;
            pct += (more && wrap) ? 1 : 0;
        };
        auto dma_piece = [&](int i) __attribute__((always_inline)) {          // i = 0..7: A rows 8 (i/2) .. of the wave's 32, then B rows
            if (i & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void *)(bbase + 8 * (i >> 1) * ROWB), 16, (int)lane_off[i >> 1], ub, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void *)(abase + 8 * (i >> 1) * ROWB), 16, (int)lane_off[i >> 1], ub, 0, 0);
        };
        // fragment addresses: lane (li, lh) reads chunk 2 kk + lh of row li (+ 32 m) -- slot (2 kk + lh) ^ ((li >> 1) & 7).
        // The fragment reads are asm: hipcc orders a C++ ds_read behind ALL pending LDS-DMA ("s_waitcnt vmcnt(0)": a DMA is a pending
        // LDS write that might alias), i.e. it drained the stage just issued -- a round trip to the L2 per stage in front of the
        // MFMAs.  The reads of step kk + 1 are issued before the MFMAs of step kk (two register sets); lgkmcnt is waited by hand.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const unsigned lds0 = (unsigned)(size_t)lds;
        unsigned fragA[S_BK / 8], fragB[S_BK / 8];
#pragma unroll
        for (int kk = 0; kk < S_BK / 8; ++kk) {
            fragA[kk] = lds0 + li * ROWB + ((2 * kk + lh) ^ ((li >> 1) & 7)) * 16;
            fragB[kk] = fragA[kk] + TILEB + wave * 32 * ROWB;
        }
        f32x4 af[2][4], bf[2];
        auto frag_read = [&](int set, int kk, unsigned bufofs) __attribute__((always_inline)) {
            asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:4096\n\t"
                         "ds_read_b128 %3, %6 offset:8192\n\tds_read_b128 %4, %6 offset:12288"
                         : "=&v"(bf[set]), "=&v"(af[set][0]), "=&v"(af[set][1]), "=&v"(af[set][2]), "=&v"(af[set][3])
                         : "v"(fragB[kk] + bufofs), "v"(fragA[kk] + bufofs));
        };
        auto frag_wait = [&](int set) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[set]), "+v"(af[set][0]), "+v"(af[set][1]), "+v"(af[set][2]), "+v"(af[set][3]));
        };
        static_assert(32 * ROWB == 4096, "fragment offsets above are written for 128-byte staged rows");
        dma_prepare(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_piece(i);
        int ct = 0, kb = 0;
        for (int s = 0; s < nstage; ++s) {
            const unsigned bufofs = (unsigned)(s & 1) * (2 * TILEB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // own part of stage s has landed
            __builtin_amdgcn_s_barrier();                               // everyone's has; everyone is done with the other buffer
            asm volatile("" ::: "memory");
            frag_read(0, 0, bufofs);
            dma_prepare((s & 1) ^ 1);
            frag_wait(0);
#pragma unroll
            for (int kk = 0; kk < S_BK / 8; ++kk) {
                const int cur = kk & 1;
                if (kk + 1 < S_BK / 8) frag_read(cur ^ 1, kk + 1, bufofs);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].x, bf[cur].x, acc[m], 0, 0, 0);
                if (kk < 2) { dma_piece(4 * kk); asm volatile("" ::: "memory"); }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].y, bf[cur].y, acc[m], 0, 0, 0);
                if (kk < 2) { dma_piece(4 * kk + 1); asm volatile("" ::: "memory"); }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].z, bf[cur].z, acc[m], 0, 0, 0);
                if (kk < 2) { dma_piece(4 * kk + 2); asm volatile("" ::: "memory"); }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m].w, bf[cur].w, acc[m], 0, 0, 0);
                if (kk < 2) { dma_piece(4 * kk + 3); asm volatile("" ::: "memory"); }
                if (kk < 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                }
                if (kk + 1 < S_BK / 8) frag_wait(cur ^ 1);
            }
            if (++kb == nkb) {
                tile_epilogue(ct);
                kb = 0;
                ++ct;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    gload();
    int ct = 0, kb = 0;
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        lstore(buf);
        __syncthreads();
        if (s + 1 < nstage) gload();
        const float *A = smem + buf * S_TILE + li * S_LD + lh * 4;
        const float *B = smem + 2 * S_TILE + buf * S_TILE + (wave * 32 + li) * S_LD + lh * 4;
#pragma unroll
        for (int kk = 0; kk < S_BK / 8; ++kk) {
            const float4 bf = ld4(B + kk * 8);
            float4 af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = ld4(A + m * 32 * S_LD + kk * 8);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf.x, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf.y, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf.z, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf.w, acc[m], 0, 0, 0);
        }
        if (++kb == nkb) {
            tile_epilogue(ct);
            kb = 0;
            ++ct;
        }
    }
    }

    // ---- join the two half-waves that share an input row (disjoint code sets)
    float pv[TOPK];
    int pi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], 32, 64); pi[j] = __shfl_xor(bi[j], 32, 64); }
#pragma unroll
    for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);

    if (lh == 0 && mypos < n) {
        if (FINAL) {
#pragma unroll
            for (int j = 0; j < TOPK; ++j)
                if (j < topk_out) { out_idx[myrow * topk_out + j] = bi[j]; out_dist[myrow * topk_out + j] = bv[j]; }
        } else {
            const long base = ((long)split * part_rows + (INDIRECT ? mypos - list_begin : myrow - list_begin)) * TOPK;
#pragma unroll
            for (int j = 0; j < TOPK; ++j) { pval[base + j] = bv[j]; pidx[base + j] = bi[j]; }
        }
    }
}

template <int TOPK, bool FINAL, bool KTAIL, bool INDIRECT>
__global__ __launch_bounds__(256, S_WPS) void search_f32_kernel(
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, long n, int k_codes, int d, int codes_per_split, int topk_out,
    float *__restrict__ pval, int *__restrict__ pidx, int64_t *__restrict__ out_idx,
    float *__restrict__ out_dist, const int *__restrict__ row_list, const int *__restrict__ row_count,
    int list_begin, int list_end)
{
    search_f32_body<TOPK, FINAL, KTAIL, INDIRECT>(xhat, xsq, what, wsq, n, k_codes, d, codes_per_split, topk_out, pval, pidx, out_idx, out_dist,
                                                  row_list, row_count, list_begin, list_end, blockIdx.x, blockIdx.y);
}

template <int TOPK, bool FINAL, bool KTAIL>
__global__ __launch_bounds__(256, S_WPS) void search_f32_excl_kernel(
    const float *__restrict__ xhat, const float *__restrict__ xsq, const float *__restrict__ what,
    const float *__restrict__ wsq, long n, int k_codes, int d, int codes_per_split, int topk_out,
    float *__restrict__ pval, int *__restrict__ pidx, int64_t *__restrict__ out_idx,
    float *__restrict__ out_dist, int list_begin, int list_end,
    const float *__restrict__ excl_d, const int64_t *__restrict__ excl_i, int excl_stride)
{
    search_f32_body<TOPK, FINAL, KTAIL, false, true>(xhat, xsq, what, wsq, n, k_codes, d, codes_per_split, topk_out, pval, pidx, out_idx, out_dist,
                                                     (const int *)nullptr, (const int *)nullptr, list_begin, list_end, blockIdx.x, blockIdx.y,
                                                     excl_d, excl_i, excl_stride);
}

// ---- several small searches in ONE launch each of three kernels (the B = 256 forward of the reference's default configuration runs
// its specific and shared searches as four calls of four launches each: 16 launches of 4-30 us for 2 GFLOP; batched: 3 launches).
// Same kernels' bodies, same arithmetic, same bits; a descriptor per search, selected by blockIdx.z.
constexpr int MS_MAX = MEDTOK_MULTI_SEARCH_MAX;   // shared (merged) + text + graph + the two aug views
struct MultiSearchOne {
    const float *x;                             // [n, d] rows to quantise
    const float *what, *wsq;                    // the normalised codebook region [k_codes, d] and its squared norms
    float *xhat, *xsq;                          // out: F.normalize(x) [n, d], its squared norms [n] (scratch)
    float *pval; int *pidx;                     // scratch: per-split lists [splits][n][TOPK]
    int64_t *idx; float *dist, *w, *zq;         // out: [n, topk] ids / distances / weights; [n, d] rows with a row stride
    float *row_sqerr;                           // out (may be NULL): [n] squared error of the soft assignment per row (training losses)
    long n, zq_stride, x_stride;                // (x rows may be a column block of a wider matrix)
    int k_codes, codes_per_split, splits, row_tiles;
};
// block_base: the searches' (row tile, code split) blocks in ONE dimension, search after search (block_base[i] = first block of
// search i, block_base[count] = all).  The search kernel's grid holds exactly the blocks that have work: a 3-D grid of
// max_tiles x max_splits x count launched 996 blocks for the 444 of a B = 256 forward, and -- blocks going to the 8 XCDs by id --
// the active ones of the two short searches all landed on XCDs 0, 1, 4, 5: 70 blocks for 64 slots there, a second generation of a few
// blocks, twice the kernel time (round 6: rocprofv3 counters, profiles/r06_pmc_small_search_*.txt).
struct MultiSearchArgs { MultiSearchOne s[MS_MAX]; int block_base[MS_MAX + 1]; int count, d, topk; };

__global__ __launch_bounds__(256) void rownorm_multi_kernel(MultiSearchArgs a)
{
    const MultiSearchOne &m = a.s[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m.n) return;
    const int d = a.d;
    const float *src = m.x + row * m.x_stride;
    float *dst = m.xhat + row * d;
    // (rownorm_kernel<true>, statement for statement: the same bits)
    float p = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 v = ld4(src + i);
        p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
    }
    const float nrm = sqrtf(wave_butterfly_sum(p));
    const float den = fmaxf(nrm, 1e-12f);
    p = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 v = ld4(src + i);
        v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
        st4(dst + i, v);
        p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
    }
    p = wave_butterfly_sum(p);
    if (lane == 0) m.xsq[row] = p;
}

template <int TOPK, bool KTAIL>
__global__ __launch_bounds__(256, S_WPS) void search_f32_multi_kernel(MultiSearchArgs a)
{
    int z = 0;
#pragma unroll
    for (int i = 1; i < MS_MAX; ++i) z += (i < a.count && (int)blockIdx.x >= a.block_base[i]) ? 1 : 0;
    const MultiSearchOne &m = a.s[z];
    const unsigned local = blockIdx.x - (unsigned)a.block_base[z];
    const unsigned bx = local % (unsigned)m.row_tiles, by = local / (unsigned)m.row_tiles;
    search_f32_body<TOPK, false, KTAIL, false>(m.xhat, m.xsq, m.what, m.wsq, m.n, m.k_codes, a.d, m.codes_per_split, a.topk, m.pval, m.pidx,
                                               (int64_t *)nullptr, (float *)nullptr, (const int *)nullptr, (const int *)nullptr, 0, 0,
                                               bx, by);
}

// Joins the per-split candidate lists of one row: 8 lanes per row, each folds every 8th split, then three shuffle rounds.
// (d, index) is a total order, so the result does not depend on who inserts what when.  (A thread per row walked
// up to 64 splits x k entries serially: 90 us for a 256-row batch.)
// LPR lanes per row: 8, or a whole wave (64) for small batches with many splits (256 rows x 128 splits: 28 -> ~12 us).
template <int TOPK, int LPR = 8>
__global__ __launch_bounds__(256) void merge_topk_kernel(const float *__restrict__ pval, const int *__restrict__ pidx,
                                                         long n, int splits, int topk_out,
                                                         int64_t *__restrict__ out_idx, float *__restrict__ out_dist,
                                                         const int *__restrict__ row_list, const int *__restrict__ row_count, int k_codes)
{
    // with a row list: partial lists are indexed by list position (extent n), results go to row_list[position]
    constexpr int RPB = 256 / LPR;                       // rows per block
    const int l8 = threadIdx.x & (LPR - 1);
    const long pos = (long)blockIdx.x * RPB + (threadIdx.x / LPR);
    const long limit = row_list ? min(n, (long)*row_count) : n;
    if ((long)blockIdx.x * RPB >= limit) return;         // (block-uniform: the redo of the rows the filter gave up on normally has none)
    const long row = min(pos, n - 1);                    // lanes past the end keep shuffling with their group, write nothing
    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffff; }
    for (int s = l8; s < splits; s += LPR) {
        const long base = ((long)s * n + row) * TOPK;
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pval[base + j], pidx[base + j]);
    }
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) {
        float pv[TOPK];
        int pi[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], off, LPR); pi[j] = __shfl_xor(bi[j], off, LPR); }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);
    }
    if (l8 != 0 || pos >= limit) return;
    const long orow = row_list ? (long)row_list[pos] : row;
#pragma unroll
    for (int j = 0; j < TOPK; ++j)
        if (j < topk_out) { out_idx[orow * topk_out + j] = valid_code(bi[j], j, k_codes); out_dist[orow * topk_out + j] = bv[j]; }
}

#include "filter_f16.h"

// Test hook: plan branches that the default heuristics only take at very large shapes can be forced PER CALL through the upper
// bits of the `path` argument (MEDTOK_PLAN_* in medtok_vq.h), so small parity tests cover them.  No process state: the
// workspace query and the launch decode the same argument.  The product path never reads the environment.
struct PlanOverride { long filter_splits = -1, filter_xcd = -1, filter_tail_min_blocks = -1, search_max_splits = -1, filter_rows64 = -1; };
static PlanOverride decode_plan(int path)
{
    PlanOverride o;
    const int fs = (path >> 8) & 0xFF, xcd = (path >> 16) & 3, tail = (path >> 18) & 3, ss = (path >> 20) & 0xFF;
    if (fs) o.filter_splits = fs;
    if (xcd) o.filter_xcd = xcd - 1;
    if (tail) o.filter_tail_min_blocks = tail == 1 ? 0 : 256;
    if (ss) o.search_max_splits = ss;
    const int r64 = (path >> 28) & 3;          // 1 = off, 2 = on (the default at D <= 64), 3 = on, the earlier 128 x 64 wave-tile kernel
    if (r64) o.filter_rows64 = r64 - 1;
    return o;
}
static inline int path_id(int path) { return path & MEDTOK_PATH_MASK; }

// Device geometry the launch plans are sized for, queried once per device (hipDeviceGetAttribute); without a device (a
// workspace query on a CPU-only box) the full MI355X is assumed.  `full` = the unpartitioned chip (256 CUs in 8 XCDs of 32):
// the XCD-aware block order is only used there; on a partitioned device (CPX / a CU mask) the round arithmetic follows the
// CU count that is actually visible.
struct DevInfo { int cus; bool full; };
static DevInfo dev_info()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return {256, true}; }
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        cached[dev].store(cus, std::memory_order_relaxed);
    }
    return {cus, cus == 256};
}

struct SearchPlan {
    int tslots;          // list length the kernels are instantiated with (1, 5 or 8)
    int splits;          // code-range splits (grid.y)
    int codes_per_split; // multiple of 128
    long row_tiles;
    // large searches: the first main_tiles row tiles (whole rounds of 512 resident blocks) walk ALL codes in one block; the
    // row tiles of the last, partly filled round form a second launch with tail_splits code splits (0 = no tail launch)
    long main_tiles;
    int tail_splits, tail_codes_per_split;
};

static SearchPlan plan_search(int64_t n, int64_t k_codes, int topk, const PlanOverride &ov)
{
    SearchPlan p;
    const DevInfo di = dev_info();
    p.tslots = topk == 1 ? 1 : (topk <= 5 ? 5 : 8);
    p.row_tiles = (n + S_BN - 1) / S_BN;
    const long code_tiles = (k_codes + S_BM - 1) / S_BM;
    // Blocks all cost the same and run 2-3 per CU, so a launch of B blocks wastes up to one "round" of ~2.3 blocks per CU
    // (600 on the full chip); ask for >= 64 blocks per CU (16384: tail <= ~4%) by splitting the code range.
    long want = (64L * di.cus + p.row_tiles - 1) / p.row_tiles;
    if (want < 1) want = 1;
    if (want > code_tiles) want = code_tiles;
    // at most 64 splits -- except for one or two row tiles (batches of <= 256 rows), where 64 splits would leave half
    // the CUs without a block: n = 256, K = 49152: 463 -> 276 us with 128 splits; n = 64: 459 -> 202 us with 256
    // (three to seven row tiles -- 257 .. 896 rows: 128 splits, i.e. 384 .. 896 blocks of short code ranges, keep two blocks on every CU;
    // with 64 a 512-row search left half the CUs one block: 0.376 -> 0.322 ms at K = 49152, D = 768, 0.063 -> 0.055 at the reference's
    // shape -- tools/r06/exact_splits_small.py, profiles/r06_exact_splits_small.txt)
    long cap = p.row_tiles <= 2 ? lmin(256, di.cus) / p.row_tiles : (p.row_tiles <= 7 ? 128 : 64);
    if (ov.search_max_splits > 0) cap = ov.search_max_splits;
    if (want > cap) want = cap;
    const long tiles_per_split = (code_tiles + want - 1) / want;
    p.codes_per_split = (int)(tiles_per_split * S_BM);
    p.splits = (int)((code_tiles + tiles_per_split - 1) / tiles_per_split);
    p.main_tiles = p.row_tiles; p.tail_splits = 0; p.tail_codes_per_split = 0;
    // From two full rounds of blocks up (2 blocks per CU resident: 512 per round) the code range is NOT split: every split
    // restarts the lane-local top-k lists, and a young list sends the whole wave through the insertion code for most values
    // (measured at N = 600k, K = 16384: the epilogue is 10 % of the kernel with 4 splits).  What splitting bought -- a short
    // last round -- comes from a second launch instead: the row tiles of the partly filled last round, with enough splits to be
    // one round of short blocks.
    if (p.row_tiles >= 4L * di.cus && n < (1ll << 31) && ov.search_max_splits <= 0) {
        const long round = 2L * di.cus;      // two blocks per CU resident
        const long main_tiles = p.row_tiles / round * round, tail_tiles = p.row_tiles - main_tiles;
        p.splits = 1;
        p.codes_per_split = (int)(code_tiles * S_BM);
        long ts = tail_tiles > 0 ? lmin(lmin(64, code_tiles), round / tail_tiles) : 0;
        if (ts >= 2) {
            const long tps = (code_tiles + ts - 1) / ts;
            p.main_tiles = main_tiles;
            p.tail_codes_per_split = (int)(tps * S_BM);
            p.tail_splits = (int)((code_tiles + tps - 1) / tps);
        }
    }
    return p;
}

// ---- fp16 filter path: plan + workspace layout
struct FilterPlan {
    long n_pad, k_pad, row_tiles;
    int dp, splits, codes_per_split, own_total, tslots;
    int xcd_rows;      // > 0: XCD-aware block order with this many row tiles per XCD at a time (32 / splits)
    bool rows64;       // dp == 64: the narrow-row kernels (row tiles of 128, 4-wave blocks, no tail launch)
    bool rows64_wide;  // ... of which filter_rows64_kernel (128 x 64 wave tiles, two blocks per CU) instead of filter_rows64n_kernel (plan bit)
    int row_bn;        // rows per row tile of the kernel that runs
    // tail launch: the last main_tiles..row_tiles row tiles with more, shorter splits (0 tiles = none)
    long main_tiles;
    int tail_splits, tail_codes_per_split, own_tail;
};

static FilterPlan plan_filter(int64_t n, int64_t k_codes, int d, int topk, const PlanOverride &ov)
{
    FilterPlan f;
    const DevInfo di = dev_info();
    f.tslots = topk == 1 ? 1 : (topk <= 5 ? 5 : 8);
    f.n_pad = (n + F_BN - 1) / F_BN * F_BN;
    f.k_pad = (k_codes + F_BM - 1) / F_BM * F_BM;
    // at least TWO 32-deep stages per code tile: the start values of code tile t + 2 are copied (LDS-DMA, wave 0) at the end of tile
    // t's scan and read at the end of tile t + 1's, and what guarantees that they have landed is the counted vmcnt wait of a LATER
    // stage -- with one stage per tile there is none in between (found by tools/fuzz_search.py at D = 4 / 32: one wave's 64 rows
    // wrong in ~1 % of the searches).  D <= 32 is zero-padded to 64 and takes the narrow-row kernel like every D <= 64.
    f.dp = (int)lmax(2 * F_BK, (d + F_BK - 1) / F_BK * F_BK);
    f.row_tiles = f.n_pad / F_BN;
    const long code_tiles = f.k_pad / F_BM;
    f.rows64 = f.dp == 64 && ov.filter_rows64 != 0;
    f.rows64_wide = f.rows64 && ov.filter_rows64 == 2;
    f.row_bn = f.rows64 ? R64_BN : F_BN;
    if (f.rows64) {
        // two or three blocks per CU, every one walking its whole code range with the x rows in registers: nothing is shared between
        // blocks but the codebook (K x 128 B, L2-resident), so one split as soon as the chip is full four times over --
        // every split adds candidate lists per row (four; two with 32-row wave tiles) and restarts the thresholds
        f.row_tiles = f.n_pad / R64_BN;
        const long slots = (f.rows64_wide ? 2L : 3L) * di.cus;
        long want = f.row_tiles >= 4 * slots ? 1 : (4 * slots + f.row_tiles - 1) / f.row_tiles;
        if (ov.filter_splits > 0) want = ov.filter_splits;
        if (want > code_tiles) want = code_tiles;
        if (want > 16) want = 16;
        if (want < 1) want = 1;
        const long tiles_per_split = (code_tiles + want - 1) / want;
        f.codes_per_split = (int)(tiles_per_split * R64_BM);
        f.splits = (int)((code_tiles + tiles_per_split - 1) / tiles_per_split);
        f.own_total = f.splits * (f.rows64_wide ? F_OWN_PER_SPLIT : R64N_OWN_PER_SPLIT);
        f.xcd_rows = 0;
        f.main_tiles = f.row_tiles; f.tail_splits = 0; f.tail_codes_per_split = 0; f.own_tail = 0;
        return f;
    }
    // every split adds 4 candidate lists per row, so split only as far as filling the chip needs
    // one 8-wave block per CU: a launch of B equal blocks wastes up to one round of 256
    long want = f.row_tiles >= 4L * di.cus ? 2 : (4L * di.cus + f.row_tiles - 1) / f.row_tiles;
    // With >= 1024 row tiles the blocks are ordered XCD-aware (see filter_f16_kernel): the 32 CUs of an XCD share
    // 32/splits x tiles, which then stay in its 4 MB L2 (393 KB each at D = 768) instead of being re-streamed from
    // the Infinity Cache once per code tile (measured at N = 600k, K = 49152: 6-8 % on the kernel).  Two splits: more
    // would keep more of the x tiles resident but loosen the per-split thresholds (4: +3.6 %, 8: +7.7 % kernel time).
    bool xcd = di.full && f.row_tiles >= 1024;      // (the block order below is written for 8 XCDs of 32 CUs)
    if (ov.filter_splits > 0) want = ov.filter_splits;
    if (ov.filter_xcd >= 0) xcd = di.full && ov.filter_xcd != 0;
    if (want > code_tiles) want = code_tiles;
    if (want > 16) want = 16;      // (more splits for small batches were measured: slower from 32 up)
    if (want < 1) want = 1;
    const long tiles_per_split = (code_tiles + want - 1) / want;
    f.codes_per_split = (int)(tiles_per_split * F_BM);
    f.splits = (int)((code_tiles + tiles_per_split - 1) / tiles_per_split);
    f.own_total = f.splits * F_OWN_PER_SPLIT;
    f.xcd_rows = (xcd && f.splits >= 2 && 32 % f.splits == 0) ? 32 / f.splits : 0;
    // One 8-wave block per CU and equal-cost blocks: B blocks take ceil(B / 256) rounds, the last one however few blocks it
    // holds (N = 600k: 4688 blocks = 18.3 rounds -> 19).  The row tiles of that last round go into a second launch
    // with enough splits to be ONE round of short blocks (N = 600k: 40 row tiles x 6 splits, a third of a round).
    f.main_tiles = f.row_tiles; f.tail_splits = 0; f.tail_codes_per_split = 0; f.own_tail = 0;
    const long blocks = f.row_tiles * f.splits;
    long tail_min_blocks = 8L * di.cus;
    if (ov.filter_tail_min_blocks >= 0)
        tail_min_blocks = ov.filter_tail_min_blocks > 0 ? ov.filter_tail_min_blocks : (1L << 60);   // 0 = off
    const bool tail = blocks >= tail_min_blocks && f.splits <= 4 && code_tiles >= 4L * f.splits;
    if (tail) {
        const long round = (long)di.cus / f.splits * f.splits;     // one block per CU, whole row tiles (splits <= 4)
        const long main_blocks = blocks / round * round;
        const long main_tiles = main_blocks / f.splits;
        // as many splits as make the tail ONE round of short blocks (at most 16: every split costs candidates)
        const long tail_tiles = f.row_tiles - main_tiles;
        const long ts = tail_tiles > 0 ? lmin(lmin(16, code_tiles), round / tail_tiles) : 0;
        if (ts >= 2L * f.splits) {
            const long tps = (code_tiles + ts - 1) / ts;
            f.main_tiles = main_tiles;
            f.tail_codes_per_split = (int)(tps * F_BM);
            f.tail_splits = (int)((code_tiles + tps - 1) / tps);
            f.own_tail = f.tail_splits * F_OWN_PER_SPLIT;
        }
    }
    return f;
}

struct FilterWs {
    _Float16 *xh, *wh;
    float *en_max, *wsqp, *dump;
    uint2 *cand, *cand_tail;
    int *cnt_tail;
    int *cand_cnt, *fb_count, *fb_rows, *fb_pidx;
    float *fb_pval;
    size_t total;
};

// Exact redo of the few rows the filter gives up on: the first FB_ROWS list entries are searched with the
// code range split FB_SPLITS ways (a handful of rows would otherwise crawl through all K codes in one
// block); anything beyond that many rows is plentiful enough for the plain one-block-per-128-rows form.
constexpr int FB_ROWS = 8192, FB_SPLITS = 32;

static FilterWs filter_ws_layout(void *ws, int64_t n, const FilterPlan &f)
{
    FilterWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = ws ? (char *)ws + off : nullptr; off += align_up(bytes, 256); return (void *)p; };
    w.xh = (_Float16 *)take((size_t)f.n_pad * f.dp * 2);
    w.wh = (_Float16 *)take((size_t)f.k_pad * f.dp * 2);
    w.en_max = (float *)take(4);
    w.wsqp = (float *)take((size_t)f.k_pad * 4);
    w.fb_count = (int *)take(4);
    w.fb_rows = (int *)take((size_t)n * 4);
    w.fb_pval = (float *)take((size_t)FB_SPLITS * FB_ROWS * MEDTOK_MAX_TOPK * 4);
    w.fb_pidx = (int *)take((size_t)FB_SPLITS * FB_ROWS * MEDTOK_MAX_TOPK * 4);
    const size_t n_main = (size_t)lmin(n, f.main_tiles * f.row_bn), n_tail = (size_t)n - n_main;
    w.cand_cnt = (int *)take(n_main * f.own_total * 4);
    w.cand = (uint2 *)take(n_main * f.own_total * F_CAP * 8);
    w.cnt_tail = (int *)take(n_tail * f.own_tail * 4);
    w.cand_tail = (uint2 *)take(n_tail * f.own_tail * F_CAP * 8);
    w.dump = nullptr;
    w.total = off;
    return w;
}

static bool filter_eligible(int64_t n, int64_t k_codes, int d, int topk)
{
    // the filter path has ~0.1-0.2 ms of fixed cost (operand conversion, three extra launches); measured
    // crossover against the exact kernel on MI355X is around 1e10 flop (tools/path_crossover.py)
    // (k_codes * d < 2^29: the kernel addresses a code split's fp16 image with a 32-bit buffer offset)
    const double flops = 2.0 * (double)n * (double)k_codes * (double)d;
    if (!(n >= 512 && k_codes >= 1024 && flops >= 1.0e10 && topk <= 8 && n < (1ll << 31) &&
          (double)k_codes * (double)(d + 64) < 536870912.0))
        return false;
    // Few rows: the filter has a floor -- converting the codebook, one block walking its share of the codes, the re-score and the
    // (empty) redo launches -- while the exact kernel runs at about 100 TFLOP/s from 512 rows up once D is a few hundred.  Floor
    // fitted in round 6 on MI355X (tools/r06/crossover_small.py, profiles/r06_path_crossover_small.txt: D = 768 / 256, K = 8192 ..
    // 49152, N = 128 .. 4096): 61.5 us + 0.70 us per 256-code tile and k block of a 16th of the codebook + 1.04 ns per code; the
    // round-2 floor (170 us + 1.1 us per unit) sent 512 x 49152 and 1024 x 16384 searches at D = 768 to the exact kernel, which
    // takes 0.38 / 0.24 ms there against 0.32 / 0.15 ms.  5 % in favour of the exact kernel (fewer launches).
    if (d >= 256) {
        const double filter_floor = 61.5e-6 + ((double)k_codes / 4096.0) * ((double)d / 32.0) * 0.703e-6 + (double)k_codes * 1.04e-9;
        if (flops / 1.0e14 + 20.0e-6 < 1.05 * filter_floor) return false;
    }
    return true;
}

static size_t wide_head_bytes(int64_t n);

static int resolve_path(int path, int64_t n, int64_t k_codes, int d, int topk)
{
    path = path_id(path);
    if (topk > 8) return MEDTOK_PATH_F32_MFMA;          // (the shortlist kernels keep lists of at most 8: search_wide)
    if (path == MEDTOK_PATH_AUTO) return filter_eligible(n, k_codes, d, topk) ? MEDTOK_PATH_F16_FILTER : MEDTOK_PATH_F32_MFMA;
    if (path == MEDTOK_PATH_F16_FILTER && (double)k_codes * (double)(d + 64) >= 536870912.0) return MEDTOK_PATH_F32_MFMA;
    return path;
}

extern "C" size_t medtok_search_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path)
{
    if (n <= 0 || k_codes <= 0 || topk < 1 || topk > MEDTOK_MAX_TOPK) return 0;
    if (topk > 8)            // two passes of the exact kernel with lists of 8 + the join's buffers
        return wide_head_bytes(n) + medtok_search_workspace_bytes(n, k_codes, d, 8, (path & ~MEDTOK_PATH_MASK) | MEDTOK_PATH_F32_MFMA);
    const PlanOverride ov = decode_plan(path);
    if (resolve_path(path, n, k_codes, d, topk) == MEDTOK_PATH_F16_FILTER)
        return filter_ws_layout(nullptr, n, plan_filter(n, k_codes, d, topk, ov)).total;
    SearchPlan p = plan_search(n, k_codes, topk, ov);
    if (p.tail_splits > 0) {
        const size_t tail_rows = (size_t)(n - p.main_tiles * S_BN);
        return align_up((size_t)p.tail_splits * tail_rows * p.tslots * sizeof(float), 256) +
               align_up((size_t)p.tail_splits * tail_rows * p.tslots * sizeof(int), 256);
    }
    if (p.splits == 1) return 256;
    return align_up((size_t)p.splits * n * p.tslots * sizeof(float), 256) +
           align_up((size_t)p.splits * n * p.tslots * sizeof(int), 256);
}

template <int T, bool KTAIL>
static int launch_search(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq,
                         int64_t k_codes, int d, int topk, int64_t *idx, float *dist, void *ws, size_t ws_bytes,
                         const SearchPlan &p, hipStream_t s, const float *excl_d = nullptr, const int64_t *excl_i = nullptr, int excl_stride = 0)
{
    dim3 grid((unsigned)p.row_tiles, (unsigned)p.splits), block(256);
    hipEvent_t pa = prof_wanted(1) ? prof_mark(s) : nullptr;
    const double pflops = 2.0 * (double)n * (double)k_codes * (double)d;
    // one launch of the exact kernel: FINAL (a block walks all codes and writes the row's list) or partial lists per code split;
    // with excl_d the second-pass form that skips a row's first-pass codes (lists of 8 only)
    auto go = [&](bool final_, dim3 g, int cps, float *pv, int *pi, int64_t *oi, float *od, int lb, int le) {
        if (excl_d) {
            if constexpr (T == 8) {
                if (final_) {
                    (void)set_lds_once<search_f32_excl_kernel<8, true, KTAIL>>(S_LDS_BYTES);
                    hipLaunchKernelGGL((search_f32_excl_kernel<8, true, KTAIL>), g, block, S_LDS_BYTES, s, xhat, xsq, what, wsq, (long)n, (int)k_codes, d, cps, topk,
                                       pv, pi, oi, od, lb, le, excl_d, excl_i, excl_stride);
                } else {
                    (void)set_lds_once<search_f32_excl_kernel<8, false, KTAIL>>(S_LDS_BYTES);
                    hipLaunchKernelGGL((search_f32_excl_kernel<8, false, KTAIL>), g, block, S_LDS_BYTES, s, xhat, xsq, what, wsq, (long)n, (int)k_codes, d, cps, topk,
                                       pv, pi, oi, od, lb, le, excl_d, excl_i, excl_stride);
                }
            }
            return;
        }
        if (final_) {
            (void)set_lds_once<search_f32_kernel<T, true, KTAIL, false>>(S_LDS_BYTES);
            hipLaunchKernelGGL((search_f32_kernel<T, true, KTAIL, false>), g, block, S_LDS_BYTES, s, xhat, xsq, what, wsq, (long)n, (int)k_codes, d, cps, topk,
                               pv, pi, oi, od, (const int *)nullptr, (const int *)nullptr, lb, le);
        } else {
            (void)set_lds_once<search_f32_kernel<T, false, KTAIL, false>>(S_LDS_BYTES);
            hipLaunchKernelGGL((search_f32_kernel<T, false, KTAIL, false>), g, block, S_LDS_BYTES, s, xhat, xsq, what, wsq, (long)n, (int)k_codes, d, cps, topk,
                               pv, pi, oi, od, (const int *)nullptr, (const int *)nullptr, lb, le);
        }
    };
    if (excl_d && T != 8) return fail("search: the exclusion pass runs with lists of 8");
    if (p.tail_splits > 0) {
        // whole rounds of unsplit blocks, then the last round's row tiles with their own code splits + merge
        const long tail_start = p.main_tiles * S_BN, tail_rows = n - tail_start;
        const size_t vb = align_up((size_t)p.tail_splits * tail_rows * T * sizeof(float), 256);
        const size_t ib = align_up((size_t)p.tail_splits * tail_rows * T * sizeof(int), 256);
        if (!ws || ws_bytes < vb + ib) return fail("search: workspace too small (%zu < %zu)", ws_bytes, vb + ib);
        float *pval = (float *)ws;
        int *pidx = (int *)((char *)ws + vb);
        go(true, dim3((unsigned)p.main_tiles, 1), p.codes_per_split, (float *)nullptr, (int *)nullptr, idx, dist, 0, 0);
        go(false, dim3((unsigned)(p.row_tiles - p.main_tiles), (unsigned)p.tail_splits), p.tail_codes_per_split, pval, pidx, (int64_t *)nullptr, (float *)nullptr,
           (int)tail_start, (int)tail_rows);
        if (pa) prof_push(pa, prof_mark(s), pflops, 1);
        if (check_launch("search_f32(main + tail)")) return 1;
        hipLaunchKernelGGL((merge_topk_kernel<T>), dim3((unsigned)((tail_rows + 31) / 32)), dim3(256), 0, s, pval, pidx, tail_rows,
                           p.tail_splits, topk, idx + tail_start * topk, dist + tail_start * topk, (const int *)nullptr, (const int *)nullptr,
                           (int)k_codes);
        return check_launch("merge_topk(tail)");
    }
    if (p.splits == 1) {
        go(true, grid, p.codes_per_split, (float *)nullptr, (int *)nullptr, idx, dist, 0, 0);
        if (pa) prof_push(pa, prof_mark(s), pflops, 1);
        return check_launch("search_f32");
    }
    const size_t vbytes = align_up((size_t)p.splits * n * T * sizeof(float), 256);
    const size_t ibytes = align_up((size_t)p.splits * n * T * sizeof(int), 256);
    if (!ws || ws_bytes < vbytes + ibytes) return fail("search: workspace too small (%zu < %zu)", ws_bytes, vbytes + ibytes);
    float *pval = (float *)ws;
    int *pidx = (int *)((char *)ws + vbytes);
    go(false, grid, p.codes_per_split, pval, pidx, (int64_t *)nullptr, (float *)nullptr, 0, 0);
    if (pa) prof_push(pa, prof_mark(s), pflops, 1);
    if (check_launch("search_f32(split)")) return 1;
    if (p.splits >= 64 && n <= 8192)     // few rows, many lists each: a wave per row
        hipLaunchKernelGGL((merge_topk_kernel<T, 64>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, pval, pidx, (long)n,
                           p.splits, topk, idx, dist, (const int *)nullptr, (const int *)nullptr, (int)k_codes);
    else
        hipLaunchKernelGGL((merge_topk_kernel<T>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, pval, pidx, (long)n,
                           p.splits, topk, idx, dist, (const int *)nullptr, (const int *)nullptr, (int)k_codes);
    return check_launch("merge_topk");
}

template <int T>
static int launch_search_t(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq,
                           int64_t k_codes, int d, int topk, int64_t *idx, float *dist, void *ws, size_t ws_bytes,
                           const SearchPlan &p, hipStream_t s, const float *excl_d = nullptr, const int64_t *excl_i = nullptr, int excl_stride = 0)
{
    if (d % S_BK) return launch_search<T, true>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, p, s, excl_d, excl_i, excl_stride);
    return launch_search<T, false>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, p, s, excl_d, excl_i, excl_stride);
}

// ---- more than 8 codes per row (k = 9 .. MEDTOK_MAX_TOPK; the reference takes any k, vector_quantization_soft_one_new.py:91,157,203):
// the lane-local lists of the kernels hold 8 entries, so the search runs as TWO passes of the exact kernel -- the 8 best codes, then
// the best k - 8 among the codes behind the row's 8th (distance, index) pair -- and a join.  Same total order (distance, then
// lowest index) as one list of k: bit-identical to the oracle's top-k.  Always the exact fp32 path (the shortlist kernels keep lists
// of at most 8).
constexpr int WIDE_T = 8;
static size_t wide_head_bytes(int64_t n)
{
    return 2 * align_up((size_t)n * WIDE_T * sizeof(int64_t), 256) + 2 * align_up((size_t)n * WIDE_T * sizeof(float), 256);
}

__global__ __launch_bounds__(256) void join_lists_kernel(const int64_t *__restrict__ ia, const float *__restrict__ da, const int64_t *__restrict__ ib,
                                                         const float *__restrict__ db, long n, int kb, int64_t *__restrict__ idx, float *__restrict__ dist)
{
    const int k = WIDE_T + kb;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * k; i += (long)gridDim.x * 256) {
        const long r = i / k;
        const int j = (int)(i - r * k);
        idx[i] = j < WIDE_T ? ia[r * WIDE_T + j] : ib[r * kb + (j - WIDE_T)];
        dist[i] = j < WIDE_T ? da[r * WIDE_T + j] : db[r * kb + (j - WIDE_T)];
    }
}

static int search_wide(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq, int64_t k_codes, int d, int topk,
                       int64_t *idx, float *dist, void *ws, size_t ws_bytes, int path, hipStream_t s)
{
    const PlanOverride ov = decode_plan(path);
    const SearchPlan p = plan_search(n, k_codes, WIDE_T, ov);
    const size_t head = wide_head_bytes(n);
    if (!ws || ws_bytes < head) return fail("search(k > 8): workspace too small (%zu < %zu)", ws_bytes, head);
    char *base = (char *)ws;
    int64_t *ia = (int64_t *)base;
    int64_t *ib = (int64_t *)(base + align_up((size_t)n * WIDE_T * sizeof(int64_t), 256));
    float *da = (float *)(base + 2 * align_up((size_t)n * WIDE_T * sizeof(int64_t), 256));
    float *db = (float *)((char *)da + align_up((size_t)n * WIDE_T * sizeof(float), 256));
    void *inner = base + head;
    const size_t inner_bytes = ws_bytes - head;
    if (launch_search_t<WIDE_T>(xhat, xsq, n, what, wsq, k_codes, d, WIDE_T, ia, da, inner, inner_bytes, p, s)) return 1;
    const int kb = topk - WIDE_T;
    if (launch_search_t<WIDE_T>(xhat, xsq, n, what, wsq, k_codes, d, kb, ib, db, inner, inner_bytes, p, s, da + (WIDE_T - 1), ia + (WIDE_T - 1), WIDE_T)) return 1;
    hipLaunchKernelGGL(join_lists_kernel, dim3((unsigned)lmin(4096, (n * topk + 255) / 256)), dim3(256), 0, s, ia, da, ib, db, (long)n, kb, idx, dist);
    return check_launch("join_lists");
}

// Handed by medtok_soft_vq_forward_f32 to its search call: when the filter path runs, its re-score kernel also does the
// soft assignment (and the exact-path leftovers get it from soft_assign_kernel through the row list).
struct FuseAssign { const float *xref; float *w; float *zq; long zq_stride; bool done; const int *fb_rows, *fb_count; bool xh_done;
                    // a codebook region prepared once per weight version (medtok_codebook_prepare_f32): its fp16 image, its padded start
                    // values and its largest squared norm -- the search then skips its own passes over the codebook
                    const _Float16 *p_wh = nullptr; const float *p_wsqp = nullptr; const float *p_en_max = nullptr; bool fb_zeroed = false; };

template <int T, bool KTAIL>
static int launch_filter(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq,
                         int64_t k_codes, int d, int topk, int64_t *idx, float *dist, void *ws, size_t ws_bytes,
                         hipStream_t s, FuseAssign *fuse, const PlanOverride &ov)
{
    const FilterPlan f = plan_filter(n, k_codes, d, topk, ov);
    const FilterWs w = filter_ws_layout(ws, n, f);
    if (!ws || ws_bytes < w.total) return fail("search(filter): workspace too small (%zu < %zu)", ws_bytes, w.total);
    const bool prep = fuse && fuse->p_wh && fuse->p_wsqp && fuse->p_en_max;
    const _Float16 *const c_wh = prep ? fuse->p_wh : w.wh;
    const float *const c_wsqp = prep ? fuse->p_wsqp : w.wsqp, *const c_en_max = prep ? fuse->p_en_max : w.en_max;
    if (!(fuse && fuse->xh_done))      // (the one-call forward's rownorm has already written the fp16 image of x)
        hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.n_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, xhat, (long)n, d, f.n_pad, f.dp, w.xh,
                           prep ? w.fb_count : (int *)nullptr);
    else if (prep && !fuse->fb_zeroed && hipMemsetAsync(w.fb_count, 0, 4, s) != hipSuccess)
        return fail("search(filter): memset failed");
    if (!prep) {
        hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.k_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, what, (long)k_codes, d, f.k_pad, f.dp, w.wh);
        hipLaunchKernelGGL(wsq_max_kernel, dim3(1), dim3(1024), 0, s, wsq, (int)k_codes, w.en_max, w.wsqp, (int)f.k_pad, w.fb_count);
    }
    hipEvent_t pa = prof_wanted(0) ? prof_mark(s) : nullptr;
    if (f.rows64 && !f.rows64_wide) {
        hipLaunchKernelGGL((filter_rows64n_kernel<T>), dim3((unsigned)f.row_tiles, (unsigned)f.splits), dim3(R64N_THREADS), R64N_SMEM_BYTES, s,
                           w.xh, c_wh, xsq, c_wsqp, c_en_max, (long)n, (int)k_codes, d, f.codes_per_split, f.own_total, w.cand, w.cand_cnt);
    } else if (f.rows64) {
        (void)set_lds_once<filter_rows64_kernel<T>>(R64_SMEM_BYTES);
        hipLaunchKernelGGL((filter_rows64_kernel<T>), dim3((unsigned)f.row_tiles, (unsigned)f.splits), dim3(R64_THREADS), R64_SMEM_BYTES, s,
                           w.xh, c_wh, xsq, c_wsqp, c_en_max, (long)n, (int)k_codes, d, f.codes_per_split, f.own_total, w.cand, w.cand_cnt);
    } else {
        (void)set_lds_once<filter_f16_kernel<T, false>>(F_SMEM_BYTES);
        dim3 fgrid((unsigned)f.main_tiles, (unsigned)f.splits);
        if (f.xcd_rows) fgrid = dim3((unsigned)(((f.main_tiles + 8 * f.xcd_rows - 1) / (8 * f.xcd_rows)) * 256), 1);
        hipLaunchKernelGGL((filter_f16_kernel<T, false>), fgrid, dim3(F_THREADS), F_SMEM_BYTES, s,
                           w.xh, c_wh, xsq, c_wsqp, c_en_max, (long)n, (int)k_codes, f.dp, d, f.codes_per_split, f.own_total,
                           w.cand, w.cand_cnt, (float *)nullptr, f.xcd_rows, f.splits, 0, (int)f.main_tiles);
    }
    const long tail_start = f.main_tiles * f.row_bn;
    if (f.main_tiles < f.row_tiles) {
        // the kernel indexes its lists by absolute row: bias the tail region's base pointers accordingly
        hipLaunchKernelGGL((filter_f16_kernel<T, false>), dim3((unsigned)(f.row_tiles - f.main_tiles), (unsigned)f.tail_splits), dim3(F_THREADS),
                           F_SMEM_BYTES, s, w.xh, c_wh, xsq, c_wsqp, c_en_max, (long)n, (int)k_codes, f.dp, d, f.tail_codes_per_split, f.own_tail,
                           w.cand_tail - tail_start * f.own_tail * F_CAP, w.cnt_tail - tail_start * f.own_tail, (float *)nullptr, 0, f.tail_splits,
                           (int)f.main_tiles, (int)f.row_tiles);
    }
    if (pa) prof_push(pa, prof_mark(s), 2.0 * (double)n * (double)k_codes * (double)d, 0);
    if (check_launch("filter_f16")) return 1;
    // few rows: one wavefront per row (latency-bound: rescore_wave_kernel); many rows: 32-row blocks
#define MEDTOK_RESCORE_ARGS                                                                                                      \
    w.cand, w.cand_cnt, f.own_total, w.cand_tail, w.cnt_tail, f.own_tail, f.main_tiles < f.row_tiles ? tail_start : (long)n,     \
    xhat, xsq, what, wsq, c_en_max, (long)n, (int)k_codes, d, topk, idx, dist, w.fb_count, w.fb_rows,                             \
    fuse ? fuse->xref : (const float *)nullptr, fuse ? fuse->w : (float *)nullptr,                                                \
    fuse ? fuse->zq : (float *)nullptr, fuse ? fuse->zq_stride : 0L
    if (n < 32L * 4 * dev_info().cus && f.own_total <= 64 && f.own_tail <= 64)
        hipLaunchKernelGGL((rescore_wave_kernel<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, MEDTOK_RESCORE_ARGS);
    else
        hipLaunchKernelGGL((rescore_kernel<T, 32>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, MEDTOK_RESCORE_ARGS);
#undef MEDTOK_RESCORE_ARGS
    if (check_launch("rescore")) return 1;
    // exact redo of the rows the filter gave up on (normally none: every block exits on *fb_count)
    const long code_tiles = (k_codes + S_BM - 1) / S_BM;
    const int fb_splits = (int)lmin(FB_SPLITS, code_tiles);
    const int fb_cps = (int)((code_tiles + fb_splits - 1) / fb_splits * S_BM);
    const int fb_nsplit = (int)((code_tiles * S_BM + fb_cps - 1) / fb_cps);
    const int head = (int)lmin(n, FB_ROWS);
    (void)set_lds_once<search_f32_kernel<T, false, KTAIL, true>>(S_LDS_BYTES);
    hipLaunchKernelGGL((search_f32_kernel<T, false, KTAIL, true>), dim3((unsigned)((head + S_BN - 1) / S_BN), (unsigned)fb_nsplit), dim3(256), S_LDS_BYTES, s,
                       xhat, xsq, what, wsq, (long)n, (int)k_codes, d, fb_cps, topk, w.fb_pval, w.fb_pidx, (int64_t *)nullptr, (float *)nullptr,
                       (const int *)w.fb_rows, (const int *)w.fb_count, 0, head);
    hipLaunchKernelGGL((merge_topk_kernel<T>), dim3((unsigned)((head + 31) / 32)), dim3(256), 0, s, w.fb_pval, w.fb_pidx, (long)head,
                       fb_nsplit, topk, idx, dist, (const int *)w.fb_rows, (const int *)w.fb_count, (int)k_codes);
    if (n > head) {
        (void)set_lds_once<search_f32_kernel<T, true, KTAIL, true>>(S_LDS_BYTES);
        hipLaunchKernelGGL((search_f32_kernel<T, true, KTAIL, true>), dim3((unsigned)((n - head + S_BN - 1) / S_BN), 1), dim3(256), S_LDS_BYTES, s,
                           xhat, xsq, what, wsq, (long)n, (int)k_codes, d, (int)(code_tiles * S_BM), topk,
                           (float *)nullptr, (int *)nullptr, idx, dist, (const int *)w.fb_rows, (const int *)w.fb_count, head, (int)n);
    }
    if (fuse) { fuse->done = true; fuse->fb_rows = w.fb_rows; fuse->fb_count = w.fb_count; }
    return check_launch("search_f32(fallback)");
}

template <int T>
static int launch_filter_t(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq,
                           int64_t k_codes, int d, int topk, int64_t *idx, float *dist, void *ws, size_t ws_bytes, hipStream_t s,
                           FuseAssign *fuse, const PlanOverride &ov)
{
    if (d % S_BK) return launch_filter<T, true>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, s, fuse, ov);
    return launch_filter<T, false>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, s, fuse, ov);
}

// `fuse` (internal callers only): when the filter path runs, its re-score kernel also does the soft assignment described there
static int search_impl(const float *xhat, const float *xsq, int64_t n, const float *what,
                       const float *wsq, int64_t k_codes, int d, int topk, int64_t *idx, float *dist,
                       void *ws, size_t ws_bytes, int path, void *stream, FuseAssign *fuse)
{
    if (n < 0 || k_codes <= 0 || d <= 0 || (d & 3)) return fail("search: bad shape n=%ld K=%ld d=%d (d %% 4 == 0)", (long)n, (long)k_codes, d);
    if (topk < 1 || topk > MEDTOK_MAX_TOPK || topk > k_codes) return fail("search: topk=%d unsupported (1..%d, <= K)", topk, MEDTOK_MAX_TOPK);
    if (k_codes >= (1ll << 31)) return fail("search: K too large");
    if (path_id(path) != MEDTOK_PATH_AUTO && path_id(path) != MEDTOK_PATH_F32_MFMA && path_id(path) != MEDTOK_PATH_F16_FILTER) return fail("search: unknown path %d", path_id(path));
    const PlanOverride ov = decode_plan(path);
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (topk > 8) return search_wide(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, path, s);
    const int tslots = topk == 1 ? 1 : (topk <= 5 ? 5 : 8);
    if (resolve_path(path, n, k_codes, d, topk) == MEDTOK_PATH_F16_FILTER) {
        if (n >= (1ll << 31)) return fail("search(filter): n too large");
        switch (tslots) {
        case 1: return launch_filter_t<1>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, s, fuse, ov);
        case 5: return launch_filter_t<5>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, s, fuse, ov);
        default: return launch_filter_t<8>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, s, fuse, ov);
        }
    }
    const SearchPlan p = plan_search(n, k_codes, topk, ov);
    switch (tslots) {
    case 1: return launch_search_t<1>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, p, s);
    case 5: return launch_search_t<5>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, p, s);
    default: return launch_search_t<8>(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, p, s);
    }
}

extern "C" int medtok_topk_search_f32(const float *xhat, const float *xsq, int64_t n, const float *what,
                                      const float *wsq, int64_t k_codes, int d, int topk, int64_t *idx, float *dist,
                                      void *ws, size_t ws_bytes, int path, void *stream)
{
    return search_impl(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, path, stream, nullptr);
}

// DEV probe (tools/r04/filter_probe.py): the filter kernel of one search with per-wave cycle counts of its loop segments written to
// `probe` (uint64 [blocks][8][8] -- [blocks][4][8] for D <= 64, whose kernel has four waves; blocks = *n_blocks on return).  Same plan as the search would take, main launch only; the
// candidate lists it writes into `ws` are discarded.
extern "C" int medtok_debug_filter_probe(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq, int64_t k_codes, int d,
                                         int topk, void *ws, size_t ws_bytes, void *probe, size_t probe_bytes, int64_t *n_blocks, void *stream)
{
    if (topk < 2 || topk > 5) return fail("filter_probe: topk 2..5");
    // (at D <= 64 the timed instantiation is the 128 x 64 wave-tile kernel's: size the workspace with MEDTOK_PLAN_FILTER_ROWS64_WIDE)
    const PlanOverride ov = decode_plan(MEDTOK_PATH_F16_FILTER | MEDTOK_PLAN_FILTER_ROWS64_WIDE);
    const FilterPlan f = plan_filter(n, k_codes, d, topk, ov);
    const FilterWs w = filter_ws_layout(ws, n, f);
    if (!ws || ws_bytes < w.total) return fail("filter_probe: workspace too small (%zu < %zu)", ws_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.n_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, xhat, (long)n, d, f.n_pad, f.dp, w.xh);
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.k_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, what, (long)k_codes, d, f.k_pad, f.dp, w.wh);
    hipLaunchKernelGGL(wsq_max_kernel, dim3(1), dim3(1024), 0, s, wsq, (int)k_codes, w.en_max, (float *)nullptr, 0, (int *)nullptr);
    hipLaunchKernelGGL(pad_wsq_kernel, dim3((unsigned)((f.k_pad + 255) / 256)), dim3(256), 0, s, wsq, (int)k_codes, (int)f.k_pad, w.wsqp);
    if (f.rows64) {         // D <= 64: uint64 [blocks][4][8]
        const size_t blocks = (size_t)f.row_tiles * f.splits;
        if (probe_bytes < blocks * 256) return fail("filter_probe: probe buffer too small (%zu < %zu)", probe_bytes, blocks * 256);
        if (hipMemsetAsync(probe, 0, blocks * 256, s) != hipSuccess) return fail("filter_probe: memset failed");
        (void)set_lds_once<filter_rows64_kernel<5, true>>(R64_SMEM_BYTES);
        hipLaunchKernelGGL((filter_rows64_kernel<5, true>), dim3((unsigned)f.row_tiles, (unsigned)f.splits), dim3(R64_THREADS), R64_SMEM_BYTES, s,
                           w.xh, w.wh, xsq, w.wsqp, w.en_max, (long)n, (int)k_codes, d, f.codes_per_split, f.own_total, w.cand, w.cand_cnt,
                           (unsigned long long *)probe);
        if (n_blocks) *n_blocks = (int64_t)blocks;
        return check_launch("filter_probe(rows64)");
    }
    (void)set_lds_once<filter_f16_kernel<5, false, true>>(F_SMEM_BYTES);
    dim3 fgrid((unsigned)f.main_tiles, (unsigned)f.splits);
    if (f.xcd_rows) fgrid = dim3((unsigned)(((f.main_tiles + 8 * f.xcd_rows - 1) / (8 * f.xcd_rows)) * 256), 1);
    const size_t blocks = (size_t)fgrid.x * fgrid.y;
    if (probe_bytes < blocks * 8 * 8 * 8) return fail("filter_probe: probe buffer too small (%zu < %zu)", probe_bytes, blocks * 512);
    if (hipMemsetAsync(probe, 0, blocks * 512, s) != hipSuccess) return fail("filter_probe: memset failed");
    hipLaunchKernelGGL((filter_f16_kernel<5, false, true>), fgrid, dim3(F_THREADS), F_SMEM_BYTES, s,
                       w.xh, w.wh, xsq, w.wsqp, w.en_max, (long)n, (int)k_codes, f.dp, d, f.codes_per_split, f.own_total,
                       w.cand, w.cand_cnt, (float *)probe, f.xcd_rows, f.splits, 0, (int)f.main_tiles);
    if (n_blocks) *n_blocks = (int64_t)blocks;
    return check_launch("filter_probe");
}

// Test hook: byte offset, inside a filter-path search workspace, of the int32 count of rows the filter handed to the exact kernel
// (candidate list overflow, norms outside the bound's range, NaN) -- tests and tools read it after a search; (size_t)-1 if the shape
// does not take the filter path.
extern "C" size_t medtok_debug_filter_fallback_count_offset(int64_t n, int64_t k_codes, int d, int topk, int path)
{
    if (n <= 0 || k_codes <= 0 || resolve_path(path, n, k_codes, d, topk) != MEDTOK_PATH_F16_FILTER) return (size_t)-1;
    const FilterWs w = filter_ws_layout((void *)256, n, plan_filter(n, k_codes, d, topk, decode_plan(path)));
    return (size_t)((char *)w.fb_count - (char *)256);
}

// Measurement hook (bench.py --data ..., tests): what the filter pass of a finished search left in its workspace -- how many candidates
// it shortlisted, how many of a row's lists are full, how many rows it handed to the exact kernel.  `ws` is the workspace of the
// search call (soft_vq_ws != 0: of a medtok_soft_vq_forward*_f32 call, whose search workspace starts behind |x|^2), read after
// the call on the same stream.  out[0] = candidates over all rows and lists, out[1] = lists at or over capacity, out[2] = rows
// handed to the exact kernel, out[3] = lists in total (rows x owners).
__global__ __launch_bounds__(256) void filter_stats_kernel(const int *__restrict__ cnt, long lists, const int *__restrict__ cnt_tail, long lists_tail,
                                                           const int *__restrict__ fb_count, unsigned long long *__restrict__ out)
{
    unsigned long long cand = 0, full = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < lists + lists_tail; i += (long)gridDim.x * 256) {
        const int c = i < lists ? cnt[i] : cnt_tail[i - lists];
        cand += (unsigned long long)min(c, F_CAP);
        full += c >= F_CAP ? 1ull : 0ull;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cand += __shfl_xor(cand, off, 64); full += __shfl_xor(full, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], cand); atomicAdd(&out[1], full); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[2] = (unsigned long long)*fb_count; out[3] = (unsigned long long)(lists + lists_tail); }
}

extern "C" int medtok_debug_filter_stats(const void *ws, size_t ws_bytes, int soft_vq_ws, int64_t n, int64_t k_codes, int d, int topk, int path,
                                         uint64_t *out, void *stream)
{
    if (!ws || !out || n <= 0 || k_codes <= 0) return fail("filter_stats: bad arguments");
    if (topk < 1 || topk > 8 || resolve_path(path, n, k_codes, d, topk) != MEDTOK_PATH_F16_FILTER) return fail("filter_stats: this shape does not take the filter path");
    const size_t head = soft_vq_ws ? align_up((size_t)n * 4, 256) : 0;
    const FilterPlan f = plan_filter(n, k_codes, d, topk, decode_plan(path));
    const FilterWs w = filter_ws_layout((char *)const_cast<void *>(ws) + head, n, f);
    if (ws_bytes < head + w.total) return fail("filter_stats: workspace too small (%zu < %zu)", ws_bytes, head + w.total);
    const long n_main = (long)lmin(n, f.main_tiles * f.row_bn), n_tail = (long)n - n_main;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 32, s) != hipSuccess) return fail("filter_stats: memset failed");
    const long lists = n_main * f.own_total, lists_tail = n_tail * f.own_tail;
    hipLaunchKernelGGL(filter_stats_kernel, dim3((unsigned)lmin(1024, (lists + lists_tail + 255) / 256)), dim3(256), 0, s, w.cand_cnt, lists, w.cnt_tail, lists_tail,
                       w.fb_count, (unsigned long long *)out);
    return check_launch("filter_stats");
}

// Test hook: the filter's approximate scores s~ [n, k_codes] (same MFMA sequence as the search uses),
// so tests can measure |s~ - s| against the bound filter_f16.h assumes.
extern "C" size_t medtok_debug_filter_scores_workspace_bytes(int64_t n, int64_t k_codes, int d)
{
    if (n <= 0 || k_codes <= 0 || d <= 0) return 0;
    PlanOverride gen; gen.filter_rows64 = 0;     // (the score dump is an instantiation of the general kernel: same MFMA order at any D)
    FilterPlan f = plan_filter(n, k_codes, d, 5, gen);
    return align_up((size_t)f.n_pad * f.dp * 2, 256) + align_up((size_t)f.k_pad * f.dp * 2, 256) + 512 + align_up((size_t)f.k_pad * 4, 256);
}

extern "C" int medtok_debug_filter_scores_f32(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq,
                                              int64_t k_codes, int d, float *scores, void *ws, size_t ws_bytes, void *stream)
{
    if (n <= 0 || k_codes <= 0 || d <= 0 || (d & 3)) return fail("debug_filter_scores: bad shape");
    PlanOverride gen; gen.filter_rows64 = 0;
    FilterPlan f = plan_filter(n, k_codes, d, 5, gen);
    const size_t xb = align_up((size_t)f.n_pad * f.dp * 2, 256), wb = align_up((size_t)f.k_pad * f.dp * 2, 256);
    if (!ws || ws_bytes < xb + wb + 512 + (size_t)f.k_pad * 4) return fail("debug_filter_scores: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    _Float16 *xh = (_Float16 *)ws, *wh = (_Float16 *)((char *)ws + xb);
    float *en_max = (float *)((char *)ws + xb + wb);
    float *wsqp = (float *)((char *)ws + xb + wb + 512);
    hipLaunchKernelGGL(pad_wsq_kernel, dim3((unsigned)((f.k_pad + 255) / 256)), dim3(256), 0, s, wsq, (int)k_codes, (int)f.k_pad, wsqp);
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.n_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, xhat, (long)n, d, f.n_pad, f.dp, xh);
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (f.k_pad * (f.dp / 8) + 255) / 256)), dim3(256), 0, s, what, (long)k_codes, d, f.k_pad, f.dp, wh);
    hipLaunchKernelGGL(wsq_max_kernel, dim3(1), dim3(1024), 0, s, wsq, (int)k_codes, en_max, (float *)nullptr, 0, (int *)nullptr);
    (void)set_lds_once<filter_f16_kernel<5, true>>(F_SMEM_BYTES);
    hipLaunchKernelGGL((filter_f16_kernel<5, true>), dim3((unsigned)f.row_tiles, 1), dim3(F_THREADS), F_SMEM_BYTES, s, xh, wh, xsq, wsqp, en_max,
                       (long)n, (int)k_codes, f.dp, d, (int)f.k_pad, F_OWN_PER_SPLIT, (uint2 *)nullptr, (int *)nullptr, scores, 0, 1,
                       0, (int)f.row_tiles);
    return check_launch("filter_f16(dump)");
}

// ================================================================= merge of per-shard top-k lists
// Code-sharded search (SURVEY 8e variant): every shard returns, for the same rows, its own top-k over its slice of
// the codebook (global code ids); the exact top-k over the union is the (d, index)-lexicographic merge.
__global__ __launch_bounds__(256) void merge_lists_kernel(const float *__restrict__ dist_parts, const int64_t *__restrict__ idx_parts,
                                                          long n, int parts, int topk, int64_t *__restrict__ out_idx,
                                                          float *__restrict__ out_dist)
{
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float bv[MEDTOK_MAX_TOPK];
    long bi[MEDTOK_MAX_TOPK];
#pragma unroll
    for (int j = 0; j < MEDTOK_MAX_TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffffffffffffl; }
    for (int p = 0; p < parts; ++p)
        for (int j = 0; j < topk; ++j) {
            const float v = dist_parts[((long)p * n + row) * topk + j];
            const long c = idx_parts[((long)p * n + row) * topk + j];
            // insertion by (value, index); the lists are short (parts * topk entries)
#pragma unroll
            for (int q = MEDTOK_MAX_TOPK - 1; q >= 0; --q) {
                const bool before = v < bv[q] || (v == bv[q] && c < bi[q]);
                if (before) {
                    if (q + 1 < MEDTOK_MAX_TOPK) { bv[q + 1] = bv[q]; bi[q + 1] = bi[q]; }
                    bv[q] = v; bi[q] = c;
                }
            }
        }
#pragma unroll
    for (int j = 0; j < MEDTOK_MAX_TOPK; ++j)
        if (j < topk) { out_idx[row * topk + j] = bi[j] == 0x7fffffffffffffffl ? (long)j : bi[j]; out_dist[row * topk + j] = bv[j]; }   // NaN rows: in range
}

extern "C" int medtok_merge_topk_lists_f32(const float *dist_parts, const int64_t *idx_parts, int64_t n, int parts, int topk,
                                           int64_t *idx, float *dist, void *stream)
{
    if (n < 0 || parts < 1 || topk < 1 || topk > MEDTOK_MAX_TOPK) return fail("merge_topk_lists: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(merge_lists_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dist_parts, idx_parts,
                       (long)n, parts, topk, idx, dist);
    return check_launch("merge_topk_lists");
}

// ================================================================= soft assign
// One wavefront per row; lanes stride the D axis in float4.
template <int MAXK>
__global__ __launch_bounds__(256) void soft_assign_kernel(const float *__restrict__ xref, const float *__restrict__ what,
                                                          const int64_t *__restrict__ idx, const float *__restrict__ dist,
                                                          long n, int d, int topk, int flags, float *__restrict__ w_out,
                                                          float *zq_ste, long zq_stride, float *__restrict__ row_sqerr,
                                                          const int *__restrict__ row_list, const int *__restrict__ row_count)
{
    const bool hard = flags & MEDTOK_ASSIGN_HARD, raw = flags & MEDTOK_ASSIGN_RAW;
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (row_list) {                    // only the listed rows (the filter's exact-path leftovers after a fused assignment)
        if (row >= *row_count) return;
        row = row_list[row];
    }
    float wj[MAXK];
    long cj[MAXK];
    if (hard) {
        wj[0] = 1.f;
        cj[0] = idx[row];
    } else {
        const float m = -dist[row * topk];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j < topk) { wj[j] = expf(-dist[row * topk + j] - m); sum += wj[j]; cj[j] = idx[row * topk + j]; }
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j < topk) wj[j] = wj[j] / sum;
    }
    if (w_out && lane < topk) {
        float v = wj[0];
#pragma unroll
        for (int j = 1; j < MAXK; ++j) v = (lane == j) ? wj[j] : v;
        w_out[row * topk + lane] = v;
    }
    const float *xr = xref + row * d;
    float *out = zq_ste + row * zq_stride;
    float se = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 a;
        if (hard) {
            a = ld4(what + cj[0] * d + i);
        } else {
            a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < MAXK; ++j)
                if (j < topk) {
                    const float4 e = ld4(what + cj[j] * d + i);
                    a.x = fmaf(wj[j], e.x, a.x); a.y = fmaf(wj[j], e.y, a.y);
                    a.z = fmaf(wj[j], e.z, a.z); a.w = fmaf(wj[j], e.w, a.w);
                }
        }
        const float4 x = ld4(xr + i);
        float4 df;
        df.x = a.x - x.x; df.y = a.y - x.y; df.z = a.z - x.z; df.w = a.w - x.w;
        st4(out + i, raw ? a : make_float4(x.x + df.x, x.y + df.y, x.z + df.z, x.w + df.w));
        se = fmaf(df.x, df.x, se); se = fmaf(df.y, df.y, se); se = fmaf(df.z, df.z, se); se = fmaf(df.w, df.w, se);
    }
    se = wave_butterfly_sum(se);
    if (row_sqerr && lane == 0) row_sqerr[row] = se;
}

extern "C" int medtok_soft_assign_f32(const float *xref, const float *what, const int64_t *idx, const float *dist,
                                      int64_t n, int d, int topk, int flags, float *w, float *zq_ste, int64_t zq_stride,
                                      float *row_sqerr, void *stream)
{
    if (zq_stride == 0) zq_stride = d;
    if (zq_stride < d || (zq_stride & 3)) return fail("soft_assign: zq_stride must be >= d and a multiple of 4");
    const int hard = flags & MEDTOK_ASSIGN_HARD;
    if (n < 0 || d <= 0 || (d & 3)) return fail("soft_assign: bad shape n=%ld d=%d", (long)n, d);
    if (topk < 1 || topk > MEDTOK_MAX_TOPK) return fail("soft_assign: topk=%d unsupported", topk);
    if (hard && topk != 1) return fail("soft_assign: hard assignment needs topk == 1");
    if (!hard && !dist) return fail("soft_assign: dist required");
    if (n == 0) return 0;
    if (!zq_ste) return fail("soft_assign: zq_ste required");
    // (lists of up to 8 codes -- every shipped configuration -- keep the 8-slot instantiation; 9 .. 16 take the 16-slot one)
    if (topk <= 8)
        hipLaunchKernelGGL((soft_assign_kernel<8>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           xref, what, idx, dist, (long)n, d, topk, flags, w, zq_ste, (long)zq_stride, row_sqerr,
                           (const int *)nullptr, (const int *)nullptr);
    else
        hipLaunchKernelGGL((soft_assign_kernel<MEDTOK_MAX_TOPK>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           xref, what, idx, dist, (long)n, d, topk, flags, w, zq_ste, (long)zq_stride, row_sqerr,
                           (const int *)nullptr, (const int *)nullptr);
    return check_launch("soft_assign");
}

// ================================================================= several small searches, batched (kernels next to search_f32_kernel)
// Joins the per-split lists of a row (merge_topk_kernel<T, 64>: a wavefront per row, (d, index) total order) and does the row's soft
// assignment at once (soft_assign_kernel's arithmetic on the values it would have read back: the same bits).
template <int TOPK>
__global__ __launch_bounds__(256) void merge_assign_multi_kernel(MultiSearchArgs a)
{
    const MultiSearchOne &m = a.s[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m.n) return;
    const int d = a.d, topk = a.topk;
    float bv[TOPK];
    int bi[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { bv[j] = INFINITY; bi[j] = 0x7fffffff; }
    for (int sp = lane; sp < m.splits; sp += 64) {
        const long base = ((long)sp * m.n + row) * TOPK;
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, m.pval[base + j], m.pidx[base + j]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float pv[TOPK];
        int pi[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) { pv[j] = __shfl_xor(bv[j], off, 64); pi[j] = __shfl_xor(bi[j], off, 64); }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) topk_insert_lex<TOPK>(bv, bi, pv[j], pi[j]);
    }
    long cj[TOPK];
#pragma unroll
    for (int j = 0; j < TOPK; ++j) cj[j] = valid_code(bi[j], j, m.k_codes);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk) { m.idx[row * topk + j] = cj[j]; m.dist[row * topk + j] = bv[j]; }
    }
    float wj[TOPK];
    const float mx = -bv[0];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < TOPK; ++j)
        if (j < topk) { wj[j] = expf(-bv[j] - mx); sum += wj[j]; }
#pragma unroll
    for (int j = 0; j < TOPK; ++j)
        if (j < topk) wj[j] = wj[j] / sum;
    if (m.w && lane < topk) {
        float v = wj[0];
#pragma unroll
        for (int j = 1; j < TOPK; ++j) v = (lane == j) ? wj[j] : v;
        m.w[row * topk + lane] = v;
    }
    const float *xr = m.x + row * m.x_stride;
    float *out = m.zq + row * m.zq_stride;
    float se = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TOPK; ++j)
            if (j < topk) {
                const float4 e = ld4(m.what + cj[j] * d + i);
                acc.x = fmaf(wj[j], e.x, acc.x); acc.y = fmaf(wj[j], e.y, acc.y);
                acc.z = fmaf(wj[j], e.z, acc.z); acc.w = fmaf(wj[j], e.w, acc.w);
            }
        const float4 x = ld4(xr + i);
        float4 df;
        df.x = acc.x - x.x; df.y = acc.y - x.y; df.z = acc.z - x.z; df.w = acc.w - x.w;
        st4(out + i, make_float4(x.x + df.x, x.y + df.y, x.z + df.z, x.w + df.w));
        se = fmaf(df.x, df.x, se); se = fmaf(df.y, df.y, se); se = fmaf(df.z, df.z, se); se = fmaf(df.w, df.w, se);
    }
    if (m.row_sqerr) {                              // (soft_assign_kernel's sum: 64 strided fmaf chains joined by the xor butterfly)
        se = wave_butterfly_sum(se);
        if (lane == 0) m.row_sqerr[row] = se;
    }
}

constexpr int64_t MS_MAX_ROWS = 4096;

extern "C" int medtok_soft_vq_multi_eligible(int64_t n, int64_t k_codes, int d, int topk)
{
    return n >= 1 && n <= MS_MAX_ROWS && k_codes >= 1 && k_codes < (1ll << 31) && d > 0 && !(d & 3) && topk >= 1 && topk <= 8 &&
           topk <= k_codes && resolve_path(MEDTOK_PATH_AUTO, n, k_codes, d, topk) == MEDTOK_PATH_F32_MFMA;
}

// code splits of every search of a batched call: about two rounds' worth of blocks over the whole call (two 256-thread blocks are
// resident per CU), every search's code tiles cut into splits of equal length
static void multi_plan(const medtok_search_desc *descs, int count, int *splits, int *cps)
{
    const DevInfo di = dev_info();
    long tiles_total = 0;
    for (int i = 0; i < count; ++i) tiles_total += ((descs[i].n + S_BN - 1) / S_BN) * ((descs[i].k_codes + S_BM - 1) / S_BM);
    const long per_block = lmax(1, (tiles_total + 2L * di.cus - 1) / (2L * di.cus));
    for (int i = 0; i < count; ++i) {
        const long code_tiles = (descs[i].k_codes + S_BM - 1) / S_BM;
        const long tps = lmin(per_block, code_tiles);
        cps[i] = (int)(tps * S_BM);
        splits[i] = (int)((code_tiles + tps - 1) / tps);
    }
}

extern "C" size_t medtok_soft_vq_forward_multi_workspace_bytes(const medtok_search_desc *descs, int count, int d, int topk)
{
    if (!descs || count < 1 || count > MS_MAX) return 0;
    int splits[MS_MAX], cps[MS_MAX];
    multi_plan(descs, count, splits, cps);
    const int tslots = topk == 1 ? 1 : (topk <= 5 ? 5 : 8);
    size_t total = 0;
    for (int i = 0; i < count; ++i) {
        const size_t n = (size_t)(descs[i].n > 0 ? descs[i].n : 1);
        total += align_up(n * 4, 256) + 2 * align_up((size_t)splits[i] * n * tslots * 4, 256);
    }
    return total + 256;
}

extern "C" int medtok_soft_vq_forward_multi_f32(const medtok_search_desc *descs, int count, int d, int topk, void *ws, size_t ws_bytes, void *stream)
{
    if (!descs || count < 1 || count > MS_MAX) return fail("soft_vq_forward_multi: 1..%d searches per call", MS_MAX);
    for (int i = 0; i < count; ++i) {
        const medtok_search_desc &q = descs[i];
        if (!medtok_soft_vq_multi_eligible(q.n, q.k_codes, d, topk))
            return fail("soft_vq_forward_multi: search %d (n=%ld K=%ld d=%d topk=%d) does not take the batched exact path", i, (long)q.n, (long)q.k_codes, d, topk);
        if (!q.x || !q.what || !q.wsq || !q.xhat || !q.idx || !q.dist || !q.zq) return fail("soft_vq_forward_multi: NULL argument in search %d", i);
        const int64_t zs = q.zq_stride ? q.zq_stride : d, xs = q.x_stride ? q.x_stride : d;
        if (zs < d || (zs & 3) || xs < d || (xs & 3)) return fail("soft_vq_forward_multi: zq_stride / x_stride must be >= d and multiples of 4");
        if (((uintptr_t)q.x | (uintptr_t)q.what | (uintptr_t)q.xhat | (uintptr_t)q.zq) & 15) return fail("soft_vq_forward_multi: pointers must be 16-byte aligned");
    }
    const size_t need = medtok_soft_vq_forward_multi_workspace_bytes(descs, count, d, topk);
    if (!ws || ws_bytes < need) return fail("soft_vq_forward_multi: workspace too small (%zu < %zu)", ws_bytes, need);
    int splits[MS_MAX], cps[MS_MAX];
    multi_plan(descs, count, splits, cps);
    const int tslots = topk == 1 ? 1 : (topk <= 5 ? 5 : 8);
    MultiSearchArgs a;
    memset(&a, 0, sizeof a);
    a.count = count; a.d = d; a.topk = topk;
    char *p = (char *)ws;
    auto take = [&](size_t bytes) { char *q = p; p += align_up(bytes, 256); return (void *)q; };
    long max_rows = 0;
    int max_tiles = 0, max_splits = 0;
    for (int i = 0; i < count; ++i) {
        const medtok_search_desc &q = descs[i];
        MultiSearchOne &m = a.s[i];
        m.x = q.x; m.what = q.what; m.wsq = q.wsq; m.xhat = q.xhat; m.idx = q.idx; m.dist = q.dist; m.w = q.w; m.zq = q.zq; m.row_sqerr = q.row_sqerr;
        m.n = (long)q.n; m.zq_stride = (long)(q.zq_stride ? q.zq_stride : d); m.x_stride = (long)(q.x_stride ? q.x_stride : d);
        m.k_codes = (int)q.k_codes; m.codes_per_split = cps[i]; m.splits = splits[i]; m.row_tiles = (int)((q.n + S_BN - 1) / S_BN);
        m.xsq = (float *)take((size_t)q.n * 4);
        m.pval = (float *)take((size_t)splits[i] * q.n * tslots * 4);
        m.pidx = (int *)take((size_t)splits[i] * q.n * tslots * 4);
        max_rows = lmax(max_rows, m.n); max_tiles = max_tiles > m.row_tiles ? max_tiles : m.row_tiles; max_splits = max_splits > m.splits ? max_splits : m.splits;
        a.block_base[i + 1] = a.block_base[i] + m.row_tiles * m.splits;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rownorm_multi_kernel, dim3((unsigned)((max_rows + 3) / 4), (unsigned)count), dim3(256), 0, s, a);
    hipEvent_t pa = prof_wanted(1) ? prof_mark(s) : nullptr;
    double pflops = 0.0;
    for (int i = 0; i < count; ++i) pflops += 2.0 * (double)descs[i].n * (double)descs[i].k_codes * (double)d;
    (void)max_tiles; (void)max_splits;
    const dim3 grid((unsigned)a.block_base[count]);          // exactly the blocks that have work, search after search
#define MEDTOK_MS(T)                                                                                                          \
    do {                                                                                                                      \
        if (d % S_BK) {                                                                                                       \
            (void)set_lds_once<search_f32_multi_kernel<T, true>>(S_LDS_BYTES);                                                \
            hipLaunchKernelGGL((search_f32_multi_kernel<T, true>), grid, dim3(256), S_LDS_BYTES, s, a);                       \
        } else {                                                                                                              \
            (void)set_lds_once<search_f32_multi_kernel<T, false>>(S_LDS_BYTES);                                               \
            hipLaunchKernelGGL((search_f32_multi_kernel<T, false>), grid, dim3(256), S_LDS_BYTES, s, a);                      \
        }                                                                                                                     \
        if (pa) prof_push(pa, prof_mark(s), pflops, 1);                                                                       \
        hipLaunchKernelGGL((merge_assign_multi_kernel<T>), dim3((unsigned)((max_rows + 3) / 4), (unsigned)count), dim3(256), 0, s, a); \
    } while (0)
    switch (tslots) {
    case 1: MEDTOK_MS(1); break;
    case 5: MEDTOK_MS(5); break;
    default: MEDTOK_MS(8); break;
    }
#undef MEDTOK_MS
    return check_launch("soft_vq_forward_multi");
}

// ================================================================= fixed-order fp64 sum
__global__ __launch_bounds__(1024) void sum_scale_kernel(const float *__restrict__ v, long n, double scale, float *out)
{
    __shared__ double sh[1024];
    double a = 0.0;
    long i = threadIdx.x;
    // same order of additions as the plain loop; eight loads in flight per thread instead of one (one block: latency-bound)
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (double)x[u];
    }
    for (; i < n; i += 1024) a += (double)v[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] * scale);
}

extern "C" int medtok_sum_scale_f32(const float *vals, int64_t n, double scale, float *out, void *stream)
{
    if (n < 0 || !out) return fail("sum_scale: bad args");
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, vals, (long)n, scale, out);
    return check_launch("sum_scale");
}

// ================================================================= training half: sparse backward, normalize backward, InfoNCE
#include "train_kernels.h"

extern "C" int medtok_soft_vq_backward_f32(const float *x, const float *xhat, const float *what, const int64_t *idx, const float *w,
                                           int64_t n, int d, int topk, const float *g_zq, const float *g_xhat, const float *g_out,
                                           const float *g_vq, const float *g_commit, float vq_scale, float commit_scale,
                                           float *gx, float *g_code, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3)) return fail("soft_vq_backward: bad shape n=%ld d=%d", (long)n, d);
    if (topk < 1 || topk > MEDTOK_MAX_TOPK) return fail("soft_vq_backward: topk=%d unsupported", topk);
    if (n == 0) return 0;
    if (!x || !xhat || !what || !idx || !w) return fail("soft_vq_backward: x, xhat, what, idx and w are required");
    if (!gx && !g_code) return fail("soft_vq_backward: nothing to compute (gx and g_code are both NULL)");
    if (topk <= 8)
        hipLaunchKernelGGL((soft_vq_backward_kernel<8>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, xhat, what, idx, w, (long)n, d, topk, g_zq, g_xhat, g_out, g_vq, g_commit, vq_scale, commit_scale, gx, g_code);
    else
        hipLaunchKernelGGL((soft_vq_backward_kernel<MEDTOK_MAX_TOPK>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, xhat, what, idx, w, (long)n, d, topk, g_zq, g_xhat, g_out, g_vq, g_commit, vq_scale, commit_scale, gx, g_code);
    return check_launch("soft_vq_backward");
}

extern "C" int medtok_normalize_backward_f32(const float *g, const float *vhat, const float *v, int64_t n, int d, float *out, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3)) return fail("normalize_backward: bad shape n=%ld d=%d", (long)n, d);
    if (n == 0) return 0;
    if (!g || !vhat || !v || !out) return fail("normalize_backward: NULL argument");
    hipLaunchKernelGGL(normalize_backward_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, vhat, v, (long)n, d, out);
    return check_launch("normalize_backward");
}
// ... with `live` [n]: rows whose live entry is 0 hold an all-zero g (e.g. the bins of medtok_ema_stats_f32 for a code gradient): their
// output rows are written as zeros without reading g / vhat / v -- the same bits, a third of the traffic for a sparse g
extern "C" int medtok_normalize_backward_sparse_f32(const float *g, const float *vhat, const float *v, const float *live, int64_t n, int d, float *out,
                                                    void *stream)
{
    if (n < 0 || d <= 0 || (d & 3)) return fail("normalize_backward_sparse: bad shape n=%ld d=%d", (long)n, d);
    if (n == 0) return 0;
    if (!g || !vhat || !v || !live || !out) return fail("normalize_backward_sparse: NULL argument");
    hipLaunchKernelGGL(normalize_backward_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, vhat, v, (long)n, d, out, live);
    return check_launch("normalize_backward_sparse");
}

// rows per block of the InfoNCE kernels: sharing a fetched key row among 8 / 4 / 2 rows cuts the L2 traffic, but only once
// there are enough rows to keep >= 256 blocks (at the training batch of 256, one row per block: parallelism wins)
static int info_nce_rows(int64_t b, int d)
{
    for (int rt = 8; rt > 1; rt >>= 1)
        if (b / rt >= 256 && ((size_t)rt * ((size_t)b + (size_t)d) + 4) * sizeof(float) <= 64 * 1024) return rt;
    return 1;
}
static size_t info_nce_lds_bytes(int64_t b, int d, int rt) { return ((size_t)rt * ((size_t)b + (size_t)d) + 4) * sizeof(float); }

extern "C" size_t medtok_info_nce_workspace_bytes(int64_t b, int d)
{
    return (2 * (size_t)b * (size_t)d + (size_t)b) * sizeof(float);     // qhat | khat | row_loss
}

extern "C" int medtok_info_nce_forward_f32(const float *q, const float *k, int64_t b, int d, float temperature, float *loss, float *prob,
                                           void *ws, size_t ws_bytes, void *stream)
{
    if (b <= 0 || d <= 0 || (d & 3)) return fail("info_nce: bad shape b=%ld d=%d", (long)b, d);
    if (!(temperature > 0.f)) return fail("info_nce: temperature must be positive");
    if (!q || !k || !loss || !prob || !ws) return fail("info_nce: NULL argument");
    if (ws_bytes < medtok_info_nce_workspace_bytes(b, d)) return fail("info_nce: workspace too small");
    const int rt = info_nce_rows(b, d);
    if (info_nce_lds_bytes(b, d, rt) > 64 * 1024) return fail("info_nce: b + d = %ld exceeds the 64 KB LDS row budget", (long)(b + d));
    hipStream_t s = (hipStream_t)stream;
    float *qhat = (float *)ws, *khat = qhat + b * d, *row_loss = khat + b * d;
    hipLaunchKernelGGL(info_nce_prepare_kernel, dim3((unsigned)((2 * b + 3) / 4)), dim3(256), 0, s, q, k, (long)b, d, qhat, khat);
    const dim3 fgrid((unsigned)((b + rt - 1) / rt));
    const size_t lds = info_nce_lds_bytes(b, d, rt);
    switch (rt) {
    case 8: hipLaunchKernelGGL(info_nce_forward_kernel<8>, fgrid, dim3(NCE_THREADS), lds, s, qhat, khat, (int)b, d, 1.f / temperature, prob, row_loss); break;
    case 4: hipLaunchKernelGGL(info_nce_forward_kernel<4>, fgrid, dim3(NCE_THREADS), lds, s, qhat, khat, (int)b, d, 1.f / temperature, prob, row_loss); break;
    case 2: hipLaunchKernelGGL(info_nce_forward_kernel<2>, fgrid, dim3(NCE_THREADS), lds, s, qhat, khat, (int)b, d, 1.f / temperature, prob, row_loss); break;
    default: hipLaunchKernelGGL(info_nce_forward_kernel<1>, fgrid, dim3(NCE_THREADS), lds, s, qhat, khat, (int)b, d, 1.f / temperature, prob, row_loss); break;
    }
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, s, row_loss, (long)b, 1.0 / (double)b, loss);
    return check_launch("info_nce_forward");
}

extern "C" int medtok_info_nce_backward_f32(const float *q, const float *k, const float *prob, const float *g_loss, int64_t b, int d,
                                            float temperature, float *gq, float *gk, const void *ws, size_t ws_bytes, void *stream)
{
    if (b <= 0 || d <= 0 || (d & 3)) return fail("info_nce_backward: bad shape b=%ld d=%d", (long)b, d);
    if (!q || !k || !prob || !g_loss || !gq || !gk || !ws) return fail("info_nce_backward: NULL argument");
    if (ws_bytes < medtok_info_nce_workspace_bytes(b, d)) return fail("info_nce_backward: workspace too small");
    const int rt = info_nce_rows(b, d);
    if (info_nce_lds_bytes(b, d, rt) > 64 * 1024) return fail("info_nce_backward: b + d = %ld exceeds the 64 KB LDS row budget", (long)(b + d));
    const float *qhat = (const float *)ws, *khat = qhat + b * d;
    const dim3 bgrid((unsigned)(2 * ((b + rt - 1) / rt)));
    const size_t lds = info_nce_lds_bytes(b, d, rt);
    hipStream_t s = (hipStream_t)stream;
    switch (rt) {
    case 8: hipLaunchKernelGGL(info_nce_backward_kernel<8>, bgrid, dim3(NCE_THREADS), lds, s, q, k, qhat, khat, prob, g_loss, (int)b, d, 1.f / temperature, gq, gk); break;
    case 4: hipLaunchKernelGGL(info_nce_backward_kernel<4>, bgrid, dim3(NCE_THREADS), lds, s, q, k, qhat, khat, prob, g_loss, (int)b, d, 1.f / temperature, gq, gk); break;
    case 2: hipLaunchKernelGGL(info_nce_backward_kernel<2>, bgrid, dim3(NCE_THREADS), lds, s, q, k, qhat, khat, prob, g_loss, (int)b, d, 1.f / temperature, gq, gk); break;
    default: hipLaunchKernelGGL(info_nce_backward_kernel<1>, bgrid, dim3(NCE_THREADS), lds, s, q, k, qhat, khat, prob, g_loss, (int)b, d, 1.f / temperature, gq, gk); break;
    }
    return check_launch("info_nce_backward");
}

// ================================================================= alignment / orthogonality losses
#include "loss_kernels.h"

extern "C" int medtok_row_dot_f32(const float *a, const float *b, int64_t n, int d, float *out, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3)) return fail("row_dot: bad shape n=%ld d=%d (d %% 4 == 0)", (long)n, d);
    if (n == 0) return 0;
    if (!a || !b || !out) return fail("row_dot: NULL argument");
    hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, d, out);
    return check_launch("row_dot");
}

extern "C" int medtok_small_gemm_f32(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn, int m, int n, int k,
                                     float *C, void *stream)
{
    if (m <= 0 || n <= 0 || k <= 0) return fail("small_gemm: bad shape m=%d n=%d k=%d", m, n, k);
    if (!A || !B || !C) return fail("small_gemm: NULL argument");
    const long tiles = (long)((m + 31) / 32) * ((n + 31) / 32);
    // an operand that is contiguous along k, 16-byte aligned in every row: four consecutive k per load
    const bool av = sak == 1 && (sam & 3) == 0 && ((uintptr_t)A & 15) == 0, bv = sbk == 1 && (sbn & 3) == 0 && ((uintptr_t)B & 15) == 0;
#define MEDTOK_SMALL_GEMM(AV, BV)                                                                                                 \
    hipLaunchKernelGGL((small_gemm_f32_kernel<AV, BV>), dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, A, (long)sam,              \
                       (long)sak, B, (long)sbk, (long)sbn, m, n, k, C)
    if (av && bv) MEDTOK_SMALL_GEMM(true, true); else if (av) MEDTOK_SMALL_GEMM(true, false);
    else if (bv) MEDTOK_SMALL_GEMM(false, true); else MEDTOK_SMALL_GEMM(false, false);
#undef MEDTOK_SMALL_GEMM
    return check_launch("small_gemm");
}

extern "C" size_t medtok_frobenius_workspace_bytes(int64_t rows) { return rows > 0 ? align_up((size_t)rows * 4, 256) : 256; }

extern "C" int medtok_frobenius_f32(const float *x, int64_t rows, int d, float *out, void *ws, size_t ws_bytes, void *stream)
{
    if (rows <= 0 || d <= 0 || (d & 3)) return fail("frobenius: bad shape rows=%ld d=%d (d %% 4 == 0)", (long)rows, d);
    if (!x || !out || !ws || ws_bytes < (size_t)rows * 4) return fail("frobenius: NULL argument or workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *sq = (float *)ws;
    launch_rownorm<false>(s, x, (long)rows, d, (float *)nullptr, sq);
    hipLaunchKernelGGL(frobenius_kernel, dim3(1), dim3(1024), 0, s, sq, (long)rows, out);
    return check_launch("frobenius");
}

extern "C" int medtok_scale_by_device_scalar_f32(const float *x, int64_t count, const float *num, const float *den, float c, float *out,
                                                 void *stream)
{
    if (count < 0 || !x || !num || !out) return fail("scale_by_device_scalar: bad arguments");
    if (count == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return fail("scale_by_device_scalar: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3((unsigned)lmin(2048, (count + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       x, (long)count, num, den, c, out);
    return check_launch("scale_by_device_scalar");
}

// ================================================================= cross-attention core (ragged, shared key/value rows)
#include "attention_kernels.h"

#include "attention_backward.h"
#include "attention_dma.h"
#include "attention_pp.h"
#include "attention_small.h"
#include "pack_kernels.h"

static int attention_shape_ok(int d) { return d == 64 || (d > 0 && d % 128 == 0 && d <= 768); }

// forward, shared by the inference and the training entry point
static int attention_forward(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv, const int64_t *kv_start,
                             const int64_t *kv_len, int64_t n_codes, int64_t max_q_len, int d, float scale, float *out, float *lse,
                             float dropout_p, unsigned seed, hipStream_t s)
{
    const int64_t q_tiles = (max_q_len + 31) / 32;
    if (q_tiles * n_codes >= (1ll << 31)) return fail("shared_kv_attention: n_codes * ceil(max_q_len / 32) = %ld exceeds the grid limit", (long)(q_tiles * n_codes));
    const dim3 grid((unsigned)(q_tiles * n_codes));
    hipEvent_t pa = prof_wanted(2) ? prof_mark(s) : nullptr;
    const int waves = d == 64 ? 2 : (d % 256 == 0 ? 8 : 4);
    const size_t lds = ((size_t)32 * (d + 4 * waves) + (waves + 1) * 32 * 33 + 64) * sizeof(float);   // key chunk (one padded column slice per wave) + per-wave partial scores + probabilities + row state
    const unsigned thresh = dropout_p > 0.f ? (unsigned)fmin(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    const float keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
#define MEDTOK_ATT(W, NT)                                                                                                        \
    do {                                                                                                                         \
        if (lds > 64 * 1024 && !set_lds_once<shared_kv_attention_kernel<W, NT>>(lds))                                            \
            return fail("shared_kv_attention: cannot reserve %zu bytes of LDS", lds);                                            \
        hipLaunchKernelGGL((shared_kv_attention_kernel<W, NT>), grid, dim3(64 * W), lds, s, q, q_start, q_len, kv, kv_start, kv_len, scale, out, \
                           (int)q_tiles, lse, thresh, seed, keep_scale);                                                        \
    } while (0)
    switch (d / 128) {
    case 0: MEDTOK_ATT(2, 1); break;      // d = 64, the reference's default e_dim
    case 1: MEDTOK_ATT(4, 1); break;
    case 2: MEDTOK_ATT(8, 1); break;
    case 3: MEDTOK_ATT(4, 3); break;
    case 4: MEDTOK_ATT(8, 2); break;
    case 5: MEDTOK_ATT(4, 5); break;
    default: MEDTOK_ATT(8, 3); break;
    }
#undef MEDTOK_ATT
    if (pa) prof_push(pa, prof_mark(s), 0.0, 2);      // the row / key counts live on the device: the caller prices the launch
    return check_launch("shared_kv_attention");
}

static int attention_forward_f16s(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv, const int64_t *kv_start,
                                  const int64_t *kv_len, int64_t n_codes, int64_t max_q_len, int d, float scale, float *out, _Float16 *out_h,
                                  _Float16 *out_l, hipStream_t s)
{
    const int64_t q_tiles = (max_q_len + 31) / 32;
    if (q_tiles * n_codes >= (1ll << 31)) return fail("shared_kv_attention: n_codes * ceil(max_q_len / 32) = %ld exceeds the grid limit", (long)(q_tiles * n_codes));
    const dim3 grid((unsigned)(q_tiles * n_codes));
    hipEvent_t pa = prof_wanted(2) ? prof_mark(s) : nullptr;
    if (max_q_len <= 8) {
        // a few query rows per code (the text side: one row per head): one wavefront per code, plain fp32 (attention_kernels.h)
        const dim3 fgrid((unsigned)((n_codes + 3) / 4));
        if (d > 768) {          // (D = 1024, e.g. BERT-large text features: four column chunks per lane; at most 4 query rows fit the registers)
            if (max_q_len > 4) return fail("shared_kv_attention: d=%d takes at most 4 query rows per code on this path (heads <= 4)", d);
            hipLaunchKernelGGL((shared_kv_attention_fewq_kernel<4, 4>), fgrid, dim3(256), 0, s, q, q_start, q_len, kv, kv_start, kv_len, (long)n_codes, d, scale,
                               out, out_h, out_l);
        } else if (max_q_len <= 4)
            hipLaunchKernelGGL(shared_kv_attention_fewq_kernel<4>, fgrid, dim3(256), 0, s, q, q_start, q_len, kv, kv_start, kv_len, (long)n_codes, d, scale,
                               out, out_h, out_l);
        else
            hipLaunchKernelGGL(shared_kv_attention_fewq_kernel<8>, fgrid, dim3(256), 0, s, q, q_start, q_len, kv, kv_start, kv_len, (long)n_codes, d, scale,
                               out, out_h, out_l);
        if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
        return check_launch("shared_kv_attention(few rows)");
    }
    if (d > 768) return fail("shared_kv_attention: d=%d with more than 8 query rows per code runs on the image form (medtok_shared_kv_attention_split_f32)", d);
    const int waves = d == 64 ? 2 : (d % 256 == 0 ? 8 : 4);
    // (hi, lo) key planes of W slices, 32 keys x (d / W + 8) halves each; per-wave partial scores; probabilities (hi, lo); row state
    const size_t lds = (size_t)waves * 2 * 32 * (d / waves + 8) * 2 + (size_t)waves * 32 * 33 * 4 + 2 * 32 * 40 * 2 + 64 * 4;
#define MEDTOK_ATT16(W, NT)                                                                                                      \
    do {                                                                                                                         \
        if (lds > 64 * 1024 && !set_lds_once<shared_kv_attention_f16s_kernel<W, NT>>(lds))                                       \
            return fail("shared_kv_attention: cannot reserve %zu bytes of LDS", lds);                                            \
        hipLaunchKernelGGL((shared_kv_attention_f16s_kernel<W, NT>), grid, dim3(64 * W), lds, s, q, q_start, q_len, kv, kv_start, kv_len, scale, \
                           out, out_h, out_l, (int)q_tiles);                                                                     \
    } while (0)
    switch (d / 128) {
    case 0: MEDTOK_ATT16(2, 1); break;
    case 1: MEDTOK_ATT16(4, 1); break;
    case 2: MEDTOK_ATT16(8, 1); break;
    case 3: MEDTOK_ATT16(4, 3); break;
    case 4: MEDTOK_ATT16(8, 2); break;
    case 5: MEDTOK_ATT16(4, 5); break;
    default: MEDTOK_ATT16(8, 3); break;
    }
#undef MEDTOK_ATT16
    if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
    return check_launch("shared_kv_attention(f16 x 3)");
}

// 64-row blocks, keys from (hi, lo) fp16 images by LDS-DMA (attention_dma.h)
static bool attention_dma_shape_ok(int d) { return d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024; }

static void *g_att_dbg = nullptr;      // DEV (tools/r04): per-wave cycle counts of the pp kernel's phases
extern "C" void medtok_debug_set_attention_probe(void *p) { g_att_dbg = p; }

extern "C" int medtok_shared_kv_attention_split_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const void *kv_hi, const void *kv_lo,
                                                    const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                    int d, float scale, float *out, void *out_hi, void *out_lo, int variant, void *stream)
{
    if ((out_hi == nullptr) != (out_lo == nullptr)) return fail("shared_kv_attention_split: out_hi and out_lo go together");
    if (n_codes < 0 || max_q_len < 0) return fail("shared_kv_attention_split: bad sizes n_codes=%ld max_q_len=%ld", (long)n_codes, (long)max_q_len);
    if (!attention_dma_shape_ok(d)) return fail("shared_kv_attention_split: d=%d must be 128, 256, 384, 512, 768 or 1024", d);
    if (n_codes == 0 || max_q_len == 0) return 0;
    const int vform = variant & 15;
    const bool pp_shape = vform == 2 && (d == 256 || d == 512 || d == 768);
    // MEDTOK_ATTENTION_F32_KEYS: kv_hi points at the caller's fp32 key rows, which the kernel turns into their (hi, lo) images itself
    const bool f32_keys = (variant & MEDTOK_ATTENTION_F32_KEYS) != 0;
    if (f32_keys && !pp_shape) return fail("shared_kv_attention_split: fp32 keys converted in the kernel need variant 2 and d = 256, 512 or 768");
    if (!q || !q_start || !q_len || !kv_hi || !kv_start || !kv_len || (!out && !out_hi)) return fail("shared_kv_attention_split: NULL argument");
    if (!kv_lo && !f32_keys && !pp_shape) return fail("shared_kv_attention_split: keys without a lo image (fp16 keys as they stand) need variant 2 and d = 256, 512 or 768");
    if (((uintptr_t)q | (uintptr_t)kv_hi | (uintptr_t)kv_lo | (uintptr_t)out) & 15) return fail("shared_kv_attention_split: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t pa = prof_wanted(2) ? prof_mark(s) : nullptr;
#define MEDTOK_ATT_DMA(W, NT, MT, RING)                                                                                          \
    do {                                                                                                                         \
        const size_t lds = AttDma<W, NT, MT, RING>::LDS_BYTES;                                                                   \
        const int64_t q_tiles = (max_q_len + 32 * MT - 1) / (32 * MT);                                                           \
        if (q_tiles * (n_codes + 8) >= (1ll << 31)) return fail("shared_kv_attention_split: grid limit exceeded");                \
        if (lds > 64 * 1024 && !set_lds_once<shared_kv_attention_dma_kernel<W, NT, MT, RING>>(lds))                              \
            return fail("shared_kv_attention_split: cannot reserve %zu bytes of LDS", lds);                                      \
        hipLaunchKernelGGL((shared_kv_attention_dma_kernel<W, NT, MT, RING>), dim3((unsigned)(q_tiles * ((n_codes + 7) / 8 * 8))), dim3(64 * W), \
                           lds, s, q, q_start, q_len, (const _Float16 *)kv_hi, (const _Float16 *)kv_lo, kv_start, kv_len, scale, out, \
                           (_Float16 *)out_hi, (_Float16 *)out_lo, (int)q_tiles, (int)n_codes);                                  \
    } while (0)
    if (pp_shape) {
        // two 32-row tiles of a code per block, run one phase apart on one copy of the keys (attention_pp.h)
        const int64_t q_pairs = (max_q_len + 63) / 64;
        if (q_pairs * (n_codes + 8) >= (1ll << 31)) return fail("shared_kv_attention_split: grid limit exceeded");
#define MEDTOK_ATT_PP(NT, TIMED, KLO, ...)                                                                                         \
    do {                                                                                                                          \
        const size_t lds = AttPP<NT>::LDS_BYTES;                                                                                  \
        if (!set_lds_once<shared_kv_attention_pp_kernel<NT, TIMED, KLO __VA_ARGS__>>(lds)) return fail("shared_kv_attention_split: cannot reserve %zu bytes of LDS", lds); \
        hipLaunchKernelGGL((shared_kv_attention_pp_kernel<NT, TIMED, KLO __VA_ARGS__>), dim3((unsigned)(q_pairs * ((n_codes + 7) / 8 * 8))), dim3(512), lds, s, q, q_start, \
                           q_len, (const _Float16 *)kv_hi, (const _Float16 *)kv_lo, kv_start, kv_len, scale, out, (_Float16 *)out_hi,             \
                           (_Float16 *)out_lo, (int)q_pairs, (int)n_codes, (unsigned long long *)g_att_dbg);                      \
    } while (0)
        const bool timed = ((variant >> 4) & 15) == 8 && g_att_dbg;   // (dev probe: tools/r04/att_probe.py)
        if (f32_keys) { if (d == 256) MEDTOK_ATT_PP(2, false, true, , true); else if (d == 512) MEDTOK_ATT_PP(4, false, true, , true); else MEDTOK_ATT_PP(6, false, true, , true); }
        else if (!kv_lo) { if (d == 256) MEDTOK_ATT_PP(2, false, false); else if (d == 512) MEDTOK_ATT_PP(4, false, false); else MEDTOK_ATT_PP(6, false, false); }
        else if (d == 256) MEDTOK_ATT_PP(2, false, true); else if (d == 512) MEDTOK_ATT_PP(4, false, true);
        else if (timed) MEDTOK_ATT_PP(6, true, true);
        else MEDTOK_ATT_PP(6, false, true);
#undef MEDTOK_ATT_PP
        if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
        return check_launch("shared_kv_attention_split(pp)");
    }
    switch (d) {
    case 128: MEDTOK_ATT_DMA(4, 1, 2, 2); break;
    case 256: MEDTOK_ATT_DMA(8, 1, 2, 2); break;
    case 384: MEDTOK_ATT_DMA(4, 3, 2, 2); break;
    case 512: MEDTOK_ATT_DMA(8, 2, 2, 2); break;
    case 1024: MEDTOK_ATT_DMA(4, 8, 1, 2); break;      // (BERT-large width: 32 rows per block, one wave per SIMD with 512 registers, two-deep ring)
    default:                               // 768: variant 0 = 32 rows per block, two blocks per CU; 1 = 64 rows per block, one per CU
        if (variant == 1) MEDTOK_ATT_DMA(8, 3, 2, 2); else MEDTOK_ATT_DMA(4, 6, 1, 1);
        break;
    }
#undef MEDTOK_ATT_DMA
    if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
    return check_launch("shared_kv_attention_split");
}

extern "C" int medtok_shared_kv_attention_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                              const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                              int d, float scale, float *out, void *out_hi, void *out_lo, int exact_f32, void *stream)
{
    if ((out_hi == nullptr) != (out_lo == nullptr)) return fail("shared_kv_attention: out_hi and out_lo go together");
    if (exact_f32 && out_hi) return fail("shared_kv_attention: the (hi, lo) output images come from the inference kernels (exact_f32 = 0)");
    if (!out && !out_hi && n_codes > 0 && max_q_len > 0) return fail("shared_kv_attention: no output buffer");
    if (n_codes < 0 || max_q_len < 0) return fail("shared_kv_attention: bad sizes n_codes=%ld max_q_len=%ld", (long)n_codes, (long)max_q_len);
    // (d = 1024: the inference form only, with at most 4 query rows per code -- the text side's CLS row per head; wider query sets
    // run on medtok_shared_kv_attention_split_f32; the exact fp32 kernel's 32-key chunk of fp32 rows does not fit the LDS there)
    if (!attention_shape_ok(d) && !(d == 1024 && !exact_f32)) return fail("shared_kv_attention: d=%d must be 64 or a multiple of 128, at most 768 (1024: inference only)", d);
    if (n_codes == 0 || max_q_len == 0) return 0;
    if (!q || !q_start || !q_len || !kv || !kv_start || !kv_len) return fail("shared_kv_attention: NULL argument");
    if (!exact_f32) return attention_forward_f16s(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, d, scale, out, (_Float16 *)out_hi,
                                                  (_Float16 *)out_lo, (hipStream_t)stream);
    return attention_forward(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, d, scale, out, nullptr, 0.f, 0u, (hipStream_t)stream);
}

// CrossAttention.pooled at e_dim = 64, 4 heads (the reference's default shape) in two launches (attention_small.h)
static int cross_attention_small_impl(bool exact, const float *text, const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len,
                                                const float *nodes, const int64_t *batch, int64_t n_nodes, int d, int heads, int layers,
                                                const float *weights, float scale, float ln_eps, float *y_nodes, float *pooled,
                                                int64_t pooled_stride, int64_t graph_off, int32_t *status, void *stream)
{
    if (d != XS_D || heads != XS_H) return fail("cross_attention_small: d=%d heads=%d (this path is e_dim = 64 with 4 heads)", d, heads);
    if (n_codes < 0 || seq_len <= 0 || n_nodes < 0 || layers <= 0) return fail("cross_attention_small: bad sizes n_codes=%ld seq_len=%ld n_nodes=%ld layers=%d", (long)n_codes, (long)seq_len, (long)n_nodes, layers);
    if (mask_elem_bytes != 1 && mask_elem_bytes != 4 && mask_elem_bytes != 8) return fail("cross_attention_small: mask elements of %d bytes (bool / int32 / int64 expected)", mask_elem_bytes);
    if (n_codes == 0) return 0;
    if (!text || !mask || !weights || !pooled || !status || (n_nodes && (!nodes || !batch || !y_nodes))) return fail("cross_attention_small: NULL argument");
    if (((uintptr_t)text | (uintptr_t)nodes | (uintptr_t)weights | (uintptr_t)y_nodes) & 15) return fail("cross_attention_small: pointers must be 16-byte aligned");
    if (pooled_stride < XS_D || graph_off < 0) return fail("cross_attention_small: bad output layout");
    const int64_t tiles = (n_nodes + XS_G - 1) / XS_G;
    if (tiles + n_codes >= (1ll << 31)) return fail("cross_attention_small: grid limit exceeded");
    XSmallArgs a;
    a.text = text; a.mask = mask; a.nodes = nodes; a.batch = batch; a.weights = weights; a.y_nodes = y_nodes; a.pooled = pooled; a.status = status;
    a.n_codes = (long)n_codes; a.seq_len = (long)seq_len; a.n_nodes = (long)n_nodes; a.pooled_stride = (long)pooled_stride; a.graph_off = (long)graph_off;
    a.mask_bytes = mask_elem_bytes; a.layers = layers; a.n_graph_tiles = (int)tiles; a.scale = scale; a.ln_eps = ln_eps;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t pa = prof_wanted(2) ? prof_mark(s) : nullptr;
    if (exact) hipLaunchKernelGGL(cross_attention64_kernel<false>, dim3((unsigned)(tiles + n_codes)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(cross_attention64_kernel<true>, dim3((unsigned)(tiles + n_codes)), dim3(256), 0, s, a);
    if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
    hipLaunchKernelGGL(cross_attention64_mean_kernel, dim3((unsigned)((n_codes + 3) / 4)), dim3(256), 0, s, y_nodes, batch, (long)n_nodes, (long)n_codes,
                       pooled, (long)pooled_stride, (long)graph_off);
    return check_launch("cross_attention_small");
}

extern "C" int medtok_cross_attention_small_f32(const float *text, const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len,
                                                const float *nodes, const int64_t *batch, int64_t n_nodes, int d, int heads, int layers,
                                                const float *weights, float scale, float ln_eps, float *y_nodes, float *pooled,
                                                int64_t pooled_stride, int64_t graph_off, int32_t *status, void *stream)
{
    return cross_attention_small_impl(false, text, mask, mask_elem_bytes, n_codes, seq_len, nodes, batch, n_nodes, d, heads, layers, weights, scale, ln_eps,
                                      y_nodes, pooled, pooled_stride, graph_off, status, stream);
}

// Test hook: the same call with the attention core on the fp32 matrix pipe (exact fmaf chains; the round-5 form) -- a second opinion
// on the split-fp16 core that the product runs
extern "C" int medtok_debug_cross_attention_small_exact_f32(const float *text, const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len,
                                                            const float *nodes, const int64_t *batch, int64_t n_nodes, int d, int heads, int layers,
                                                            const float *weights, float scale, float ln_eps, float *y_nodes, float *pooled,
                                                            int64_t pooled_stride, int64_t graph_off, int32_t *status, void *stream)
{
    return cross_attention_small_impl(true, text, mask, mask_elem_bytes, n_codes, seq_len, nodes, batch, n_nodes, d, heads, layers, weights, scale, ln_eps,
                                      y_nodes, pooled, pooled_stride, graph_off, status, stream);
}

// the prologue of CrossAttention.pooled (pack_kernels.h)
extern "C" size_t medtok_pack_codes_workspace_bytes(int64_t n_codes)
{
    return align_up((size_t)(n_codes > 0 ? n_codes : 1) * 4, 256) * 2 + 256;        // counts32 | order | stats32
}

extern "C" int medtok_pack_codes_checked(const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len, const int64_t *batch, int64_t n_nodes,
                                         int heads, int lpt, int64_t *valid_len, int64_t *counts, int64_t *starts, int64_t *t_start, int64_t *t_len,
                                         int64_t *g_start, int64_t *g_len, int64_t *tok_start, int64_t *g_kv_len, int64_t *stats,
                                         int64_t count_bound, int *status, void *ws, size_t ws_bytes, void *stream);
extern "C" int medtok_pack_codes(const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len, const int64_t *batch, int64_t n_nodes,
                                 int heads, int lpt, int64_t *valid_len, int64_t *counts, int64_t *starts, int64_t *t_start, int64_t *t_len,
                                 int64_t *g_start, int64_t *g_len, int64_t *tok_start, int64_t *g_kv_len, int64_t *stats, void *ws, size_t ws_bytes,
                                 void *stream)
{
    return medtok_pack_codes_checked(mask, mask_elem_bytes, n_codes, seq_len, batch, n_nodes, heads, lpt, valid_len, counts, starts, t_start, t_len,
                                     g_start, g_len, tok_start, g_kv_len, stats, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int medtok_pack_codes_checked(const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len, const int64_t *batch, int64_t n_nodes,
                                         int heads, int lpt, int64_t *valid_len, int64_t *counts, int64_t *starts, int64_t *t_start, int64_t *t_len,
                                         int64_t *g_start, int64_t *g_len, int64_t *tok_start, int64_t *g_kv_len, int64_t *stats,
                                         int64_t count_bound, int *status, void *ws, size_t ws_bytes, void *stream)
{
    if (count_bound < 0) return fail("pack_codes: count_bound=%ld must be >= 0 (0 = none)", (long)count_bound);
    if (n_codes <= 0 || seq_len < 0 || n_nodes < 0 || heads <= 0) return fail("pack_codes: bad sizes n_codes=%ld seq_len=%ld n_nodes=%ld heads=%d", (long)n_codes, (long)seq_len, (long)n_nodes, heads);
    if (n_codes >= (1ll << 31) - 1 || n_nodes >= (1ll << 31) * 256ll) return fail("pack_codes: too many codes / nodes");
    if (mask_elem_bytes != 1 && mask_elem_bytes != 4 && mask_elem_bytes != 8) return fail("pack_codes: mask elements of %d bytes (bool / int32 / int64 expected)", mask_elem_bytes);
    if (!mask || (!batch && n_nodes) || !valid_len || !counts || !starts || !t_start || !t_len || !g_start || !g_len || !tok_start || !g_kv_len || !stats)
        return fail("pack_codes: NULL argument");
    const size_t need = medtok_pack_codes_workspace_bytes(n_codes);
    if (!ws || ws_bytes < need) return fail("pack_codes: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const size_t seg = align_up((size_t)n_codes * 4, 256);
    int *counts32 = (int *)ws, *order = (int *)((char *)ws + seg), *stats32 = (int *)((char *)ws + 2 * seg);
    const unsigned mgrid = (unsigned)((n_codes + 3) / 4);
    if (mask_elem_bytes == 1) hipLaunchKernelGGL(pack_mask_len_kernel<uint8_t>, dim3(mgrid), dim3(256), 0, s, (const uint8_t *)mask, (long)n_codes, (long)seq_len, valid_len, counts32, stats32);
    else if (mask_elem_bytes == 4) hipLaunchKernelGGL(pack_mask_len_kernel<int32_t>, dim3(mgrid), dim3(256), 0, s, (const int32_t *)mask, (long)n_codes, (long)seq_len, valid_len, counts32, stats32);
    else hipLaunchKernelGGL(pack_mask_len_kernel<int64_t>, dim3(mgrid), dim3(256), 0, s, (const int64_t *)mask, (long)n_codes, (long)seq_len, valid_len, counts32, stats32);
    if (n_nodes > 0)
        hipLaunchKernelGGL(pack_count_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, s, batch, (long)n_nodes, (long)n_codes, counts32, stats32);
    const bool sort = lpt && seq_len < PACK_MAX_KEYS && n_codes <= PACK_SORT_MAX_CODES;
    size_t pw = 1;
    while ((int64_t)pw < n_codes) pw <<= 1;
    hipLaunchKernelGGL(pack_lists_kernel, dim3(1), dim3(PACK_THREADS), sort ? pw * 4 : 0, s, counts32, stats32, valid_len, (long)n_codes,
                       (long)seq_len, heads, lpt, order, counts, starts, t_start, t_len, g_start, g_len, tok_start, g_kv_len, stats, (long)count_bound, status);
    return check_launch("pack_codes");
}

// the layer tail and the node mean around the attention core (attention_kernels.h)
extern "C" int medtok_residual_layernorm_split_f32(const float *a, const float *b, const float *gamma, const float *beta, int64_t n, int d,
                                                   float eps, float *y, void *y_hi, void *y_lo, int dp, void *stream)
{
    if (n < 0 || d <= 0 || d % 4 != 0 || d > 4 * 64 * LN_MAXV) return fail("residual_layernorm: bad shape n=%ld d=%d (d %% 4 == 0, d <= %d)", (long)n, d, 4 * 64 * LN_MAXV);
    if (!(eps >= 0.f)) return fail("residual_layernorm: eps=%g must be >= 0", (double)eps);
    if ((y_hi == nullptr) != (y_lo == nullptr)) return fail("residual_layernorm: y_hi and y_lo go together");
    if (y_hi && (dp < d || dp % 8 != 0 || dp > 4 * 64 * LN_MAXV)) return fail("residual_layernorm: image width dp=%d (>= d, a multiple of 8)", dp);
    if (n == 0) return 0;
    if (!a || !b || !gamma || !beta || !y) return fail("residual_layernorm: NULL argument");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)y_hi | (uintptr_t)y_lo) & 15) return fail("residual_layernorm: pointers must be 16-byte aligned");
    if ((n + 3) / 4 >= (1ll << 31)) return fail("residual_layernorm: too many rows");
    if (y_hi)
        hipLaunchKernelGGL(residual_layernorm_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta, (long)n, d, eps, y,
                           (_Float16 *)y_hi, (_Float16 *)y_lo, dp);
    else
        hipLaunchKernelGGL(residual_layernorm_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta, (long)n, d, eps, y,
                           (_Float16 *)nullptr, (_Float16 *)nullptr, 0);
    return check_launch("residual_layernorm");
}

extern "C" int medtok_residual_layernorm_f32(const float *a, const float *b, const float *gamma, const float *beta, int64_t n, int d,
                                             float eps, float *y, void *stream)
{
    return medtok_residual_layernorm_split_f32(a, b, gamma, beta, n, d, eps, y, nullptr, nullptr, 0, stream);
}

// One CrossAttentionLayer at inference in ONE call: the seven launches of CrossAttention._folded_rows_split (rows -> images ->
// in_proj -> per-head fold -> attention core -> per-head W_v -> out_proj -> residual + LayerNorm), composed from the entry points
// above with the intermediates in caller-provided scratch.  Same kernels, same arguments, same bits as the seven separate calls;
// what it saves is the host side of six of them (a B = 256 forward is host-bound: ~55 calls of ~15 us against 0.45 ms of kernels).
extern "C" size_t medtok_cross_attention_layer_workspace_bytes(int64_t n_rows, int d, int dw, int heads, int hp)
{
    const size_t r = (size_t)(n_rows > 0 ? n_rows : 1);
    return 2 * align_up(r * dw * 2, 256) + 2 * align_up(r * heads * hp * 2, 256) + align_up(r * heads * dw * 4, 256) +
           2 * align_up(r * heads * dw * 2, 256) + 2 * align_up(r * heads * hp * 2, 256) + align_up(r * d * 4, 256);
}

extern "C" int medtok_cross_attention_layer_f32(
    const float *rows, const void *rows_hi, const void *rows_lo, int64_t n_rows, int d, int dw, int heads, int hp,
    const void *wq_hi, const void *wq_lo, float wq_unscale, const float *bq, const void *wk_hi, const void *wk_lo, float wk_unscale,
    const void *wv_hi, const void *wv_lo, float wv_unscale, const float *bv, const void *wo_hi, const void *wo_lo, float wo_unscale, const float *bo,
    const int64_t *q_start, const int64_t *q_len, int64_t n_codes, int64_t max_q_len, const float *kv, const void *kv_hi, const void *kv_lo,
    const int64_t *kv_start, const int64_t *kv_len, float scale, int variant, const float *ln_gamma, const float *ln_beta, float ln_eps,
    float *y, void *y_hi, void *y_lo, void *ws, size_t ws_bytes, void *stream)
{
    if (n_rows < 0 || d <= 0 || dw < d || heads <= 0 || hp <= 0) return fail("cross_attention_layer: bad shape n_rows=%ld d=%d dw=%d heads=%d hp=%d", (long)n_rows, d, dw, heads, hp);
    if (n_rows == 0) return 0;
    const size_t need = medtok_cross_attention_layer_workspace_bytes(n_rows, d, dw, heads, hp);
    if (!ws || ws_bytes < need) return fail("cross_attention_layer: workspace too small (%zu < %zu)", ws_bytes, need);
    if (!rows || !y || (!kv && !kv_hi)) return fail("cross_attention_layer: NULL argument");
    char *p = (char *)ws;
    auto take = [&](size_t bytes) { char *q = p; p += align_up(bytes, 256); return (void *)q; };
    const size_t r = (size_t)n_rows;
    void *x_hi = take(r * dw * 2), *x_lo = take(r * dw * 2);
    void *q_hi = take(r * heads * hp * 2), *q_lo = take(r * heads * hp * 2);
    float *qf = (float *)take(r * heads * dw * 4);
    void *c_hi = take(r * heads * dw * 2), *c_lo = take(r * heads * dw * 2);
    void *a_hi = take(r * heads * hp * 2), *a_lo = take(r * heads * hp * 2);
    float *o = (float *)take(r * d * 4);
    if (rows_hi && rows_lo) { x_hi = (void *)rows_hi; x_lo = (void *)rows_lo; }
    else if (medtok_split_half_f32(rows, n_rows, d, d, dw, 1.0f, x_hi, x_lo, nullptr, 0, stream)) return 1;
    if (medtok_split_gemm_f16(x_hi, x_lo, n_rows, dw, 0, wq_hi, wq_lo, (int64_t)heads * hp, dw, 0, heads * hp, dw, 1, bq, wq_unscale, nullptr, 0, q_hi, q_lo,
                              heads * hp, stream)) return 1;
    if (medtok_split_gemm_f16(q_hi, q_lo, n_rows, heads * hp, hp, wk_hi, wk_lo, (int64_t)heads * dw, hp, dw, dw, hp, heads, nullptr, wk_unscale, qf, heads * dw,
                              nullptr, nullptr, 0, stream)) return 1;
    if (kv_hi) {
        if (medtok_shared_kv_attention_split_f32(qf, q_start, q_len, kv_hi, kv_lo, kv_start, kv_len, n_codes, max_q_len, dw, scale, nullptr, c_hi, c_lo, variant, stream))
            return 1;
    } else if (medtok_shared_kv_attention_f32(qf, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, dw, scale, nullptr, c_hi, c_lo, 0, stream)) {
        return 1;
    }
    if (medtok_split_gemm_f16(c_hi, c_lo, n_rows, heads * dw, dw, wv_hi, wv_lo, (int64_t)heads * hp, dw, hp, hp, dw, heads, bv, wv_unscale, nullptr, 0, a_hi, a_lo,
                              heads * hp, stream)) return 1;
    if (medtok_split_gemm_f16(a_hi, a_lo, n_rows, heads * hp, 0, wo_hi, wo_lo, d, heads * hp, 0, d, heads * hp, 1, bo, wo_unscale, o, d, nullptr, nullptr, 0, stream))
        return 1;
    return medtok_residual_layernorm_split_f32(rows, o, ln_gamma, ln_beta, n_rows, d, ln_eps, y, y_hi, y_lo, y_hi ? dw : 0, stream);
}

extern "C" int medtok_segment_mean_f32(const float *x, const int64_t *seg_start, const int64_t *seg_len, int64_t n_seg, int d, float *out,
                                       void *stream)
{
    if (n_seg < 0 || d <= 0 || d % 4 != 0 || n_seg >= (1ll << 31)) return fail("segment_mean: bad shape n_seg=%ld d=%d (d %% 4 == 0)", (long)n_seg, d);
    if (n_seg == 0) return 0;
    if (!x || !seg_start || !seg_len || !out) return fail("segment_mean: NULL argument");
    if (((uintptr_t)x | (uintptr_t)out) & 15) return fail("segment_mean: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)n_seg, (unsigned)((d / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, seg_start,
                       seg_len, d, out);
    return check_launch("segment_mean");
}

// ================================================================= split-fp16 dense products (split_gemm.h)
#include "split_gemm.h"

extern "C" int medtok_split_half_f32(const float *src, int64_t n, int d, int64_t src_stride, int dp, float scale, void *hi, void *lo,
                                     const int64_t *seg_len, int seg_rows, void *stream)
{
    if (seg_len && (seg_rows <= 0 || n % seg_rows)) return fail("split_half: n=%ld is not a whole number of segments of %d rows", (long)n, seg_rows);
    if (n < 0 || d <= 0 || (d & 3) || dp < d || (dp & 7) || src_stride < d || (src_stride & 3))
        return fail("split_half: bad shape n=%ld d=%d stride=%ld dp=%d (d %% 4 == 0, dp %% 8 == 0, dp >= d)", (long)n, d, (long)src_stride, dp);
    if (n == 0) return 0;
    if (!src || !hi || !lo) return fail("split_half: NULL argument");
    if (((uintptr_t)src | (uintptr_t)hi | (uintptr_t)lo) & 15) return fail("split_half: pointers must be 16-byte aligned");
    const long total = n * (dp / 8);
    // (the segmented form is the text-image pass of a forward, gigabytes of HBM-bound streaming: six blocks per CU keep the memory
    // system full and leave a quarter of the wave slots for the small launches that run beside it; in-box A/B of the forward: 2 per CU
    // 382 k, 4: 394 k, 6: 398 k, 32: 390-397 k codes/s)
    const long grid = seg_len ? lmin(6L * dev_info().cus, (total + 1023) / 1024) : lmin(8192, (total + 1023) / 1024);
    hipLaunchKernelGGL(split_half_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, (long)n, d,
                       (long)src_stride, dp, scale, (_Float16 *)hi, (_Float16 *)lo, seg_len, seg_rows > 0 ? seg_rows : 1);
    return check_launch("split_half");
}

extern "C" int medtok_absmax_f32(const float *x, int64_t count, float *amax, void *stream)
{
    if (count < 0 || !amax || (!x && count)) return fail("absmax: bad arguments");
    if (((uintptr_t)x) & 15) return fail("absmax: x must be 16-byte aligned");
    if (hipMemsetAsync(amax, 0, 4, (hipStream_t)stream) != hipSuccess) return fail("absmax: memset failed");
    if (count == 0) return 0;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)lmin(512, (count + 4095) / 4096)), dim3(256), 0, (hipStream_t)stream, x, (long)count, amax);
    return check_launch("absmax");
}

extern "C" int medtok_split_half_scaled_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, const float *amax, int transpose,
                                            int64_t group_cols, void *hi, void *lo, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3) || (dp & 7) || src_stride < d || (src_stride & 3) || dp < (transpose ? n : (int64_t)d))
        return fail("split_half_scaled: bad shape n=%ld d=%d stride=%ld dp=%ld transpose=%d", (long)n, d, (long)src_stride, (long)dp, transpose);
    if (group_cols == 0) group_cols = dp;
    if (transpose && (group_cols <= 0 || dp % group_cols || (group_cols != dp && group_cols % 64)))
        return fail("split_half_scaled: group_cols=%ld must divide dp=%ld and be a multiple of 64", (long)group_cols, (long)dp);
    if (n == 0 && !transpose) return 0;
    if ((!src && n) || !hi || !lo) return fail("split_half_scaled: NULL argument");
    if (((uintptr_t)src | (uintptr_t)hi | (uintptr_t)lo) & 15) return fail("split_half_scaled: pointers must be 16-byte aligned");
    if (!transpose) {
        if (dp >= (1ll << 31)) return fail("split_half_scaled: dp too large");
        const long total = n * (dp / 8);
        hipLaunchKernelGGL(split_half_kernel<4>, dim3((unsigned)lmin(8192, (total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src, (long)n, d,
                           (long)src_stride, (int)dp, 1.0f, (_Float16 *)hi, (_Float16 *)lo, (const int64_t *)nullptr, 1, amax);
        return check_launch("split_half_scaled");
    }
    // transposed: the images are [d, dp], dp >= n columns (the zero tail is written by the tiles that cover it)
    const long row_tiles = (dp + 63) / 64;
    if (row_tiles >= (1ll << 31) || (d + 63) / 64 > 65535) return fail("split_half_scaled: too large");
    hipLaunchKernelGGL(split_half_t_kernel, dim3((unsigned)row_tiles, (unsigned)((d + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, (long)n, d,
                       (long)src_stride, (long)dp, (long)group_cols, 1.0f, (_Float16 *)hi, (_Float16 *)lo, amax);
    return check_launch("split_half_scaled(transposed)");
}

// The 16-bit image (fp16, or bf16 with bf16 != 0) of an fp32 matrix [n, d]: row-major [n, dp] with zero columns past d, or (transpose) the
// image of the transpose [d, dp], dp >= n, grouped along the rows like medtok_split_half_scaled_f32 -- operands of medtok_half_gemm_f32.
static int half_image_impl(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int transpose, int64_t group_cols, int bf16,
                           void *out, float *col_partials, void *stream);
extern "C" int medtok_half_image_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int transpose, int64_t group_cols, int bf16,
                                     void *out, void *stream)
{
    return half_image_impl(src, n, d, src_stride, dp, transpose, group_cols, bf16, out, nullptr, stream);
}
// the transposed image and, from the same pass, col_partials [(dp + 63) / 64, d]: the sums over every 64-row tile of each column of src
// (medtok_half_image_pair_sums_f32's, for a product that needs no row-major image of its upstream gradient)
extern "C" int medtok_half_image_t_sums_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int64_t group_cols, int bf16,
                                            void *out, float *col_partials, void *stream)
{
    if (!col_partials) return fail("half_image_t_sums: NULL argument");
    return half_image_impl(src, n, d, src_stride, dp, 1, group_cols, bf16, out, col_partials, stream);
}
static int half_image_impl(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int transpose, int64_t group_cols, int bf16,
                           void *out, float *col_partials, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3) || (dp & 7) || src_stride < d || (src_stride & 3) || dp < (transpose ? n : (int64_t)d))
        return fail("half_image: bad shape n=%ld d=%d stride=%ld dp=%ld transpose=%d", (long)n, d, (long)src_stride, (long)dp, transpose);
    if (group_cols == 0) group_cols = dp;
    if (transpose && (group_cols <= 0 || dp % group_cols || (group_cols != dp && group_cols % 64)))
        return fail("half_image: group_cols=%ld must divide dp=%ld and be a multiple of 64", (long)group_cols, (long)dp);
    if (n == 0 && !transpose) return 0;
    if ((!src && n) || !out) return fail("half_image: NULL argument");
    if (((uintptr_t)src | (uintptr_t)out) & 15) return fail("half_image: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (!transpose) {
        if (dp >= (1ll << 31)) return fail("half_image: dp too large");
        const long total = n * (dp / 8);
        const dim3 grid((unsigned)lmin(8192, (total + 255) / 256));
        if (bf16) hipLaunchKernelGGL(half_image_kernel<true>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (int)dp, (unsigned short *)out);
        else hipLaunchKernelGGL(half_image_kernel<false>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (int)dp, (unsigned short *)out);
        return check_launch("half_image");
    }
    const long row_tiles = (dp + 63) / 64;
    if (row_tiles >= (1ll << 31) || (d + 63) / 64 > 65535) return fail("half_image: too large");
    const dim3 grid((unsigned)row_tiles, (unsigned)((d + 63) / 64));
    if (bf16) hipLaunchKernelGGL(half_image_t_kernel<true>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (long)dp, (long)group_cols, (unsigned short *)out, (unsigned short *)nullptr, 0, col_partials);
    else hipLaunchKernelGGL(half_image_t_kernel<false>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (long)dp, (long)group_cols, (unsigned short *)out, (unsigned short *)nullptr, 0, col_partials);
    return check_launch("half_image(transposed)");
}

// both images of one matrix in one pass over it: out_plain [n, dp_plain] (zero columns past d; dp_plain <= d rounded up to 64) and
// out_t = the transposed image of medtok_half_image_f32(transpose = 1, dp = np, group_cols)
static int half_image_pair_impl(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                                int bf16, void *out_plain, void *out_t, float *col_partials, void *stream);
extern "C" int medtok_half_image_pair_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                                          int bf16, void *out_plain, void *out_t, void *stream)
{
    return half_image_pair_impl(src, n, d, src_stride, dp_plain, np, group_cols, bf16, out_plain, out_t, nullptr, stream);
}
// ... and, from the same pass, col_partials [(np + 63) / 64, d] fp32: the sums over every 64-row tile of each column (fixed order): the
// column sums of src -- a Linear's bias gradient when src is its upstream gradient -- are the sum of these rows
extern "C" int medtok_half_image_pair_sums_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                                               int bf16, void *out_plain, void *out_t, float *col_partials, void *stream)
{
    if (!col_partials) return fail("half_image_pair_sums: NULL argument");
    return half_image_pair_impl(src, n, d, src_stride, dp_plain, np, group_cols, bf16, out_plain, out_t, col_partials, stream);
}
static int half_image_pair_impl(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                                int bf16, void *out_plain, void *out_t, float *col_partials, void *stream)
{
    if (n <= 0 || d <= 0 || (d & 3) || (np & 7) || (dp_plain & 7) || src_stride < d || (src_stride & 3) || np < n || dp_plain < d || dp_plain > (d + 63) / 64 * 64)
        return fail("half_image_pair: bad shape n=%ld d=%d stride=%ld dp_plain=%ld np=%ld", (long)n, d, (long)src_stride, (long)dp_plain, (long)np);
    if (group_cols == 0) group_cols = np;
    if (group_cols <= 0 || np % group_cols || (group_cols != np && group_cols % 64)) return fail("half_image_pair: group_cols=%ld must divide np=%ld and be a multiple of 64", (long)group_cols, (long)np);
    if (!src || !out_plain || !out_t) return fail("half_image_pair: NULL argument");
    if (((uintptr_t)src | (uintptr_t)out_plain | (uintptr_t)out_t) & 15) return fail("half_image_pair: pointers must be 16-byte aligned");
    const long row_tiles = (np + 63) / 64;
    if (row_tiles >= (1ll << 31) || (d + 63) / 64 > 65535) return fail("half_image_pair: too large");
    const dim3 grid((unsigned)row_tiles, (unsigned)((d + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    if (bf16) hipLaunchKernelGGL(half_image_t_kernel<true>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (long)np, (long)group_cols, (unsigned short *)out_t, (unsigned short *)out_plain, (int)dp_plain, col_partials);
    else hipLaunchKernelGGL(half_image_t_kernel<false>, grid, dim3(256), 0, s, src, (long)n, d, (long)src_stride, (long)np, (long)group_cols, (unsigned short *)out_t, (unsigned short *)out_plain, (int)dp_plain, col_partials);
    return check_launch("half_image_pair");
}

static bool g_gemm_k32 = false;        // DEV (tools/r06): the one-pass products with 32-deep stages, as before round 6
static bool g_gemm_by_row_tile = false; // DEV: tile ids by row tile whatever the row-tile count (bit 1 of the switch)
extern "C" void medtok_debug_set_half_gemm_k32(int on) { g_gemm_k32 = (on & 1) != 0; g_gemm_by_row_tile = (on & 2) != 0; }

static int split_gemm_impl(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols,
                           const void *b_hi, const void *b_lo, int64_t b_rows, int ldb, int b_group_rows,
                           int n_g, int k_g, int groups, const float *bias, float unscale, const float *amax_a, const float *amax_b,
                           float *c, int ldc, void *c_hi, void *c_lo, int ldch, void *stream, int one_pass = 0);

// C = unscale * (A . B^T) + bias in ONE half-precision pass with fp32 accumulation: a [m, lda], b [b_rows, ldb] are fp16 (bf16 = 0) or
// bf16 (bf16 = 1) matrices -- the product torch.autocast makes of an nn.Linear (train_MedTok.py:212,394); grouped as medtok_split_gemm_f16.
extern "C" int medtok_half_gemm_f32(const void *a, int64_t m, int lda, int a_group_cols, const void *b, int64_t b_rows, int ldb, int b_group_rows,
                                    int n_g, int k_g, int groups, const float *bias, float unscale, float *c, int ldc, int bf16, void *stream)
{
    return split_gemm_impl(a, a, m, lda, a_group_cols, b, b, b_rows, ldb, b_group_rows, n_g, k_g, groups, bias, unscale, nullptr, nullptr, c, ldc,
                           nullptr, nullptr, 0, stream, bf16 ? 2 : 1);
}

extern "C" int medtok_split_gemm_scaled_f16(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols, const void *b_hi, const void *b_lo,
                                            int64_t b_rows, int ldb, int b_group_rows, int n_g, int k_g, int groups, const float *bias, float unscale,
                                            const float *amax_a, const float *amax_b, float *c, int ldc, void *stream)
{
    return split_gemm_impl(a_hi, a_lo, m, lda, a_group_cols, b_hi, b_lo, b_rows, ldb, b_group_rows, n_g, k_g, groups, bias, unscale, amax_a, amax_b, c, ldc,
                           nullptr, nullptr, 0, stream);
}

extern "C" int medtok_split_gemm_f16(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols,
                                     const void *b_hi, const void *b_lo, int64_t b_rows, int ldb, int b_group_rows,
                                     int n_g, int k_g, int groups, const float *bias, float unscale,
                                     float *c, int ldc, void *c_hi, void *c_lo, int ldch, void *stream)
{
    return split_gemm_impl(a_hi, a_lo, m, lda, a_group_cols, b_hi, b_lo, b_rows, ldb, b_group_rows, n_g, k_g, groups, bias, unscale, nullptr, nullptr,
                           c, ldc, c_hi, c_lo, ldch, stream);
}

static int split_gemm_impl(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols,
                           const void *b_hi, const void *b_lo, int64_t b_rows, int ldb, int b_group_rows,
                           int n_g, int k_g, int groups, const float *bias, float unscale, const float *amax_a, const float *amax_b,
                           float *c, int ldc, void *c_hi, void *c_lo, int ldch, void *stream, int one_pass)
{
    if (m < 0 || groups < 1 || n_g <= 0 || k_g <= 0 || (n_g & 3) || (k_g % G_BK) || (lda & 7) || (ldb & 7) || (a_group_cols & 7) || lda < k_g || ldb < k_g)
        return fail("split_gemm: bad shape m=%ld groups=%d n_g=%d k_g=%d lda=%d ldb=%d a_group_cols=%d (n_g %% 4 == 0, k_g %% 32 == 0, strides %% 8 == 0)",
                    (long)m, groups, n_g, k_g, lda, ldb, a_group_cols);
    if ((long)(groups - 1) * a_group_cols + k_g > lda || (long)(groups - 1) * b_group_rows + n_g > b_rows)
        return fail("split_gemm: the last group reads past its operand (A columns %ld > lda %d or B rows %ld > %ld)",
                    (long)(groups - 1) * a_group_cols + k_g, lda, (long)(groups - 1) * b_group_rows + n_g, (long)b_rows);
    if (m == 0) return 0;
    if (!a_hi || !a_lo || !b_hi || !b_lo || (!c && !c_hi)) return fail("split_gemm: NULL argument");
    if ((c_hi == nullptr) != (c_lo == nullptr)) return fail("split_gemm: c_hi and c_lo go together");
    if (c && ((ldc & 3) || ldc < groups * n_g)) return fail("split_gemm: ldc=%d must be a multiple of 4 and >= groups * n_g", ldc);
    if (c_hi && ((ldch & 3) || ldch < groups * n_g)) return fail("split_gemm: ldch=%d must be a multiple of 4 and >= groups * n_g", ldch);
    if (((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)b_hi | (uintptr_t)b_lo | (uintptr_t)c | (uintptr_t)c_hi | (uintptr_t)c_lo | (uintptr_t)bias) & 15)
        return fail("split_gemm: pointers must be 16-byte aligned");
    if ((double)m * lda * 2 >= 2147483647.0 * 256 || (double)b_rows * ldb * 2 >= 2147483647.0) return fail("split_gemm: operand too large");
    SplitGemmArgs p;
    p.ah = (const _Float16 *)a_hi; p.al = (const _Float16 *)a_lo; p.bh = (const _Float16 *)b_hi; p.bl = (const _Float16 *)b_lo;
    p.bias = bias; p.c = c; p.ch = (_Float16 *)c_hi; p.cl = (_Float16 *)c_lo;
    p.M = (long)m; p.a_bytes = (long)m * lda * 2; p.b_bytes = (long)b_rows * ldb * 2;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldch = ldch; p.n_g = n_g; p.k_g = k_g; p.groups = groups;
    p.a_group_cols = a_group_cols; p.b_group_rows = b_group_rows; p.unscale = unscale; p.amax_a = amax_a; p.amax_b = amax_b;
    // tile height: 256 features, or 192 where that pads the group's features by > 10 % less (n_g = 192: the per-head W_v product)
    const long pad4 = (long)((n_g + 255) / 256) * 256, pad3 = (long)((n_g + 191) / 192) * 192;
    const bool mt3 = pad3 * 10 < pad4 * 9;
    p.row_tiles = (int)((m + G_BN - 1) / G_BN);
    const long row_ids = (long)((p.row_tiles + 7) / 8) * 8;
    // ... or fewer where the 256-feature tiles would leave most of the chip without a block (a product of a few thousand rows: the
    // text side of a forward, the node side of a training batch): 128 or 64 features per tile double / quadruple the blocks at the
    // same bytes per block-stage of the activation tile -- a launch of that size is latency, not throughput
    int mt = mt3 ? 3 : 4;
    const long half_chip = dev_info().cus / 2;
    if (!mt3 && row_ids * ((n_g + 255) / 256) * groups < half_chip) mt = row_ids * ((n_g + 127) / 128) * groups < half_chip ? 1 : 2;
    const int bm = 64 * mt;
    p.ftiles = (n_g + bm - 1) / bm;
    // few row tiles (less than 0.8 of the 8-padded count: a weight-gradient product's 3 or 12, the 1 of a small batch): the dense order
    // (split_gemm.h) instead of "XCD = row tile mod 8", which would leave XCDs without a tile
    const long n_tiles = (long)p.row_tiles * p.ftiles * groups;
    p.per_xcd = (!g_gemm_by_row_tile && p.row_tiles * 10 < row_ids * 8) ? (int)((n_tiles + 7) / 8) : 0;
    const long ids = p.per_xcd ? (long)p.per_xcd * 8 : row_ids * p.ftiles * groups;
    if (ids >= (1ll << 31)) return fail("split_gemm: grid too large");
    // persistent: one block per CU (a multiple of 8: the XCD round-robin; never fewer than 8 -- a CU-masked or partitioned device
    // with < 8 CUs still gets a valid launch, its blocks just share CUs)
    const long blocks = lmin(ids, lmax(8, (long)(dev_info().cus / 8) * 8));
    const size_t lds = mt == 3 ? GemmShape<3>::LDS_BYTES : mt == 4 ? GemmShape<4>::LDS_BYTES : mt == 2 ? GemmShape<2>::LDS_BYTES : GemmShape<1>::LDS_BYTES;
    hipEvent_t pa = nullptr;
#define MEDTOK_GEMM_LAUNCH(...)                                                                                                   \
    do {                                                                                                                          \
        if (!set_lds_once<split_gemm_kernel<__VA_ARGS__>>(lds)) return fail("split_gemm: cannot reserve %zu bytes of LDS", lds);  \
        pa = prof_wanted(4) ? prof_mark((hipStream_t)stream) : nullptr;                                                                \
        hipLaunchKernelGGL((split_gemm_kernel<__VA_ARGS__>), dim3((unsigned)blocks), dim3(G_THREADS), lds, (hipStream_t)stream, p); \
    } while (0)
#define MEDTOK_GEMM_BY_MT(...)                                                                                                    \
    do {                                                                                                                          \
        if (mt == 3) MEDTOK_GEMM_LAUNCH(3 __VA_ARGS__); else if (mt == 4) MEDTOK_GEMM_LAUNCH(4 __VA_ARGS__);                      \
        else if (mt == 2) MEDTOK_GEMM_LAUNCH(2 __VA_ARGS__); else MEDTOK_GEMM_LAUNCH(1 __VA_ARGS__);                              \
    } while (0)
    // (one pass: 64-deep stages wherever the depth allows -- a copy then has 32 MFMAs of time to arrive instead of 16)
    const bool k64 = one_pass != 0 && k_g % (2 * G_BK) == 0 && !g_gemm_k32;
    if (one_pass == 0) MEDTOK_GEMM_BY_MT();
    else if (one_pass == 1) { if (k64) MEDTOK_GEMM_BY_MT(, true, false, true); else MEDTOK_GEMM_BY_MT(, true, false); }
    else { if (k64) MEDTOK_GEMM_BY_MT(, true, true, true); else MEDTOK_GEMM_BY_MT(, true, true); }
#undef MEDTOK_GEMM_BY_MT
#undef MEDTOK_GEMM_LAUNCH
    if (pa) prof_push(pa, prof_mark((hipStream_t)stream), 2.0 * (double)m * (double)n_g * (double)k_g * (double)groups, 4);     // fp32-equivalent flops (x3 on the fp16 pipe)
    return check_launch("split_gemm");
}

extern "C" int medtok_shared_kv_attention_train_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                    const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                    int d, float scale, float dropout_p, uint32_t seed, float *out, float *lse, void *stream)
{
    if (n_codes < 0 || max_q_len < 0) return fail("shared_kv_attention_train: bad sizes n_codes=%ld max_q_len=%ld", (long)n_codes, (long)max_q_len);
    if (!attention_shape_ok(d)) return fail("shared_kv_attention_train: d=%d must be 64 or a multiple of 128, at most 768", d);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail("shared_kv_attention_train: dropout_p=%g must be in [0, 1)", (double)dropout_p);
    if (n_codes == 0 || max_q_len == 0) return 0;
    if (!q || !q_start || !q_len || !kv || !kv_start || !kv_len || !out || !lse) return fail("shared_kv_attention_train: NULL argument");
    return attention_forward(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, d, scale, out, lse, dropout_p, seed, (hipStream_t)stream);
}

// The training forward on the three-pass fp16 products (attention_pp.h, TRAIN form): two 32-row query tiles of a code per block on one
// copy of its keys, which the kernel turns from the caller's fp32 rows into (hi, lo) images itself.  Same outputs as
// medtok_shared_kv_attention_train_f32 to ~1e-6 relative (same dropout mask bits), about a third of its time at d = 768; d = 256, 512, 768.
extern "C" int medtok_shared_kv_attention_train_split_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                          const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                          int d, float scale, float dropout_p, uint32_t seed, float *out, float *lse, void *stream)
{
    if (n_codes < 0 || max_q_len < 0) return fail("shared_kv_attention_train_split: bad sizes n_codes=%ld max_q_len=%ld", (long)n_codes, (long)max_q_len);
    if (d != 256 && d != 512 && d != 768) return fail("shared_kv_attention_train_split: d=%d must be 256, 512 or 768", d);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail("shared_kv_attention_train_split: dropout_p=%g must be in [0, 1)", (double)dropout_p);
    if (n_codes == 0 || max_q_len == 0) return 0;
    if (!q || !q_start || !q_len || !kv || !kv_start || !kv_len || !out || !lse) return fail("shared_kv_attention_train_split: NULL argument");
    if (((uintptr_t)q | (uintptr_t)kv | (uintptr_t)out) & 15) return fail("shared_kv_attention_train_split: pointers must be 16-byte aligned");
    const int64_t q_pairs = (max_q_len + 63) / 64;
    if (q_pairs * (n_codes + 8) >= (1ll << 31)) return fail("shared_kv_attention_train_split: grid limit exceeded");
    hipStream_t s = (hipStream_t)stream;
    const unsigned thresh = dropout_p > 0.f ? (unsigned)fmin(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    const float keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    hipEvent_t pa = prof_wanted(2) ? prof_mark(s) : nullptr;
#define MEDTOK_ATT_PP_TRAIN(NT)                                                                                                   \
    do {                                                                                                                          \
        const size_t lds = AttPP<NT>::LDS_BYTES;                                                                                  \
        if (!set_lds_once<shared_kv_attention_pp_kernel<NT, false, true, true, true>>(lds))                                        \
            return fail("shared_kv_attention_train_split: cannot reserve %zu bytes of LDS", lds);                                 \
        hipLaunchKernelGGL((shared_kv_attention_pp_kernel<NT, false, true, true, true>), dim3((unsigned)(q_pairs * ((n_codes + 7) / 8 * 8))), dim3(512), lds, s, \
                           q, q_start, q_len, (const _Float16 *)kv, (const _Float16 *)nullptr, kv_start, kv_len, scale, out, (_Float16 *)nullptr,  \
                           (_Float16 *)nullptr, (int)q_pairs, (int)n_codes, (unsigned long long *)nullptr, lse, thresh, seed, keep_scale); \
    } while (0)
    if (d == 256) MEDTOK_ATT_PP_TRAIN(2); else if (d == 512) MEDTOK_ATT_PP_TRAIN(4); else MEDTOK_ATT_PP_TRAIN(6);
#undef MEDTOK_ATT_PP_TRAIN
    if (pa) prof_push(pa, prof_mark(s), 0.0, 2);
    return check_launch("shared_kv_attention_train_split");
}

extern "C" size_t medtok_shared_kv_attention_backward_workspace_bytes(int64_t q_rows) { return q_rows > 0 ? align_up((size_t)q_rows * 4, 256) : 256; }

static int attention_backward_impl(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                   const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                   int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                   uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                   float *dkv, void *ws, size_t ws_bytes, void *stream, int hm, int acc_dkv = 0);

extern "C" int medtok_shared_kv_attention_backward_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                       const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                       int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                                       uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                                       float *dkv, void *ws, size_t ws_bytes, void *stream)
{
    return attention_backward_impl(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, max_kv_len, q_rows, kv_rows, d, scale, dropout_p, seed, out,
                                   lse, d_out, dq, dkv, ws, ws_bytes, stream, 0);
}

// the same backward with its four matrix products in ONE half-precision pass (fp16; bf16 != 0: bf16) with fp32 accumulation -- the
// precision class of nn.MultiheadAttention under torch.autocast (train_MedTok.py:212,394); operands, softmax rebuild and outputs fp32
extern "C" int medtok_shared_kv_attention_backward_half_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                            const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                            int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                                            uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                                            float *dkv, void *ws, size_t ws_bytes, int bf16, void *stream)
{
    return attention_backward_impl(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, max_kv_len, q_rows, kv_rows, d, scale, dropout_p, seed, out,
                                   lse, d_out, dq, dkv, ws, ws_bytes, stream, bf16 ? 2 : 1);
}

// ... the same (mode 0: exact fp32, 1: fp16, 2: bf16 products) with dKV ADDED to what dkv already holds (accumulate_dkv != 0): the
// layers of CrossAttention all attend to the ORIGINAL other modality (:83,86), so their key gradients land in one buffer -- no second
// [kv_rows, d] tensor, no zero fill of it, no pass that adds the two.  Rows no block owns are then left as they are.
extern "C" int medtok_shared_kv_attention_backward_acc_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                           const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                           int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                                           uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                                           float *dkv, void *ws, size_t ws_bytes, int mode, int accumulate_dkv, void *stream)
{
    if (mode < 0 || mode > 2) return fail("shared_kv_attention_backward_acc: mode=%d must be 0 (fp32), 1 (fp16) or 2 (bf16)", mode);
    return attention_backward_impl(q, q_start, q_len, kv, kv_start, kv_len, n_codes, max_q_len, max_kv_len, q_rows, kv_rows, d, scale, dropout_p, seed, out,
                                   lse, d_out, dq, dkv, ws, ws_bytes, stream, mode, accumulate_dkv == 2 ? 2 : (accumulate_dkv != 0));
}

static int attention_backward_impl(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                   const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                   int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                   uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                   float *dkv, void *ws, size_t ws_bytes, void *stream, int hm, int acc_dkv)
{
    if (n_codes < 0 || max_q_len < 0 || max_kv_len < 0 || q_rows < 0 || kv_rows < 0) return fail("shared_kv_attention_backward: bad sizes");
    if (!attention_shape_ok(d)) return fail("shared_kv_attention_backward: d=%d must be 64 or a multiple of 128, at most 768", d);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail("shared_kv_attention_backward: dropout_p=%g must be in [0, 1)", (double)dropout_p);
    if (q_rows == 0 && kv_rows == 0) return 0;                 // nothing to write
    const bool dq_only = acc_dkv == 2;                         // (the caller takes the key gradient later: medtok_shared_kv_attention_dkv_multi_f32)
    if (!q_start || !q_len || !kv_start || !kv_len || (q_rows > 0 && (!q || !out || !lse || !d_out || !dq)) || (kv_rows > 0 && (!kv || (!dkv && !dq_only))))
        return fail("shared_kv_attention_backward: NULL argument");
    if (!ws || ws_bytes < (size_t)q_rows * 4) return fail("shared_kv_attention_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    // rows no block owns (key rows past a code's kv_len inside its slot, query rows of no code) get a zero gradient
    if (q_rows > 0 && hipMemsetAsync(dq, 0, (size_t)q_rows * d * 4, s) != hipSuccess) return fail("shared_kv_attention_backward: memset failed");
    if (kv_rows > 0 && !acc_dkv && hipMemsetAsync(dkv, 0, (size_t)kv_rows * d * 4, s) != hipSuccess) return fail("shared_kv_attention_backward: memset failed");
    DkvSources srcs;
    memset(&srcs, 0, sizeof srcs);
    if (n_codes == 0 || q_rows == 0) return 0;
    float *delta = (float *)ws;
    hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)((q_rows + 3) / 4)), dim3(256), 0, s, d_out, out, (long)q_rows, d, delta);
    const int64_t q_tiles = (max_q_len + 31) / 32, kv_tiles = (max_kv_len + 31) / 32;
    if (q_tiles * n_codes >= (1ll << 31) || kv_tiles * n_codes >= (1ll << 31)) return fail("shared_kv_attention_backward: grid limit exceeded");
    const unsigned thresh = dropout_p > 0.f ? (unsigned)fmin(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    const float keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    srcs.count = 1;
    srcs.s[0] = DkvSource{q, d_out, lse, delta, q_start, q_len, scale, keep_scale, thresh, seed};
    hipEvent_t pa_bwd = prof_wanted(3) ? prof_mark(s) : nullptr;
#define MEDTOK_ATT_BWD_HM(W, NT, HM)                                                                                             \
    do {                                                                                                                         \
        const size_t lds = AttShape<W, NT>::LDS_FLOATS * sizeof(float);                                                          \
        if (lds > 64 * 1024 && (!set_lds_once<shared_kv_attention_dq_kernel<W, NT, HM>>(lds) || !set_lds_once<shared_kv_attention_dkv_kernel<W, NT, HM>>(lds))) \
            return fail("shared_kv_attention_backward: cannot reserve %zu bytes of LDS", lds);                                   \
        if (q_tiles > 0)                                                                                                         \
            hipLaunchKernelGGL((shared_kv_attention_dq_kernel<W, NT, HM>), dim3((unsigned)(q_tiles * n_codes)), dim3(64 * W), lds, s, q, q_start, q_len, \
                               kv, kv_start, kv_len, d_out, lse, delta, scale, dq, (int)q_tiles, thresh, seed, keep_scale);      \
        if (kv_tiles > 0 && !dq_only)                                                                                            \
            hipLaunchKernelGGL((shared_kv_attention_dkv_kernel<W, NT, HM>), dim3((unsigned)(kv_tiles * n_codes)), dim3(64 * W), lds, s, srcs, \
                               kv, kv_start, kv_len, dkv, (int)kv_tiles, acc_dkv);                                               \
    } while (0)
#define MEDTOK_ATT_BWD(W, NT)                                                                                                    \
    do {                                                                                                                         \
        if (hm == 0) MEDTOK_ATT_BWD_HM(W, NT, 0); else if (hm == 1) MEDTOK_ATT_BWD_HM(W, NT, 1); else MEDTOK_ATT_BWD_HM(W, NT, 2); \
    } while (0)
    switch (d / 128) {
    case 0: MEDTOK_ATT_BWD(2, 1); break;
    case 1: MEDTOK_ATT_BWD(4, 1); break;
    case 2: MEDTOK_ATT_BWD(8, 1); break;
    case 3: MEDTOK_ATT_BWD(4, 3); break;
    case 4: MEDTOK_ATT_BWD(8, 2); break;
    case 5: MEDTOK_ATT_BWD(4, 5); break;
    default: MEDTOK_ATT_BWD(8, 3); break;
    }
#undef MEDTOK_ATT_BWD_HM
#undef MEDTOK_ATT_BWD
    if (pa_bwd) prof_push(pa_bwd, prof_mark(s), 0.0, 3);     // dQ + dKV; the caller prices the pair (ragged counts live on the device)
    return check_launch("shared_kv_attention_backward");
}

// The key gradient of SEVERAL attention calls over the same keys in one launch (attention_backward.h: DkvSources): every source brings its
// queries, upstream gradient, log-sum-exp, delta (= <d_out, out> per query row: the workspace a dQ-only call of
// medtok_shared_kv_attention_backward_acc_f32 filled), dropout parameters and its own q_start / q_len; kv, kv_start, kv_len are common.
// dkv [kv_rows, d] is written once (rows no block owns zeroed).  mode: 0 exact fp32, 1 fp16, 2 bf16 products.
extern "C" int medtok_shared_kv_attention_dkv_multi_f32(const medtok_dkv_source *sources, int count, const float *kv, const int64_t *kv_start,
                                                        const int64_t *kv_len, int64_t n_codes, int64_t max_kv_len, int64_t kv_rows, int d,
                                                        float *dkv, int mode, void *stream)
{
    if (!sources || count < 1 || count > DKV_SRC_MAX) return fail("shared_kv_attention_dkv_multi: 1..%d sources per call", DKV_SRC_MAX);
    if (n_codes < 0 || max_kv_len < 0 || kv_rows < 0) return fail("shared_kv_attention_dkv_multi: bad sizes");
    if (!attention_shape_ok(d)) return fail("shared_kv_attention_dkv_multi: d=%d must be 64 or a multiple of 128, at most 768", d);
    if (mode < 0 || mode > 2) return fail("shared_kv_attention_dkv_multi: mode=%d must be 0 (fp32), 1 (fp16) or 2 (bf16)", mode);
    if (kv_rows == 0) return 0;
    if (!kv || !kv_start || !kv_len || !dkv) return fail("shared_kv_attention_dkv_multi: NULL argument");
    DkvSources srcs;
    memset(&srcs, 0, sizeof srcs);
    srcs.count = count;
    for (int i = 0; i < count; ++i) {
        const medtok_dkv_source &m = sources[i];
        if (!m.q || !m.d_out || !m.lse || !m.delta || !m.q_start || !m.q_len) return fail("shared_kv_attention_dkv_multi: NULL argument in source %d", i);
        if (!(m.dropout_p >= 0.f && m.dropout_p < 1.f)) return fail("shared_kv_attention_dkv_multi: dropout_p=%g must be in [0, 1)", (double)m.dropout_p);
        const unsigned thresh = m.dropout_p > 0.f ? (unsigned)fmin(4294967295.0, (double)m.dropout_p * 4294967296.0) : 0u;
        srcs.s[i] = DkvSource{m.q, m.d_out, m.lse, m.delta, m.q_start, m.q_len, m.scale, m.dropout_p > 0.f ? 1.f / (1.f - m.dropout_p) : 1.f, thresh, m.seed};
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dkv, 0, (size_t)kv_rows * d * 4, s) != hipSuccess) return fail("shared_kv_attention_dkv_multi: memset failed");
    const int64_t kv_tiles = (max_kv_len + 31) / 32;
    if (n_codes == 0 || kv_tiles == 0) return 0;
    if (kv_tiles * n_codes >= (1ll << 31)) return fail("shared_kv_attention_dkv_multi: grid limit exceeded");
    hipEvent_t pa = prof_wanted(3) ? prof_mark(s) : nullptr;
#define MEDTOK_DKV_HM(W, NT, HM)                                                                                                 \
    do {                                                                                                                         \
        const size_t lds = AttShape<W, NT>::LDS_FLOATS * sizeof(float);                                                          \
        if (lds > 64 * 1024 && !set_lds_once<shared_kv_attention_dkv_kernel<W, NT, HM>>(lds))                                    \
            return fail("shared_kv_attention_dkv_multi: cannot reserve %zu bytes of LDS", lds);                                  \
        hipLaunchKernelGGL((shared_kv_attention_dkv_kernel<W, NT, HM>), dim3((unsigned)(kv_tiles * n_codes)), dim3(64 * W), lds, s, srcs, \
                           kv, kv_start, kv_len, dkv, (int)kv_tiles, 0);                                                         \
    } while (0)
#define MEDTOK_DKV(W, NT)                                                                                                        \
    do {                                                                                                                         \
        if (mode == 0) MEDTOK_DKV_HM(W, NT, 0); else if (mode == 1) MEDTOK_DKV_HM(W, NT, 1); else MEDTOK_DKV_HM(W, NT, 2);        \
    } while (0)
    switch (d / 128) {
    case 0: MEDTOK_DKV(2, 1); break;
    case 1: MEDTOK_DKV(4, 1); break;
    case 2: MEDTOK_DKV(8, 1); break;
    case 3: MEDTOK_DKV(4, 3); break;
    case 4: MEDTOK_DKV(8, 2); break;
    case 5: MEDTOK_DKV(4, 5); break;
    default: MEDTOK_DKV(8, 3); break;
    }
#undef MEDTOK_DKV_HM
#undef MEDTOK_DKV
    if (pa) prof_push(pa, prof_mark(s), 0.0, 3);
    return check_launch("shared_kv_attention_dkv_multi");
}

// ================================================================= EMA statistics
// bins: integer histogram.  embed_sum: rows are ordered by (code, row) with a stable LSD radix
// sort (8-bit digits), then one wavefront per code adds its rows in increasing row order.
constexpr int SORT_BLOCKS = 64;        // x 4 waves = 256 sorting waves (the digit x wave count table is scanned by ONE block)
constexpr int SORT_WAVES = SORT_BLOCKS * 4;

// ids outside [0, K) are clamped (same clamp in the sort) so the layout stays consistent
__device__ __forceinline__ uint32_t clamp_code(int64_t c, int k_codes)
{
    return (uint32_t)(c < 0 ? 0 : (c >= k_codes ? k_codes - 1 : c));
}

__global__ __launch_bounds__(256) void hist_kernel(const int64_t *__restrict__ idx, long n, int k_codes, int *__restrict__ counts)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        atomicAdd(&counts[clamp_code(idx[i], k_codes)], 1);
}

// exclusive scan of counts[0..k) -> offsets[0..k] (single block, fixed order).  A thread owns 16 consecutive entries of each of
// up to four 16384-entry slabs per pass (a wave's loads cover 4 KB of consecutive memory): per-thread sums, a shuffle scan inside
// the wave, one scan of the 64 (slab, wave) totals by wave 0, then the entries are written with their running offsets.  (The first
// form gave every thread one long contiguous chunk -- 64 lanes 256 bytes apart -- and took 30-40 us for the 65536-entry digit
// table of a radix pass.)
__global__ __launch_bounds__(1024) void scan_kernel(const int *__restrict__ counts, int k, int *__restrict__ offsets)
{
    constexpr int PER = 16, SLABS = 4;
    __shared__ int wave_tot[SLABS * 16], wave_pre[SLABS * 16], carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool vec = ((reinterpret_cast<uintptr_t>(counts) | reinterpret_cast<uintptr_t>(offsets)) & 15) == 0;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < k; base += SLABS * 1024 * PER) {
        int v[SLABS][PER], incl[SLABS], sum[SLABS];
#pragma unroll
        for (int j = 0; j < SLABS; ++j) {
            const int start = base + (j * 1024 + t) * PER;
            if (vec && start + PER <= k) {
#pragma unroll
                for (int c = 0; c < PER / 4; ++c) {
                    const int4 q = *reinterpret_cast<const int4 *>(counts + start + 4 * c);
                    v[j][4 * c] = q.x; v[j][4 * c + 1] = q.y; v[j][4 * c + 2] = q.z; v[j][4 * c + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < PER; ++c) v[j][c] = start + c < k ? counts[start + c] : 0;
            }
            int sm = 0;
#pragma unroll
            for (int c = 0; c < PER; ++c) sm += v[j][c];
            sum[j] = sm;
            int in = sm;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(in, off, 64);
                if (lane >= off) in += o;
            }
            incl[j] = in;
            if (lane == 63) wave_tot[j * 16 + wave] = in;
        }
        __syncthreads();
        if (t < 64) {                                   // the 64 (slab, wave) totals in order
            const int carry = carry_s;                  // (only this wave touches carry_s inside the loop: read, then written below, in program order)
            const int x = wave_tot[t];
            int in = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(in, off, 64);
                if (lane >= off) in += o;
            }
            wave_pre[t] = carry + in - x;
            if (t == 63) carry_s = carry + in;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SLABS; ++j) {
            const int start = base + (j * 1024 + t) * PER;
            int run = wave_pre[j * 16 + wave] + incl[j] - sum[j];
            if (vec && start + PER <= k) {
#pragma unroll
                for (int c = 0; c < PER / 4; ++c) {
                    int4 o;
                    o.x = run; o.y = o.x + v[j][4 * c]; o.z = o.y + v[j][4 * c + 1]; o.w = o.z + v[j][4 * c + 2];
                    run = o.w + v[j][4 * c + 3];
                    *reinterpret_cast<int4 *>(offsets + start + 4 * c) = o;
                }
            } else {
#pragma unroll
                for (int c = 0; c < PER; ++c)
                    if (start + c < k) { offsets[start + c] = run; run += v[j][c]; }
            }
        }
        __syncthreads();                                // wave_tot / wave_pre are reused by the next pass
    }
    if (t == 0) offsets[k] = carry_s;
}

__device__ __forceinline__ void wave_chunk(long n, int gw, long &lo, long &hi)
{
    const long per = ((n + SORT_WAVES - 1) / SORT_WAVES + 63) / 64 * 64;
    lo = min(n, (long)gw * per);
    hi = min(n, lo + per);
}

// pass 0 reads keys from idx (payload = position); later passes read (key,payload) pairs
template <bool FIRST>
__global__ __launch_bounds__(256) void radix_count_kernel(const int64_t *__restrict__ idx, const uint32_t *__restrict__ keys_in,
                                                          long n, int k_codes, int shift, int *__restrict__ wave_counts)
{
    __shared__ int cnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w;
    for (int i = lane; i < 256; i += 64) cnt[w][i] = 0;
    __syncthreads();
    long lo, hi;
    wave_chunk(n, gw, lo, hi);
    for (long i = lo + lane; i < hi; i += 64) {
        const uint32_t key = FIRST ? clamp_code(idx[i], k_codes) : keys_in[i];
        atomicAdd(&cnt[w][(key >> shift) & 255], 1);
    }
    __syncthreads();
    // digit-major so the scan below yields stable destinations
    for (int i = lane; i < 256; i += 64) wave_counts[(long)i * SORT_WAVES + gw] = cnt[w][i];
}

template <bool FIRST>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const int64_t *__restrict__ idx, const uint32_t *__restrict__ keys_in,
                                                            const uint32_t *__restrict__ vals_in, long n, int k_codes, int shift,
                                                            const int *__restrict__ wave_offsets, uint32_t *__restrict__ keys_out,
                                                            uint32_t *__restrict__ vals_out)
{
    __shared__ int base[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w;
    for (int i = lane; i < 256; i += 64) base[w][i] = wave_offsets[(long)i * SORT_WAVES + gw];
    __syncthreads();
    long lo, hi;
    wave_chunk(n, gw, lo, hi);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (long i0 = lo; i0 < hi; i0 += 64) {
        const long i = i0 + lane;
        const bool live = i < hi;
        uint32_t key = 0, val = 0;
        if (live) { key = FIRST ? clamp_code(idx[i], k_codes) : keys_in[i]; val = FIRST ? (uint32_t)i : vals_in[i]; }
        const uint32_t dg = (key >> shift) & 255;
        unsigned long long eq = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((dg >> b) & 1);
            eq &= ((dg >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(eq & lt_mask);
        const int total = __popcll(eq);
        int dst = 0;
        if (live) dst = base[w][dg] + rank;
        __builtin_amdgcn_wave_barrier();
        if (live && rank == total - 1) base[w][dg] += total;   // one lane per digit group
        __builtin_amdgcn_wave_barrier();
        if (live) { keys_out[dst] = key; vals_out[dst] = val; }
    }
}

// One wavefront per code: add that code's rows in increasing row order.
__global__ __launch_bounds__(256) void segsum_kernel(const float *__restrict__ zhat, const uint32_t *__restrict__ sorted_rows,
                                                     const int *__restrict__ offsets, int k_codes, int d,
                                                     float *__restrict__ bins, float *__restrict__ embed_sum)
{
    const int lane = threadIdx.x & 63;
    for (long code = (long)blockIdx.x * 4 + (threadIdx.x >> 6); code < k_codes; code += (long)gridDim.x * 4) {
        const int lo = offsets[code], hi = offsets[code + 1];
        if (lane == 0) bins[code] = (float)(hi - lo);
        for (int i = lane * 4; i < d; i += 256) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int r = lo;
            for (; r + 4 <= hi; r += 4) {      // 4 loads in flight, adds stay in row order
                const float4 v0 = ld4(zhat + (long)sorted_rows[r] * d + i);
                const float4 v1 = ld4(zhat + (long)sorted_rows[r + 1] * d + i);
                const float4 v2 = ld4(zhat + (long)sorted_rows[r + 2] * d + i);
                const float4 v3 = ld4(zhat + (long)sorted_rows[r + 3] * d + i);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
            }
            for (; r < hi; ++r) {
                const float4 v = ld4(zhat + (long)sorted_rows[r] * d + i);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            st4(embed_sum + code * d + i, a);
        }
    }
}

struct EmaWs {
    int *counts, *offsets, *wave_counts, *wave_offsets;
    uint32_t *keys[2], *vals[2];
    size_t total;
};

static EmaWs ema_ws_layout(void *ws, int64_t n, int64_t k_codes)
{
    EmaWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = ws ? (char *)ws + off : nullptr; off += align_up(bytes, 256); return (void *)p; };
    w.counts = (int *)take((size_t)k_codes * 4);
    w.offsets = (int *)take((size_t)(k_codes + 1) * 4);
    w.wave_counts = (int *)take((size_t)256 * SORT_WAVES * 4);
    w.wave_offsets = (int *)take(((size_t)256 * SORT_WAVES + 1) * 4);
    for (int i = 0; i < 2; ++i) { w.keys[i] = (uint32_t *)take((size_t)n * 4); w.vals[i] = (uint32_t *)take((size_t)n * 4); }
    w.total = off;
    return w;
}

extern "C" size_t medtok_ema_stats_workspace_bytes(int64_t n, int64_t k_codes)
{
    if (n < 0 || k_codes <= 0) return 0;
    return ema_ws_layout(nullptr, n, k_codes).total;
}

extern "C" int medtok_ema_stats_f32(const float *zhat, const int64_t *idx, int64_t n, int d, int64_t k_codes, float *bins,
                                    float *embed_sum, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || d <= 0 || (d & 3) || k_codes <= 0 || k_codes >= (1ll << 31) || n >= (1ll << 31))
        return fail("ema_stats: bad shape n=%ld d=%d K=%ld", (long)n, d, (long)k_codes);
    EmaWs w = ema_ws_layout(ws, n, k_codes);
    if (!ws || ws_bytes < w.total) return fail("ema_stats: workspace too small (%zu < %zu)", ws_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(w.counts, 0, (size_t)k_codes * 4, s) != hipSuccess) return fail("ema_stats: memset failed");
    if (n > 0) {
        hipLaunchKernelGGL(hist_kernel, dim3((unsigned)lmin(2048, (n + 255) / 256)), dim3(256), 0, s, idx, (long)n, (int)k_codes, w.counts);
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, w.counts, (int)k_codes, w.offsets);
    int cur = 0;
    bool first = true;
    for (int shift = 0; n > 0 && (first || (k_codes - 1) >> shift); shift += 8) {
        if (first) hipLaunchKernelGGL((radix_count_kernel<true>), dim3(SORT_BLOCKS), dim3(256), 0, s, idx, (const uint32_t *)nullptr, (long)n, (int)k_codes, shift, w.wave_counts);
        else hipLaunchKernelGGL((radix_count_kernel<false>), dim3(SORT_BLOCKS), dim3(256), 0, s, idx, w.keys[cur], (long)n, (int)k_codes, shift, w.wave_counts);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, w.wave_counts, 256 * SORT_WAVES, w.wave_offsets);
        if (first) hipLaunchKernelGGL((radix_scatter_kernel<true>), dim3(SORT_BLOCKS), dim3(256), 0, s, idx, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (long)n, (int)k_codes, shift, w.wave_offsets, w.keys[cur ^ 1], w.vals[cur ^ 1]);
        else hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(SORT_BLOCKS), dim3(256), 0, s, idx, w.keys[cur], w.vals[cur], (long)n, (int)k_codes, shift, w.wave_offsets, w.keys[cur ^ 1], w.vals[cur ^ 1]);
        cur ^= 1;
        first = false;
    }
    hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)lmin(4096, (k_codes + 3) / 4)), dim3(256), 0, s, zhat, w.vals[cur], w.offsets, (int)k_codes, d, bins, embed_sum);
    return check_launch("ema_stats");
}

// bins only (eval branch of the reference, norm_ema_quantizer.py:185-188)
__global__ __launch_bounds__(256) void counts_to_float_kernel(const int *__restrict__ counts, long k, float *__restrict__ bins)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < k) bins[i] = (float)counts[i];
}

extern "C" size_t medtok_code_histogram_workspace_bytes(int64_t k_codes)
{
    return k_codes > 0 ? align_up((size_t)k_codes * 4, 256) : 0;
}

extern "C" int medtok_code_histogram_f32(const int64_t *idx, int64_t n, int64_t k_codes, float *bins, void *ws, size_t ws_bytes,
                                         void *stream)
{
    if (n < 0 || k_codes <= 0 || k_codes >= (1ll << 31)) return fail("code_histogram: bad shape n=%ld K=%ld", (long)n, (long)k_codes);
    if (!ws || ws_bytes < (size_t)k_codes * 4) return fail("code_histogram: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int *counts = (int *)ws;
    if (hipMemsetAsync(counts, 0, (size_t)k_codes * 4, s) != hipSuccess) return fail("code_histogram: memset failed");
    if (n > 0) hipLaunchKernelGGL(hist_kernel, dim3((unsigned)lmin(2048, (n + 255) / 256)), dim3(256), 0, s, idx, (long)n, (int)k_codes, counts);
    hipLaunchKernelGGL(counts_to_float_kernel, dim3((unsigned)((k_codes + 255) / 256)), dim3(256), 0, s, counts, (long)k_codes, bins);
    return check_launch("code_histogram");
}

// ================================================================= EMA apply
__global__ __launch_bounds__(256) void ema_apply_kernel(float *E, float *cluster_size, const float *__restrict__ bins,
                                                        const float *__restrict__ embed_sum, int k_codes, int d, float decay, float omd)
{
    const int lane = threadIdx.x & 63;
    const long code = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (code >= k_codes) return;
    const float b = bins[code];
    if (lane == 0) {
        const float a0 = cluster_size[code] * decay;
        const float a1 = b * omd;
        cluster_size[code] = a0 + a1;
    }
    float *e = E + code * d;
    const float *sm = embed_sum + code * d;
    const bool keep = (b == 0.0f);
    float den1 = 1.f;
    if (!keep) {
        float p = 0.f;
        for (int i = lane * 4; i < d; i += 256) {
            float4 v = ld4(sm + i);
            v.x = v.x / b; v.y = v.y / b; v.z = v.z / b; v.w = v.w / b;
            p = fmaf(v.x, v.x, p); p = fmaf(v.y, v.y, p); p = fmaf(v.z, v.z, p); p = fmaf(v.w, v.w, p);
        }
        den1 = fmaxf(sqrtf(wave_butterfly_sum(p)), 1e-12f);
    }
    auto mixed = [&](int i) {
        float4 nw;
        const float4 ev = ld4(e + i);
        if (keep) nw = ev;
        else {
            nw = ld4(sm + i);
            nw.x = (nw.x / b) / den1; nw.y = (nw.y / b) / den1; nw.z = (nw.z / b) / den1; nw.w = (nw.w / b) / den1;
        }
        float4 m;
        float t0, t1;
        t0 = ev.x * decay; t1 = nw.x * omd; m.x = t0 + t1;
        t0 = ev.y * decay; t1 = nw.y * omd; m.y = t0 + t1;
        t0 = ev.z * decay; t1 = nw.z * omd; m.z = t0 + t1;
        t0 = ev.w * decay; t1 = nw.w * omd; m.w = t0 + t1;
        return m;
    };
    float p = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 m = mixed(i);
        p = fmaf(m.x, m.x, p); p = fmaf(m.y, m.y, p); p = fmaf(m.z, m.z, p); p = fmaf(m.w, m.w, p);
    }
    const float den2 = fmaxf(sqrtf(wave_butterfly_sum(p)), 1e-12f);
    for (int i = lane * 4; i < d; i += 256) {
        float4 m = mixed(i);
        m.x = m.x / den2; m.y = m.y / den2; m.z = m.z / den2; m.w = m.w / den2;
        st4(e + i, m);
    }
}

extern "C" int medtok_ema_apply_f32(float *E, float *cluster_size, const float *bins, const float *embed_sum, int64_t k_codes,
                                    int d, float decay, float one_minus_decay, void *stream)
{
    if (k_codes <= 0 || d <= 0 || (d & 3)) return fail("ema_apply: bad shape K=%ld d=%d", (long)k_codes, d);
    hipLaunchKernelGGL(ema_apply_kernel, dim3((unsigned)((k_codes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, E, cluster_size, bins,
                       embed_sum, (int)k_codes, d, decay, one_minus_decay);
    return check_launch("ema_apply");
}

__global__ __launch_bounds__(256) void ema_cluster_size_kernel(float *cs, const float *__restrict__ bins, long k, float decay, float omd)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    const float a0 = cs[i] * decay;
    const float a1 = bins[i] * omd;
    cs[i] = a0 + a1;
}

extern "C" int medtok_ema_cluster_size_f32(float *cluster_size, const float *bins, int64_t k_codes, float decay,
                                           float one_minus_decay, void *stream)
{
    if (k_codes <= 0) return fail("ema_cluster_size: bad K");
    hipLaunchKernelGGL(ema_cluster_size_kernel, dim3((unsigned)((k_codes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cluster_size, bins, (long)k_codes, decay, one_minus_decay);
    return check_launch("ema_cluster_size");
}

// ================================================================= codebook usage window
__global__ __launch_bounds__(256) void usage_shift_kernel(const float *__restrict__ win, long wlen, const int64_t *__restrict__ ids,
                                                          long m, float *__restrict__ tmp)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < wlen; i += (long)gridDim.x * 256) {
        float v;
        if (m >= wlen) v = (float)ids[m - wlen + i];
        else v = (i < wlen - m) ? win[i + m] : (float)ids[i - (wlen - m)];
        tmp[i] = v;
    }
}

// Distinct count without atomics on the hot words: every window entry stores 1 into its code's flag byte (all
// writers write the same value, so the race is benign), then one block sums the n_codes + 1 flags.  (300 000 atomicOr
// operations on a 21 000-bit map serialise on ~650 words: 1.6 ms; this form takes microseconds.)
__global__ __launch_bounds__(256) void usage_mark_kernel(const float *__restrict__ tmp, long wlen, long n_codes, float *__restrict__ win,
                                                         unsigned char *__restrict__ flags)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < wlen; i += (long)gridDim.x * 256) {
        const float v = tmp[i];
        win[i] = v;
        long c = (long)v;
        if (c < 0 || c >= n_codes) c = n_codes;
        flags[c] = 1;
    }
}

__global__ __launch_bounds__(1024) void usage_sum_kernel(const unsigned char *__restrict__ flags, long n, int *__restrict__ count)
{
    __shared__ int sh[1024];
    int a = 0;
    for (long i = threadIdx.x; i < n; i += 1024) a += flags[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = sh[0];
}

extern "C" size_t medtok_usage_workspace_bytes(int64_t window_len, int64_t n_codes)
{
    if (window_len <= 0 || n_codes <= 0) return 0;
    return align_up((size_t)window_len * 4, 256) + align_up((size_t)n_codes + 1, 256);
}

extern "C" int medtok_usage_update(float *window, int64_t window_len, const int64_t *ids, int64_t m, int64_t n_codes,
                                   int32_t *count_out, void *ws, size_t ws_bytes, void *stream)
{
    if (window_len <= 0 || n_codes <= 0 || m < 0 || !count_out) return fail("usage_update: bad args");
    const size_t need = medtok_usage_workspace_bytes(window_len, n_codes);
    if (!ws || ws_bytes < need) return fail("usage_update: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    float *tmp = (float *)ws;
    unsigned char *flags = (unsigned char *)ws + align_up((size_t)window_len * 4, 256);
    if (hipMemsetAsync(flags, 0, (size_t)n_codes + 1, s) != hipSuccess) return fail("usage_update: memset failed");
    const unsigned blocks = (unsigned)lmin(1024, (window_len + 255) / 256);
    hipLaunchKernelGGL(usage_shift_kernel, dim3(blocks), dim3(256), 0, s, window, (long)window_len, ids, (long)m, tmp);
    hipLaunchKernelGGL(usage_mark_kernel, dim3(blocks), dim3(256), 0, s, tmp, (long)window_len, (long)n_codes, window, flags);
    hipLaunchKernelGGL(usage_sum_kernel, dim3(1), dim3(1024), 0, s, flags, (long)n_codes + 1, count_out);
    return check_launch("usage_update");
}

// ---- the forward's three to five window updates (:241-250: shared, text, graph, and the aug views') in ONE call of two launches
// (twelve to twenty before: each update a memset and three launches over the 300 000-entry window).  Conceptually the updates slide
// one window over U = [old window | ids_1 | ids_2 | ...]: after update u the window is U[off_u, off_u + W), off_u = m_1 + ... + m_u.
// One pass over U marks every entry in the flag map of each update whose window holds it and writes the final window
// U[M, M + W) to scratch; a second pass copies it back and sums the maps.  Same values as the updates one by one.
constexpr int USAGE_MULTI_MAX = 6;
struct UsageMultiArgs { const int64_t *ids[USAGE_MULTI_MAX]; long m[USAGE_MULTI_MAX]; int count; };

__global__ __launch_bounds__(256) void usage_multi_mark_kernel(const float *__restrict__ win, long wlen, UsageMultiArgs a, long n_codes,
                                                               float *__restrict__ tmp, unsigned char *__restrict__ flags)
{
    long total = 0;
    for (int u = 0; u < a.count; ++u) total += a.m[u];
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < wlen + total; p += (long)gridDim.x * 256) {
        float v;
        if (p < wlen) v = win[p];
        else {
            long q = p - wlen;
            int u = 0;
            while (q >= a.m[u]) { q -= a.m[u]; ++u; }
            v = (float)a.ids[u][q];
        }
        long c = (long)v;
        if (c < 0 || c >= n_codes) c = n_codes;
        long off = 0;
        for (int u = 0; u < a.count; ++u) {
            off += a.m[u];
            if (p >= off && p < off + wlen) flags[(long)u * (n_codes + 1) + c] = 1;
        }
        if (p >= total) tmp[p - total] = v;
    }
}

__global__ __launch_bounds__(256) void usage_multi_finish_kernel(const float *__restrict__ tmp, long wlen, float *__restrict__ win,
                                                                 const unsigned char *__restrict__ flags, long n_codes, int count, int *__restrict__ counts,
                                                                 const int *__restrict__ extra_word = nullptr)
{
    // (extra_word: a device word the caller wants behind the counts -- its one host read then fetches both.  A NON-ZERO word vetoes
    // the window write: the caller's device-side input checks failed, the ids are not to be trusted, and the caller -- who sees the
    // word with the counts -- repeats the forward on repaired inputs against the window as it was)
    const bool veto = extra_word && extra_word[0] != 0;
    if (extra_word && blockIdx.x == 0 && threadIdx.x == 0) counts[count] = extra_word[0];
    if (!veto)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < wlen; i += (long)gridDim.x * 256) win[i] = tmp[i];
    const int lane = threadIdx.x & 63;
    for (int u = 0; u < count; ++u) {
        int part = 0;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_codes + 1; i += (long)gridDim.x * 256) part += flags[(long)u * (n_codes + 1) + i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0 && part) atomicAdd(&counts[u], part);                // (integer: exact in any order)
    }
}

extern "C" size_t medtok_usage_multi_workspace_bytes(int64_t window_len, int64_t n_codes, int count)
{
    if (window_len <= 0 || n_codes <= 0 || count < 1 || count > USAGE_MULTI_MAX) return 0;
    return align_up((size_t)window_len * 4, 256) + align_up((size_t)count * ((size_t)n_codes + 1), 256);
}

extern "C" int medtok_usage_update_multi_word(float *window, int64_t window_len, const int64_t *const *ids, const int64_t *m, int count, int64_t n_codes,
                                              int32_t *counts_out, const int32_t *extra_word, void *ws, size_t ws_bytes, void *stream);
extern "C" int medtok_usage_update_multi(float *window, int64_t window_len, const int64_t *const *ids, const int64_t *m, int count, int64_t n_codes,
                                         int32_t *counts_out, void *ws, size_t ws_bytes, void *stream)
{
    return medtok_usage_update_multi_word(window, window_len, ids, m, count, n_codes, counts_out, nullptr, ws, ws_bytes, stream);
}

// ... with a device word copied behind the counts (counts_out [count + 1]): the caller's one host read of the usage counts also brings
// e.g. the cross-attention's status word, without a concatenation launch
extern "C" int medtok_usage_update_multi_word(float *window, int64_t window_len, const int64_t *const *ids, const int64_t *m, int count, int64_t n_codes,
                                              int32_t *counts_out, const int32_t *extra_word, void *ws, size_t ws_bytes, void *stream)
{
    if (window_len <= 0 || n_codes <= 0 || count < 1 || count > USAGE_MULTI_MAX || !ids || !m || !counts_out || !window)
        return fail("usage_update_multi: bad args (1..%d updates)", USAGE_MULTI_MAX);
    const size_t need = medtok_usage_multi_workspace_bytes(window_len, n_codes, count);
    if (!ws || ws_bytes < need) return fail("usage_update_multi: workspace too small (%zu < %zu)", ws_bytes, need);
    UsageMultiArgs a;
    a.count = count;
    long total = 0;
    for (int u = 0; u < USAGE_MULTI_MAX; ++u) {
        a.ids[u] = u < count ? ids[u] : nullptr;
        a.m[u] = u < count ? (long)m[u] : 0;
        if (u < count && (m[u] < 0 || (m[u] > 0 && !ids[u]))) return fail("usage_update_multi: update %d has no ids", u);
        total += a.m[u];
    }
    hipStream_t s = (hipStream_t)stream;
    float *tmp = (float *)ws;
    unsigned char *flags = (unsigned char *)ws + align_up((size_t)window_len * 4, 256);
    if (hipMemsetAsync(flags, 0, (size_t)count * ((size_t)n_codes + 1), s) != hipSuccess) return fail("usage_update_multi: memset failed");
    if (hipMemsetAsync(counts_out, 0, (size_t)count * 4, s) != hipSuccess) return fail("usage_update_multi: memset failed");
    const unsigned blocks = (unsigned)lmin(1024, (window_len + total + 255) / 256);
    hipLaunchKernelGGL(usage_multi_mark_kernel, dim3(blocks), dim3(256), 0, s, window, (long)window_len, a, (long)n_codes, tmp, flags);
    hipLaunchKernelGGL(usage_multi_finish_kernel, dim3((unsigned)lmin(256, (window_len + 255) / 256)), dim3(256), 0, s, tmp, (long)window_len, window, flags,
                       (long)n_codes, count, counts_out, (const int *)extra_word);
    return check_launch("usage_update_multi");
}

// ================================================================= one-call soft VQ forward
// l2norm + nearest codes in one call (the head of NormEMAVectorQuantizer.forward): on the filter path the fp16 image of the
// normalised rows that the shortlist pass streams comes out of the pass that normalises them (one read of z less).
extern "C" size_t medtok_normalized_search_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path)
{
    return medtok_search_workspace_bytes(n, k_codes, d, topk, path);
}

extern "C" int medtok_normalized_search_f32(const float *z, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes, int topk,
                                            int path, float *zhat, float *zsq, int64_t *idx, float *dist, void *ws, size_t ws_bytes,
                                            void *stream)
{
    if (n < 0 || k_codes <= 0 || d <= 0 || (d & 3)) return fail("normalized_search: bad shape n=%ld K=%ld d=%d (d %% 4 == 0)", (long)n, (long)k_codes, d);
    if (n == 0) return 0;
    if (!z || !zhat || !zsq || !what || !wsq || !idx || !dist) return fail("normalized_search: NULL argument");
    FuseAssign fuse = {nullptr, nullptr, nullptr, 0L, false, nullptr, nullptr, false};
    const bool filter_path = topk >= 1 && topk <= MEDTOK_MAX_TOPK && k_codes < (1ll << 31) && n < (1ll << 31) &&
                             resolve_path(path, n, k_codes, d, topk) == MEDTOK_PATH_F16_FILTER;
    if (filter_path) {
        const FilterPlan f = plan_filter(n, k_codes, d, topk, decode_plan(path));
        const FilterWs fw = filter_ws_layout(ws, n, f);
        if (!ws || ws_bytes < fw.total) return fail("normalized_search: workspace too small (%zu < %zu)", ws_bytes, fw.total);
        hipStream_t s = (hipStream_t)stream;
        launch_rownorm<true>(s, z, (long)n, d, zhat, zsq, fw.xh, f.dp);
        if (f.n_pad > n && hipMemsetAsync(fw.xh + (size_t)n * f.dp, 0, (size_t)(f.n_pad - n) * f.dp * 2, s) != hipSuccess)
            return fail("normalized_search: memset failed");
        if (check_launch("rownorm(+fp16)")) return 1;
        fuse.xh_done = true;
    } else if (medtok_rownorm_f32(z, n, d, 1, zhat, zsq, stream)) {
        return 1;
    }
    return search_impl(zhat, zsq, n, what, wsq, k_codes, d, topk, idx, dist, ws, ws_bytes, path, stream, filter_path ? &fuse : nullptr);
}

extern "C" size_t medtok_soft_vq_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path)
{
    if (n <= 0) return 256;
    return align_up((size_t)n * 4, 256) + medtok_search_workspace_bytes(n, k_codes, d, topk, path);
}

static int soft_vq_forward_impl(const float *x, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes,
                                int topk, int path, float *xhat, int64_t *idx, float *dist, float *w, float *zq_ste,
                                int64_t zq_stride, float *row_sqerr, void *ws, size_t ws_bytes, void *stream,
                                const _Float16 *p_wh, const float *p_wsqp, const float *p_en_max);

extern "C" int medtok_soft_vq_forward_f32(const float *x, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes,
                                          int topk, int path, float *xhat, int64_t *idx, float *dist, float *w, float *zq_ste,
                                          int64_t zq_stride, float *row_sqerr, void *ws, size_t ws_bytes, void *stream)
{
    return soft_vq_forward_impl(x, n, d, what, wsq, k_codes, topk, path, xhat, idx, dist, w, zq_ste, zq_stride, row_sqerr, ws, ws_bytes, stream,
                                nullptr, nullptr, nullptr);
}

// ---- a codebook prepared once per weight version
extern "C" int medtok_filter_image_width(int d)
{
    return d > 0 ? (int)lmax(2 * F_BK, (d + F_BK - 1) / F_BK * F_BK) : 0;
}

extern "C" int medtok_rownorm_image_f32(const float *x, int64_t n, int d, float *xhat, float *sqn, void *image, int64_t image_rows, int dp, void *stream)
{
    if (n <= 0 || d <= 0 || (d & 3)) return fail("rownorm_image: need n > 0, d > 0, d %% 4 == 0 (n=%ld d=%d)", (long)n, d);
    if (!x || !xhat || !sqn || !image) return fail("rownorm_image: NULL argument");
    if (dp != medtok_filter_image_width(d)) return fail("rownorm_image: dp=%d, the filter reads images of width %d at d=%d", dp, medtok_filter_image_width(d), d);
    if (image_rows < n) return fail("rownorm_image: image_rows=%ld < n=%ld", (long)image_rows, (long)n);
    if (((uintptr_t)x | (uintptr_t)xhat | (uintptr_t)image) & 15) return fail("rownorm_image: pointers must be 16-byte aligned");
    launch_rownorm<true>((hipStream_t)stream, x, (long)n, d, xhat, sqn, (_Float16 *)image, dp, (long)image_rows);
    return check_launch("rownorm_image");
}

extern "C" int medtok_codebook_image_f32(const float *what, int64_t n, int d, void *image, int64_t image_rows, int dp, void *stream)
{
    if (n <= 0 || d <= 0 || (d & 3) || !what || !image) return fail("codebook_image: bad arguments (n=%ld d=%d)", (long)n, d);
    if (dp != medtok_filter_image_width(d) || image_rows < n) return fail("codebook_image: dp=%d (expected %d), image_rows=%ld (>= %ld)", dp, medtok_filter_image_width(d), (long)image_rows, (long)n);
    if (((uintptr_t)what | (uintptr_t)image) & 15) return fail("codebook_image: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)lmin(4096, (image_rows * (dp / 8) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, what, (long)n, d,
                       (long)image_rows, dp, (_Float16 *)image, (int *)nullptr);
    return check_launch("codebook_image");
}

extern "C" int medtok_search_resolved_path(int64_t n, int64_t k_codes, int d, int topk, int path)
{
    if (n <= 0 || k_codes <= 0 || d <= 0 || topk < 1 || topk > MEDTOK_MAX_TOPK) return MEDTOK_PATH_F32_MFMA;
    return resolve_path(path, n, k_codes, d, topk);
}

extern "C" int medtok_codebook_prepare_f32(const float *wsq, const medtok_region_desc *regions, int count, void *stream)
{
    if (!wsq || !regions || count < 1 || count > PREP_MAX_REGIONS) return fail("codebook_prepare: 1..%d regions", PREP_MAX_REGIONS);
    RegionPrepArgs a;
    for (int i = 0; i < count; ++i) {
        const medtok_region_desc &r = regions[i];
        if (r.lo < 0 || r.k <= 0 || r.lo + r.k >= (1ll << 31) || !r.wsqp || !r.en_max) return fail("codebook_prepare: bad region %d", i);
        a.r[i] = RegionPrep{wsq + r.lo, (int)r.k, r.en_max, r.wsqp, (int)((r.k + F_BM - 1) / F_BM * F_BM)};
    }
    hipLaunchKernelGGL(wsq_max_regions_kernel, dim3((unsigned)count), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("codebook_prepare");
}

extern "C" int medtok_soft_vq_forward_prepared_f32(const float *x, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes,
                                                   int topk, int path, const void *image, const float *wsqp, const float *en_max,
                                                   float *xhat, int64_t *idx, float *dist, float *w, float *zq_ste,
                                                   int64_t zq_stride, void *ws, size_t ws_bytes, void *stream)
{
    if (!image || !wsqp || !en_max) return fail("soft_vq_forward_prepared: NULL prepared argument");
    if ((uintptr_t)image & 15) return fail("soft_vq_forward_prepared: the image must be 16-byte aligned");
    return soft_vq_forward_impl(x, n, d, what, wsq, k_codes, topk, path, xhat, idx, dist, w, zq_ste, zq_stride, nullptr, ws, ws_bytes, stream,
                                (const _Float16 *)image, wsqp, en_max);
}

static int soft_vq_forward_impl(const float *x, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes,
                                int topk, int path, float *xhat, int64_t *idx, float *dist, float *w, float *zq_ste,
                                int64_t zq_stride, float *row_sqerr, void *ws, size_t ws_bytes, void *stream,
                                const _Float16 *p_wh, const float *p_wsqp, const float *p_en_max)
{
    if (n == 0) return 0;
    const size_t need = medtok_soft_vq_workspace_bytes(n, k_codes, d, topk, path);
    if (!ws || ws_bytes < need) return fail("soft_vq_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    float *xsq = (float *)ws;
    void *sws = (char *)ws + align_up((size_t)n * 4, 256);
    // Without the squared-error output (its summation order is the stand-alone kernel's) the filter path's re-score kernel
    // does the soft assignment itself, bit for bit the same, while the top-k code rows are hot in the L2.
    if (zq_stride == 0) zq_stride = d;
    FuseAssign fuse = {x, w, zq_ste, (long)zq_stride, false, nullptr, nullptr, false};
    fuse.p_wh = p_wh; fuse.p_wsqp = p_wsqp; fuse.p_en_max = p_en_max;
    const bool try_fuse = !row_sqerr && zq_ste && zq_stride >= d && !(zq_stride & 3) && topk <= 8;
    const bool filter_path = topk >= 1 && topk <= MEDTOK_MAX_TOPK && resolve_path(path, n, k_codes, d, topk) == MEDTOK_PATH_F16_FILTER;
    if (try_fuse && filter_path && xhat && !(d & 3)) {
        // the filter's fp16 image of the normalised rows comes out of the same pass that normalises them
        const FilterPlan f = plan_filter(n, k_codes, d, topk, decode_plan(path));
        const FilterWs fw = filter_ws_layout(sws, n, f);
        hipStream_t s = (hipStream_t)stream;
        // (the image's padding rows are written by the same launch; with a prepared codebook it also clears the count of rows the
        // filter hands to the exact kernel -- nothing else of the search's preparation runs then)
        launch_rownorm<true>(s, x, (long)n, d, xhat, xsq, fw.xh, f.dp, (long)f.n_pad, p_wh ? fw.fb_count : (int *)nullptr);
        if (check_launch("rownorm(+fp16)")) return 1;
        fuse.xh_done = true;
        fuse.fb_zeroed = p_wh != nullptr;
    } else if (medtok_rownorm_f32(x, n, d, 1, xhat, xsq, stream)) {
        return 1;
    }
    const int rc = search_impl(xhat, xsq, n, what, wsq, k_codes, d, topk, idx, dist, sws, ws_bytes - align_up((size_t)n * 4, 256), path, stream,
                               try_fuse ? &fuse : nullptr);
    if (rc) return 1;
    if (!fuse.done) return medtok_soft_assign_f32(x, what, idx, dist, n, d, topk, 0, w, zq_ste, zq_stride, row_sqerr, stream);
    // rows the filter handed to the exact kernel: assignment through the device-side row list (normally empty)
    hipLaunchKernelGGL((soft_assign_kernel<8>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,          // (the filter path: topk <= 8)
                       x, what, idx, dist, (long)n, d, topk, 0, w, zq_ste, (long)zq_stride, (float *)nullptr, fuse.fb_rows, fuse.fb_count);
    return check_launch("soft_assign(list)");
}
