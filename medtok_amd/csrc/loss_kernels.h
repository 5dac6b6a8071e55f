// loss_kernels.h -- the two regularisers of MedTok/loss.py next to InfoNCE (train_kernels.h): alignment_loss (:59-64) and
// orthogonal_loss (:66-83), forward and backward.  Included by medtok_vq.hip; gfx950 only.
//
//   alignment_loss(mu1, mu2) = mean_b <mu1[b], mu2[b]>             row_dot_kernel + the fixed-order fp64 sum
//   orthogonal_loss(z, z*)   = || z^T z* ||_F                      small_gemm_f32_kernel (exact fp32 MFMA) + frobenius_kernel
//   their gradients          = scaled copies / two more small GEMMs with G = g M / ||M||
//
// Arithmetic (restated by oracle/medtok_oracle.c): a row dot product is 64 strided fmaf chains joined by the xor butterfly
// (as |v|^2 in rownorm); a GEMM entry is ONE fmaf chain over k in increasing order from +0 -- what v_mfma_f32_32x32x2_f32
// computes when the two half-waves carry k and k + 1; the Frobenius sum is rownorm's row sums added in fp64 in row order.
#pragma once

// out[r] = <a[r], b[r]>; one wavefront per row
__global__ __launch_bounds__(256) void row_dot_kernel(const float *__restrict__ a, const float *__restrict__ b, long n, int d,
                                                      float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float p = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 x = ld4(a + row * d + i), y = ld4(b + row * d + i);
        p = fmaf(x.x, y.x, p); p = fmaf(x.y, y.y, p); p = fmaf(x.z, y.z, p); p = fmaf(x.w, y.w, p);
    }
    p = wave_butterfly_sum(p);
    if (lane == 0) out[row] = p;
}

// C[m, n] = sum_k A[m * sam + k * sak] * B[k * sbk + n * sbn], row-major C [M, N]; any transposition is a choice of strides.
// One wavefront (= one block) per 32 x 32 tile of C; lane (i = lane & 31, h = lane >> 5) feeds A[m0 + i][k + h] and B[k + h][n0 + i] to
// v_mfma_f32_32x32x2_f32, so every entry is the fmaf chain over k = 0, 1, 2, ... .  The operands of these losses are a few
// hundred KB (B = 256 rows of D <= 1536) and live in the L2; a tile's chain is K / 2 dependent MFMAs (768: ~10 us of matrix pipe)
// whatever the launch looks like, so the kernel's job is to keep the loads out of that chain: trips of 32 k, the NEXT trip's
// operands requested before this trip's 16 MFMAs (two register sets), 16-byte loads of four consecutive k where an operand is
// contiguous along k (AV / BV: a lane then reads the 32 k of its row and picks its half-wave's parity).  Round 5's form loaded
// 8 k, waited, multiplied: one L2 round trip per four MFMAs -- 38 us for K = 256, 90 us for K = 768 (six launches and 0.44 ms of
// a cfg 4 training step).  Same MFMA sequence per entry, same bits.
template <bool AV, bool BV>
__global__ __launch_bounds__(64) void small_gemm_f32_kernel(const float *A, long sam, long sak, const float *B,
                                                             long sbk, long sbn, int M, int N, int K, float *C)
{
    constexpr int T = 32;                                              // k per trip
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int tiles_n = (N + 31) / 32;
    // ONE wavefront per block: a tile's loads are 32 rows x 128 bytes per instruction (strided rows: 32 cache lines each), and four
    // waves on one CU queued four tiles' worth of them on its single L1 (1.9 us per 32-k trip against 0.45 us of MFMAs); the few
    // hundred tiles of these products spread over as many CUs instead
    const long tile = blockIdx.x;
    const int m0 = (int)(tile / tiles_n) * 32, n0 = (int)(tile % tiles_n) * 32;
    const int am = min(m0 + li, M - 1), bn = min(n0 + li, N - 1);      // clamped lanes compute entries that are never stored
    const float *pa = A + am * sam, *pb = B + bn * sbn;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // one trip's operands: scalar form = element j is k + 2 j + lh; vector form = float4 #j holds k + 4 j .. + 3 (both half-waves
    // load the same 32 values and pick theirs below)
    struct Trip { float a[AV ? 1 : 16]; float4 a4[AV ? 8 : 1]; float b[BV ? 1 : 16]; float4 b4[BV ? 8 : 1]; };
    auto load = [&](Trip &t, int k) __attribute__((always_inline)) {
        if (AV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) t.a4[j] = ld4(pa + k + 4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) t.a[j] = pa[(k + 2 * j + lh) * sak];
        }
        if (BV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) t.b4[j] = ld4(pb + k + 4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) t.b[j] = pb[(k + 2 * j + lh) * sbk];
        }
    };
    auto compute = [&](const Trip &t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float av, bv;
            if (AV) { const float4 v = t.a4[j / 2]; av = (j & 1) ? (lh ? v.w : v.z) : (lh ? v.y : v.x); } else av = t.a[j];
            if (BV) { const float4 v = t.b4[j / 2]; bv = (j & 1) ? (lh ? v.w : v.z) : (lh ? v.y : v.x); } else bv = t.b[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    };
    // (the prefetches are unconditional -- past the last trip they re-read it -- so that no branch stands between a load and the
    // wait for it: hipcc's wait-count pass merges the paths of a conditional load into "wait for everything")
    Trip t0, t1;
    const int trips = K / T;
    if (trips > 0) {
        load(t0, 0);
        for (int t = 0; t < trips; t += 2) {          // t0 holds trip t
            load(t1, min(t + 1, trips - 1) * T);
            // (the loads may not sink below this statement, the MFMAs -- through acc -- may not rise above it: left alone hipcc
            // puts every load just in front of its MFMA, or this trip's MFMAs in front of the next trip's loads)
            asm volatile("" : "+v"(acc) : : "memory");
            compute(t0);
            if (t + 1 >= trips) break;
            load(t0, min(t + 2, trips - 1) * T);
            asm volatile("" : "+v"(acc) : : "memory");
            compute(t1);
        }
    }
    int k = trips * T;
    for (; k < K; k += 2) {
        const bool live = k + lh < K;         // odd K: the upper half-wave adds +0 * b = nothing
        const float av = live ? pa[(k + lh) * sak] : 0.f, bv = live ? pb[(k + lh) * sbk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + li;
        if (m < M && n < N) C[(long)m * N + n] = acc[r];
    }
}

// out[0] = sqrt(sum_r rows[r]) in fp64, fixed order (rows = rownorm's per-row sums of squares)
__global__ __launch_bounds__(1024) void frobenius_kernel(const float *__restrict__ rows, long n, float *__restrict__ out)
{
    __shared__ double sh[1024];
    double a = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) a += (double)rows[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(sh[0]);
}

// out[i] = x[i] * (c * num[0] / den[0])   (den may be NULL: 1); num / den are device scalars, so nothing synchronises the host.
// den == 0 gives 0 (the subgradient torch uses for the norm at the origin).
__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(const float *__restrict__ x, long count, const float *__restrict__ num,
                                                                     const float *__restrict__ den, float c, float *__restrict__ out)
{
    float f = c * num[0];
    if (den) f = den[0] != 0.f ? f / den[0] : 0.f;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += (long)gridDim.x * 1024) {
        if (i + 4 <= count) {
            const float4 v = ld4(x + i);
            st4(out + i, make_float4(v.x * f, v.y * f, v.z * f, v.w * f));
        } else {
            for (long j = i; j < count; ++j) out[j] = x[j] * f;
        }
    }
}
