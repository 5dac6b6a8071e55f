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
// One wavefront per 32 x 32 tile of C; lane (i = lane & 31, h = lane >> 5) feeds A[m0 + i][k + h] and B[k + h][n0 + i] to
// v_mfma_f32_32x32x2_f32, so every entry is the fmaf chain over k = 0, 1, 2, ... .  The operands of these losses are a few
// hundred KB (B = 256 rows of D <= 1536): they live in the L2, the loads are whatever the strides make them, and the whole
// product is a few microseconds -- the point is one exact, reproducible kernel behind the C ABI, not a tuned GEMM.
__global__ __launch_bounds__(256) void small_gemm_f32_kernel(const float *__restrict__ A, long sam, long sak, const float *__restrict__ B,
                                                             long sbk, long sbn, int M, int N, int K, float *__restrict__ C)
{
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int tiles_n = (N + 31) / 32;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= (long)tiles_n * ((M + 31) / 32)) return;
    const int m0 = (int)(tile / tiles_n) * 32, n0 = (int)(tile % tiles_n) * 32;
    const int am = min(m0 + li, M - 1), bn = min(n0 + li, N - 1);      // clamped lanes compute entries that are never stored
    const float *pa = A + am * sam, *pb = B + bn * sbn;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {              // four MFMAs per trip, loads issued together
        float av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { av[j] = pa[(k + 2 * j + lh) * sak]; bv[j] = pb[(k + 2 * j + lh) * sbk]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    }
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
