// train_kernels.h -- gfx950 kernels of the training half of the VQ path: the sparse backward of the soft
// top-k assignment, the backward of F.normalize, and InfoNCE (forward + backward) of loss.py.
// Included by medtok_vq.hip after its helpers (ld4 / st4 / wave_butterfly_sum / fail / check_launch).
//
// All of these are HBM/L2-bound row kernels: one wavefront per row, lanes stride the D axis in float4,
// wave reductions by xor butterfly.  Nothing here is GEMM-shaped enough to pay for MFMA at training
// batch sizes (B = 256 rows/GPU, vector_quantization_soft_one_new.py callers in train_MedTok.py:196-238).
#pragma once

// ================================================================= soft top-k assignment: backward
// Forward (vector_quantization_soft_one_new.py:157-182,203-214), per row:
//   xhat = x / max(|x|, 1e-12);  d_j = |xhat|^2 + |e_j|^2 - 2 xhat.e_j  (e_j = what[idx_j]);
//   w = softmax(-d);  zq = sum_j w_j e_j;
//   vq = mean((zq - sg(x))^2);  commit = beta * mean((sg(zq) - x)^2);  out = x + sg(zq - x).
// The reference differentiates a dense N x K distance matrix; only the k selected columns are non-zero.
// Given upstream gradients  g_out (w.r.t. out), g_xhat, and the scalars g_vq / g_commit, this kernel forms
//   geff   = g_zq + cv * (zq - x)                       cv = g_vq * 2 / (n d)
//   gw_j   = geff . e_j ;   gd_j = -w_j (gw_j - sum_i w_i gw_i)          (softmax(-d) backward)
//   gxh    = 2 (sum_j gd_j) xhat - 2 sum_j gd_j e_j + g_xhat
//   gx     = (gxh - xhat (xhat . gxh)) / max(|x|, 1e-12) + g_out - cc * (zq - x)     cc = g_commit * 2 beta / (n d)
//   g_code[row, j, :] = w_j geff + 2 gd_j (e_j - xhat)   (gradient w.r.t. the NORMALISED code e_j)
// g_code rows are then segment-summed by code id in row order (medtok_ema_stats_f32) -- deterministic,
// no atomics -- and pushed through normalize_backward_kernel to reach codebook.weight.
template <int MAXK>
__global__ __launch_bounds__(256) void soft_vq_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ xhat, const float *__restrict__ what,
    const int64_t *__restrict__ idx, const float *__restrict__ w, long n, int d, int topk,
    const float *__restrict__ g_zq, const float *__restrict__ g_xhat, const float *__restrict__ g_out,
    const float *__restrict__ g_vq, const float *__restrict__ g_commit, float vq_scale, float commit_scale,
    float *__restrict__ gx, float *__restrict__ g_code)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float cv = g_vq ? vq_scale * g_vq[0] : 0.f;
    const float cc = g_commit ? commit_scale * g_commit[0] : 0.f;
    float wj[MAXK], gw[MAXK], gd[MAXK];
    const float *ej[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        wj[j] = 0.f; gw[j] = 0.f; gd[j] = 0.f; ej[j] = what;
        if (j < topk) { wj[j] = w[row * topk + j]; ej[j] = what + idx[row * topk + j] * (long)d; }
    }
    const float *xr = x + row * d, *xh = xhat + row * d;
    const float *gz = g_zq ? g_zq + row * d : nullptr;
    const float *gh = g_xhat ? g_xhat + row * d : nullptr;
    const float *go = g_out ? g_out + row * d : nullptr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // zq, diff = zq - x and geff for one float4 column group (same fmaf order as soft_assign_kernel)
    auto column = [&](int i, float4 e[MAXK], float4 &diff, float4 &geff) {
        float4 a = zero4;
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j < topk) {
                e[j] = ld4(ej[j] + i);
                a.x = fmaf(wj[j], e[j].x, a.x); a.y = fmaf(wj[j], e[j].y, a.y);
                a.z = fmaf(wj[j], e[j].z, a.z); a.w = fmaf(wj[j], e[j].w, a.w);
            }
        const float4 xv = ld4(xr + i);
        diff = make_float4(a.x - xv.x, a.y - xv.y, a.z - xv.z, a.w - xv.w);
        const float4 g = gz ? ld4(gz + i) : zero4;
        geff = make_float4(fmaf(cv, diff.x, g.x), fmaf(cv, diff.y, g.y), fmaf(cv, diff.z, g.z), fmaf(cv, diff.w, g.w));
    };

    // pass 1: gw_j = geff . e_j and |x|^2
    float xx = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 e[MAXK], diff, geff;
        column(i, e, diff, geff);
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j < topk) {
                gw[j] = fmaf(geff.x, e[j].x, gw[j]); gw[j] = fmaf(geff.y, e[j].y, gw[j]);
                gw[j] = fmaf(geff.z, e[j].z, gw[j]); gw[j] = fmaf(geff.w, e[j].w, gw[j]);
            }
        const float4 xv = ld4(xr + i);
        xx = fmaf(xv.x, xv.x, xx); xx = fmaf(xv.y, xv.y, xx); xx = fmaf(xv.z, xv.z, xx); xx = fmaf(xv.w, xv.w, xx);
    }
    xx = wave_butterfly_sum(xx);
    float sw = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        if (j < topk) { gw[j] = wave_butterfly_sum(gw[j]); sw = fmaf(wj[j], gw[j], sw); }
    float sd = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        if (j < topk) { gd[j] = -wj[j] * (gw[j] - sw); sd += gd[j]; }

    // gxh for one column group
    auto gxh_of = [&](int i, const float4 e[MAXK]) {
        const float4 h = ld4(xh + i);
        float4 acc = zero4;
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j < topk) {
                acc.x = fmaf(gd[j], e[j].x, acc.x); acc.y = fmaf(gd[j], e[j].y, acc.y);
                acc.z = fmaf(gd[j], e[j].z, acc.z); acc.w = fmaf(gd[j], e[j].w, acc.w);
            }
        const float4 g = gh ? ld4(gh + i) : zero4;
        return make_float4(fmaf(2.f, sd * h.x - acc.x, g.x), fmaf(2.f, sd * h.y - acc.y, g.y),
                           fmaf(2.f, sd * h.z - acc.z, g.z), fmaf(2.f, sd * h.w - acc.w, g.w));
    };

    // pass 2: per-slot code gradients, and xhat . gxh
    float dp = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        float4 e[MAXK], diff, geff;
        column(i, e, diff, geff);
        const float4 h = ld4(xh + i);
        if (g_code) {
#pragma unroll
            for (int j = 0; j < MAXK; ++j)
                if (j < topk) {
                    const float t = 2.f * gd[j];
                    st4(g_code + (row * topk + j) * (long)d + i,
                        make_float4(fmaf(wj[j], geff.x, t * (e[j].x - h.x)), fmaf(wj[j], geff.y, t * (e[j].y - h.y)),
                                    fmaf(wj[j], geff.z, t * (e[j].z - h.z)), fmaf(wj[j], geff.w, t * (e[j].w - h.w))));
                }
        }
        const float4 q = gxh_of(i, e);
        dp = fmaf(h.x, q.x, dp); dp = fmaf(h.y, q.y, dp); dp = fmaf(h.z, q.z, dp); dp = fmaf(h.w, q.w, dp);
    }
    if (!gx) return;
    dp = wave_butterfly_sum(dp);
    const float inv = 1.f / fmaxf(sqrtf(xx), 1e-12f);

    // pass 3: through F.normalize, plus the straight-through and commitment terms
    for (int i = lane * 4; i < d; i += 256) {
        float4 e[MAXK], diff, geff;
        column(i, e, diff, geff);
        const float4 h = ld4(xh + i);
        const float4 q = gxh_of(i, e);
        const float4 g = go ? ld4(go + i) : zero4;
        st4(gx + row * d + i, make_float4(fmaf(q.x - h.x * dp, inv, g.x) - cc * diff.x, fmaf(q.y - h.y * dp, inv, g.y) - cc * diff.y,
                                          fmaf(q.z - h.z * dp, inv, g.z) - cc * diff.z, fmaf(q.w - h.w * dp, inv, g.w) - cc * diff.w));
    }
}

// Backward of F.normalize(v, dim=-1, eps=1e-12): out = (g - vhat (vhat . g)) / max(|v|, 1e-12).  One wave per row.
// live (optional, [n]): rows with live[row] == 0 are known to hold an all-zero g -- their output row is +0 whatever v is, so it is
// written without reading the three rows (the codebook's gradient through F.normalize at training: a step touches at most
// B x searches x k of the 49 152 codes; the kernel read 450 MB to write 151 MB of mostly zeros).
__global__ __launch_bounds__(256) void normalize_backward_kernel(const float *__restrict__ g, const float *__restrict__ vhat,
                                                                 const float *__restrict__ v, long n, int d, float *__restrict__ out,
                                                                 const float *__restrict__ live = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (live && live[row] == 0.f) {
        for (int i = lane * 4; i < d; i += 256) st4(out + row * d + i, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const float *gr = g + row * d, *hr = vhat + row * d, *vr = v + row * d;
    float dp = 0.f, vv = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = ld4(gr + i), h = ld4(hr + i), b = ld4(vr + i);
        dp = fmaf(h.x, a.x, dp); dp = fmaf(h.y, a.y, dp); dp = fmaf(h.z, a.z, dp); dp = fmaf(h.w, a.w, dp);
        vv = fmaf(b.x, b.x, vv); vv = fmaf(b.y, b.y, vv); vv = fmaf(b.z, b.z, vv); vv = fmaf(b.w, b.w, vv);
    }
    dp = wave_butterfly_sum(dp);
    vv = wave_butterfly_sum(vv);
    const float inv = 1.f / fmaxf(sqrtf(vv), 1e-12f);
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = ld4(gr + i), h = ld4(hr + i);
        st4(out + row * d + i, make_float4((a.x - h.x * dp) * inv, (a.y - h.y * dp) * inv, (a.z - h.z * dp) * inv, (a.w - h.w * dp) * inv));
    }
}

// ================================================================= InfoNCE (loss.py:40-56)
// loss = CE([pos | off-diagonal negatives] / T, label 0) over normalised q, k  ==  CE(qhat khat^T / T, diagonal).
// Forward: one block per query row i; its 4 waves walk the key rows; logits stay in LDS.
//   row_loss[i] = logsumexp_j(l_ij) - l_ii,  prob[i, j] = softmax_j(l_ij)  (kept for the backward)
// qhat / khat / the inverse norms are written once by info_nce_prepare_kernel.
__global__ __launch_bounds__(256) void info_nce_prepare_kernel(const float *__restrict__ q, const float *__restrict__ k, long b, int d,
                                                               float *__restrict__ qhat, float *__restrict__ khat)
{
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= 2 * b) return;
    const float *src = r < b ? q + r * d : k + (r - b) * d;
    float *dst = r < b ? qhat + r * d : khat + (r - b) * d;
    float p = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = ld4(src + i);
        p = fmaf(a.x, a.x, p); p = fmaf(a.y, a.y, p); p = fmaf(a.z, a.z, p); p = fmaf(a.w, a.w, p);
    }
    const float nrm = fmaxf(sqrtf(wave_butterfly_sum(p)), 1e-12f);
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = ld4(src + i);
        st4(dst + i, make_float4(a.x / nrm, a.y / nrm, a.z / nrm, a.w / nrm));
    }
}

__device__ __forceinline__ float block256_reduce(float v, float *sh, bool is_max)
{
    // xor butterfly inside the wave, then the 4 wave results through LDS; every thread gets the result.  (The InfoNCE kernels run
    // their heavy phases with 16 waves: waves 4 .. 15 pass through here for the barriers and contribute nothing -- the sums keep the
    // 256-thread partition the oracle restates.)
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = sh[0], b2 = sh[1], c = sh[2], e = sh[3];
    return is_max ? fmaxf(fmaxf(a, b2), fmaxf(c, e)) : (a + b2) + (c + e);
}

// RT query rows per block: every key row fetched from L2 serves RT dot products (one block per row re-read the whole
// key matrix, B x D x 4 bytes, B times: 400 MB of L2 traffic at B = 256, D = 1536 -- 90 us for 0.2 GFLOP).
// 16 waves per block for the dot products (round 6): with 4 a wave walked 64 key groups one L2 round trip each with nothing else
// resident on its SIMD (46 - 78 us per call at B = 256); the softmax phase keeps its 256-thread sums.
constexpr int NCE_THREADS = 1024;
template <int RT>
__global__ __launch_bounds__(NCE_THREADS) void info_nce_forward_kernel(const float *__restrict__ qhat, const float *__restrict__ khat, int b, int d,
                                                               float inv_temp, float *__restrict__ prob, float *__restrict__ row_loss)
{
    extern __shared__ float sm[];          // [RT][d] query rows, [RT][b] logits, [4] reduction scratch
    float *qs = sm, *lg = sm + RT * d, *red = lg + RT * b;
    const int i0 = blockIdx.x * RT, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const bool low = threadIdx.x < 256;           // the threads of the fixed-order sums below
    for (int t = threadIdx.x * 4; t < RT * d; t += blockDim.x * 4) {
        const int r = t / d, c = t - r * d;
        st4(qs + t, ld4(qhat + (long)min(i0 + r, b - 1) * d + c));
    }
    __syncthreads();
    // KU key rows per step: their loads are independent, so one L2 round trip covers KU dot products
    constexpr int KU = 4;
    for (int j0 = wave * KU; j0 < b; j0 += waves * KU) {
        float p[RT][KU];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int u = 0; u < KU; ++u) p[r][u] = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            float4 a[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) a[u] = ld4(khat + (long)min(j0 + u, b - 1) * d + c);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float4 q4 = *reinterpret_cast<const float4 *>(qs + r * d + c);
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    p[r][u] = fmaf(a[u].x, q4.x, p[r][u]); p[r][u] = fmaf(a[u].y, q4.y, p[r][u]);
                    p[r][u] = fmaf(a[u].z, q4.z, p[r][u]); p[r][u] = fmaf(a[u].w, q4.w, p[r][u]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float v = wave_butterfly_sum(p[r][u]);
                if (lane == 0 && j0 + u < b) lg[r * b + j0 + u] = v * inv_temp;
            }
    }
    __syncthreads();
    for (int r = 0; r < RT; ++r) {          // (uniform trip count: block256_reduce holds barriers)
        const int i = i0 + r;
        const float *l = lg + r * b;
        float m = -INFINITY;
        if (low)
            for (int j = threadIdx.x; j < b; j += 256) m = fmaxf(m, l[j]);
        m = block256_reduce(m, red, true);
        float sum = 0.f;
        if (low)
            for (int j = threadIdx.x; j < b; j += 256) sum += expf(l[j] - m);
        sum = block256_reduce(sum, red, false);
        const float lse = m + logf(sum);
        if (i < b) {
            for (int j = threadIdx.x; j < b; j += blockDim.x) prob[(long)i * b + j] = expf(l[j] - lse);
            if (threadIdx.x == 0) row_loss[i] = lse - l[i];
        }
    }
}

// Backward: dl_ij = g (prob_ij - [i == j]) / b.  Blocks [0, nb) produce g_q rows, blocks [nb, 2 nb) g_k rows (RT rows each):
//   g_qhat_i = (1/T) sum_j dl_ij khat_j,   g_khat_j = (1/T) sum_i dl_ij qhat_i,   then through F.normalize.
template <int RT>
__global__ __launch_bounds__(NCE_THREADS) void info_nce_backward_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                const float *__restrict__ qhat, const float *__restrict__ khat,
                                                                const float *__restrict__ prob, const float *__restrict__ g_loss,
                                                                int b, int d, float inv_temp, float *__restrict__ gq, float *__restrict__ gk)
{
    extern __shared__ float sm[];          // [RT][b] coefficients, [RT][d] gradients w.r.t. the normalised rows, [4] scratch
    float *cf = sm, *gh = sm + RT * b, *red = gh + RT * d;
    const int nb = (b + RT - 1) / RT;
    const bool qside = (int)blockIdx.x < nb;
    const int r0 = (qside ? blockIdx.x : blockIdx.x - nb) * RT;
    const float scale = g_loss[0] * inv_temp / (float)b;
    const bool low = threadIdx.x < 256;           // the threads of the fixed-order sums of the normalisation's backward
    for (int t = threadIdx.x; t < RT * b; t += blockDim.x) {
        const int rr = t / b, j = t - rr * b, r = min(r0 + rr, b - 1);
        const float p = qside ? prob[(long)r * b + j] : prob[(long)j * b + r];
        cf[t] = scale * (p - (j == r ? 1.f : 0.f));
    }
    __syncthreads();
    const float *other = qside ? khat : qhat;
    // a column's chain over j is sequential (the oracle's order); the columns are independent: one per thread of 16 waves (with 4
    // waves a thread walked three columns one after the other, 96 L2 round trips of eight loads: 38 - 74 us per call)
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float a[RT];
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) a[rr] = 0.f;
#pragma unroll 16
        for (int j = 0; j < b; ++j) {            // (unrolled: sixteen independent loads per L2 round trip)
            const float o = other[(long)j * d + c];
#pragma unroll
            for (int rr = 0; rr < RT; ++rr) a[rr] = fmaf(cf[rr * b + j], o, a[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) gh[rr * d + c] = a[rr];
    }
    __syncthreads();
    for (int rr = 0; rr < RT; ++rr) {
        const int r = min(r0 + rr, b - 1);
        const float *mine_hat = (qside ? qhat : khat) + (long)r * d, *mine = (qside ? q : k) + (long)r * d;
        float dp = 0.f, vv = 0.f;
        if (low)
            for (int c = threadIdx.x; c < d; c += 256) {
                dp = fmaf(mine_hat[c], gh[rr * d + c], dp);
                vv = fmaf(mine[c], mine[c], vv);
            }
        dp = block256_reduce(dp, red, false);
        vv = block256_reduce(vv, red, false);
        const float inv = 1.f / fmaxf(sqrtf(vv), 1e-12f);
        if (r0 + rr < b) {
            float *out = (qside ? gq : gk) + (long)r * d;
            for (int c = threadIdx.x; c < d; c += blockDim.x) out[c] = (gh[rr * d + c] - mine_hat[c] * dp) * inv;
        }
    }
}
