/*
 * medtok_oracle.c -- CPU restatement of MedTok's vector-quantisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under medtok_amd/ may include, link or
 * dlopen this file; it is the checker for tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against tests/golden/ (.npz), which oracle/gen_golden.py produced in the
 * dev container by importing the reference's unmodified Python files on CPU.
 *
 * Reference algorithm (file:line under /root/reference):
 *   l2 normalise            MedTok/norm_ema_quantizer.py:8-9, F.normalize eps=1e-12
 *   distance                MedTok/vector_quantization_soft_one_new.py:120-125
 *                           MedTok/norm_ema_quantizer.py:175-177
 *   top-k / softmax / mix   MedTok/vector_quantization_soft_one_new.py:157-165,203-205
 *   vq / commit loss, STE   MedTok/vector_quantization_soft_one_new.py:168-182,207-214
 *   argmin + EMA update     MedTok/norm_ema_quantizer.py:179-214
 *   codebook usage window   MedTok/vector_quantization_soft_one_new.py:219-236
 *
 * The reference evaluates  d = (|x|^2 + |e|^2) - 2 * (x . e)  with whatever
 * summation order its BLAS picks.  This restatement fixes ONE order, chosen so
 * the MI355X kernels can reproduce it bit for bit:
 *   - x.e is a single fp32 fmaf chain starting from +0 that visits every group of 8
 *     consecutive elements in the order 0,4,1,5,2,6,3,7 (indices >= D skipped).  That is
 *     exactly what v_mfma_f32_32x32x2_f32 computes when each half-wave reads one float4 of
 *     the group (lanes 0-31: elements 0-3, lanes 32-63: elements 4-7) and the four MFMAs
 *     consume register 0,1,2,3 in turn: no data permutation anywhere on the GPU side;
 *   - |v|^2 is 64 strided fmaf chains (element i goes to chain (i/4)%64)
 *     combined by an xor butterfly (offsets 32,16,8,4,2,1), i.e. one wavefront
 *     reading float4 per lane;
 *   - ties in the search resolve to the lowest code index (torch.argmin's
 *     rule; torch.topk's tie order is implementation-defined).
 * Near-ties (top-k gap below ~1e-6) may therefore order differently from the
 * reference's BLAS; the golden fixtures carry fp64 gaps so tests can tell.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off, FMA clone).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#define ORACLE_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORACLE_CLONES
#endif

#define ORACLE_MAX_TOPK 16

/* position p of the canonical dot-product chain -> element index (see header) */
static inline int chain_index(int p)
{
    static const int perm[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    return (p & ~7) + perm[p & 7];
}
#define ORACLE_CODE_CHUNK 64

/* ---- canonical |v|^2 : 64 strided fmaf chains + xor butterfly ------------ */
static float canon_sumsq(const float *v, int d)
{
    float p[64], q[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0f;
    for (int i = 0; i < d; ++i) {
        int l = (i >> 2) & 63;
        p[l] = fmaf(v[i], v[i], p[l]);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof p);
    }
    return p[0];
}

/* F.normalize(p=2, dim=-1, eps=1e-12): x / max(||x||, eps); also returns the
 * squared norm of the *output* row (the term get_distance re-derives). */
int oracle_rownorm_f32(const float *x, int64_t n, int d, int normalize,
                       float *xhat, float *sqn)
{
    for (int64_t r = 0; r < n; ++r) {
        const float *src = x + r * d;
        float *dst = xhat + r * d;
        if (normalize) {
            float nrm = sqrtf(canon_sumsq(src, d));
            float den = nrm > 1e-12f ? nrm : 1e-12f;
            for (int i = 0; i < d; ++i) dst[i] = src[i] / den;
        } else if (dst != src) {
            memcpy(dst, src, sizeof(float) * (size_t)d);
        }
        if (sqn) sqn[r] = canon_sumsq(dst, d);
    }
    return 0;
}

/* One row against a chunk of codes stored transposed ([d][chunk]) so the
 * compiler can run one fmaf chain per SIMD lane; per-lane arithmetic is the
 * scalar chain, so the result does not depend on the vector width. */
ORACLE_CLONES
static void dot_chunk(const float *xrow, const float *wt, int d, float *acc)
{
    float a[ORACLE_CODE_CHUNK];
    for (int c = 0; c < ORACLE_CODE_CHUNK; ++c) a[c] = 0.0f;
    const int dpad = (d + 7) & ~7;
    for (int p = 0; p < dpad; ++p) {
        const int i = chain_index(p);
        if (i >= d) continue;
        const float xi = xrow[i];
        const float *w = wt + (size_t)i * ORACLE_CODE_CHUNK;
        for (int c = 0; c < ORACLE_CODE_CHUNK; ++c)
            a[c] = fmaf(xi, w[c], a[c]);
    }
    for (int c = 0; c < ORACLE_CODE_CHUNK; ++c) acc[c] = a[c];
}

/* Nearest-code search: for each row the `topk` smallest
 *   d = (xsq + wsq[c]) - 2 * dot(xhat, what[c])
 * ascending, ties to the lowest code index.  idx is int64 like torch. */
int oracle_topk_search_f32(const float *xhat, const float *xsq, int64_t n,
                           const float *what, const float *wsq, int64_t k_codes,
                           int d, int topk, int64_t *idx, float *dist)
{
    if (topk < 1 || topk > ORACLE_MAX_TOPK || topk > k_codes) return -1;
    const int64_t n_chunks = (k_codes + ORACLE_CODE_CHUNK - 1) / ORACLE_CODE_CHUNK;
    float *wt = (float *)calloc((size_t)n_chunks * d * ORACLE_CODE_CHUNK, sizeof(float));
    if (!wt) return -2;
    for (int64_t c = 0; c < k_codes; ++c) {
        int64_t ch = c / ORACLE_CODE_CHUNK, lane = c % ORACLE_CODE_CHUNK;
        for (int i = 0; i < d; ++i)
            wt[((size_t)ch * d + i) * ORACLE_CODE_CHUNK + lane] = what[c * d + i];
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float bv[ORACLE_MAX_TOPK];
        int64_t bi[ORACLE_MAX_TOPK];
        float acc[ORACLE_CODE_CHUNK];
        for (int j = 0; j < topk; ++j) { bv[j] = INFINITY; bi[j] = 0; }
        const float *xr = xhat + r * d;
        const float xn = xsq[r];
        for (int64_t ch = 0; ch < n_chunks; ++ch) {
            int64_t base = ch * ORACLE_CODE_CHUNK;
            int cnt = (int)((k_codes - base) < ORACLE_CODE_CHUNK ? (k_codes - base) : ORACLE_CODE_CHUNK);
            dot_chunk(xr, wt + (size_t)ch * d * ORACLE_CODE_CHUNK, d, acc);
            for (int c = 0; c < cnt; ++c) {
                float s = xn + wsq[base + c];
                float t = 2.0f * acc[c];
                float dv = s - t;
                if (dv < bv[topk - 1]) {
                    int j = topk - 1;
                    while (j > 0 && dv < bv[j - 1]) { bv[j] = bv[j - 1]; bi[j] = bi[j - 1]; --j; }
                    bv[j] = dv; bi[j] = base + c;
                }
            }
        }
        for (int j = 0; j < topk; ++j) { idx[r * topk + j] = bi[j]; dist[r * topk + j] = bv[j]; }
    }
    free(wt);
    return 0;
}

/* Full distance matrix (small cases only): used to derive fp64 gaps and to
 * cross-check the search above. */
int oracle_distance_f32(const float *xhat, const float *xsq, int64_t n,
                        const float *what, const float *wsq, int64_t k_codes,
                        int d, float *out)
{
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < k_codes; ++c) {
            float acc = 0.0f;
            for (int p = 0; p < ((d + 7) & ~7); ++p) {
                const int i = chain_index(p);
                if (i < d) acc = fmaf(xhat[r * d + i], what[c * d + i], acc);
            }
            float s = xsq[r] + wsq[c];
            float t = 2.0f * acc;
            out[r * k_codes + c] = s - t;
        }
    return 0;
}

/* The canonical chain scores s[r][c] themselves (small cases; used to measure the fp16 filter's error). */
int oracle_scores_f32(const float *xhat, int64_t n, const float *what, int64_t k_codes, int d, float *out)
{
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < k_codes; ++c) {
            float acc = 0.0f;
            for (int p = 0; p < ((d + 7) & ~7); ++p) {
                const int i = chain_index(p);
                if (i < d) acc = fmaf(xhat[r * d + i], what[c * d + i], acc);
            }
            out[r * k_codes + c] = acc;
        }
    return 0;
}

/* Soft assignment.  w = softmax(-dist) over the topk entries;
 * zq = sum_j w_j * what[idx_j]; out = xref + (zq - xref) (straight-through
 * forward value, vector_quantization_soft_one_new.py:181-182,214);
 * row_sqerr[r] = sum_i (zq - xref)^2 (the un-normalised numerator of the vq
 * and commitment losses, :169-173,208-209).
 * flags bit0 selects the NormEMA form (topk==1, w=1, zq = what[idx]);
 * bit1 stores zq itself instead of the straight-through value. */
int oracle_soft_assign_f32(const float *xref, const float *what,
                           const int64_t *idx, const float *dist, int64_t n,
                           int d, int topk, int flags, float *w, float *zq_ste,
                           float *row_sqerr)
{
    const int hard = flags & 1, raw = flags & 2;
    if (topk < 1 || topk > ORACLE_MAX_TOPK) return -1;
    for (int64_t r = 0; r < n; ++r) {
        float wj[ORACLE_MAX_TOPK];
        if (hard) {
            wj[0] = 1.0f;
        } else {
            float m = -dist[r * topk];
            float sum = 0.0f;
            for (int j = 0; j < topk; ++j) { wj[j] = expf(-dist[r * topk + j] - m); sum += wj[j]; }
            for (int j = 0; j < topk; ++j) wj[j] = wj[j] / sum;
        }
        if (w) for (int j = 0; j < topk; ++j) w[r * topk + j] = wj[j];
        double se = 0.0;
        for (int i = 0; i < d; ++i) {
            float acc = 0.0f;
            if (hard) acc = what[idx[r] * d + i];
            else for (int j = 0; j < topk; ++j) acc = fmaf(wj[j], what[idx[r * topk + j] * d + i], acc);
            float xr = xref[r * d + i];
            float diff = acc - xr;
            if (zq_ste) zq_ste[r * d + i] = raw ? acc : xr + diff;
            se += (double)diff * (double)diff;
        }
        if (row_sqerr) row_sqerr[r] = (float)se;
    }
    return 0;
}

/* Backward of the soft assignment (what autograd does to the graph the reference builds at
 * vector_quantization_soft_one_new.py:157-182,203-214, restricted to the k selected codes -- every
 * other column of the N x K distance matrix receives an exactly-zero gradient).  Per row, with
 * e_j = what[idx_j], zq = sum_j w_j e_j, cv = g_vq * vq_scale, cc = g_commit * commit_scale:
 *   geff = g_zq + cv (zq - x);  gw_j = geff . e_j;  gd_j = -w_j (gw_j - sum_i w_i gw_i)
 *   gxh  = 2 (sum_j gd_j) xhat - 2 sum_j gd_j e_j + g_xhat
 *   gx   = (gxh - xhat (xhat . gxh)) / max(|x|, 1e-12) + g_out - cc (zq - x)
 *   g_code[r*topk + j] = w_j geff + 2 gd_j (e_j - xhat)
 * Reductions are accumulated in double: this is a tolerance checker (1e-5), not a bit oracle. */
int oracle_soft_vq_backward_f32(const float *x, const float *xhat, const float *what, const int64_t *idx,
                                const float *w, int64_t n, int d, int topk, const float *g_zq,
                                const float *g_xhat, const float *g_out, float g_vq, float g_commit,
                                float vq_scale, float commit_scale, float *gx, float *g_code)
{
    if (topk < 1 || topk > ORACLE_MAX_TOPK) return -1;
    const double cv = (double)g_vq * vq_scale, cc = (double)g_commit * commit_scale;
    double *geff = (double *)malloc(sizeof(double) * d), *diff = (double *)malloc(sizeof(double) * d);
    double *gxh = (double *)malloc(sizeof(double) * d);
    for (int64_t r = 0; r < n; ++r) {
        const float *xr = x + r * d, *hr = xhat + r * d;
        double gw[ORACLE_MAX_TOPK], gd[ORACLE_MAX_TOPK], xx = 0.0, sw = 0.0, sd = 0.0, dp = 0.0;
        for (int i = 0; i < d; ++i) {
            double zq = 0.0;
            for (int j = 0; j < topk; ++j) zq += (double)w[r * topk + j] * what[idx[r * topk + j] * d + i];
            diff[i] = zq - xr[i];
            geff[i] = (g_zq ? g_zq[r * d + i] : 0.0) + cv * diff[i];
            xx += (double)xr[i] * xr[i];
        }
        for (int j = 0; j < topk; ++j) {
            const float *e = what + idx[r * topk + j] * d;
            gw[j] = 0.0;
            for (int i = 0; i < d; ++i) gw[j] += geff[i] * e[i];
            sw += (double)w[r * topk + j] * gw[j];
        }
        for (int j = 0; j < topk; ++j) { gd[j] = -(double)w[r * topk + j] * (gw[j] - sw); sd += gd[j]; }
        for (int i = 0; i < d; ++i) {
            double acc = 0.0;
            for (int j = 0; j < topk; ++j) acc += gd[j] * what[idx[r * topk + j] * d + i];
            gxh[i] = 2.0 * (sd * hr[i] - acc) + (g_xhat ? g_xhat[r * d + i] : 0.0);
            dp += (double)hr[i] * gxh[i];
        }
        double nrm = sqrt(xx);
        if (nrm < 1e-12) nrm = 1e-12;
        if (gx)
            for (int i = 0; i < d; ++i)
                gx[r * d + i] = (float)((gxh[i] - hr[i] * dp) / nrm + (g_out ? g_out[r * d + i] : 0.0) - cc * diff[i]);
        if (g_code)
            for (int j = 0; j < topk; ++j) {
                const float *e = what + idx[r * topk + j] * d;
                for (int i = 0; i < d; ++i)
                    g_code[(r * topk + j) * d + i] = (float)((double)w[r * topk + j] * geff[i] + 2.0 * gd[j] * ((double)e[i] - hr[i]));
            }
    }
    free(geff); free(diff); free(gxh);
    return 0;
}

/* Backward of F.normalize(v, dim=-1, eps=1e-12): (g - vhat (vhat . g)) / max(|v|, 1e-12). */
int oracle_normalize_backward_f32(const float *g, const float *vhat, const float *v, int64_t n, int d, float *out)
{
    for (int64_t r = 0; r < n; ++r) {
        double dp = 0.0, vv = 0.0;
        for (int i = 0; i < d; ++i) { dp += (double)vhat[r * d + i] * g[r * d + i]; vv += (double)v[r * d + i] * v[r * d + i]; }
        double nrm = sqrt(vv);
        if (nrm < 1e-12) nrm = 1e-12;
        for (int i = 0; i < d; ++i) out[r * d + i] = (float)(((double)g[r * d + i] - vhat[r * d + i] * dp) / nrm);
    }
    return 0;
}

/* info_nce_loss (loss.py:40-56): normalise q and k, logits = [q.k_pos | q.k_neg (off-diagonal)] / T,
 * cross entropy against column 0 -- restated literally (positive first, then the negatives in
 * column order), plus its gradient w.r.t. q and k for upstream gradient g_loss. */
int oracle_info_nce_f32(const float *q, const float *k, int64_t b, int d, float temperature, float g_loss,
                        float *loss, float *gq, float *gk)
{
    double *qh = (double *)malloc(sizeof(double) * b * d), *kh = (double *)malloc(sizeof(double) * b * d);
    double *qn = (double *)malloc(sizeof(double) * b), *kn = (double *)malloc(sizeof(double) * b);
    double *gqh = (double *)calloc((size_t)b * d, sizeof(double)), *gkh = (double *)calloc((size_t)b * d, sizeof(double));
    double *lg = (double *)malloc(sizeof(double) * b);
    for (int64_t r = 0; r < b; ++r) {
        double a = 0.0, c = 0.0;
        for (int i = 0; i < d; ++i) { a += (double)q[r * d + i] * q[r * d + i]; c += (double)k[r * d + i] * k[r * d + i]; }
        qn[r] = sqrt(a) < 1e-12 ? 1e-12 : sqrt(a);
        kn[r] = sqrt(c) < 1e-12 ? 1e-12 : sqrt(c);
        for (int i = 0; i < d; ++i) { qh[r * d + i] = q[r * d + i] / qn[r]; kh[r * d + i] = k[r * d + i] / kn[r]; }
    }
    double total = 0.0;
    for (int64_t i = 0; i < b; ++i) {
        /* column 0 = positive, then negatives j != i */
        double m = -INFINITY;
        for (int64_t j = 0; j < b; ++j) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) s += qh[i * d + c] * kh[j * d + c];
            lg[j] = s / temperature;
            if (lg[j] > m) m = lg[j];
        }
        double se = exp(lg[i] - m);
        for (int64_t j = 0; j < b; ++j) if (j != i) se += exp(lg[j] - m);
        const double lse = m + log(se);
        total += lse - lg[i];
        for (int64_t j = 0; j < b; ++j) {
            const double dl = (double)g_loss * (exp(lg[j] - lse) - (j == i ? 1.0 : 0.0)) / (double)b / temperature;
            for (int c = 0; c < d; ++c) { gqh[i * d + c] += dl * kh[j * d + c]; gkh[j * d + c] += dl * qh[i * d + c]; }
        }
    }
    if (loss) *loss = (float)(total / (double)b);
    for (int64_t r = 0; r < b; ++r) {
        double dq = 0.0, dk = 0.0;
        for (int i = 0; i < d; ++i) { dq += qh[r * d + i] * gqh[r * d + i]; dk += kh[r * d + i] * gkh[r * d + i]; }
        for (int i = 0; i < d; ++i) {
            if (gq) gq[r * d + i] = (float)((gqh[r * d + i] - qh[r * d + i] * dq) / qn[r]);
            if (gk) gk[r * d + i] = (float)((gkh[r * d + i] - kh[r * d + i] * dk) / kn[r]);
        }
    }
    free(qh); free(kh); free(qn); free(kn); free(gqh); free(gkh); free(lg);
    return 0;
}

/* Ragged shared-key/value attention core: out[r] = softmax_j(scale <q[r], kv[j]>) . kv over each code's own
 * query rows [q_start[b], +q_len[b]) and key rows [kv_start[b], +kv_len[b]).  This is what
 * nn.MultiheadAttention (vector_quantization_soft_one_new.py:30,45) computes once its key/value projections are
 * folded into the queries; double accumulation (tolerance checker). */
int oracle_shared_kv_attention_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                   const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int d, float scale,
                                   float *out)
{
    for (int64_t b = 0; b < n_codes; ++b) {
        const int64_t kl = kv_len[b];
        double *sc = (double *)malloc(sizeof(double) * (kl > 0 ? kl : 1));
        for (int64_t r = q_start[b]; r < q_start[b] + q_len[b]; ++r) {
            double m = -INFINITY, l = 0.0;
            for (int64_t j = 0; j < kl; ++j) {
                double a = 0.0;
                for (int i = 0; i < d; ++i) a += (double)q[r * d + i] * kv[(kv_start[b] + j) * d + i];
                sc[j] = a * scale;
                if (sc[j] > m) m = sc[j];
            }
            for (int64_t j = 0; j < kl; ++j) { sc[j] = exp(sc[j] - m); l += sc[j]; }
            for (int i = 0; i < d; ++i) {
                double a = 0.0;
                for (int64_t j = 0; j < kl; ++j) a += sc[j] * kv[(kv_start[b] + j) * d + i];
                out[r * d + i] = l > 0.0 ? (float)(a / l) : 0.0f;      /* no key rows: the code attends to nothing (build's rule) */
            }
        }
        free(sc);
    }
    return 0;
}

/* The dropout mask of the training-mode attention core: a stateless 32-bit hash of (seed, packed query row, key of the code).
 * Bit for bit the device function att_keep() (attention_kernels.h) -- the mask is part of the function being checked. */
static int att_keep(uint32_t seed, int64_t qrow, int key, uint32_t thresh)
{
    uint32_t h = seed ^ ((uint32_t)qrow * 0x9E3779B1u) ^ ((uint32_t)((uint64_t)qrow >> 32) * 0x7F4A7C15u) ^ ((uint32_t)key * 0x85EBCA77u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h >= thresh;
}

/* Training-mode attention core and its backward (nn.MultiheadAttention with dropout on the attention weights, reference :21,30,45,
 * after the projections are folded into the queries): per code
 *   P = softmax(scale Q KV^T), M = keep / (1 - p), O = (P o M) KV, lse = log sum exp(scale Q KV^T);
 *   dQ = scale dS KV, dKV = (P o M)^T dO + scale dS^T Q with dS = P o ((dO KV^T) o M - <dO, O>).
 * Double accumulation (tolerance checker).  dq / dkv / lse may be NULL (forward only when d_out is NULL). */
int oracle_shared_kv_attention_train_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                         const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int d, float scale,
                                         float dropout_p, uint32_t seed, float *out, float *lse, const float *d_out, float *dq, float *dkv,
                                         int64_t kv_rows)
{
    double t = (double)dropout_p * 4294967296.0;
    const uint32_t thresh = dropout_p > 0.f ? (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t) : 0u;
    const double ks = dropout_p > 0.f ? (double)(1.f / (1.f - dropout_p)) : 1.0;
    if (d_out && dkv) memset(dkv, 0, sizeof(float) * (size_t)kv_rows * d);
    for (int64_t b = 0; b < n_codes; ++b) {
        const int64_t kl = kv_len[b], k0 = kv_start[b];
        double *p = (double *)malloc(sizeof(double) * (kl > 0 ? kl : 1));
        double *acc_kv = d_out && dkv ? (double *)calloc((size_t)(kl > 0 ? kl : 1) * d, sizeof(double)) : NULL;
        for (int64_t r = q_start[b]; r < q_start[b] + q_len[b]; ++r) {
            double m = -INFINITY, l = 0.0;
            for (int64_t j = 0; j < kl; ++j) {
                double a = 0.0;
                for (int i = 0; i < d; ++i) a += (double)q[r * d + i] * kv[(k0 + j) * d + i];
                p[j] = a * scale;
                if (p[j] > m) m = p[j];
            }
            for (int64_t j = 0; j < kl; ++j) { p[j] = exp(p[j] - m); l += p[j]; }
            for (int64_t j = 0; j < kl; ++j) p[j] /= l;
            if (lse) lse[r] = kl > 0 ? (float)(m + log(l)) : -INFINITY;
            for (int i = 0; i < d; ++i) {
                double a = 0.0;
                for (int64_t j = 0; j < kl; ++j)
                    if (!thresh || att_keep(seed, r, (int)j, thresh)) a += p[j] * ks * kv[(k0 + j) * d + i];
                out[r * d + i] = kl > 0 ? (float)a : 0.0f;
            }
            if (!d_out) continue;
            double delta = 0.0;
            for (int i = 0; i < d; ++i) delta += (double)d_out[r * d + i] * out[r * d + i];
            for (int i = 0; i < d && dq; ++i) dq[r * d + i] = 0.0f;
            double *dqa = (double *)calloc(d, sizeof(double));
            for (int64_t j = 0; j < kl; ++j) {
                const int keep = !thresh || att_keep(seed, r, (int)j, thresh);
                double dp = 0.0;
                for (int i = 0; i < d; ++i) dp += (double)d_out[r * d + i] * kv[(k0 + j) * d + i];
                dp = keep ? dp * ks : 0.0;
                const double ds = p[j] * (dp - delta) * scale;
                const double pm = keep ? p[j] * ks : 0.0;
                for (int i = 0; i < d; ++i) {
                    dqa[i] += ds * kv[(k0 + j) * d + i];
                    if (acc_kv) acc_kv[j * d + i] += pm * d_out[r * d + i] + ds * q[r * d + i];
                }
            }
            if (dq) for (int i = 0; i < d; ++i) dq[r * d + i] = (float)dqa[i];
            free(dqa);
        }
        if (acc_kv) {
            for (int64_t j = 0; j < kl; ++j)
                for (int i = 0; i < d; ++i) dkv[(k0 + j) * d + i] = (float)acc_kv[j * d + i];
            free(acc_kv);
        }
        free(p);
    }
    return 0;
}

/* EMA statistics (norm_ema_quantizer.py:194,202): bins[c] = #rows assigned to
 * c; embed_sum[c][:] = sum of those rows of zhat, added in increasing row
 * order (layout [K,D]; the reference's [D,K] is its transpose). */
int oracle_ema_stats_f32(const float *zhat, const int64_t *idx, int64_t n, int d,
                         int64_t k_codes, float *bins, float *embed_sum)
{
    memset(bins, 0, sizeof(float) * (size_t)k_codes);
    memset(embed_sum, 0, sizeof(float) * (size_t)k_codes * d);
    for (int64_t r = 0; r < n; ++r) {
        int64_t c = idx[r];
        if (c < 0 || c >= k_codes) return -1;
        bins[c] += 1.0f;
        float *dst = embed_sum + c * d;
        const float *src = zhat + r * d;
        for (int i = 0; i < d; ++i) dst[i] = dst[i] + src[i];
    }
    return 0;
}

/* EMA apply (norm_ema_quantizer.py:197-210,136-138,11-12):
 *   cluster_size <- decay*cluster_size + (1-decay)*bins
 *   new = l2norm(embed_sum / max(bins,1)); rows with bins==0 keep E
 *   E <- l2norm(decay*E + (1-decay)*new)
 * one_minus_decay is passed in because the reference forms it in Python
 * double precision (1 - 0.99) before it is cast to fp32. */
int oracle_ema_apply_f32(float *E, float *cluster_size, const float *bins,
                         const float *embed_sum, int64_t k_codes, int d,
                         float decay, float one_minus_decay)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)d);
    if (!tmp) return -2;
    for (int64_t c = 0; c < k_codes; ++c) {
        float b = bins[c];
        float a0 = cluster_size[c] * decay;
        float a1 = b * one_minus_decay;
        cluster_size[c] = a0 + a1;
        float *e = E + c * d;
        if (b == 0.0f) {
            memcpy(tmp, e, sizeof(float) * (size_t)d);
        } else {
            const float *s = embed_sum + c * d;
            for (int i = 0; i < d; ++i) tmp[i] = s[i] / b;
            float nrm = sqrtf(canon_sumsq(tmp, d));
            float den = nrm > 1e-12f ? nrm : 1e-12f;
            for (int i = 0; i < d; ++i) tmp[i] = tmp[i] / den;
        }
        for (int i = 0; i < d; ++i) {
            float m0 = e[i] * decay;
            float m1 = tmp[i] * one_minus_decay;
            tmp[i] = m0 + m1;
        }
        float nrm = sqrtf(canon_sumsq(tmp, d));
        float den = nrm > 1e-12f ? nrm : 1e-12f;
        for (int i = 0; i < d; ++i) e[i] = tmp[i] / den;
    }
    free(tmp);
    return 0;
}

/* Eval-mode branch (norm_ema_quantizer.py:185-189): only cluster_size moves. */
int oracle_ema_cluster_size_f32(float *cluster_size, const float *bins,
                                int64_t k_codes, float decay, float one_minus_decay)
{
    for (int64_t c = 0; c < k_codes; ++c) {
        float a0 = cluster_size[c] * decay;
        float a1 = bins[c] * one_minus_decay;
        cluster_size[c] = a0 + a1;
    }
    return 0;
}

/* codebook_usage (vector_quantization_soft_one_new.py:219-236): slide the
 * fp32 window left by m, append the new ids, count distinct values.
 * Returns the distinct count (the reference divides it by n_e). When m exceeds
 * the window the reference raises; here the last `window` ids are kept. */
int64_t oracle_usage_update(float *window, int64_t wlen, const int64_t *ids,
                            int64_t m, int64_t n_codes)
{
    if (m >= wlen) {
        for (int64_t i = 0; i < wlen; ++i) window[i] = (float)ids[m - wlen + i];
    } else {
        memmove(window, window + m, sizeof(float) * (size_t)(wlen - m));
        for (int64_t i = 0; i < m; ++i) window[wlen - m + i] = (float)ids[i];
    }
    unsigned char *seen = (unsigned char *)calloc((size_t)n_codes + 1, 1);
    if (!seen) return -2;
    int64_t cnt = 0;
    for (int64_t i = 0; i < wlen; ++i) {
        int64_t v = (int64_t)window[i];
        if (v < 0 || v >= n_codes) v = n_codes;
        if (!seen[v]) { seen[v] = 1; ++cnt; }
    }
    free(seen);
    return cnt;
}

/* ---- alignment / orthogonality losses (loss.py:59-83) -------------------------------------------------
 * row dot: 64 strided fmaf chains joined by the xor butterfly (canon_sumsq with two operands). */
int oracle_row_dot_f32(const float *a, const float *b, int64_t n, int d, float *out)
{
    for (int64_t r = 0; r < n; ++r) {
        float p[64], q[64];
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int i = 0; i < d; ++i) {
            int l = (i >> 2) & 63;
            p[l] = fmaf(a[r * d + i], b[r * d + i], p[l]);
        }
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
            memcpy(p, q, sizeof p);
        }
        out[r] = p[0];
    }
    return 0;
}

/* C[m, n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]: ONE fmaf chain over k = 0, 1, 2, ... from +0 per entry
 * (torch.mm(z.T, z_star), loss.py:79, and the two products of its backward). */
int oracle_small_gemm_f32(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn,
                          int m, int n, int k, float *C)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            float acc = 0.0f;
            for (int t = 0; t < k; ++t) acc = fmaf(A[i * sam + t * sak], B[t * sbk + j * sbn], acc);
            C[(int64_t)i * n + j] = acc;
        }
    return 0;
}

/* ||x||_F (torch.norm(., p='fro'), loss.py:82): canonical per-row sums of squares, added in fp64 in row order
 * by the kernel's fixed tree (1024 strided partial sums, pairwise), square root in fp64. */
int oracle_frobenius_f32(const float *x, int64_t rows, int d, float *out)
{
    double part[1024];
    for (int t = 0; t < 1024; ++t) part[t] = 0.0;
    for (int64_t r = 0; r < rows; ++r) part[r % 1024] += (double)canon_sumsq(x + r * d, d);
    for (int off = 512; off >= 1; off >>= 1)
        for (int t = 0; t < off; ++t) part[t] += part[t + off];
    out[0] = (float)sqrt(part[0]);
    return 0;
}

int oracle_abi_version(void) { return 1; }

/* ---- around the attention core -----------------------------------------------------------------------
 * CrossAttentionLayer's tail (vector_quantization_soft_one_new.py:47-50): y = LayerNorm(a + b) * gamma + beta, nn.LayerNorm's
 * biased variance and eps inside the square root.  Summation order of the kernel: element i belongs to chain (i / 4) % 64,
 * chains run in increasing i and meet in the xor butterfly (as canon_sumsq); the variance is that of the centred values. */
int oracle_residual_layernorm_f32(const float *a, const float *b, const float *gamma, const float *beta, int64_t n, int d,
                                  float eps, float *y)
{
    float *v = (float *)malloc(sizeof(float) * (size_t)d);
    if (!v) return -2;
    for (int64_t r = 0; r < n; ++r) {
        float p[64], q[64];
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int i = 0; i < d; ++i) {
            v[i] = a[r * d + i] + b[r * d + i];
            p[(i >> 2) & 63] += v[i];
        }
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
            memcpy(p, q, sizeof p);
        }
        const float mean = p[0] / (float)d;
        for (int i = 0; i < d; ++i) v[i] -= mean;
        const float rstd = 1.0f / sqrtf(canon_sumsq(v, d) / (float)d + eps);
        for (int i = 0; i < d; ++i) y[r * d + i] = fmaf(v[i] * rstd, gamma[i], beta[i]);
    }
    free(v);
    return 0;
}

/* `.mean(dim=0)` over each code's attended graph nodes (:140-141): rows added in order, one fp32 chain per column. */
int oracle_segment_mean_f32(const float *x, const int64_t *seg_start, const int64_t *seg_len, int64_t n_seg, int d, float *out)
{
    for (int64_t b = 0; b < n_seg; ++b) {
        const float den = (float)(seg_len[b] > 1 ? seg_len[b] : 1);
        for (int c = 0; c < d; ++c) {
            float acc = 0.0f;
            for (int64_t r = 0; r < seg_len[b]; ++r) acc += x[(seg_start[b] + r) * d + c];
            out[b * d + c] = acc / den;
        }
    }
    return 0;
}
