/*
 * medtok_vq.h -- C ABI of the MI355X (gfx950) MedTok vector-quantisation library.
 *
 * This is the drop-in boundary for the hot path BASELINE.json:north_star names.
 * The reference has no FFI of its own: its boundary is a set of Python classes
 * that call ATen ops.  Each entry point below replaces one group of those ATen
 * call sites (cited as file:line under the reference tree); the Python classes in
 * medtok_amd/ bind them with ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous row-major buffer owned by
 *     the caller; float buffers are 16-byte aligned and D % 4 == 0;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all
 *     work is enqueued on it and nothing synchronises the host;
 *   - functions return 0 on success, non-zero on error; the message is available
 *     from medtok_last_error() (thread-local).  No exceptions cross the ABI;
 *   - no global state: scratch memory comes from the caller (`ws`), sized by the
 *     *_workspace_bytes() queries.  Calls are re-entrant across streams as long
 *     as they use different workspaces.
 *
 * Arithmetic contract (identical to oracle/medtok_oracle.c, which tests use as
 * the checker): x.e is one fp32 fmaf chain that visits each group of 8 elements in the
 * order 0,4,1,5,2,6,3,7 (the order v_mfma_f32_32x32x2_f32 consumes two float4 halves);
 * |v|^2 is 64 strided fmaf chains joined by an xor butterfly; d = (|x|^2 + |e|^2) - 2*(x.e); ties go
 * to the lowest code index.  Search results are therefore bit-reproducible and
 * independent of tile shape, code sharding and which search path ran.
 */
#ifndef MEDTOK_VQ_H
#define MEDTOK_VQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEDTOK_VQ_ABI_VERSION 2
/* codes per row a search returns: 1 .. 16.  Lists of up to 8 run on either search path; 9 .. 16 (the reference takes any k,
 * vector_quantization_soft_one_new.py:91) always take the exact fp32 path, as two passes of lists of 8 (the 8 best, then the best
 * among the codes behind the row's 8th (distance, index) pair) -- the same total order as one list of k. */
#define MEDTOK_MAX_TOPK 16

/* search path selector for medtok_topk_search_f32 */
#define MEDTOK_PATH_AUTO 0          /* library picks the fastest exact path          */
#define MEDTOK_PATH_F32_MFMA 1      /* brute force on v_mfma_f32_32x32x2_f32          */
#define MEDTOK_PATH_F16_FILTER 2    /* fp16-MFMA shortlist + exact fp32 re-score: same bits, ~16x the matrix rate */
#define MEDTOK_PATH_MASK 0xF        /* the selector proper; the bits above it are the test hooks below */
/* Test hooks, OR-ed into any `path` argument: force launch-plan branches the default heuristics only take at very large shapes
 * (code-range splits, the XCD-aware block order, the tail launch; the split cap of the exact kernel) so that small parity tests
 * cover them.  Per call -- the library keeps no plan state; the *_workspace_bytes query must be given the same `path`.
 * Results are bit-identical under every plan (tests/test_gpu_filter.py). */
#define MEDTOK_PLAN_FILTER_SPLITS(s) (((s) & 0xFF) << 8)            /* 0 = default */
#define MEDTOK_PLAN_FILTER_XCD(on) (((on) ? 2 : 1) << 16)           /* XCD-aware block order on / off (full 256-CU device only) */
#define MEDTOK_PLAN_FILTER_TAIL(on) (((on) ? 2 : 1) << 18)          /* tail launch from 256 blocks up / never */
#define MEDTOK_PLAN_SEARCH_MAX_SPLITS(s) (((s) & 0xFF) << 20)       /* 0 = default */
#define MEDTOK_PLAN_FILTER_ROWS64(on) (((on) ? 2 : 1) << 28)        /* D <= 64: the 128-byte-row filter kernels on / off (default on) */
#define MEDTOK_PLAN_FILTER_ROWS64_WIDE (3 << 28)                    /* D <= 64: the 128 x 64 wave-tile form (two blocks per CU) instead of the 128 x 32 one */

/* flags for medtok_soft_assign_f32 */
#define MEDTOK_ASSIGN_HARD 1        /* NormEMA form: topk == 1, zq = what[idx]        */
#define MEDTOK_ASSIGN_RAW 2         /* write zq itself instead of xref + (zq - xref)  */

int medtok_abi_version(void);
const char *medtok_last_error(void);

/* Optional self-profiling for bench.py: between _begin and _end every launch of the three matrix-pipe
 * kernels is bracketed by HIP events on its launch stream (no host sync until _end).  _end fills, for
 * kind 0 = filter_f16_kernel, 1 = search_f32_kernel, 2 = the attention forward kernels, 3 = the attention backward
 * pair (dQ + dKV), 4 = split_gemm_kernel: total milliseconds, total algorithmic flops (2*n*K*D per search launch,
 * 2*m*n*k per dense product -- its fp32-equivalent work, computed as three fp16 passes; 0 for kinds 2 and 3, whose ragged
 * row/key counts live on the device -- the caller prices them) and the number of launches.  Process-wide (autograd launches the
 * backward kernels from its own thread); meant for one profiling client at a time. */
#define MEDTOK_PROFILE_KINDS 5
int medtok_profile_begin(void);
/* ... bracketing only the launches of the kinds whose bit is set in `kinds` (bit k = kind k): a training step is ~180 library launches,
 * and two event records per launch are ~0.5 ms of an 11 ms step; a timed region then carries the events of its dominant kernel only. */
int medtok_profile_begin_kinds(unsigned kinds);
int medtok_profile_end(double ms[MEDTOK_PROFILE_KINDS], double flops[MEDTOK_PROFILE_KINDS],
                       int launches[MEDTOK_PROFILE_KINDS]);

/* Several SMALL soft top-k searches in one call and three launches (normalise, search, merge + assign) instead of four launches each:
 * the B = 256 forward of the reference's default configuration (train_MedTok.py:363-368,387) runs its two modality-specific
 * searches (:187-217) and its shared searches (:147-165) on 256-512 rows each -- 16 launches of 4-30 us for 2 GFLOP.  Every
 * descriptor is one medtok_soft_vq_forward_f32 call (row_sqerr NULL: no squared-error output) and yields the same bits; all searches share
 * d and topk.  Only searches that medtok_soft_vq_multi_eligible() accepts (the exact fp32-MFMA path, at most 4096 rows).
 * xhat [n, d], idx [n, topk], dist [n, topk], w [n, topk] (may be NULL), zq [n, d] with row stride zq_stride (0 = d). */
typedef struct medtok_search_desc {
    const float *x; int64_t n;
    const float *what, *wsq; int64_t k_codes;
    float *xhat; int64_t *idx; float *dist, *w, *zq; int64_t zq_stride;
    int64_t x_stride;                       /* row stride of x in floats (0 = d): x may be a column block of a wider matrix */
    float *row_sqerr;                       /* [n] or NULL: per-row squared error of the soft assignment (medtok_soft_vq_forward_f32's
                                               row_sqerr: the training forward's vq / commitment losses are its fixed-order sums) */
} medtok_search_desc;
#define MEDTOK_MULTI_SEARCH_MAX 6
int medtok_soft_vq_multi_eligible(int64_t n, int64_t k_codes, int d, int topk);
size_t medtok_soft_vq_forward_multi_workspace_bytes(const medtok_search_desc *descs, int count, int d, int topk);
int medtok_soft_vq_forward_multi_f32(const medtok_search_desc *descs, int count, int d, int topk, void *ws, size_t ws_bytes, void *stream);

/* The forward's three to five updates of the usage window (vector_quantization_soft_one_new.py:219-236 called at :183,216 via
 * :241-250: shared, text, graph and, with an aug view, its two) in one call of two launches: ids[u] (int64 [m[u]]) are appended in
 * order; counts_out[u] = distinct window values after update u (what medtok_usage_update reports for each, one by one). */
#define MEDTOK_USAGE_MULTI_MAX 6
size_t medtok_usage_multi_workspace_bytes(int64_t window_len, int64_t n_codes, int count);
int medtok_usage_update_multi(float *window, int64_t window_len, const int64_t *const *ids, const int64_t *m, int count, int64_t n_codes,
                              int32_t *counts_out, void *ws, size_t ws_bytes, void *stream);
/* the same with a device word copied behind the counts (counts_out [count + 1]; extra_word NULL: as above): the caller's one host read
 * of the usage counts then also brings e.g. the cross-attention's status word.  A NON-ZERO word vetoes the window write (the counts
 * are still made): the caller's device-side input checks failed, and it repeats the forward on repaired inputs against the window
 * as it was */
int medtok_usage_update_multi_word(float *window, int64_t window_len, const int64_t *const *ids, const int64_t *m, int count, int64_t n_codes,
                                   int32_t *counts_out, const int32_t *extra_word, void *ws, size_t ws_bytes, void *stream);

/* C = unscale * (A . B^T) + bias as ONE half-precision pass with fp32 accumulation: what torch.autocast makes of nn.Linear and of
 * nn.MultiheadAttention's projections (train_MedTok.py:212,394 -> vector_quantization_soft_one_new.py:30,45,106-107).  a [m, lda] and
 * b [b_rows, ldb]: fp16 (bf16 = 0) or bf16 (bf16 = 1) matrices; grouping, shapes and alignment as medtok_split_gemm_f16 (k_g % 32 == 0,
 * strides % 8 == 0, n_g % 4 == 0); c fp32 [m, ldc]. */
int medtok_half_gemm_f32(const void *a, int64_t m, int lda, int a_group_cols, const void *b, int64_t b_rows, int ldb, int b_group_rows,
                         int n_g, int k_g, int groups, const float *bias, float unscale, float *c, int ldc, int bf16, void *stream);

/* The 16-bit image (fp16; bf16 != 0: bf16) of the fp32 matrix src [n, d] (row stride src_stride): [n, dp] with zero columns past d, or with
 * transpose != 0 the image of src^T, [d, dp] with dp >= n, written as dp / group_cols groups of [d, group_cols] stacked along the rows
 * (group_cols = 0: one group) -- the operands of medtok_half_gemm_f32, made without torch's strided copies. */
int medtok_half_image_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int transpose, int64_t group_cols, int bf16,
                          void *out, void *stream);
/* both images of one matrix in one pass over it (a training-mode product needs its upstream gradient and its input both ways):
 * out_plain = medtok_half_image_f32(transpose = 0, dp = dp_plain <= d rounded up to 64), out_t = medtok_half_image_f32(transpose = 1, dp = np, group_cols) */
int medtok_half_image_pair_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                               int bf16, void *out_plain, void *out_t, void *stream);
/* ... and col_partials [(np + 63) / 64, d] fp32 from the same pass: the sums over every 64-row tile of each column of src, in a fixed
 * order -- their sum over the tiles is the column sum of src (a Linear's bias gradient when src is its upstream gradient). */
int medtok_half_image_pair_sums_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp_plain, int64_t np, int64_t group_cols,
                                    int bf16, void *out_plain, void *out_t, float *col_partials, void *stream);
/* ... the transposed image alone (medtok_half_image_f32 with transpose = 1) with the same col_partials [(dp + 63) / 64, d]. */
int medtok_half_image_t_sums_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, int64_t group_cols, int bf16,
                                 void *out, float *col_partials, void *stream);

/* Shader-clock probe for bench.py: one idle wavefront on each of 8 blocks (one per XCD on the full chip) samples the shader-cycle
 * counter and the constant 100 MHz counter from launch until *stop_flag (a word of PINNED HOST memory the device polls) becomes
 * non-zero, or max_ticks_100mhz ticks have passed, whichever is first; out = uint64 [8][4] (device memory): shader cycles, 100 MHz
 * ticks, XCC id, polls.  Launch on a stream of its own that shares no hardware queue with the measured work.  No reference
 * counterpart: measurement infrastructure (the chip clocks to its power budget; a bench line should say at which clock it ran). */
int medtok_debug_clock_probe(const int *stop_flag, uint64_t max_ticks_100mhz, uint64_t *out, void *stream);

/* F.normalize(x, p=2, dim=-1, eps=1e-12) and the squared norm of the result.
 * Replaces vector_quantization_soft_one_new.py:148,150-151,196,198,200 and
 * norm_ema_quantizer.py:8-9,170 plus the two torch.sum(..**2) terms of
 * get_distance (:121-122; norm_ema_quantizer.py:175-176).
 * normalize == 0: rows are taken as they are (xhat may be NULL or == x).
 * xhat may alias x.  sqn may be NULL. */
int medtok_rownorm_f32(const float *x, int64_t n, int d, int normalize,
                       float *xhat, float *sqn, void *stream);

/* Scratch bytes medtok_topk_search_f32 needs for this problem size. */
size_t medtok_search_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path);

/* Nearest-code search.  For every row the `topk` smallest
 *   d = (xsq[r] + wsq[c]) - 2 * <xhat[r], what[c]>
 * ascending, ties to the lowest c.  Replaces get_distance + torch.topk
 * (vector_quantization_soft_one_new.py:120-125,157,159,203) and
 * the distance + torch.argmin of norm_ema_quantizer.py:175-179 (topk = 1).
 * idx  [n, topk] int64 (indices local to `what`), dist [n, topk] fp32.
 * The N x K distance matrix is never materialised. */
int medtok_topk_search_f32(const float *xhat, const float *xsq, int64_t n,
                           const float *what, const float *wsq, int64_t k_codes,
                           int d, int topk, int64_t *idx, float *dist,
                           void *ws, size_t ws_bytes, int path, void *stream);

/* Exact merge of per-shard results of a code-sharded search: dist_parts / idx_parts are
 * [parts, n, topk] (idx already global code ids), output is the top-k of the union by (d, index).
 * Because the arithmetic contract does not depend on where a code lives, this equals the
 * single-GPU result bit for bit. */
int medtok_merge_topk_lists_f32(const float *dist_parts, const int64_t *idx_parts, int64_t n,
                                int parts, int topk, int64_t *idx, float *dist, void *stream);

/* Test hook: the approximate scores s~ [n, k_codes] the fp16 filter works with, so the error
 * bound it relies on (medtok_amd/csrc/filter_f16.h) can be measured. Not part of the product path. */
size_t medtok_debug_filter_scores_workspace_bytes(int64_t n, int64_t k_codes, int d);
int medtok_debug_filter_scores_f32(const float *xhat, const float *xsq, int64_t n,
                                   const float *what, const float *wsq, int64_t k_codes, int d,
                                   float *scores, void *ws, size_t ws_bytes, void *stream);

/* Test hook: where, inside the workspace of a filter-path search, the int32 count of rows that were handed to the exact kernel
 * lives (candidate-list overflow, out-of-range norms, NaN); (size_t)-1 when the shape does not take the filter path. */
size_t medtok_debug_filter_fallback_count_offset(int64_t n, int64_t k_codes, int d, int topk, int path);

/* Measurement hook (bench.py --data, tests): what the shortlist pass of a FINISHED filter-path search left in its workspace `ws`
 * (soft_vq_ws != 0: the workspace of a medtok_soft_vq_forward*_f32 call), read on the same stream behind the call:
 * out[0] = candidates over all rows and lists, out[1] = lists at or over capacity, out[2] = rows handed to the exact kernel,
 * out[3] = lists in total (uint64 [4], device memory).  Fails for shapes that do not take the filter path. */
int medtok_debug_filter_stats(const void *ws, size_t ws_bytes, int soft_vq_ws, int64_t n, int64_t k_codes, int d, int topk, int path,
                              uint64_t *out, void *stream);

/* Soft assignment: w = softmax(-dist), zq = sum_j w_j * what[idx_j],
 * zq_ste = xref + (zq - xref), row_sqerr[r] = sum_i (zq - xref)^2.
 * Replaces vector_quantization_soft_one_new.py:158,160,164-165,169-173,181-182,
 * 204-205,208-209,214.  MEDTOK_ASSIGN_HARD is the NormEMA form (topk == 1,
 * zq = what[idx]; norm_ema_quantizer.py:181,212,214); MEDTOK_ASSIGN_RAW stores zq
 * (the tensor autograd differentiates) instead of the straight-through value.
 * w and row_sqerr may be NULL.  zq_out may alias xref; zq_stride is its row stride in floats
 * (0 = d), so four searches can write straight into the columns of one [n, 4d] embedding
 * (the torch.cat of tokenizer.py:246).  idx entries must lie in [0, K) like F.embedding's; the ids the
 * search entry points produce always do, also for rows whose distances are NaN or infinite. */
int medtok_soft_assign_f32(const float *xref, const float *what, const int64_t *idx,
                           const float *dist, int64_t n, int d, int topk, int flags,
                           float *w, float *zq_out, int64_t zq_stride, float *row_sqerr, void *stream);

/* out[0] = scale * sum(vals[0..n)), accumulated in fp64 in a fixed order
 * (the mean of the squared error: :169-173,208-209; F.mse_loss at
 * norm_ema_quantizer.py:212). */
int medtok_sum_scale_f32(const float *vals, int64_t n, double scale, float *out, void *stream);

/* ---- fp32-accurate dense products on the fp16 matrix pipe (the projections around the cross-attention core:
 * nn.MultiheadAttention's in_proj / out_proj and the folded W_k / W_v products, vector_quantization_soft_one_new.py:17-51).
 * Every operand is a PAIR of fp16 images (hi, lo) with x = hi + lo; a product runs as three fp16 MFMA passes
 * (hi hi + hi lo + lo hi, fp32 accumulation): ~2^-22 relative, the order of a plain fp32 GEMM's own round-off.
 *
 * medtok_split_half_f32: src [n, d] fp32 (row stride src_stride floats) -> hi, lo [n, dp] fp16, dp >= d a multiple of 8, columns
 * past d zero; `scale` (an exact power of two) is applied first -- 1 for activations, the per-matrix prescale for weights. */
int medtok_split_half_f32(const float *src, int64_t n, int d, int64_t src_stride, int dp, float scale,
                          void *hi, void *lo, const int64_t *seg_len, int seg_rows, void *stream);
/* (seg_len != NULL: the rows are segments of seg_rows rows -- a [B, L, d] batch -- and only the first seg_len[b] rows of segment b
 * are converted: padding tokens, which are never read as keys, are skipped.) */

/* C = unscale * (A . B^T) + bias for `groups` independent problems that share the row range [0, m):
 *   group g:  A_g = columns [g * a_group_cols, + k_g) of A [m, lda],  B_g = rows [g * b_group_rows, + n_g) of B [b_rows, ldb]
 *             (depth k_g, k contiguous in both),  C_g = columns [g * n_g, + n_g) of C.
 * Outputs: c (fp32 [m, ldc]) and/or the (hi, lo) images c_hi / c_lo ([m, ldch] fp16) that a following product reads directly.
 * k_g % 32 == 0, n_g % 4 == 0, lda / ldb / a_group_cols % 8 == 0, ldc / ldch % 4 == 0; bias [groups * n_g] or NULL.
 * groups = 1: a plain GEMM.  The per-head products of the folded attention form are groups = heads. */
int medtok_split_gemm_f16(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols,
                          const void *b_hi, const void *b_lo, int64_t b_rows, int ldb, int b_group_rows,
                          int n_g, int k_g, int groups, const float *bias, float unscale,
                          float *c, int ldc, void *c_hi, void *c_lo, int ldch, void *stream);

/* The same products for TRAINING (the projections of the cross-attention under autograd: forward, data gradient and weight gradient
 * are all "A . B^T" with fp32-accurate split operands; reference: nn.MultiheadAttention's in/out projections,
 * vector_quantization_soft_one_new.py:30,45).  Operands whose magnitude is not known on the host (activations, upstream gradients
 * -- 1e-9 .. 1e+5 under a GradScaler) are prescaled by the power of two that brings their largest magnitude into [2^11, 2^12),
 * taken from a DEVICE-side |x|_max: no host read anywhere.
 *   medtok_absmax_f32            amax[0] = max |x[i]| (device float; non-finite inputs give a non-finite amax -> prescale 1)
 *   medtok_split_half_scaled_f32 the (hi, lo) images of src [n, d] prescaled by pow2(amax) (amax NULL: 1); transpose = 0: [n, dp]
 *                                (dp >= d); transpose = 1: the images of src^T, [d, dp] with dp >= n (weight gradients contract
 *                                over the rows) -- with group_cols (0 = dp; else a multiple of 64 dividing dp) written as
 *                                dp / group_cols groups of [d, group_cols] stacked along the rows, the layout the grouped product
 *                                reads as "group g = k chunk g": a split-K weight gradient in one launch; zero padding throughout
 *   medtok_split_gemm_scaled_f16 medtok_split_gemm_f16 (fp32 output) with unscale / (pow2(amax_a) pow2(amax_b)) applied to the
 *                                accumulators (either amax may be NULL) */
int medtok_absmax_f32(const float *x, int64_t count, float *amax, void *stream);
int medtok_split_half_scaled_f32(const float *src, int64_t n, int d, int64_t src_stride, int64_t dp, const float *amax, int transpose,
                                 int64_t group_cols, void *hi, void *lo, void *stream);
int medtok_split_gemm_scaled_f16(const void *a_hi, const void *a_lo, int64_t m, int lda, int a_group_cols, const void *b_hi, const void *b_lo,
                                 int64_t b_rows, int ldb, int b_group_rows, int n_g, int k_g, int groups, const float *bias, float unscale,
                                 const float *amax_a, const float *amax_b, float *c, int ldc, void *stream);

/* ---- training half ------------------------------------------------------------------------------
 * Sparse backward of the soft top-k assignment (the reference back-propagates through a dense
 * N x K distance matrix built at vector_quantization_soft_one_new.py:120-125,157-182,203-214; only
 * the k selected columns are non-zero).  With e_j = what[idx_j], zq = sum_j w_j e_j:
 *   geff = g_zq + (*g_vq) * vq_scale * (zq - x)
 *   gx   = dF.normalize^T[ d(dist)/d(xhat)^T . dsoftmax^T . (geff . e_j)  + g_xhat ] + g_out
 *          - (*g_commit) * commit_scale * (zq - x)
 *   g_code[row*topk + j, :] = gradient w.r.t. the NORMALISED code e_j
 * g_zq / g_xhat / g_out are [n, d] or NULL (= zero); g_vq / g_commit are DEVICE scalars or NULL, so
 * the upstream loss gradients never visit the host.  For the reference losses vq_scale = 2/(n d)
 * and commit_scale = 2 beta/(n d).  gx or g_code may be NULL when not wanted.  Sum g_code per code
 * id with medtok_ema_stats_f32 (row order, deterministic), then medtok_normalize_backward_f32
 * takes the result to codebook.weight. */
int medtok_soft_vq_backward_f32(const float *x, const float *xhat, const float *what,
                                const int64_t *idx, const float *w, int64_t n, int d, int topk,
                                const float *g_zq, const float *g_xhat, const float *g_out,
                                const float *g_vq, const float *g_commit,
                                float vq_scale, float commit_scale,
                                float *gx, float *g_code, void *stream);

/* Backward of F.normalize(v, p=2, dim=-1, eps=1e-12) (:148-151,196-200):
 * out = (g - vhat (vhat . g)) / max(|v|, 1e-12), row-wise. */
int medtok_normalize_backward_f32(const float *g, const float *vhat, const float *v,
                                  int64_t n, int d, float *out, void *stream);
/* ... with live [n]: rows whose entry is 0 hold an all-zero g (the bins of medtok_ema_stats_f32 for a per-code gradient): written as zeros
 * without reading g / vhat / v -- the same bits as the call above. */
int medtok_normalize_backward_sparse_f32(const float *g, const float *vhat, const float *v, const float *live, int64_t n, int d, float *out,
                                         void *stream);

/* info_nce_loss (loss.py:40-56): cross entropy of [positive | off-diagonal negatives] / T with
 * label 0 over normalised q, k [b, d]  ==  CE(qhat khat^T / T, diagonal).  loss is a device
 * scalar; prob [b, b] (softmax rows) and the workspace (qhat | khat | row losses) are what
 * medtok_info_nce_backward_f32 needs to produce gq, gk [b, d] from the device scalar g_loss. */
size_t medtok_info_nce_workspace_bytes(int64_t b, int d);
int medtok_info_nce_forward_f32(const float *q, const float *k, int64_t b, int d, float temperature,
                                float *loss, float *prob, void *ws, size_t ws_bytes, void *stream);
int medtok_info_nce_backward_f32(const float *q, const float *k, const float *prob,
                                 const float *g_loss, int64_t b, int d, float temperature,
                                 float *gq, float *gk, const void *ws, size_t ws_bytes, void *stream);

/* The two regularisers of loss.py next to InfoNCE, forward and backward.
 *   alignment_loss(mu1, mu2) = mean_b <mu1[b], mu2[b]>  (loss.py:59-64): medtok_row_dot_f32 -> out[b], then
 *     medtok_sum_scale_f32(out, B, 1/B); its gradients are mu2 * g / B and mu1 * g / B (medtok_scale_by_device_scalar_f32).
 *   orthogonal_loss(z, z*) = || z^T z* ||_F  (loss.py:66-83): M = z^T z* by medtok_small_gemm_f32 (strides pick the
 *     transposition), medtok_frobenius_f32(M); backward: G = M * g / ||M|| (scale_by_device_scalar with den = the norm),
 *     dz = z* G^T and dz* = z G, two more small GEMMs.
 * medtok_small_gemm_f32: C[m, n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn], C row-major [m, n]; every entry is one fp32 fmaf
 * chain over k in increasing order (v_mfma_f32_32x32x2_f32), bit-reproducible.  Meant for the loss-sized operands
 * (a few hundred rows): a correctness-first kernel, not a tuned GEMM.  d % 4 == 0 for the row kernels. */
int medtok_row_dot_f32(const float *a, const float *b, int64_t n, int d, float *out, void *stream);
int medtok_small_gemm_f32(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn,
                          int m, int n, int k, float *C, void *stream);
size_t medtok_frobenius_workspace_bytes(int64_t rows);
int medtok_frobenius_f32(const float *x, int64_t rows, int d, float *out, void *ws, size_t ws_bytes, void *stream);
/* out[i] = x[i] * (c * num[0] / den[0]); num, den: device scalars (den may be NULL = 1; den[0] == 0 gives 0). */
int medtok_scale_by_device_scalar_f32(const float *x, int64_t count, const float *num, const float *den, float c,
                                      float *out, void *stream);

/* Cross-attention core of get_shared_info (vector_quantization_soft_one_new.py:17-88,133-142) for ragged
 * batches.  With nn.MultiheadAttention's key/value projections folded into the queries on the host
 * (q_h.(Wk_h t + bk_h) = (Wk_h^T q_h).t + const;  sum_j p_j (Wv_h t_j + bv_h) = Wv_h (sum_j p_j t_j) + bv_h)
 * what remains per medical code b is
 *     out[r, :] = softmax_j( scale * <q[r, :], kv[j, :]> ) . kv
 * over the code's own query rows [q_start[b], +q_len[b]) of q and key rows [kv_start[b], +kv_len[b]) of kv
 * (raw rows of the other modality; every head is just another query row).  Nothing is padded and the
 * rows x keys matrix never reaches memory.  max_q_len >= max_b q_len[b] sizes the grid (rows beyond a
 * code's q_len cost nothing); d = 64 or d % 128 == 0, d <= 768 (d = 1024 -- BERT-large features -- with exact_f32 = 0 and
 * max_q_len <= 4 only: wider query sets at that width take medtok_shared_kv_attention_split_f32).  A code with kv_len == 0 attends to nothing: its
 * rows are zero (the reference's per-code loop would take a softmax over an empty set there).
 * q_start/q_len/kv_start/kv_len are DEVICE int64[n_codes]; any number of codes per call.
 * exact_f32 = 0 (what the modules use at inference): both products on the fp16 matrix pipe as three MFMAs over (hi, lo) fp16
 * pairs -- fp32 inputs, fp32 softmax, ~2^-22-relative products, 16/3 of the fp32 pipe's rate; |q|, |kv| entries must stay
 * below 65504 (beyond it the result is NaN, not a silently wrong number).  exact_f32 = 1: both products on
 * v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains), the kernel the training forward shares. */
int medtok_shared_kv_attention_f32(const float *q, const int64_t *q_start, const int64_t *q_len,
                                   const float *kv, const int64_t *kv_start, const int64_t *kv_len,
                                   int64_t n_codes, int64_t max_q_len, int d, float scale,
                                   float *out, void *out_hi, void *out_lo, int exact_f32, void *stream);
/* (out_hi / out_lo, exact_f32 = 0 only: the (hi, lo) fp16 images [Rq, d] of the result, written by the kernel itself for the
 * dense product that follows (medtok_split_gemm_f16); out may then be NULL.
 * exact_f32 = 0 and max_q_len <= 8 -- the text side of get_shared_info, one query row per code and head -- runs one wavefront per
 * code in plain fp32 FMA arithmetic instead of 32-row matrix tiles: same function, same tolerance class.) */

/* The same core for wide inference batches: 64 query rows per block and the keys given as the (hi, lo) fp16 images of
 * medtok_split_half_f32 (kv_hi / kv_lo [Rk, d], made once per forward: a key row serves every query tile of its code and both
 * layers), copied into LDS by DMA -- no per-block conversion, half the key traffic per query row.  Same arithmetic as
 * medtok_shared_kv_attention_f32 with exact_f32 = 0 (three fp16 MFMA passes per product, fp32 softmax); d = 128, 256, 384, 512, 768
 * or 1024 (32 rows per block there, whatever `variant`).  Rows of the images that no (kv_start, kv_len) range covers are never read. */
int medtok_shared_kv_attention_split_f32(const float *q, const int64_t *q_start, const int64_t *q_len,
                                         const void *kv_hi, const void *kv_lo, const int64_t *kv_start, const int64_t *kv_len,
                                         int64_t n_codes, int64_t max_q_len, int d, float scale, float *out,
                                         void *out_hi, void *out_lo, int variant, void *stream);
/* variant: 0 = 32 query rows per block and (d = 768) two blocks per CU -- one block's softmax and copy waits overlap the other's
 * matrix work; 1 (d = 768) = 64 rows per block, one block per CU, double-buffered key ring; 2 (d = 256, 512, 768; what the Python
 * layer passes) = two 32-row query tiles of a code per block, run one phase apart on ONE two-deep ring of key chunks
 * (attention_pp.h).  Same function; 0 / 1 agree to the last bits, 2 within 2e-6 of them (exp on v_exp_f32).
 * kv_lo = NULL (variant 2 only): the keys are fp16 as they stand -- a caller under fp16 autocast hands over half-precision text
 * features (train_MedTok.py:212,394): no lo image is read, two matrix passes per product instead of three; equal to the call with
 * an all-zero lo image bit for bit.
 * variant | MEDTOK_ATTENTION_F32_KEYS (variant 2 only): kv_hi points at the fp32 key rows [Rk, d] themselves and kv_lo is ignored;
 * the kernel forms the (hi, lo) images of every key chunk in LDS (the arithmetic of medtok_split_half_f32: same bits as the call
 * on its images) -- no image pass over the key batch, no image buffers. */
#define MEDTOK_ATTENTION_F32_KEYS 0x100

/* Dev probes (tools/r04/att_probe.py, tools/r04/filter_probe.py): per-wave cycle counts (s_memtime) of a kernel's loop segments,
 * written by a TIMED instantiation that only these entry points launch.  medtok_debug_set_attention_probe(p) arms the next
 * medtok_shared_kv_attention_split_f32 calls with variant = 2 | (8 << 4), d = 768 (uint64 [blocks][8 waves][8]; NULL disarms; the one
 * piece of process state besides the bench profiler); medtok_debug_filter_probe runs the filter kernel of one search
 * (uint64 [blocks][8][8]: MFMA group 1, wait for own copies, stage barrier, MFMA group 2, tile epilogue, stages, code tiles). */
void medtok_debug_set_attention_probe(void *probe);
/* DEV: bit 0 of `on`: run the one-pass half-precision products (medtok_half_gemm_f32) with 32-deep stages as before round 6 instead of
 * 64-deep ones; bit 1: give the dense products' tiles to the XCDs by row tile whatever the row-tile count (before round 6: a product of 3
 * row tiles ran on 3 of the 8 XCDs) -- the A/B switches of tools/r06/ab_half_gemm_k64.py; same results either way. */
void medtok_debug_set_half_gemm_k32(int on);
int medtok_debug_filter_probe(const float *xhat, const float *xsq, int64_t n, const float *what, const float *wsq, int64_t k_codes,
                              int d, int topk, void *ws, size_t ws_bytes, void *probe, size_t probe_bytes, int64_t *n_blocks,
                              void *stream);

/* CrossAttention over a whole batch at the reference's default width (vector_quantization_soft_one_new.py:17-88,133-142 with
 * e_dim = 64, num_head = 4: train_MedTok.py:363-368) in TWO launches and no host round trip: for every code b
 *     pooled[b * pooled_stride ...]             = the CLS row of text[b] after `layers` cross-attention layers against b's nodes
 *     pooled[b * pooled_stride + graph_off ...] = the mean over b's nodes after `layers` layers against text[b, :valid_b]
 * text [n_codes, seq_len, 64] fp32; mask [n_codes, seq_len] (1 / 4 / 8 bytes per element, non-zero = valid, left-aligned: the
 * number of non-zero entries of a row is its token count -- the reference's mask.sum()); nodes [n_nodes, 64] with a NON-DECREASING
 * batch vector (PyG-style); weights: per layer 4 * 64 * 64 + 6 * 64 floats = Wq^T [in][out] | Wk [out][in] | Wv^T [in][out] |
 * Wo^T [in][out] | bq | bv | bo | ln gamma | ln beta | 64 unused; y_nodes [n_nodes, 64]: the attended node rows (scratch the
 * mean reads).  status: int32 [4] the CALLER zeroes once; the kernels OR bit 0 (batch vector not sorted) / bit 1 (id outside
 * [0, n_codes)) into word 0 -- results are then undefined for the codes involved, but every access stays in bounds.
 * The 64 x 64 products are fp32 FMAs; the attention core runs on v_mfma_f32_32x32x16_f16 as three passes over (hi, lo) fp16 pairs
 * (a_hi b_hi + a_hi b_lo + a_lo b_hi: ~2^-22 relative, fp32 accumulation and softmax -- the wide kernels' arithmetic; within the
 * 1e-5 bar of the attention outputs.  Inputs are expected in fp16 range, |x| < 65504, like the wide inference kernels'). */
int medtok_cross_attention_small_f32(const float *text, const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len,
                                     const float *nodes, const int64_t *batch, int64_t n_nodes, int d, int heads, int layers,
                                     const float *weights, float scale, float ln_eps, float *y_nodes, float *pooled,
                                     int64_t pooled_stride, int64_t graph_off, int32_t *status, void *stream);
/* Test hook: the same call with the attention core on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fmaf chains) instead of the
 * three-pass split-fp16 core the entry point above runs -- a second opinion for the tests */
int medtok_debug_cross_attention_small_exact_f32(const float *text, const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len,
                                     const float *nodes, const int64_t *batch, int64_t n_nodes, int d, int heads, int layers,
                                     const float *weights, float scale, float ln_eps, float *y_nodes, float *pooled,
                                     int64_t pooled_stride, int64_t graph_off, int32_t *status, void *stream);

/* A codebook prepared once per weight version.  Every filter-path search turns its codebook region into an fp16 image, pads its
 * squared norms into accumulator start values and takes their maximum: three passes over the region PER SEARCH, four to six times
 * per forward on one codebook (vector_quantization_soft_one_new.py:107-117 searches three overlapping regions of one weight).
 *   medtok_filter_image_width(d)   the image's row width dp (fp16 elements) at row width d;
 *   medtok_rownorm_image_f32       medtok_rownorm_f32(normalize = 1) that also writes the image of the normalised rows,
 *                                  [image_rows, dp] with rows >= n zero: image_rows >= n + 255 so that every region's last code tile
 *                                  is inside it.  A region [lo, lo + k) of the codebook reads the image from row lo on;
 *   medtok_codebook_prepare_f32    per region: wsqp [k rounded up to 256] = -2^15 wsq (padding -inf) and en_max [1], one launch;
 *   medtok_soft_vq_forward_prepared_f32   medtok_soft_vq_forward_f32 (row_sqerr = NULL) for one region with its prepared image /
 *                                  wsqp / en_max: the same bits, without the three passes.  `what` / `wsq` are still the region's
 *                                  fp32 rows and norms (the exact re-score reads them); a search that takes the exact path
 *                                  ignores the prepared arguments. */
typedef struct medtok_region_desc { int64_t lo, k; float *wsqp; float *en_max; } medtok_region_desc;
int medtok_filter_image_width(int d);
int medtok_rownorm_image_f32(const float *x, int64_t n, int d, float *xhat, float *sqn, void *image, int64_t image_rows, int dp, void *stream);
int medtok_codebook_image_f32(const float *what, int64_t n, int d, void *image, int64_t image_rows, int dp, void *stream);   /* the image alone, from rows that are normalised already */
int medtok_codebook_prepare_f32(const float *wsq, const medtok_region_desc *regions, int count, void *stream);
int medtok_search_resolved_path(int64_t n, int64_t k_codes, int d, int topk, int path);   /* MEDTOK_PATH_F32_MFMA or MEDTOK_PATH_F16_FILTER: what `path` (e.g. AUTO) comes to at this shape */
int medtok_soft_vq_forward_prepared_f32(const float *x, int64_t n, int d, const float *what, const float *wsq, int64_t k_codes,
                                        int topk, int path, const void *image, const float *wsqp, const float *en_max,
                                        float *xhat, int64_t *idx, float *dist, float *w, float *zq_ste,
                                        int64_t zq_stride, void *ws, size_t ws_bytes, void *stream);

/* The prologue of the batched cross-attention: what the reference's per-code loop reads back with `.item()` and `batch == idx`
 * (vector_quantization_soft_one_new.py:133-142), for all codes at once, in three small launches and no host round trip:
 *   valid_len[b] = non-zero entries of mask row b (mask [n_codes, seq_len], elements of mask_elem_bytes = 1 (bool), 4 or 8 bytes);
 *   counts[b] = nodes with batch id b (exact), starts = their exclusive scan (nodes of a code are adjacent in a sorted `batch`);
 *   the (start, length) lists of the two attention launches, `heads` query rows per node / per code:
 *     text side  t_start[b] = b heads, t_len[b] = heads                                  (code order)
 *     graph side g_start[p] = starts[c] heads, g_len[p] = counts[c] heads, tok_start[p] = c seq_len, g_kv_len[p] = valid_len[c]
 *                with c = the p-th code in list order: longest key set first when lpt != 0 (a block's time is its key count),
 *                else code order;
 *   stats[4] = {largest count, smallest batch id, largest batch id, 1 if `batch` is not sorted} for ONE host read (ids outside
 *   [0, n_codes) are counted into the nearest valid code and reported through the id range: the caller rejects them).
 * All outputs are DEVICE int64 buffers of n_codes entries (stats: 4). */
size_t medtok_pack_codes_workspace_bytes(int64_t n_codes);
int medtok_pack_codes(const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len, const int64_t *batch, int64_t n_nodes,
                      int heads, int lpt, int64_t *valid_len, int64_t *counts, int64_t *starts, int64_t *t_start, int64_t *t_len,
                      int64_t *g_start, int64_t *g_len, int64_t *tok_start, int64_t *g_kv_len, int64_t *stats,
                      void *ws, size_t ws_bytes, void *stream);
/* The same for a caller that does NOT read stats back (a forward recorded into a HIP graph sizes its attention launches from a bound
 * on the nodes per code that it brings along): what the host read would have checked is OR-ed into the device word status[0]
 * (int32, zeroed by the caller once; NULL: no check) -- bit 0: `batch` is not sorted, bit 1: an id outside [0, n_codes), bit 2: a
 * code with more than count_bound nodes (count_bound = 0: no bound).  With a bit set the attention results of the codes involved
 * are wrong; the caller reads the word wherever it synchronises anyway. */
int medtok_pack_codes_checked(const void *mask, int mask_elem_bytes, int64_t n_codes, int64_t seq_len, const int64_t *batch, int64_t n_nodes,
                              int heads, int lpt, int64_t *valid_len, int64_t *counts, int64_t *starts, int64_t *t_start, int64_t *t_len,
                              int64_t *g_start, int64_t *g_len, int64_t *tok_start, int64_t *g_kv_len, int64_t *stats,
                              int64_t count_bound, int *status, void *ws, size_t ws_bytes, void *stream);

/* Around the core, for packed rows (no batch axis):
 *   medtok_residual_layernorm_f32: the tail of CrossAttentionLayer.forward (vector_quantization_soft_one_new.py:47-50),
 *     y[r] = LayerNorm(a[r] + b[r]) * gamma + beta  with nn.LayerNorm's biased variance and eps inside the square root
 *     (a = the layer's input rows, b = out_proj(attended); dropout is the identity in eval).  d % 4 == 0, d <= 4096.
 *   medtok_segment_mean_f32: `.mean(dim=0)` of each code's attended graph nodes (:140-141),
 *     out[b] = sum of rows [seg_start[b], seg_start[b] + seg_len[b]) of x, added in row order, / max(seg_len[b], 1).
 *     seg_start / seg_len are DEVICE int64[n_seg]; d % 4 == 0.  An empty segment gives a zero row. */
int medtok_residual_layernorm_f32(const float *a, const float *b, const float *gamma, const float *beta,
                                  int64_t n, int d, float eps, float *y, void *stream);
/* (the same with the (hi, lo) fp16 images [n, dp] of y as a second output -- what medtok_split_half_f32(y, dp) would make, for the
 * next layer's first dense product; dp >= d, a multiple of 8, zero columns appended; y_hi = y_lo = NULL: the function above) */
int medtok_residual_layernorm_split_f32(const float *a, const float *b, const float *gamma, const float *beta,
                                        int64_t n, int d, float eps, float *y, void *y_hi, void *y_lo, int dp, void *stream);
int medtok_segment_mean_f32(const float *x, const int64_t *seg_start, const int64_t *seg_len, int64_t n_seg, int d,
                            float *out, void *stream);

/* One CrossAttentionLayer (vector_quantization_soft_one_new.py:17-51, projections folded into the queries) at inference in ONE call:
 * rows [n_rows, d] -> (hi, lo) images (or the caller's, rows_hi / rows_lo [n_rows, dw], e.g. the previous layer's y_hi / y_lo)
 * -> in_proj -> per-head fold -> attention core over the ragged (q_start, q_len, kv_start, kv_len) lists (keys: kv fp32 [Rk, dw], or
 * their fp16 images kv_hi / kv_lo as in medtok_shared_kv_attention_split_f32) -> per-head W_v -> out_proj -> LayerNorm(rows + .)
 * -> y [n_rows, d] (+ its images [n_rows, dw] when y_hi is given).  Weight images and biases as CrossAttention._split_weights lays
 * them out: wq, wv [heads hp, dw], wk [heads dw, hp], wo [d, heads hp], bq, bv [heads hp], bo [d]; hp = the head width padded to 32.
 * The seven launches are the entry points above with the same arguments -- the same bits as seven separate calls; the
 * intermediates live in `ws` (medtok_cross_attention_layer_workspace_bytes). */
size_t medtok_cross_attention_layer_workspace_bytes(int64_t n_rows, int d, int dw, int heads, int hp);
int medtok_cross_attention_layer_f32(
    const float *rows, const void *rows_hi, const void *rows_lo, int64_t n_rows, int d, int dw, int heads, int hp,
    const void *wq_hi, const void *wq_lo, float wq_unscale, const float *bq, const void *wk_hi, const void *wk_lo, float wk_unscale,
    const void *wv_hi, const void *wv_lo, float wv_unscale, const float *bv, const void *wo_hi, const void *wo_lo, float wo_unscale,
    const float *bo, const int64_t *q_start, const int64_t *q_len, int64_t n_codes, int64_t max_q_len, const float *kv,
    const void *kv_hi, const void *kv_lo, const int64_t *kv_start, const int64_t *kv_len, float scale, int variant,
    const float *ln_gamma, const float *ln_beta, float ln_eps, float *y, void *y_hi, void *y_lo, void *ws, size_t ws_bytes, void *stream);

/* The same core for TRAINING, and its backward: dropout on the attention probabilities (nn.MultiheadAttention(dropout=0.1),
 * reference :21,30) by a stateless hash mask of (seed, packed query row, key) -- P(keep) = 1 - dropout_p, kept probabilities
 * scaled by 1 / (1 - dropout_p); the forward also returns lse[r] = log sum_j exp(scale <q_r, kv_j>) (-inf for an empty key
 * set) from which the backward rebuilds the probabilities.  Backward (autograd of :45 through the folded form):
 *   dq [q_rows, d]  = scale * dS . kv            dkv [kv_rows, d] = (P o M)^T . d_out + scale * dS^T . q
 * with dS = P o ((d_out . kv^T) o M - <d_out, out>).  Rows of dq / dkv that belong to no code (or to no key of a code's slot)
 * are zeroed.  max_kv_len >= max_b kv_len[b]; ws from medtok_shared_kv_attention_backward_workspace_bytes(q_rows). */
int medtok_shared_kv_attention_train_f32(const float *q, const int64_t *q_start, const int64_t *q_len,
                                         const float *kv, const int64_t *kv_start, const int64_t *kv_len,
                                         int64_t n_codes, int64_t max_q_len, int d, float scale, float dropout_p,
                                         uint32_t seed, float *out, float *lse, void *stream);
/* ... the same forward on the three-pass fp16 products (two 32-row query tiles of a code per block on one LDS copy of its keys, which
 * the kernel splits into (hi, lo) fp16 images itself): out and lse equal medtok_shared_kv_attention_train_f32's to ~1e-6 relative, the
 * dropout mask is bit-identical; d = 256, 512 or 768; q, kv, out 16-byte aligned.  The autocast trainer's form (train_MedTok.py:212):
 * a third of the exact kernel's time at d = 768. */
int medtok_shared_kv_attention_train_split_f32(const float *q, const int64_t *q_start, const int64_t *q_len,
                                               const float *kv, const int64_t *kv_start, const int64_t *kv_len,
                                               int64_t n_codes, int64_t max_q_len, int d, float scale, float dropout_p,
                                               uint32_t seed, float *out, float *lse, void *stream);
size_t medtok_shared_kv_attention_backward_workspace_bytes(int64_t q_rows);
int medtok_shared_kv_attention_backward_f32(const float *q, const int64_t *q_start, const int64_t *q_len,
                                            const float *kv, const int64_t *kv_start, const int64_t *kv_len,
                                            int64_t n_codes, int64_t max_q_len, int64_t max_kv_len, int64_t q_rows,
                                            int64_t kv_rows, int d, float scale, float dropout_p, uint32_t seed,
                                            const float *out, const float *lse, const float *d_out, float *dq, float *dkv,
                                            void *ws, size_t ws_bytes, void *stream);
/* ... the same with its four matrix products as ONE half-precision pass (fp16; bf16 != 0: bf16) with fp32 accumulation: what
 * torch.autocast makes of nn.MultiheadAttention's Q K^T and P V (train_MedTok.py:212,394).  Operands, the softmax rebuilt from the
 * forward's log-sum-exp, and the outputs stay fp32. */
int medtok_shared_kv_attention_backward_half_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                 const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                 int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                                 uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                                 float *dkv, void *ws, size_t ws_bytes, int bf16, void *stream);
/* ... either form (mode 0: exact fp32, 1: fp16, 2: bf16 products) with dKV ADDED to what dkv already holds when accumulate_dkv != 0.
 * Every layer of CrossAttention attends to the ORIGINAL other modality (vector_quantization_soft_one_new.py:83,86): the key gradients of
 * all layers land in one [kv_rows, d] buffer, in launch order (a block owns its key rows: the sum is ordered, held + this launch's).
 * Rows no block owns are then left untouched instead of zeroed.  accumulate_dkv == 0: exactly the two entries above. */
/* accumulate_dkv == 2: dQ ONLY -- dkv may be NULL, nothing of the key gradient is computed; ws (q_rows floats) then holds delta = <d_out, out>
 * per query row, which medtok_shared_kv_attention_dkv_multi_f32 below takes with the call's other tensors as one of its sources. */
int medtok_shared_kv_attention_backward_acc_f32(const float *q, const int64_t *q_start, const int64_t *q_len, const float *kv,
                                                const int64_t *kv_start, const int64_t *kv_len, int64_t n_codes, int64_t max_q_len,
                                                int64_t max_kv_len, int64_t q_rows, int64_t kv_rows, int d, float scale, float dropout_p,
                                                uint32_t seed, const float *out, const float *lse, const float *d_out, float *dq,
                                                float *dkv, void *ws, size_t ws_bytes, int mode, int accumulate_dkv, void *stream);

/* The key gradient of SEVERAL attention calls over the same keys (kv, kv_start, kv_len) in ONE launch: the layers of CrossAttention all
 * attend to the original other modality (:83,86), so dKV is one sum -- a block walks the sources' queries one source after the other into
 * one accumulator and stores its key rows once (one launch per layer: a zero fill, a store and a read-add-store pass over [kv_rows, d]).
 * A source = one earlier dQ-only call (accumulate_dkv == 2 above): its q, d_out, lse, its workspace as delta, its q_start / q_len and
 * dropout parameters.  dkv [kv_rows, d]: rows no block owns are zeroed.  mode as above. */
typedef struct medtok_dkv_source {
    const float *q, *d_out, *lse, *delta;
    const int64_t *q_start, *q_len;
    float scale, dropout_p;
    uint32_t seed, reserved_;
} medtok_dkv_source;
#define MEDTOK_DKV_SOURCES_MAX 4
int medtok_shared_kv_attention_dkv_multi_f32(const medtok_dkv_source *sources, int count, const float *kv, const int64_t *kv_start,
                                             const int64_t *kv_len, int64_t n_codes, int64_t max_kv_len, int64_t kv_rows, int d,
                                             float *dkv, int mode, void *stream);

/* EMA statistics of norm_ema_quantizer.py:183,194,202 without the one-hot:
 * bins[c] = #rows with idx == c (exact), embed_sum[c][:] = sum of those rows of
 * zhat added in increasing row order (deterministic).  embed_sum is [K, D]
 * (the transpose of the reference's [D, K]; all-reduce is layout-agnostic). */
size_t medtok_ema_stats_workspace_bytes(int64_t n, int64_t k_codes);
int medtok_ema_stats_f32(const float *zhat, const int64_t *idx, int64_t n, int d,
                         int64_t k_codes, float *bins, float *embed_sum,
                         void *ws, size_t ws_bytes, void *stream);

/* bins only: the eval branch (norm_ema_quantizer.py:185-188) needs no embed_sum. */
size_t medtok_code_histogram_workspace_bytes(int64_t k_codes);
int medtok_code_histogram_f32(const int64_t *idx, int64_t n, int64_t k_codes, float *bins,
                              void *ws, size_t ws_bytes, void *stream);

/* EMA apply of norm_ema_quantizer.py:197-210 (+ :11-12,136-138), in place:
 * cluster_size <- decay*cs + (1-decay)*bins; rows with bins == 0 keep their
 * code; E <- l2norm(decay*E + (1-decay)*l2norm(embed_sum/bins)). */
int medtok_ema_apply_f32(float *E, float *cluster_size, const float *bins,
                         const float *embed_sum, int64_t k_codes, int d,
                         float decay, float one_minus_decay, void *stream);

/* Eval-mode branch (norm_ema_quantizer.py:185-189): cluster_size only. */
int medtok_ema_cluster_size_f32(float *cluster_size, const float *bins, int64_t k_codes,
                                float decay, float one_minus_decay, void *stream);

/* codebook_usage (vector_quantization_soft_one_new.py:219-236): slide the fp32
 * window left by m, append ids, count distinct values into *count_out (device
 * int32).  The caller divides by n_e. */
size_t medtok_usage_workspace_bytes(int64_t window_len, int64_t n_codes);
int medtok_usage_update(float *window, int64_t window_len, const int64_t *ids, int64_t m,
                        int64_t n_codes, int32_t *count_out, void *ws, size_t ws_bytes,
                        void *stream);

/* The head of NormEMAVectorQuantizer.forward (norm_ema_quantizer.py:169-179) in one call: zhat = F.normalize(z) [n, d], its squared
 * norms zsq [n], and the topk nearest rows of the codebook (what, wsq as for medtok_topk_search_f32; topk = 1 is the reference's
 * argmin).  Same bits as medtok_rownorm_f32 followed by medtok_topk_search_f32; on the fp16-shortlist path the normalising pass
 * also writes the fp16 image the shortlist streams (one pass over z less). */
size_t medtok_normalized_search_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path);
int medtok_normalized_search_f32(const float *z, int64_t n, int d, const float *what, const float *wsq,
                                 int64_t k_codes, int topk, int path, float *zhat, float *zsq, int64_t *idx,
                                 float *dist, void *ws, size_t ws_bytes, void *stream);

/* One-call forward of VectorQuantizer.specific_embedding / the search half of
 * get_shared_info (vector_quantization_soft_one_new.py:147-182,194-214) for
 * rows x [n, d] against an already normalised codebook slice:
 * rownorm(x) -> search -> soft assign.  Outputs as in the pieces above (on the filter path without row_sqerr the
 * assignment is fused into the re-score kernel: same bits, one gather of the top-k code rows less). */
size_t medtok_soft_vq_workspace_bytes(int64_t n, int64_t k_codes, int d, int topk, int path);
int medtok_soft_vq_forward_f32(const float *x, int64_t n, int d,
                               const float *what, const float *wsq, int64_t k_codes,
                               int topk, int path,
                               float *xhat, int64_t *idx, float *dist, float *w,
                               float *zq_ste, int64_t zq_stride, float *row_sqerr,
                               void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDTOK_VQ_H */
