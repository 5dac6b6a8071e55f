// pack_kernels.h -- the prologue of CrossAttention.pooled: what the reference's per-code loop gets from `mask[idx].sum().item()` and
// `batch == idx` (vector_quantization_soft_one_new.py:133-142), for all codes at once and without a host round trip per code.
// Included by medtok_vq.hip; gfx950 only.
//
// Three small launches replace ~30 torch ops (scatter_add_, cumsum, argsort, stack, index, arange, ...: each a launch of a few
// microseconds that the forward's first dense product waits for):
//   pack_mask_len_kernel   valid_len[b] = number of non-zero entries of mask row b           one wavefront per row
//   pack_count_kernel      counts[b] = nodes whose batch id is b (integer atomics: exact); id range; is `batch` sorted?
//   pack_lists_kernel      ONE block: starts = exclusive scan of counts, the largest count, the codes ordered longest key set
//                          first (counting sort by valid_len; blocks of the attention launch are taken in list order and a block's
//                          time is its key count), and the (start, length) lists of both attention sides.
#pragma once

template <typename M>
__global__ __launch_bounds__(256) void pack_mask_len_kernel(const M *__restrict__ mask, long n_codes, long seq_len, int64_t *__restrict__ valid_len)
{
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_codes) return;
    const int lane = threadIdx.x & 63;
    const M *row = mask + b * seq_len;
    int cnt = 0;
    for (long i = lane; i < seq_len; i += 64) cnt += row[i] != (M)0 ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) valid_len[b] = cnt;
}

// stats: [0] largest count (filled by pack_lists_kernel), [1] smallest id, [2] largest id, [3] 1 if batch[i] < batch[i - 1] anywhere
__global__ __launch_bounds__(256) void pack_count_kernel(const int64_t *__restrict__ batch, long n_nodes, long n_codes, int *__restrict__ counts32,
                                                         int *__restrict__ stats32)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    long lo = 0x7fffffffL, hi = -0x7fffffffL;
    int uns = 0;
    if (i < n_nodes) {
        const long id = batch[i];
        lo = hi = id < -0x7fffffffL ? -0x7fffffffL : (id > 0x7fffffffL ? 0x7fffffffL : id);
        const long c = id < 0 ? 0 : (id >= n_codes ? n_codes - 1 : id);     // (out-of-range ids are reported through the id range)
        atomicAdd(&counts32[c], 1);
        if (i > 0 && id < batch[i - 1]) uns = 1;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
        uns |= __shfl_xor(uns, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats32[1], (int)lo);
        atomicMax(&stats32[2], (int)hi);
        if (uns) atomicOr(&stats32[3], 1);
    }
}

constexpr int PACK_THREADS = 1024;
constexpr int PACK_MAX_KEYS = 8192;        // longest key set the longest-first order is computed for (beyond: list order = code order)

__global__ __launch_bounds__(PACK_THREADS) void pack_lists_kernel(
    const int *__restrict__ counts32, const int *__restrict__ stats32, const int64_t *__restrict__ valid_len, long n_codes, long seq_len, int heads, int lpt,
    int *__restrict__ order, int64_t *__restrict__ counts, int64_t *__restrict__ starts, int64_t *__restrict__ t_start, int64_t *__restrict__ t_len,
    int64_t *__restrict__ g_start, int64_t *__restrict__ g_len, int64_t *__restrict__ tok_start, int64_t *__restrict__ g_kv_len, int64_t *__restrict__ stats)
{
    __shared__ long s_wave[PACK_THREADS / 64];
    __shared__ long s_carry;
    __shared__ int s_max;
    extern __shared__ int s_hist[];                // [seq_len + 2] when lpt
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_carry = 0; s_max = 0; }
    __syncthreads();
    // ---- exclusive scan of the counts, 1024 codes per pass
    int my_max = 0;
    for (long base = 0; base < n_codes; base += PACK_THREADS) {
        const long c = base + tid;
        const long v = c < n_codes ? counts32[c] : 0;
        my_max = max(my_max, (int)v);
        long incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        long before = s_carry;
        for (int k = 0; k < wv; ++k) before += s_wave[k];
        if (c < n_codes) {
            counts[c] = v;
            starts[c] = before + incl - v;
            t_start[c] = c * heads;
            t_len[c] = heads;
        }
        __syncthreads();
        if (tid == PACK_THREADS - 1) s_carry = before + incl;
        __syncthreads();
    }
    atomicMax(&s_max, my_max);
    // ---- the list order: longest key set first (counting sort over the key counts), or code order
    const bool sort = lpt && seq_len < PACK_MAX_KEYS;
    if (sort) {
        for (long i = tid; i < seq_len + 2; i += PACK_THREADS) s_hist[i] = 0;
        __syncthreads();
        for (long c = tid; c < n_codes; c += PACK_THREADS) {
            const long len = valid_len[c];
            atomicAdd(&s_hist[len < 0 ? 0 : (len > seq_len ? seq_len : len)], 1);
        }
        __syncthreads();
        if (tid == 0) {                            // descending offsets: bucket seq_len first
            int run = 0;
            for (long len = seq_len; len >= 0; --len) { const int h = s_hist[len]; s_hist[len] = run; run += h; }
        }
        __syncthreads();
        for (long c = tid; c < n_codes; c += PACK_THREADS) {
            const long len = valid_len[c];
            order[atomicAdd(&s_hist[len < 0 ? 0 : (len > seq_len ? seq_len : len)], 1)] = (int)c;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (long p = tid; p < n_codes; p += PACK_THREADS) {
        const long c = sort ? order[p] : p;
        g_start[p] = starts[c] * heads;
        g_len[p] = counts[c] * heads;
        tok_start[p] = c * seq_len;
        g_kv_len[p] = valid_len[c];
    }
    if (tid == 0) {
        stats[0] = s_max;
        stats[1] = stats32[1];
        stats[2] = stats32[2];
        stats[3] = stats32[3];
    }
}
