// pack_kernels.h -- the prologue of CrossAttention.pooled: what the reference's per-code loop gets from `mask[idx].sum().item()` and
// `batch == idx` (vector_quantization_soft_one_new.py:133-142), for all codes at once and without a host round trip per code.
// Included by medtok_vq.hip; gfx950 only.
//
// Three small launches replace ~30 torch ops (scatter_add_, cumsum, argsort, stack, index, arange, ...: each a launch of a few
// microseconds that the forward's first dense product waits for):
//   pack_mask_len_kernel   valid_len[b] = number of non-zero entries of mask row b           one wavefront per row
//   pack_count_kernel      counts[b] = nodes whose batch id is b (integer atomics: exact); id range; is `batch` sorted?
//   pack_lists_kernel      ONE block: starts = exclusive scan of counts, the largest count, the codes ordered longest key set
//                          first (ties in code order: a stable, run-to-run identical list; blocks of the attention launch are
//                          taken in list order and a block's time is its key count), and the (start, length) lists of both sides.
#pragma once

// (also the first launch of the three: it zeroes the node counts and sets the id-range statistics to their start values -- a
// hipMemsetAsync and a host-to-device copy of four ints before; the copy, from the caller's stack, could not be captured into a graph)
template <typename M>
__global__ __launch_bounds__(256) void pack_mask_len_kernel(const M *__restrict__ mask, long n_codes, long seq_len, int64_t *__restrict__ valid_len,
                                                            int *__restrict__ counts32, int *__restrict__ stats32)
{
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    for (long c = gid; c < n_codes; c += (long)gridDim.x * 256) counts32[c] = 0;
    if (gid < 4) stats32[gid] = gid == 1 ? 0x7fffffff : (gid == 2 ? -0x7fffffff : 0);
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
constexpr int PACK_SORT_MAX_CODES = 8192;  // ... and the largest batch (a bitonic sort of one 32-bit key per code in LDS; beyond: code order)

__global__ __launch_bounds__(PACK_THREADS) void pack_lists_kernel(
    const int *__restrict__ counts32, const int *__restrict__ stats32, const int64_t *__restrict__ valid_len, long n_codes, long seq_len, int heads, int lpt,
    int *__restrict__ order, int64_t *__restrict__ counts, int64_t *__restrict__ starts, int64_t *__restrict__ t_start, int64_t *__restrict__ t_len,
    int64_t *__restrict__ g_start, int64_t *__restrict__ g_len, int64_t *__restrict__ tok_start, int64_t *__restrict__ g_kv_len, int64_t *__restrict__ stats,
    long count_bound = 0, int *__restrict__ status = nullptr)
{
    __shared__ long s_wave[PACK_THREADS / 64];
    __shared__ long s_carry;
    __shared__ int s_max;
    extern __shared__ unsigned s_key[];            // [next power of two >= n_codes] when the list is ordered
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
    // ---- the list order: longest key set first, codes of equal length in code order (a bitonic sort of (seq_len - length, code)
    // keys: the SAME list on every run and rank -- the counting sort of round 4 placed equal-length codes by atomicAdd, i.e. in an
    // order that changed from run to run: same outputs, but different block-to-code assignments, timings and profiles), or code order.
    // valid_len counts the NON-ZERO mask entries of a row; the reference takes mask.sum() (:135): the same for 0/1 masks.
    const bool sort = lpt && seq_len < PACK_MAX_KEYS && n_codes <= PACK_SORT_MAX_CODES;
    if (sort) {
        long pw = 1;
        while (pw < n_codes) pw <<= 1;
        for (long i = tid; i < pw; i += PACK_THREADS) {
            unsigned key = 0xffffffffu;
            if (i < n_codes) {
                const long len = valid_len[i];
                key = (unsigned)(seq_len - (len < 0 ? 0 : (len > seq_len ? seq_len : len))) * (unsigned)PACK_SORT_MAX_CODES + (unsigned)i;
            }
            s_key[i] = key;
        }
        __syncthreads();
        for (long k = 2; k <= pw; k <<= 1)
            for (long j = k >> 1; j > 0; j >>= 1) {
                for (long t = tid; t < pw / 2; t += PACK_THREADS) {
                    const long i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;       // the pair (i, i + j), i without bit j
                    const unsigned a = s_key[i], b = s_key[p];
                    if ((a > b) == ((i & k) == 0)) { s_key[i] = b; s_key[p] = a; }
                }
                __syncthreads();
            }
        for (long i = tid; i < n_codes; i += PACK_THREADS) order[i] = (int)(s_key[i] % (unsigned)PACK_SORT_MAX_CODES);
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
        // a caller that sizes its launches from a BOUND on the node count instead of reading stats back (no host synchronisation: the
        // forward records into a HIP graph) finds out here whether the batch kept to it: bit 0 = batch vector not sorted, bit 1 = an
        // id outside [0, n_codes), bit 2 = a code with more nodes than the bound
        if (status) {
            int f = stats32[3] ? 1 : 0;
            if (stats32[1] <= stats32[2] && (stats32[1] < 0 || stats32[2] >= n_codes)) f |= 2;
            if (count_bound > 0 && s_max > count_bound) f |= 4;
            if (f) atomicOr(status, f);
        }
    }
}
