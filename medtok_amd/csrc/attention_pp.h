// attention_pp.h -- the ragged attention core for wide inference batches, "ping-pong" form: TWO 32-row query tiles of one code per
// block, one copy of the code's keys.  Included by medtok_vq.hip after attention_dma.h; gfx950 only.
//
//     out[r, :] = softmax_j( scale * <q[r, :], kv[j, :]> ) . kv        (vector_quantization_soft_one_new.py:17-88,133-142, folded)
//
// The kernel of attention_dma.h runs a chunk of 16 keys as three phases separated by block-wide barriers -- S: partial scores of
// the block's rows over each wave's column slice (matrix pipe), X: join the partials + online softmax (VALU, LDS), V: out += P . KV
// (matrix pipe) -- and with D = 768 a block's LDS (one 48 KB chunk) leaves room for two blocks per CU but not for a second chunk
// buffer: every chunk's copy is issued when the previous chunk is done and waited for in full (measured on the `full` workload:
// ~9000 cycles per 16-key chunk and block of which the matrix pipe works 1150; two blocks per CU overlap each other by luck).
// Here a block is EIGHT waves = two groups of four; group g owns query tile 2 p + g of the code and the two groups run the same
// three phases ONE PHASE APART on the same key chunks:
//
//         slot 3c      slot 3c+1    slot 3c+2    slot 3c+3
//   g0    S(c)         X(c)         V(c)         S(c+1)
//   g1    V(c-1)       S(c)         X(c)         V(c)            (one s_barrier per slot)
//
// so that (1) while one group is in its softmax the other one has the matrix pipe, by construction instead of by luck: every slot
// holds exactly one S or V per SIMD beside an X, or an S beside a V; (2) a key chunk is copied into LDS ONCE for 64 query rows
// (half the L2 -> LDS traffic and DMA issue of two independent 32-row blocks); (3) the LDS that the second block's chunk took is
// the second buffer of a two-deep ring: chunk c + 2 is copied (LDS-DMA, by all eight waves: group g copies plane g -- hi / lo --
// of its column slice) while chunk c + 1 is being worked on -- issued at slot 3c+4, when the last reader of its buffer (g1's
// V(c)) is done, waited for at the end of slot 3c+5.  Everything else -- the (hi, lo) fp16 images of the keys, the swizzled LDS
// image, transposed value reads, the three-pass split products, the arithmetic order -- is attention_dma.h's <4, NT, 1, *> form:
// the results are bit-identical to it.  A code's last, odd tile runs with group 1 idle (it still copies its share of the keys).
#pragma once

template <int NT>
struct AttPP {
    static constexpr int W = 4, D = 32 * W * NT;
    static constexpr int PIECE = 1024;                    // bytes one DMA instruction writes: 16 keys x 64 B
    static constexpr int PLANEB = NT * PIECE;             // one plane (hi or lo) of a wave's slice of a chunk
    static constexpr int CHUNKB = W * 2 * PLANEB;         // a 16-key chunk: W slices x (hi, lo)
    static constexpr int PSL = 24;                        // halves per probability row: 16 keys + 8 (conflict-free b128 reads)
    static constexpr int PART_FLOATS = W * 32 * 17;       // [W][32][17] partial scores [row][key], per group
    static constexpr int HALVES = 2;                      // column passes of the output staging
    static constexpr size_t RING_B = 2 * (size_t)CHUNKB;
    static constexpr size_t GROUP_B = (size_t)PART_FLOATS * 4 + (size_t)2 * 32 * PSL * 2 + 2 * 32 * 4;   // part | ph, pl | alpha, l
    static constexpr size_t STAGE_FLOATS = (size_t)32 * (D / HALVES + 4);                                // one group's output staging
    static constexpr size_t LDS_BYTES = (RING_B + 2 * GROUP_B > 2 * STAGE_FLOATS * 4 ? RING_B + 2 * GROUP_B : 2 * STAGE_FLOATS * 4) + 64;
};

// exp on the transcendental unit: v_exp_f32 (2^x, <= 1 ulp) of x log2(e) -- ~1e-6 relative on the probabilities, inside the 1e-5
// bar of the attention outputs; expf()'s range reduction was a third of the softmax step.
__device__ __forceinline__ float pp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// TIMED (dev probe, tools/r04/att_probe.py through medtok_debug_set_attention_probe; the product instantiates TIMED = false): per-wave
// cycle counts (s_memtime) of the three phases and of the waits between them, written to `dbg` (uint64 [blocks][8 waves][8]).
// KLO = false: the keys are fp16 as they stand (a caller under fp16 autocast hands over half-precision text features, the
// reference's default training mode train_MedTok.py:212,394): there is no lo image -- kvl is not read, a chunk is one plane (half
// the bytes from HBM and through the DMA), and the passes against it drop out of both products (two MFMAs per step instead of
// three).  With lo = 0 the three-pass form adds exact zeros, so the results equal the KLO form on the widened keys bit for bit.
// KF32: the keys are the caller's fp32 rows (kvh points at them, kvl is not read) and become their (hi, lo) fp16 images INSIDE the
// kernel: the image pass over the whole text batch (read 4 B + write 4 B per element, 1.4 ms at BASELINE sizes, and its 3.2 GB of
// images read back by both layers) is gone, the kernel reads the same 4 B per element it read as images.  A wave's share of a
// chunk's copy is then both planes of HALF its group's pieces, and every 16-byte DMA slot of the hi plane receives the first four
// floats of the eight columns whose hi image belongs there, the slot of the lo plane at the same position the other four: when
// its own copies have landed, a lane reads its two slots, forms hi = fp16(x), lo = fp16(x - hi) (the arithmetic of
// split_half_kernel: same bits), and writes the two 16-byte images back into the two slots it read -- lane-local and in place, no
// staging area, no extra barrier; the slot's closing barrier publishes the images.
// TRAIN: the training forward under torch.autocast (the exact fp32 kernel of attention_kernels.h stays the fp32 trainer's): dropout on
// the probabilities by the stateless hash mask of att_keep() -- a dropped probability does not reach the value product, the
// normaliser is that of the full softmax, kept ones are scaled by keep_scale = 1 / (1 - p) -- and the log-sum-exp of every query row
// for the backward kernels' softmax rebuild (attention_backward.h).  Same mask bits as the fp32 kernel (same hash of the same
// (packed query row, key) pair); the three-pass products differ from its fp32 ones by ~2^-22 relative.
template <int NT, bool TIMED = false, bool KLO = true, bool KF32 = false, bool TRAIN = false>
__global__ __launch_bounds__(512, 1) void shared_kv_attention_pp_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const _Float16 *__restrict__ kvh, const _Float16 *__restrict__ kvl, const int64_t *__restrict__ kv_start,
    const int64_t *__restrict__ kv_len, float scale, float *__restrict__ out, _Float16 *__restrict__ out_h, _Float16 *__restrict__ out_l,
    int q_pairs, int n_codes, unsigned long long *__restrict__ dbg = nullptr, float *__restrict__ lse = nullptr, unsigned drop_thresh = 0,
    unsigned seed = 0, float keep_scale = 1.f)
{
    using S = AttPP<NT>;
    constexpr int D = S::D, PIECE = S::PIECE, PLANEB = S::PLANEB, CHUNKB = S::CHUNKB, PSL = S::PSL, W = S::W;
    constexpr int EPT = 2, TPR = 8;                // softmax step: 256 threads of a group on 32 rows x 16 keys
    static_assert(NT % 2 == 0 && NT <= 6, "column tiles come in pairs; D <= 768");
    static_assert(!KF32 || KLO, "fp32 keys become (hi, lo) images");
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    char *ring = reinterpret_cast<char *>(att_sm);                                              // [2][W][2][NT][1 KB]
    // block -> (code, tile pair): as attention_dma.h -- round-robin over the 8 XCDs, the pairs of one code consecutive within an XCD
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int b = (jx / q_pairs) * 8 + xcd, qp = jx % q_pairs;
    if (b >= n_codes) return;
    const int nq = (int)q_len[b];
    if (qp * 64 >= nq) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, gt = tid & 255;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int qt = 2 * qp + grp;
    const bool active = qt * 32 < nq;              // wave-uniform
    const int slice = w * 32 * NT;
    char *gbase = ring + S::RING_B + grp * S::GROUP_B;
    float *part = reinterpret_cast<float *>(gbase);
    _Float16 *ph = reinterpret_cast<_Float16 *>(part + S::PART_FLOATS), *pl = ph + 32 * PSL;     // probabilities [32][PSL], hi and lo
    float *alpha_s = reinterpret_cast<float *>(pl + 32 * PSL), *l_s = alpha_s + 32;

    // ---- key chunks by LDS-DMA: this wave copies plane `grp` of column slice `w` (attention_dma.h: lane = (key, piece position))
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        KF32 ? (void *)(reinterpret_cast<const float *>(kvh) + ks * (long)D) : (void *)(((KLO && grp) ? kvl : kvh) + ks * (long)D), 0, (int)0x7fffffff, 0x00020000);
    constexpr int DMA_PER_CHUNK = KLO ? NT : NT / 2;     // (one plane: group g copies the k blocks p = g, g + 2, ... of it)
    const int d_key = lane >> 2, d_col = slice + 8 * ((lane & 3) ^ ((0 - (lane >> 4)) & 3));
    auto stage = [&](int c) __attribute__((always_inline)) {
        const int key = min(16 * c + d_key, kl - 1);                    // past the last key: re-read it (its probability is zero)
        if (KF32) {     // pieces grp NT/2 .. of BOTH planes: the two halves of each 8-column unit (see the head comment)
            const int voff = (key * D + d_col) * 4;
            char *base = ring + (c & 1) * CHUNKB + w * 2 * PLANEB;
#pragma unroll
            for (int pp = 0; pp < NT / 2; ++pp) {
                const int p = grp * (NT / 2) + pp;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(base + p * PIECE), 16, voff, 128 * p, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(base + PLANEB + p * PIECE), 16, voff, 128 * p + 16, 0, 0);
            }
            return;
        }
        const int voff = (key * D + d_col) * 2;
        char *base = ring + (c & 1) * CHUNKB + w * 2 * PLANEB + (KLO ? grp * PLANEB : 0);
#pragma unroll
        for (int p = 0; p < NT; ++p)
            if (KLO || (p & 1) == grp)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(base + p * PIECE), 16, voff, 64 * p, 0, 0);
    };
    // KF32: this wave's pieces of chunk c from fp32 to (hi, lo), in place (its own copies have landed: the caller has waited)
    auto convert = [&](int c) __attribute__((always_inline)) {
        if (!KF32) return;
        const unsigned a0 = (unsigned)(size_t)ring + (unsigned)((c & 1) * CHUNKB + w * 2 * PLANEB + grp * (NT / 2) * PIECE + lane * 16);
        // (all reads first: one LDS round trip for the wave's NT / 2 units, not one each)
        f32x4v fa_[NT / 2], fb_[NT / 2];
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp)
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                         : "=&v"(fa_[pp]), "=&v"(fb_[pp]) : "v"(a0), "i"(pp * PIECE), "i"(PLANEB + pp * PIECE) : "memory");
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp) {
            f32x4v &fa = fa_[pp], &fb = fb_[pp];
            // (in-order returns: the 2 (NT / 2 - 1 - pp) reads behind this unit's may still be out)
            if (NT / 2 - 1 - pp == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa), "+v"(fb) : : "memory");
            else if (NT / 2 - 1 - pp == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa), "+v"(fb) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa), "+v"(fb) : : "memory");
            const float f[8] = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
            half8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hh[e] = (_Float16)f[e];
                ll[e] = (_Float16)(f[e] - (float)hh[e]);
            }
            const u32x4 hv = __builtin_bit_cast(u32x4, hh), lv = __builtin_bit_cast(u32x4, ll);
            // (s_nop: a wide LDS store needs a wait state behind a VALU write to its data registers and two in front of the next one,
            // which hipcc counts for its own stores only: tools/audit_asm_waits.py checks every asm store of the library)
            asm volatile("s_nop 0\n\tds_write_b128 %0, %1 offset:%3\n\tds_write_b128 %0, %2 offset:%4\n\ts_nop 1"
                         : : "v"(a0), "v"(hv), "v"(lv), "i"(pp * PIECE), "i"(PLANEB + pp * PIECE) : "memory");
        }
    };
    const int nchunk = (kl + 15) >> 4;

    // ---- query slice as MFMA A operands of the 16 x 16 x 32 score product (attention_dma.h)
    half8 qh[2][NT], qlo[2][NT];
    {
        const int qr = lane & 15, qg = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float *qrow = q + (qs + min(qt * 32 + 16 * i + qr, nq - 1)) * (long)D + slice + 8 * qg;
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const float4 a = ld4(qrow + 32 * s), c4 = ld4(qrow + 32 * s + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    qh[i][s][e] = (_Float16)v[e];
                    qlo[i][s][e] = (_Float16)(v[e] - (float)qh[i][s][e]);
                }
            }
        }
    }
    asm volatile("" ::: "memory");                 // (the query loads in front of the first DMA: attention_dma.h)
    if (nchunk > 0) stage(0);
    if (nchunk > 1) stage(1);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane LDS byte addresses (32-bit), as in attention_dma.h
    const unsigned ring_a = (unsigned)(size_t)ring + (unsigned)(w * 2 * PLANEB);
    const unsigned k_adr = ring_a + (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ ((0 - (lane >> 2)) & 3)) << 4));
    unsigned v_adr[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int t = lane & 15, g = lane >> 4;
        const int key = 8 * (g >> 1) + 4 * rd + (t >> 2);
        const int pos = (2 * (g & 1) + ((t & 3) >> 1)) ^ ((0 - (key >> 2)) & 3);
        v_adr[rd] = ring_a + (unsigned)(key * 64 + pos * 16 + 8 * (t & 1));
    }
    const unsigned pw_adr = (unsigned)(size_t)part + (unsigned)(((w * 32 + 4 * (lane >> 4)) * 17 + (lane & 15)) * 4);
    // softmax step: thread -> row gt / TPR, keys EPT (gt % TPR) .. + EPT - 1
    const int xrow = gt / TPR, xk0 = (gt % TPR) * EPT;
    const unsigned px_adr = (unsigned)(size_t)part + (unsigned)((xrow * 17 + xk0) * 4);
    const unsigned pp_adr = (unsigned)(size_t)ph + (unsigned)((xrow * PSL + xk0) * 2);

    float m_run = -INFINITY, l_run = 0.f;          // online-softmax state of row gt / TPR, replicated in its TPR threads

    // chunk 0 has landed everywhere (chunk 1, issued behind it, may still be in flight: loads complete in order)
    if (nchunk > 1) {
        if (DMA_PER_CHUNK == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (DMA_PER_CHUNK == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (DMA_PER_CHUNK == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (DMA_PER_CHUNK == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nchunk > 0) convert(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#define PP_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define PP_MFMA32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
    unsigned long long t_s = 0, t_x = 0, t_v = 0, t_w = 0, t_0 = 0;
    if (TIMED) t_0 = __builtin_amdgcn_s_memtime();
    auto phase_s = [&](int c) __attribute__((always_inline)) {
        const unsigned cb = (unsigned)((c & 1) * CHUNKB);
        // ---- S: partial scores of the group's 32 rows x 16 keys over this wave's columns: NT k steps x 2 tiles x 3 passes
        f32x4v sacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) sacc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // operands one k step ahead: two register sets; a step's statement issues the NEXT step's two reads and waits for its own
        // (LDS reads return in order: at most the two just issued may still be out).  The wait and the reads it covers share a
        // statement with the registers as in/out operands, so that hipcc cannot schedule a use (or a copy) in front of it.
        const unsigned ka = k_adr + cb;
        u32x4 kbuf[2][2];
        if (KLO) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(kbuf[0][0]), "=&v"(kbuf[0][1]) : "v"(ka), "i"(PLANEB) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=&v"(kbuf[0][0]) : "v"(ka) : "memory");
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            u32x4 &kh = kbuf[s & 1][0], &kq = kbuf[s & 1][1];
            if (KLO) {
                if (s + 1 < NT)
                    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\ts_waitcnt lgkmcnt(2)"
                                 : "=&v"(kbuf[(s + 1) & 1][0]), "=&v"(kbuf[(s + 1) & 1][1]), "+v"(kh), "+v"(kq)
                                 : "v"(ka), "i"((s + 1) * PIECE), "i"(PLANEB + (s + 1) * PIECE) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh), "+v"(kq) : : "memory");
            } else {
                if (s + 1 < NT)
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\ts_waitcnt lgkmcnt(1)" : "=&v"(kbuf[(s + 1) & 1][0]), "+v"(kh) : "v"(ka), "i"((s + 1) * PIECE) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh) : : "memory");
            }
            const half8 bh = __builtin_bit_cast(half8, kh);
#pragma unroll
            for (int i = 0; i < 2; ++i) sacc[i] = PP_MFMA16(qlo[i][s], bh, sacc[i], 0, 0, 0);
            if (KLO) {
                const half8 bl = __builtin_bit_cast(half8, kq);
#pragma unroll
                for (int i = 0; i < 2; ++i) sacc[i] = PP_MFMA16(qh[i][s], bl, sacc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) sacc[i] = PP_MFMA16(qh[i][s], bh, sacc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            // (scaled by a compiler-visible VALU op: an asm store fed straight from an MFMA accumulator would read it early)
            for (int r = 0; r < 4; ++r) lds_st32(pw_adr + (unsigned)((16 * i + r) * 17 * 4), sacc[i][r] * scale);
    };
    auto phase_x = [&](int c) __attribute__((always_inline)) {
        // ---- X: join the W partials, online softmax (asm reads: a C++ ds_read would be ordered behind the pending LDS-DMA)
        float pv[W][EPT];
        asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\t"
                     "ds_read_b32 %2, %8 offset:%9\n\tds_read_b32 %3, %8 offset:%10\n\t"
                     "ds_read_b32 %4, %8 offset:%11\n\tds_read_b32 %5, %8 offset:%12\n\t"
                     "ds_read_b32 %6, %8 offset:%13\n\tds_read_b32 %7, %8 offset:%14\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(pv[0][0]), "=&v"(pv[0][1]), "=&v"(pv[1][0]), "=&v"(pv[1][1]), "=&v"(pv[2][0]), "=&v"(pv[2][1]),
                       "=&v"(pv[3][0]), "=&v"(pv[3][1])
                     : "v"(px_adr), "i"(32 * 17 * 4), "i"(32 * 17 * 4 + 4), "i"(2 * 32 * 17 * 4), "i"(2 * 32 * 17 * 4 + 4),
                       "i"(3 * 32 * 17 * 4), "i"(3 * 32 * 17 * 4 + 4) : "memory");
        float v[EPT], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) sc += pv[w2][j];                 // (already scaled)
            v[j] = (16 * c + xk0 + j < kl) ? sc : -INFINITY;
            mx = fmaxf(mx, v[j]);
        }
        mx = att_group_max<TPR>(mx);
        const float m_new = fmaxf(m_run, mx);           // finite: every chunk holds at least one valid key
        float psum = 0.f;
        _Float16 hv[EPT], lv[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const float pr = pp_exp(v[j] - m_new);      // exp(-inf) = 0 for masked keys
            const float pk = (TRAIN && drop_thresh && !att_keep(seed, qs + qt * 32 + xrow, 16 * c + xk0 + j, drop_thresh)) ? 0.f : pr;
            hv[j] = (_Float16)pk;
            lv[j] = (_Float16)(pk - (float)hv[j]);
            psum += pr;
        }
        lds_st32u(pp_adr, pack_h2(hv[0], hv[1]));
        lds_st32u(pp_adr + 32 * PSL * 2, pack_h2(lv[0], lv[1]));
        psum = att_group_sum<TPR>(psum);
        const float a = pp_exp(m_run - m_new);          // 0 on the first chunk (m_run = -inf)
        l_run = fmaf(l_run, a, psum);
        m_run = m_new;
        if (gt % TPR == 0) lds_st32((unsigned)(size_t)alpha_s + (unsigned)(xrow * 4), a);
    };
    auto phase_v = [&](int c) __attribute__((always_inline)) {
        const unsigned cb = (unsigned)((c & 1) * CHUNKB);
        // ---- V: out = alpha * out + P . KV: the group's row tile x this wave's NT column tiles, one k step over the 16 keys
        f32x4v aflag;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(aflag) : "v"((unsigned)(size_t)alpha_s + (unsigned)(16 * (lane & 7))) : "memory");
        if (__builtin_amdgcn_ballot_w64(aflag[0] != 1.0f || aflag[1] != 1.0f || aflag[2] != 1.0f || aflag[3] != 1.0f)) {
            const unsigned al_adr = (unsigned)(size_t)alpha_s + (unsigned)(16 * lh);
            f32x4v a0, a1, a2, a3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(al_adr) : "memory");
            const float a16[16] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3]};
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] *= a16[r];
        }
        // the probabilities of the row tile as A operands, and the value operands two column tiles ahead of their MFMAs: three
        // register sets of four transposed reads each, every step's statement issues the reads of tile tt + 2 and waits for tile tt
        // (in-order returns: at most the eight younger reads may still be out)
        u32x4 px[2];
        const unsigned p_adr = (unsigned)(size_t)ph + (unsigned)((li * PSL + 8 * lh) * 2);
        const unsigned va0 = v_adr[0] + cb, va1 = v_adr[1] + cb;
        u32x2 vb[3][4];                                // [set][h0, h1, l0, l1]  (KLO = false: h0, h1 only)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(px[0]), "=&v"(px[1]) : "v"(p_adr), "i"(32 * PSL * 2) : "memory");
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (KLO)
                asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %5 offset:%6\n\t"
                             "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
                             : "=&v"(vb[tt][0]), "=&v"(vb[tt][1]), "=&v"(vb[tt][2]), "=&v"(vb[tt][3])
                             : "v"(va0), "v"(va1), "i"(tt * PIECE), "i"(PLANEB + tt * PIECE) : "memory");
            else
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                             : "=&v"(vb[tt][0]), "=&v"(vb[tt][1]) : "v"(va0), "v"(va1), "i"(tt * PIECE) : "memory");
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            u32x2 (&cur)[4] = vb[tt % 3];
            if (KLO) {
                if (tt + 2 < NT) {
                    u32x2 (&nxt)[4] = vb[(tt + 2) % 3];
                    asm volatile("ds_read_b64_tr_b16 %0, %10 offset:%12\n\tds_read_b64_tr_b16 %1, %11 offset:%12\n\t"
                                 "ds_read_b64_tr_b16 %2, %10 offset:%13\n\tds_read_b64_tr_b16 %3, %11 offset:%13\n\ts_waitcnt lgkmcnt(8)"
                                 : "=&v"(nxt[0]), "=&v"(nxt[1]), "=&v"(nxt[2]), "=&v"(nxt[3]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]),
                                   "+v"(px[0]), "+v"(px[1])
                                 : "v"(va0), "v"(va1), "i"((tt + 2) * PIECE), "i"(PLANEB + (tt + 2) * PIECE) : "memory");
                } else if (tt + 1 < NT) {
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(px[0]), "+v"(px[1]) : : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(px[0]), "+v"(px[1]) : : "memory");
                }
            } else {
                if (tt + 2 < NT) {
                    u32x2 (&nxt)[4] = vb[(tt + 2) % 3];
                    asm volatile("ds_read_b64_tr_b16 %0, %6 offset:%8\n\tds_read_b64_tr_b16 %1, %7 offset:%8\n\ts_waitcnt lgkmcnt(4)"
                                 : "=&v"(nxt[0]), "=&v"(nxt[1]), "+v"(cur[0]), "+v"(cur[1]), "+v"(px[0]), "+v"(px[1])
                                 : "v"(va0), "v"(va1), "i"((tt + 2) * PIECE) : "memory");
                } else if (tt + 1 < NT) {
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(px[0]), "+v"(px[1]) : : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(px[0]), "+v"(px[1]) : : "memory");
                }
            }
            const half8 vh = __builtin_bit_cast(half8, __builtin_shufflevector(cur[0], cur[1], 0, 1, 2, 3));
            const half8 ph0 = __builtin_bit_cast(half8, px[0]), pl0 = __builtin_bit_cast(half8, px[1]);
            acc[tt] = PP_MFMA32(pl0, vh, acc[tt], 0, 0, 0);
            if (KLO) {
                const half8 vl = __builtin_bit_cast(half8, __builtin_shufflevector(cur[2], cur[3], 0, 1, 2, 3));
                acc[tt] = PP_MFMA32(ph0, vl, acc[tt], 0, 0, 0);
            }
            acc[tt] = PP_MFMA32(ph0, vh, acc[tt], 0, 0, 0);
        }
    };
    // a slot ends with one block-wide barrier; at the end of slots 3c'+2 this wave's share of chunk c'+1's copy must have landed
    // (copies = the chunk whose copy this slot's end waits for, or -1: with fp32 keys the wave then turns its pieces into images)
    auto slot_end = [&](int copies) __attribute__((always_inline)) {
        if (copies >= 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (KF32 && copies < nchunk) {
                convert(copies);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // the ring: at slot 3c'+1 (c' >= 1) the buffer of chunk c'-1 is free -- its last reader, g1's V(c'-1), ran in slot 3c'
    auto refill = [&](int cp) __attribute__((always_inline)) {
        if (cp >= 1 && cp + 1 < nchunk) stage(cp + 1);
    };
    auto timed = [&](unsigned long long &acc_t, auto &&fn) __attribute__((always_inline)) {
        if (TIMED) {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            fn();
            acc_t += __builtin_amdgcn_s_memtime() - a;
        } else {
            fn();
        }
    };
    if (grp == 0) {
        for (int c = 0; c < nchunk; ++c) {
            if (active) timed(t_s, [&]() __attribute__((always_inline)) { phase_s(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });
            refill(c);
            if (active) timed(t_x, [&]() __attribute__((always_inline)) { phase_x(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });
            if (active) timed(t_v, [&]() __attribute__((always_inline)) { phase_v(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(c + 1); });
        }
        timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });                           // (slot 3 nchunk: g1's last V)
    } else {
        timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });                           // (slot 0: g0's first S)
        for (int c = 0; c < nchunk; ++c) {
            refill(c);
            if (active) timed(t_s, [&]() __attribute__((always_inline)) { phase_s(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });
            if (active) timed(t_x, [&]() __attribute__((always_inline)) { phase_x(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(c + 1); });
            if (active) timed(t_v, [&]() __attribute__((always_inline)) { phase_v(c); });
            timed(t_w, [&]() __attribute__((always_inline)) { slot_end(-1); });
        }
    }
#undef PP_MFMA16
#undef PP_MFMA32
    if (TIMED && dbg && lane == 0) {
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
        o[0] = t_s; o[1] = t_x; o[2] = t_v; o[3] = t_w; o[4] = __builtin_amdgcn_s_memtime() - t_0; o[5] = (unsigned long long)nchunk; o[6] = active;
        o[7] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (gt % TPR == 0) {
        l_s[xrow] = l_run;
        // log-sum-exp of the scaled scores; -inf for an empty key set
        if (TRAIN && lse && active && qt * 32 + xrow < nq) lse[qs + qt * 32 + xrow] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
    }
    __syncthreads();
    // the row tiles leave through LDS (ring and score tiles are free now), each group through its own staging area
    float l16[16];
    const float unkeep = TRAIN ? 1.f / keep_scale : 1.f;         // (the store divides by l16: kept probabilities count 1 / (1 - p))
#pragma unroll
    for (int r = 0; r < 16; ++r) l16[r] = TRAIN ? l_s[(r & 3) + 8 * (r >> 2) + 4 * lh] * unkeep : l_s[(r & 3) + 8 * (r >> 2) + 4 * lh];
    __syncthreads();
    att_store_tile<W, NT, S::HALVES>(att_sm + grp * S::STAGE_FLOATS, acc, l16, active ? nq - qt * 32 : 0, qs + qt * 32, out, out_h, out_l, slice, li, lh, gt);
}

// Built and measured in round 4, not kept: the same kernel with TWO phases per chunk -- every wave of a group runs the softmax of all
// 32 rows itself (lane = (row, 8 keys): joins the four partial tiles, meets the other half through v_permlane32_swap) and ends with
// P[row][8 keys] in registers, which is the A operand of the 32 x 32 x 16 value MFMA: no probability tile in LDS, no barrier between
// softmax and values, two barriers per chunk instead of three (the ring then has one slot, not two, for a copy to arrive; with and
// without a one-dword-per-line L2 touch of the chunk after next).  Correct (same tests), 1.68 ms (1.75 with the touch) against
// 1.65 ms for the three-phase form above and 1.65-1.78 for attention_dma.h's on the `full` workload's graph-side launch: three
// different phase structures land within 5 % of each other.
