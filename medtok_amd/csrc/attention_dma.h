// attention_dma.h -- the ragged attention core for wide inference batches: 64 query rows per block, keys delivered by LDS-DMA from
// (hi, lo) fp16 images.  Included by medtok_vq.hip after attention_kernels.h; gfx950 only.
//
//     out[r, :] = softmax_j( scale * <q[r, :], kv[j, :]> ) . kv        (vector_quantization_soft_one_new.py:17-88,133-142, folded)
//
// The 32-row kernels of attention_kernels.h pay, per 32-key chunk, for things that are proportional to the KEYS -- fetching the
// chunk, converting it, parking it in LDS, reading it back as MFMA operands -- and with the products on the fp16 pipe those, not
// the matrix work, are the chunk's time (measured at D = 768: 6.0 us per chunk of which ~1 us MFMA).  This kernel halves them
// per query row and removes most of the rest:
//   * block = 64 query rows (two 32-row tiles) of one code; every K / V operand read from LDS feeds both tiles;
//   * the keys arrive as (hi, lo) fp16 IMAGES made once per forward by medtok_split_half_f32 (a key row is used by every query
//     tile of its code and by both layers) and are copied by LDS-DMA from L2 into a two-deep ring of 16-key chunks -- no staging
//     registers (that is what makes room for the second query tile), no conversion VALU, no ds_write pass.  A wave copies and reads
//     only its own D / W column slice, so no barrier guards the ring; the copy of chunk c + 1 is in flight under all of chunk c
//     (counted vmcnt);
//   * the LDS image of a slice is [k block of 32 columns][16 keys][64 B], its 16-byte pieces XOR-swizzled by f(key >> 2), f(x) =
//     -x & 3, through the SOURCE address (the DMA writes lane-linearly): conflict-free both for the score product's b128 operand
//     reads (one key, 8 consecutive columns; ds_read_b128 is serviced in the non-contiguous lane groups {0-3, 12-15, 20-27}, ...:
//     with f(x) = x two keys of a group shared a bank slot, SQ_LDS_BANK_CONFLICT was 35 % of the LDS cycles) and for the value product's transposed reads (ds_read_b64_tr_b16: a 16-lane group fetches a
//     [4 keys][16 columns] tile and every lane receives 4 consecutive KEYS of one column -- the B operand of a product that
//     contracts over keys, from keys stored row-major; semantics pinned by tools/probes/tr_probe.hip);
//   * scores on v_mfma_f32_16x16x32_f16 (a 16-key chunk is one tile wide), values on v_mfma_f32_32x32x16_f16 (16 keys = one k
//     step), each as three passes over (hi, lo) pairs (split_gemm.h: ~2^-22 relative); fp32 online softmax as in the other kernels.
// Every LDS STORE inside the loop is an asm statement: hipcc orders its own ds_writes behind ALL pending LDS-DMA
// (s_waitcnt vmcnt(0)), which would drain the prefetched chunk at the first store of every iteration.
#pragma once

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_st32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st32u(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}

// W waves x NT 32-column k blocks per wave (D = 32 W NT); MT 32-row query tiles per block; RING chunk buffers.
//   <8, 3, 2, 2> (D = 768): 64 rows per block, one block per CU, the next chunk's copy under this chunk's work;
//   <4, 6, 1, 1> (D = 768): 32 rows per block, 60 KB of LDS: TWO blocks per CU -- a block's phases (copy wait | scores | softmax |
//   values) are separated by barriers and cannot overlap each other, but they overlap the other block's: the matrix pipe works
//   on one block while the other is in its softmax or waits for its keys.
template <int W, int NT, int MT, int RING>
struct AttDma {
    static constexpr int D = 32 * W * NT, ROWS = 32 * MT;
    static constexpr int PIECE = 1024;                    // bytes one DMA instruction writes: 16 keys x 64 B
    static constexpr int PLANEB = NT * PIECE;             // one plane (hi or lo) of a wave's slice of a chunk
    static constexpr int CHUNKB = W * 2 * PLANEB;         // a 16-key chunk: W slices x (hi, lo)
    static constexpr int PSL = 24;                        // halves per probability row: 16 keys + 8 (conflict-free b128 reads)
    static constexpr int PART_FLOATS = W * ROWS * 17;     // [W][ROWS][17] partial scores [row][key]
    static constexpr int HALVES = (32 * (D + 4) * 4 <= RING * CHUNKB + PART_FLOATS * 4) ? 1 : 2;      // column passes of the output staging
    static constexpr size_t LDS_BYTES = (size_t)RING * CHUNKB + (size_t)PART_FLOATS * 4 + (size_t)2 * ROWS * PSL * 2 + 2 * ROWS * 4 + 64;
};

template <int W, int NT, int MT, int RING>
// (<4, 8, 1, 2>, D = 1024: 128 registers of query operands + 128 of accumulators per lane -- one wave per SIMD, arch + acc registers)
__global__ __launch_bounds__(64 * W, (MT == 1 && W * NT <= 24 ? 2 : 1) * W / 4) void shared_kv_attention_dma_kernel(
    const float *__restrict__ q, const int64_t *__restrict__ q_start, const int64_t *__restrict__ q_len,
    const _Float16 *__restrict__ kvh, const _Float16 *__restrict__ kvl, const int64_t *__restrict__ kv_start,
    const int64_t *__restrict__ kv_len, float scale, float *__restrict__ out, _Float16 *__restrict__ out_h, _Float16 *__restrict__ out_l,
    int q_tiles, int n_codes)
{
    using S = AttDma<W, NT, MT, RING>;
    constexpr int D = S::D, ROWS = S::ROWS, PIECE = S::PIECE, PLANEB = S::PLANEB, CHUNKB = S::CHUNKB, PSL = S::PSL;
    constexpr int THREADS = 64 * W;
    constexpr int EPT = ROWS * 16 / THREADS;       // score elements per thread in the softmax step (ROWS x 16 keys): 2 or 4
    constexpr int TPR = 16 / EPT;                  // threads per score row: 8 or 4
    static_assert((W == 4 || W == 8) && (MT == 1 || MT == 2) && (RING == 1 || RING == 2) && (EPT == 2 || EPT == 4), "supported shapes");
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    char *ring = reinterpret_cast<char *>(att_sm);                                              // [RING][W][2][NT][1 KB]
    float *part = reinterpret_cast<float *>(ring + RING * CHUNKB);
    _Float16 *ph = reinterpret_cast<_Float16 *>(part + S::PART_FLOATS), *pl = ph + ROWS * PSL;  // probabilities [ROWS][PSL], hi and lo
    float *alpha_s = reinterpret_cast<float *>(pl + ROWS * PSL), *l_s = alpha_s + ROWS;
    // block -> (code, query tile): consecutive block ids go round-robin to the 8 XCDs; the tiles of one code are made consecutive
    // WITHIN an XCD (ids x, x + 8, ...), so that they run at about the same time on CUs that share an L2: the keys of a code with
    // several query tiles come from HBM once
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int b = (jx / q_tiles) * 8 + xcd, qt = jx % q_tiles;
    if (b >= n_codes) return;
    const int nq = (int)q_len[b];
    if (qt * ROWS >= nq) return;
    const long qs = q_start[b], ks = kv_start[b];
    const int kl = (int)kv_len[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int slice = wave * 32 * NT;

    // ---- key chunks by LDS-DMA: lane (key = lane / 4, piece position c = lane % 4) of instruction (plane, k block p) writes LDS
    // bytes [16 lane, + 16) of that 1 KB piece and reads the 16 bytes at columns slice + 32 p + 8 (c ^ (key / 4 & 3)) of its key
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void *)(kvh + ks * (long)D), 0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void *)(kvl + ks * (long)D), 0, (int)0x7fffffff, 0x00020000);
    const int d_key = lane >> 2, d_col = slice + 8 * ((lane & 3) ^ ((0 - (lane >> 4)) & 3));        // piece position c holds columns 8 (c ^ f(key / 4)), f(x) = -x & 3
    auto stage = [&](int c) __attribute__((always_inline)) {
        const int key = min(16 * c + d_key, kl - 1);                    // past the last key: re-read it (its probability is zero)
        const int voff = (key * D + d_col) * 2;
        char *base = ring + (c & (RING - 1)) * CHUNKB + wave_s * 2 * PLANEB;
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void *)(base + p * PIECE), 16, voff, 64 * p, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, (__attribute__((address_space(3))) void *)(base + PLANEB + p * PIECE), 16, voff, 64 * p, 0, 0);
        }
    };
    const int nchunk = (kl + 15) >> 4;

    // ---- query slice as MFMA A operands of the 16 x 16 x 32 score product: tile i = rows 16 i .. 16 i + 15 of the block, k step s =
    // columns slice + 32 s .. + 31; lane (row = lane % 16, k group = lane / 16) holds 8 consecutive columns
    half8 qh[2 * MT][NT], qlo[2 * MT][NT];
    {
        const int qr = lane & 15, qg = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const float *qrow = q + (qs + min(qt * ROWS + 16 * i + qr, nq - 1)) * (long)D + slice + 8 * qg;
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
    // keep the query conversion in front of the first DMA: an ordinary load that is still outstanding beside LDS-DMA makes hipcc
    // wait vmcnt(0) at its use, i.e. drain the ring
    asm volatile("" ::: "memory");
    if (nchunk > 0) stage(0);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    // per-lane LDS byte addresses (32-bit)
    const unsigned ring_a = (unsigned)(size_t)ring + (unsigned)(wave * 2 * PLANEB);
    // score product B operand: key = lane % 16, k group = lane / 16: 16 bytes at piece position (kg ^ (key / 4 & 3))
    const unsigned k_adr = ring_a + (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ ((0 - (lane >> 2)) & 3)) << 4));
    // value product B operand by transposed reads: lane t = lane % 16 of group g = lane / 16 supplies the address of the 4 columns
    // 16 (g & 1) + 4 (t & 3) .. + 3 of key 8 (g >> 1) + 4 rd + (t >> 2); their 16-byte piece 2 (g & 1) + ((t & 3) >> 1) sits at the
    // position swizzled by (key >> 2) & 3 = (2 (g >> 1) + rd) & 3
    unsigned v_adr[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int t = lane & 15, g = lane >> 4;
        const int key = 8 * (g >> 1) + 4 * rd + (t >> 2);
        const int pos = (2 * (g & 1) + ((t & 3) >> 1)) ^ ((0 - (key >> 2)) & 3);
        v_adr[rd] = ring_a + (unsigned)(key * 64 + pos * 16 + 8 * (t & 1));
    }
    // partial-score stores: 16 x 16 C tile of v_mfma_f32_16x16x32: lane holds key = lane % 16, rows 4 (lane / 16) + r
    const unsigned pw_adr = (unsigned)(size_t)part + (unsigned)(((wave * ROWS + 4 * (lane >> 4)) * 17 + (lane & 15)) * 4);
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    float m_run = -INFINITY, l_run = 0.f;          // online-softmax state of row tid / TPR, replicated in its TPR threads
    for (int c = 0; c < nchunk; ++c) {
        if (RING == 2) {
            // chunk c + 1 into the other half of the ring (last read by this wave's value product of chunk c - 1: a wave's LDS
            // operations complete in order and nobody else touches its slice); then wait for chunk c only
            if (c + 1 < nchunk) {
                stage(c + 1);
                if (NT == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (NT == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // one buffer: this chunk's copy was issued when the previous chunk's last read had been waited for (below); what hides
            // its latency is the CU's other block
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned cb = (unsigned)((c & (RING - 1)) * CHUNKB);

        // ---- partial scores of the 64 rows x 16 keys over this wave's columns: NT k steps x 4 tiles x 3 passes
        // (one accumulator per tile: the three passes of a tile are four MFMAs apart, behind those of the other three tiles)
        f32x4v sacc[2 * MT];
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) sacc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            // (one k step's operands at a time: a second register set for the next step's does not fit beside the 192 registers of
            // query operands and output accumulators; the SIMD's other wave covers the read latency)
            u32x4 kh, kq;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(kh), "=&v"(kq) : "v"(k_adr + cb), "i"(s * PIECE), "i"(PLANEB + s * PIECE) : "memory");
            const half8 bh = __builtin_bit_cast(half8, kh), bl = __builtin_bit_cast(half8, kq);
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qlo[i][s], bh, sacc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[i][s], bl, sacc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[i][s], bh, sacc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i)
#pragma unroll
            // (the scale is applied HERE, by a compiler-visible VALU op: hipcc inserts the MFMA -> reader wait states for its own
            // instructions only -- an asm store fed straight from the accumulator reads it before the matrix pipe has written it)
            for (int r = 0; r < 4; ++r) lds_st32(pw_adr + (unsigned)((16 * i + r) * 17 * 4), sacc[i][r] * scale);
        lds_barrier();

        // ---- join the W partials, online softmax: thread -> row tid / TPR, keys EPT (tid % TPR) .. + EPT - 1
        {
            const int row = tid / TPR, kq0 = (tid % TPR) * EPT;
            float v[EPT], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < W; ++w2) sc += part[(w2 * ROWS + row) * 17 + kq0 + j];       // (already scaled)
                v[j] = (16 * c + kq0 + j < kl) ? sc : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
            mx = att_group_max<TPR>(mx);
            const float m_new = fmaxf(m_run, mx);           // finite: every chunk holds at least one valid key
            float psum = 0.f;
            _Float16 hv[EPT], lv[EPT];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float pr = expf(v[j] - m_new);        // exp(-inf) = 0 for masked keys
                hv[j] = (_Float16)pr;
                lv[j] = (_Float16)(pr - (float)hv[j]);
                psum += pr;
            }
            const unsigned p_adr = (unsigned)(size_t)ph + (unsigned)((row * PSL + kq0) * 2);
#pragma unroll
            for (int j = 0; j < EPT; j += 2) {
                lds_st32u(p_adr + 2 * j, pack_h2(hv[j], hv[j + 1]));
                lds_st32u(p_adr + ROWS * PSL * 2 + 2 * j, pack_h2(lv[j], lv[j + 1]));
            }
            psum = att_group_sum<TPR>(psum);
            const float a = expf(m_run - m_new);            // 0 on the first chunk (m_run = -inf)
            l_run = fmaf(l_run, a, psum);
            m_run = m_new;
            if (tid % TPR == 0) lds_st32((unsigned)(size_t)alpha_s + (unsigned)(row * 4), a);
        }
        lds_barrier();

        // ---- out = alpha * out + P . KV: the block's row tiles x this wave's NT column tiles, one k step over the chunk's 16 keys
        {
            // (asm reads: the first C++ ds_read behind the barrier would be ordered behind the pending LDS-DMA too)
            // A row's rescale factor is exactly 1 unless its running maximum moved in this chunk; once the maxima have settled (a few
            // chunks into a code's keys) nothing moves and the MT x NT x 16 multiplications per lane are skipped -- multiplying by
            // 1.0f is the identity, so the result is bit-identical either way (wave-uniform branch).  Lane % (ROWS / 4) checks rows
            // 4 (lane % (ROWS / 4)) .. + 3.
            f32x4v aflag;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(aflag) : "v"((unsigned)(size_t)alpha_s + (unsigned)(16 * (lane & (ROWS / 4 - 1)))) : "memory");
            if (__builtin_amdgcn_ballot_w64(aflag[0] != 1.0f || aflag[1] != 1.0f || aflag[2] != 1.0f || aflag[3] != 1.0f)) {
                const unsigned al_adr = (unsigned)(size_t)alpha_s + (unsigned)(16 * lh);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4v a0, a1, a2, a3;
                    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"
                                 "ds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                                 : "v"(al_adr), "i"(128 * m), "i"(128 * m + 32), "i"(128 * m + 64), "i"(128 * m + 96) : "memory");
                    const float a16[16] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3]};
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][t][r] *= a16[r];
                }
            }
            // the probabilities of the row tiles as A operands (waited for together with the first value operands)
            u32x4 px[2 * MT];
            {
                const unsigned p_adr = (unsigned)(size_t)ph + (unsigned)((li * PSL + 8 * lh) * 2);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(px[2 * m]), "=&v"(px[2 * m + 1])
                                 : "v"(p_adr), "i"(32 * m * PSL * 2), "i"(ROWS * PSL * 2 + 32 * m * PSL * 2) : "memory");
            }
            const unsigned va0 = v_adr[0] + cb, va1 = v_adr[1] + cb;
            // Two independent accumulators alternate, so that an MFMA never waits for the one issued just before it: the two row
            // tiles on one column tile (MT = 2), or two column tiles of the one row tile (MT = 1).  One group's value operands at
            // a time: a second register set does not fit beside the 192 registers of query operands and output accumulators.
            constexpr int TSTEP = MT == 2 ? 1 : 2;
            static_assert(NT % TSTEP == 0, "column tiles come in pairs when there is one row tile");
#pragma unroll
            for (int t = 0; t < NT; t += TSTEP) {
                u32x2 h0[TSTEP], h1[TSTEP], l0[TSTEP], l1[TSTEP];
#pragma unroll
                for (int u = 0; u < TSTEP; ++u)
                    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %5 offset:%6\n\t"
                                 "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
                                 : "=&v"(h0[u]), "=&v"(h1[u]), "=&v"(l0[u]), "=&v"(l1[u])
                                 : "v"(va0), "v"(va1), "i"((t + u) * PIECE), "i"(PLANEB + (t + u) * PIECE) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                half8 vh[TSTEP], vl[TSTEP];
#pragma unroll
                for (int u = 0; u < TSTEP; ++u) {
                    vh[u] = __builtin_bit_cast(half8, __builtin_shufflevector(h0[u], h1[u], 0, 1, 2, 3));
                    vl[u] = __builtin_bit_cast(half8, __builtin_shufflevector(l0[u], l1[u], 0, 1, 2, 3));
                }
                if constexpr (MT == 2) {
                    const half8 ph0 = __builtin_bit_cast(half8, px[0]), pl0 = __builtin_bit_cast(half8, px[1]);
                    const half8 ph1 = __builtin_bit_cast(half8, px[2]), pl1 = __builtin_bit_cast(half8, px[3]);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl0, vh[0], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl1, vh[0], acc[1][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vl[0], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph1, vl[0], acc[1][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vh[0], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph1, vh[0], acc[1][t], 0, 0, 0);
                } else {
                    const half8 ph0 = __builtin_bit_cast(half8, px[0]), pl0 = __builtin_bit_cast(half8, px[1]);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl0, vh[0], acc[0][t], 0, 0, 0);
                    acc[0][t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl0, vh[TSTEP - 1], acc[0][t + 1], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vl[0], acc[0][t], 0, 0, 0);
                    acc[0][t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vl[TSTEP - 1], acc[0][t + 1], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vh[0], acc[0][t], 0, 0, 0);
                    acc[0][t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph0, vh[TSTEP - 1], acc[0][t + 1], 0, 0, 0);
                }
            }
            if (RING == 1 && c + 1 < nchunk) stage(c + 1);      // (this wave's reads of the buffer have all been waited for)
        }
        // (no barrier here: the ring slices are wave-private; the score / probability tiles are next written behind the two
        // barriers of the next chunk)
    }
    if (tid % TPR == 0) l_s[tid / TPR] = l_run;
    __syncthreads();
    // the row tiles leave through LDS (the ring and the score tiles are free now): full 16-byte pieces per row, fp32 and/or the
    // (hi, lo) images the next dense product reads.  The row sums move to registers first: the staging tile may overwrite them.
    float l16[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) l16[m][r] = l_s[32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
        att_store_tile<W, NT, S::HALVES>(att_sm, acc[m], l16[m], nq - qt * ROWS - 32 * m, qs + qt * ROWS + 32 * m, out, out_h, out_l, slice, li, lh, tid);
}
