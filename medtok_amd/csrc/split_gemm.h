// split_gemm.h -- fp32-accurate dense products on the fp16 matrix pipe: C = A . B^T (+ bias) with both operands carried as a
// pair of fp16 images (hi, lo), x = hi + lo.  Included by medtok_vq.hip; gfx950 only.
//
// Where it is used.  The cross-attention layers of get_shared_info (vector_quantization_soft_one_new.py:17-88,133-142) are, around
// their attention core, four dense D x D products per packed query row and layer (in_proj of the queries, the fold of W_k into
// them, W_v on the attended rows, out_proj).  On the fp32 matrix pipe (157 TFLOP/s) they were 39 % of a forward at BASELINE
// sizes; v_mfma_f32_32x32x16_f16 runs 16x faster, and three of them reproduce an fp32 product to ~2^-22 relative:
//     a b  ~  a_hi b_hi + a_hi b_lo + a_lo b_hi          (a_lo b_lo ~ 2^-22 |a b| is dropped)
// with a_hi = fp16(a), a_lo = fp16(a - a_hi): products of two fp16 are exact in the fp32 accumulator, so the only errors are the
// dropped term, the 2^-22-relative (or 2^-25 absolute: fp16 subnormal step) residual of the split, and the fp32 accumulation
// itself -- the same order as a plain fp32 GEMM's round-off (tests/test_gpu_split_gemm.py measures <= 2e-6 of the row scale;
// the bar for these tolerance items is 1e-5).  Weights are prescaled by an exact power of two chosen per matrix (so that small
// weights use the fp16 normal range; undone in the epilogue); activations are taken as they are: |x| must stay below 65504,
// beyond that the hi part is inf and the outputs are NaN -- loud, not silently wrong.
//
// Kernel.  The operand pipeline is the fp16 filter's (filter_f16.h): block = 8 waves, tile 256 output features x 256 rows, wave
// tile 128 x 64 = 4 x 2 MFMA tiles; operands by buffer-addressed LDS-DMA into a 4-stage ring of 32-deep k blocks, XOR-swizzled
// through the SOURCE address, counted vmcnt + raw s_barrier.  The k loop runs three passes over K -- (B_hi, A_hi), (B_lo, A_hi),
// (B_hi, A_lo) -- i.e. it is one fp16 GEMM over 3 K whose stage picks its two source images by pass (four buffer descriptors,
// exact sizes: rows past M and features past the last group read as zeros).  Weights are the MFMA A operand, so a lane ends up
// with runs of four consecutive output features of ONE row: the epilogue adds the bias and stores fp32 (float4) and/or the
// (hi, lo) fp16 images of the result (8 bytes each) that the next product reads -- no separate conversion pass between the
// products of a layer.
// Grouped form: G independent problems (the heads) that differ by a column offset into A, a row offset into B and a column
// offset into C -- the per-head fold and the per-head W_v product are one launch each.
#pragma once

constexpr int G_BM = 256, G_BN = 256, G_BK = 32;              // output features x rows x k (fp16 elements) per stage
constexpr int G_THREADS = 512;
constexpr int G_ROWB = G_BK * 2;                              // bytes per staged tile row (64)
constexpr int G_TILEB = G_BM * G_ROWB;                        // 16 KB per operand tile
constexpr int G_STAGEB = 2 * G_TILEB;                         // weights + activations = 32 KB
constexpr int G_RING = 4;
constexpr size_t G_LDS_BYTES = (size_t)G_RING * G_STAGEB;     // 128 KB -> 1 block (8 waves) / CU
constexpr int G_WN = 4, G_MT = 4, G_NT = 2;                   // row-side waves; 32-feature / 32-row MFMA tiles per wave

struct SplitGemmArgs {
    const _Float16 *ah, *al;        // activations [M, lda]: hi and lo images
    const _Float16 *bh, *bl;        // weights [G * b_group_rows, ldb], prescaled by 1 / unscale
    const float *bias;              // [G * n_g] or nullptr
    float *c;                       // fp32 result [M, ldc] or nullptr
    _Float16 *ch, *cl;              // (hi, lo) images of the result [M, ldch] or nullptr
    long M;
    long a_bytes, b_bytes;          // size of ONE image (descriptor bound: reads past it return zeros)
    int lda, ldb, ldc, ldch;        // row strides in elements (lda, ldb, ldch multiples of 8)
    int n_g, k_g, groups;           // per group: output features (multiple of 4), depth (multiple of 32); number of groups
    int a_group_cols;               // group g reads A columns [g * a_group_cols, + k_g)
    int b_group_rows;               // ... B rows [g * b_group_rows, + n_g) and writes C columns [g * n_g, + n_g)
    float unscale;                  // applied to the accumulators (undoes the weights' power-of-two prescale)
    int row_tiles, ftiles;          // M / 256 rounded up; feature tiles per group
};

__global__ __launch_bounds__(G_THREADS, 2) void split_gemm_kernel(const SplitGemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / G_WN, wn = wave % G_WN;
    const int li = lane & 31, lh = lane >> 5;
    // PERSISTENT blocks: block b works through the tile ids b, b + gridDim.x, ... (gridDim.x a multiple of 8, so all of them run on
    // XCD b % 8).  id -> (row tile, group, feature tile): the ids that share one row tile (all feature tiles of all groups) are
    // consecutive WITHIN an XCD, so the activation tile is fetched into that L2 once.  The operand ring never drains between
    // tiles: while one tile's accumulators are stored, the first stages of the block's next tile are already in flight -- a tile
    // costs neither a pipeline fill nor an exposed epilogue (K = 192 per-head products: 18 stages per tile, where fill + epilogue
    // were 40 % of a block's life).
    const int per_row = p.ftiles * p.groups;
    const int n_ids = (p.row_tiles + 7) / 8 * 8 * per_row;
    const int stride = (int)gridDim.x;
    const int nkb = p.k_g / G_BK;
    const int nstage = 3 * nkb;
    auto row_tile_of = [&](int id) { return (long)((id >> 3) / per_row) * 8 + (id & 7); };
    auto next_tile = [&](int id) {                       // the block's next id with a real row tile (ids of the padded last 8 are skipped)
        for (id += stride; id < n_ids && row_tile_of(id) >= p.row_tiles; id += stride) {}
        return id;
    };
    int first = (int)blockIdx.x;
    if (row_tile_of(first) >= p.row_tiles) first = next_tile(first);
    if (first >= n_ids) return;

    // ---- staging (as filter_f16_kernel): wave w copies tile rows [32w, 32w+32) of both operands, 16 rows per instruction
    const int s_r = lane >> 2, s_c = lane & 3;
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = wave * 32 + q * 16 + s_r;
        const int c = s_c ^ ((r >> 2) & 3);
        a_off[q] = (unsigned)(r * p.lda + c * 8) * 2u;
        b_off[q] = (unsigned)(r * p.ldb + c * 8) * 2u;
    }
    // descriptors start at the tile's first row / the group's first column; num_records = what is left of the image from there
    auto rsrc = [](const _Float16 *img, long base, long bytes) {
        const long left = bytes - base;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(img) + base), 0,
                                                 (int)(left < 0 ? 0 : (left > 0x7fffffffL ? 0x7fffffffL : left)), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t ah_rs, al_rs, bh_rs, bl_rs;
    auto bind_tile = [&](int id) __attribute__((always_inline)) {
        const int gf = (id >> 3) % per_row, grp = gf / p.ftiles, ft = gf % p.ftiles;
        const long a_base = (row_tile_of(id) * G_BN * p.lda + (long)grp * p.a_group_cols) * 2;
        const long b_base = ((long)(grp * p.b_group_rows + ft * G_BM) * p.ldb) * 2;
        ah_rs = rsrc(p.ah, a_base, p.a_bytes); al_rs = rsrc(p.al, a_base, p.a_bytes);
        bh_rs = rsrc(p.bh, b_base, p.b_bytes); bl_rs = rsrc(p.bl, b_base, p.b_bytes);
    };
    bind_tile(first);
    const int wave_lds = wave_s * 32 * G_ROWB;
    int iid = first, ikb = 0, ipass = 0, islot = 0;      // the stage to issue next: tile id, k block, pass; ring position
    bool idone = false;
    // Issues the next stage into ring slot islot % 4 and advances -- across tile boundaries.  Past the block's last stage it
    // re-issues that stage into the slot that already holds it (same bytes) so the loop has no branch around its DMA and the
    // counted vmcnt waits see the same number of instructions in every iteration.
    auto stage = [&]() __attribute__((always_inline)) {
        char *base = gsm + (islot & (G_RING - 1)) * G_STAGEB + wave_lds;
        const int uk = __builtin_amdgcn_readfirstlane(ikb * G_BK * 2);
        // pass 0: (B_hi, A_hi), 1: (B_lo, A_hi), 2: (B_hi, A_lo) -- wave-uniform selects of SGPR descriptors
        const __amdgpu_buffer_rsrc_t wrs = ipass == 1 ? bl_rs : bh_rs;
        const __amdgpu_buffer_rsrc_t xrs = ipass == 2 ? al_rs : ah_rs;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(base + q * 16 * G_ROWB), 16,
                                                     (int)b_off[q], uk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(base + G_TILEB + q * 16 * G_ROWB), 16,
                                                     (int)a_off[q], uk, 0, 0);
        }
        if (idone) return;
        if (ikb + 1 < nkb) { ++ikb; ++islot; return; }
        if (ipass < 2) { ikb = 0; ++ipass; ++islot; return; }
        const int nid = next_tile(iid);                  // this was the tile's last stage
        if (nid >= n_ids) { idone = true; return; }      // (cursor and slot stay on the block's last stage)
        iid = nid; ikb = 0; ipass = 0; ++islot;
        bind_tile(iid);
    };

    f32x16 acc[G_MT][G_NT];
    int a_adr[2], b_adr[2];                 // fragment addresses within a stage (a = weights / MFMA A operand, b = activations)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int ia = wm * (32 * G_MT) + li, ib = wn * (32 * G_NT) + li;
        a_adr[tt] = ia * G_ROWB + (((2 * tt + lh) ^ ((ia >> 2) & 3)) << 4);
        b_adr[tt] = G_TILEB + ib * G_ROWB + (((2 * tt + lh) ^ ((ib >> 2) & 3)) << 4);
    }
    half8 fa[G_MT], fbA[G_NT], fbB[G_NT];
    auto read_a = [&](int m, int slot, int tt) __attribute__((always_inline)) {
        fa[m] = *reinterpret_cast<const half8 *>(gsm + slot * G_STAGEB + a_adr[tt] + m * 32 * G_ROWB);
    };
    auto read_b = [&](half8 (&fb)[G_NT], int slot, int tt) __attribute__((always_inline)) {
#pragma unroll
        for (int nn = 0; nn < G_NT; ++nn) fb[nn] = *reinterpret_cast<const half8 *>(gsm + slot * G_STAGEB + b_adr[tt] + nn * 32 * G_ROWB);
    };
    auto step = [&](const half8 (&fb_cur)[G_NT], half8 (&fb_nxt)[G_NT], int slot, int tt) __attribute__((always_inline)) {
        read_b(fb_nxt, slot, tt);
#pragma unroll
        for (int m = 0; m < G_MT; ++m) {
#pragma unroll
            for (int nn = 0; nn < G_NT; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb_cur[nn], acc[m][nn], 0, 0, 0);
            read_a(m, slot, tt);
        }
    };

    constexpr int LGKM0 = 0xC07F;           // s_waitcnt lgkmcnt(0) only
    stage(); stage(); stage();              // the first three stages (re-issues when the block has fewer)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(fbA, 0, 0);
#pragma unroll
    for (int m = 0; m < G_MT; ++m) read_a(m, 0, 0);
    const bool late = wave_s >= 4;          // the two waves of a SIMD issue their DMA at different points of the stage (filter_f16.h)
    if (late) __builtin_amdgcn_s_setprio(3);
    int gs = 0;                             // stages computed so far = ring position of the stage in the registers
    for (int cid = first; cid < n_ids; cid = next_tile(cid)) {
#pragma unroll
        for (int m = 0; m < G_MT; ++m)
#pragma unroll
            for (int nn = 0; nn < G_NT; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
        for (int s = 0; s < nstage; ++s, ++gs) {
            if (late && gs > 0) stage();        // waves 4-7: stage gs+2 (slot gs-2, free since the barrier of iteration gs-1)
            step(fbA, fbB, gs & (G_RING - 1), 1);
            __builtin_amdgcn_sched_group_barrier(0x100, G_NT, 0);
#pragma unroll
            for (int i = 0; i < G_MT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, G_NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_s_waitcnt(LGKM0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!late) stage();                 // waves 0-3: stage gs+3 (slot gs-1: everyone is past reading it)
            step(fbB, fbA, (gs + 1) & (G_RING - 1), 0);     // (past a tile's last stage these are the NEXT tile's first operands)
            __builtin_amdgcn_sched_group_barrier(0x100, G_NT, 0);
#pragma unroll
            for (int i = 0; i < G_MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, G_NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }

        // ---- epilogue of tile cid (the next tile's first stages are in flight meanwhile; stores only make the counted vmcnt waits
        // more conservative): lane (li, lh) holds, for row wn*64 + nn*32 + li, the features wm*128 + m*32 + 8 g + 4 lh + {0..3}
        const int gf = (cid >> 3) % per_row, grp = gf / p.ftiles, f0 = (gf % p.ftiles) * G_BM;
        const long row0 = row_tile_of(cid) * G_BN;
        const int cbase = grp * p.n_g;
#pragma unroll
        for (int nn = 0; nn < G_NT; ++nn) {
            const long row = row0 + wn * (32 * G_NT) + nn * 32 + li;
            if (row >= p.M) continue;
#pragma unroll
            for (int m = 0; m < G_MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = f0 + wm * (32 * G_MT) + m * 32 + 8 * g + 4 * lh;
                    if (f >= p.n_g) continue;
                    float4 v = make_float4(acc[m][nn][4 * g] * p.unscale, acc[m][nn][4 * g + 1] * p.unscale, acc[m][nn][4 * g + 2] * p.unscale,
                                           acc[m][nn][4 * g + 3] * p.unscale);
                    if (p.bias) {
                        const float4 b4 = ld4(p.bias + cbase + f);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
                    if (p.c) st4(p.c + row * p.ldc + cbase + f, v);
                    if (p.ch) {
                        half4v hi, lo;
                        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
                        lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
                        lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
                        *reinterpret_cast<half4v *>(p.ch + row * p.ldch + cbase + f) = hi;
                        *reinterpret_cast<half4v *>(p.cl + row * p.ldch + cbase + f) = lo;
                    }
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the re-issues of the last stage may still be in flight
}

// fp32 rows [n, d] -> (hi, lo) fp16 images [n, dp] (dp >= d, multiple of 8; columns past d are zero), scaled by `scale` first
// (an exact power of two; 1 for activations).  One thread per 8 elements.  With seg_len: the rows come in segments of seg_rows
// (a [B, L, d] batch) of which only the first seg_len[b] are converted -- padding tokens are never read as keys.
__global__ __launch_bounds__(256) void split_half_kernel(const float *__restrict__ src, long n, int d, long src_stride, int dp, float scale,
                                                         _Float16 *__restrict__ hi, _Float16 *__restrict__ lo,
                                                         const int64_t *__restrict__ seg_len, int seg_rows)
{
    const int cpr = dp / 8;
    const long total = n * cpr;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / cpr;
        const int c = (int)(t - r * cpr) * 8;
        if (seg_len && (r % seg_rows) >= seg_len[r / seg_rows]) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (c < d) { const float4 a = ld4(src + r * src_stride + c); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
        if (c + 4 < d) { const float4 b = ld4(src + r * src_stride + c + 4); v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * scale;
            h[j] = (_Float16)x;
            l[j] = (_Float16)(x - (float)h[j]);
        }
        *reinterpret_cast<half8 *>(hi + r * dp + c) = h;
        *reinterpret_cast<half8 *>(lo + r * dp + c) = l;
    }
}
