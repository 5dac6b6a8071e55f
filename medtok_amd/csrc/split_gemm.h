// split_gemm.h -- fp32-accurate dense products on the fp16 matrix pipe: C = A . B^T (+ bias) with both operands carried as a
// pair of fp16 images (hi, lo), x = hi + lo.  Included by medtok_vq.hip; gfx950 only.
//
// Where it is used.  The cross-attention layers of get_shared_info (vector_quantization_soft_one_new.py:17-88,133-142) are, around
// their attention core, four dense D x D products per packed query row and layer (in_proj of the queries, the fold of W_k into
// them, W_v on the attended rows, out_proj); proj_text / proj_graph of the specific searches (:190,192) are two more.  On the fp32
// matrix pipe (157 TFLOP/s) they were 39 % of a forward at BASELINE sizes; v_mfma_f32_32x32x16_f16 runs 16x faster, and three of
// them reproduce an fp32 product to ~2^-22 relative:
//     a b  ~  a_hi b_hi + a_hi b_lo + a_lo b_hi          (a_lo b_lo ~ 2^-22 |a b| is dropped)
// with a_hi = fp16(a), a_lo = fp16(a - a_hi): products of two fp16 are exact in the fp32 accumulator, so the only errors are the
// dropped term, the 2^-22-relative (or 2^-25 absolute: fp16 subnormal step) residual of the split, and the fp32 accumulation
// itself -- the same order as a plain fp32 GEMM's round-off (tests/test_gpu_split_gemm.py measures <= 2e-6 of the row scale;
// the bar for these tolerance items is 1e-5).  Weights are prescaled by an exact power of two chosen per matrix (so that small
// weights use the fp16 normal range; undone in the epilogue); activations are taken as they are: |x| must stay below 65504,
// beyond that the hi part is inf and the outputs are NaN -- loud, not silently wrong.
//
// Kernel.  Block = 8 waves (2 x 4), tile = 64 MT output features x 256 rows (MT = 4, or 3 where that wastes less: the per-head
// W_v product has 192 features per head), wave tile 32 MT x 64 = MT x 2 MFMA tiles; operands by buffer-addressed LDS-DMA,
// XOR-swizzled through the SOURCE address (filter_f16.h), raw s_barrier.  A stage is one 32-deep k block of all FOUR images
// (W_hi | W_lo | X_hi | X_lo, 64 KB at MT = 4; two stages), and the wave forms the three products from it:
//     per 16-deep k step:   W_hi . X_hi   |   W_hi . X_lo   |   W_lo . X_hi        (2 MT MFMAs each)
// with the weight fragments rolling through one register set (W_hi stays for two groups, W_lo replaces it fragment by fragment
// behind its last use, the next step's W_hi replaces that) and the activation fragments in three (X_hi of this step, X_lo, X_hi
// of the next).  Round 3's first form ran the three products as three passes over K -- an fp16 GEMM over 3 K that copied six
// 16 KB tiles and read 36 fragments per 48 MFMAs; as in the fp16 filter the copies through the CU's L1 path then take
// as long as the MFMAs they feed.  The six tiles are four different ones: this form copies 64 KB and reads 24 fragments per 48
// MFMAs -- two thirds of the L1 and LDS traffic -- with ONE block-wide barrier per 48 MFMAs instead of three.  The barrier sits in
// front of a stage's last group, whose operands are in registers by then: behind it the other slot is known to hold the next
// stage (every wave waited for its own copies) and this stage's slot is free for the stage after next.  Alone on a CU the main
// loop keeps the matrix pipe 87 % busy (35.4 us per 256 x 256 x 768 tile); with all 256 CUs at the sagged clock 58 us.
// Weights are the MFMA A operand, so a lane ends up with runs of four consecutive output features of ONE row; the epilogue
// transposes every 32 x 32 tile through the wave's own 4 KB of LDS and stores whole rows: fp32 and/or the (hi, lo) fp16 images of
// the result that the next product reads -- no separate conversion pass between the products of a layer.
// Grouped form: G independent problems (the heads) that differ by a column offset into A, a row offset into B and a column
// offset into C -- the per-head fold and the per-head W_v product are one launch each.
#pragma once
#include <type_traits>

constexpr int G_BN = 256, G_BK = 32;                          // rows x k (fp16 elements) per stage; features per tile: 64 MT
constexpr int G_THREADS = 512;
constexpr int G_ROWB = G_BK * 2;                              // bytes per staged tile row (64)
constexpr int G_XTILEB = G_BN * G_ROWB;                       // 16 KB per activation tile
constexpr int G_WN = 4, G_NT = 2;                             // row-side waves; 32-row MFMA tiles per wave
typedef float g_f4 __attribute__((ext_vector_type(4)));

template <int MT>
struct GemmShape {
    static constexpr int BM = 64 * MT;                        // 2 feature-side waves x MT x 32
    static constexpr int WTILEB = BM * G_ROWB;
    static constexpr int STG = 2 * WTILEB + 2 * G_XTILEB;     // W_hi | W_lo | X_hi | X_lo
    static constexpr size_t LDS_BYTES = (size_t)2 * STG + 8 * 4096;      // two stages + 4 KB of epilogue staging per wave (MT = 4: all 160 KB)
};

// The power of two that brings a tensor's largest magnitude into [2^11, 2^12): the (hi, lo) fp16 pair of a prescaled value then
// carries 22 significant bits of every element down to amax 2^-26 (smaller ones fall into the fp16 subnormals -- they move a dot
// product by < 1e-7 of its scale).  Weights get theirs from a host read once per weight version; the operands of TRAINING products
// (activations, upstream gradients: anything from 1e-9 to 1e+5 under a GradScaler) take it from a device-side |x|_max, no host
// read.  1 for amax = 0 / non-finite.  Exact: exponent arithmetic only.
__device__ __forceinline__ float pow2_prescale(float amax)
{
    const unsigned bits = __float_as_uint(amax) & 0x7fffffffu;
    if (bits == 0u || bits >= 0x7f800000u) return 1.0f;
    int e = (int)(bits >> 23) - 127;                 // floor(log2 amax) for normal amax (subnormal: -127)
    int se = 11 - e;                                 // scale = 2^se
    se = se > 126 ? 126 : (se < -126 ? -126 : se);
    return __uint_as_float((unsigned)(se + 127) << 23);
}

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
    const float *amax_a, *amax_b;   // (training products) the operands were prescaled by pow2_prescale(*amax): undone here too; or nullptr
    int row_tiles, ftiles;          // M / 256 rounded up; feature tiles (of 64 MT) per group
    int per_xcd;                    // 0: tile ids by row tile (below); > 0: the DENSE order for few row tiles, this many ids per XCD
};

template <int MT, int FRONT, bool PER_M, int VMEM>
__device__ __forceinline__ void gemm_weave()            // scheduling hints for one group of MT x G_NT MFMAs
{
    if constexpr (FRONT > 0) __builtin_amdgcn_sched_group_barrier(0x100, FRONT, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, G_NT, 0);
        if constexpr (PER_M) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (VMEM > 0) __builtin_amdgcn_sched_group_barrier(0x020, VMEM, 0);
    }
}

// ONE (the autocast form): ONE pass over the hi images alone -- a plain half-precision product with fp32 accumulation, the precision
// class of the reference's projections under torch.autocast (train_MedTok.py:212,394: nn.Linear / nn.MultiheadAttention compute in
// fp16 or bf16 there).  BF: the images are bf16 (v_mfma_f32_32x32x16_bf16, same rate; fp32's exponent range: no prescale).  Same
// ring, same tiles, same epilogue; a stage copies two tiles instead of four and holds two groups of MFMAs instead of six.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ f32x16 gemm_mfma(half8 a, half8 b, f32x16 c)
{
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// K64 (with ONE): a stage is 64 k deep instead of 32 -- the slots that hold the lo images in the three-pass form hold the second 32-k
// block of the hi images, and the stage is four groups of MFMAs (block 0 step 0 / 1, block 1 step 0 / 1).  A copy has one stage of
// matrix work to arrive (it is issued behind a stage's barrier and waited for in front of the next one): 16 MFMAs ~ 0.5 us in the
// 32-deep ONE form, which an HBM round trip does not fit into -- the half-precision products of a training step ran 1.5 - 3.3 us per
// stage for 0.5 us of matrix work; 32 MFMAs ~ 1 us here (the three-pass form has 48).  k_g % 64 == 0.
template <int MT, bool ONE = false, bool BF = false, bool K64 = false>
__global__ __launch_bounds__(G_THREADS, 2) void split_gemm_kernel(const SplitGemmArgs p)
{
    static_assert(!K64 || ONE, "the 64-deep stage is a form of the one-pass product");
    using S = GemmShape<MT>;
    constexpr int BM = S::BM, WTILEB = S::WTILEB, STG = S::STG;
    constexpr int XOFF = 2 * WTILEB;                     // activation images within a stage
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / G_WN, wn = wave % G_WN;
    const int li = lane & 31, lh = lane >> 5;
    // PERSISTENT blocks: block b works through the tile ids b, b + gridDim.x, ... (gridDim.x a multiple of 8, so all of them run on
    // XCD b % 8).  id -> (row tile, group, feature tile): the ids that share one row tile (all feature tiles of all groups) are
    // consecutive WITHIN an XCD, so the activation tile is fetched into that L2 once.  The operand ring never drains between
    // tiles: while one tile's accumulators are stored, the first stage of the block's next tile is already in LDS and the second
    // in flight.
    // DENSE order (per_xcd > 0; the host picks it where the row tiles do not fill the 8 XCDs: a weight-gradient product has 3 or 12 of
    // them, and by row tile its 252 tiles ran on the 96 CUs of three XCDs -- 406 us for the flops its forward does in 224): XCD x
    // takes the tiles t = x per_xcd .. of the order (group, row tile, feature tile), so the tiles of one group -- which share its
    // operand columns -- still meet in one L2, and every XCD has the same number of tiles.
    const int per_row = p.ftiles * p.groups;
    const int per_grp = p.row_tiles * p.ftiles, n_tiles = per_grp * p.groups;
    const bool dense = p.per_xcd > 0;
    const int n_ids = dense ? p.per_xcd * 8 : (p.row_tiles + 7) / 8 * 8 * per_row;
    const int stride = (int)gridDim.x;
    const int nkb = p.k_g / (K64 ? 2 * G_BK : G_BK);
    auto dense_t = [&](int id) { return (id & 7) * p.per_xcd + (id >> 3); };
    auto row_tile_of = [&](int id) {                     // (>= row_tiles: not a tile)
        if (dense) { const int t = dense_t(id); return t < n_tiles ? (long)((t % per_grp) / p.ftiles) : (long)p.row_tiles; }
        return (long)((id >> 3) / per_row) * 8 + (id & 7);
    };
    auto group_feature_of = [&](int id, int &grp, int &ft) {
        if (dense) { const int t = dense_t(id); grp = t / per_grp; ft = (t % per_grp) % p.ftiles; return; }
        const int gf = (id >> 3) % per_row;
        grp = gf / p.ftiles; ft = gf % p.ftiles;
    };
    auto next_tile = [&](int id) {                       // the block's next id with a real row tile (ids of the padded last 8 are skipped)
        for (id += stride; id < n_ids && row_tile_of(id) >= p.row_tiles; id += stride) {}
        return id;
    };
    int first = (int)blockIdx.x;
    if (first < n_ids && row_tile_of(first) >= p.row_tiles) first = next_tile(first);
    if (first >= n_ids) return;

    // ---- staging: one LDS-DMA instruction copies 16 tile rows x 64 B; wave w copies row blocks w and w + 8 of every tile (the
    // weight tile of MT = 3 has 12: its second block exists for waves 0-3 only)
    const int s_r = lane >> 2, s_c = lane & 3;
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wave + 8 * q) * 16 + s_r;
        const int c = s_c ^ ((r >> 2) & 3);
        a_off[q] = (unsigned)(r * p.lda + c * 8) * 2u;
        b_off[q] = (unsigned)(r * p.ldb + c * 8) * 2u;
    }
    const bool w_first = wave_s < BM / 16, w_second = wave_s + 8 < BM / 16;     // (MT = 1: the weight tile is four row blocks, waves 0-3)
    // descriptors start at the tile's first row / the group's first column; num_records = what is left of the image from there
    auto rsrc = [](const _Float16 *img, long base, long bytes) {
        const long left = bytes - base;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(img) + base), 0,
                                                 (int)(left < 0 ? 0 : (left > 0x7fffffffL ? 0x7fffffffL : left)), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t ah_rs, al_rs, bh_rs, bl_rs;
    auto bind_tile = [&](int id) __attribute__((always_inline)) {
        int grp, ft;
        group_feature_of(id, grp, ft);
        const long a_base = (row_tile_of(id) * G_BN * p.lda + (long)grp * p.a_group_cols) * 2;
        const long b_base = ((long)(grp * p.b_group_rows + ft * BM) * p.ldb) * 2;
        ah_rs = rsrc(p.ah, a_base, p.a_bytes); bh_rs = rsrc(p.bh, b_base, p.b_bytes);
        if constexpr (K64) {                             // the "lo" slots: the same images, one 32-k block (64 bytes) further
            al_rs = rsrc(p.ah, a_base + G_ROWB, p.a_bytes); bl_rs = rsrc(p.bh, b_base + G_ROWB, p.b_bytes);
        } else {
            al_rs = rsrc(p.al, a_base, p.a_bytes); bl_rs = rsrc(p.bl, b_base, p.b_bytes);
        }
    };
    bind_tile(first);
    const int wave_lds = wave_s * 16 * G_ROWB;
    int iid = first, ikb = 0, islot = 0;                 // the stage to issue next: tile id, k block; slot parity
    bool more = true;                                    // false once the block's last stage has been issued
    auto stage = [&]() __attribute__((always_inline)) {
        char *base = gsm + (islot & 1) * STG + wave_lds;
        const int uk = __builtin_amdgcn_readfirstlane(ikb * (K64 ? 2 : 1) * G_BK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char *dst = base + q * 8 * 16 * G_ROWB;
            if (q == 0 ? w_first : w_second) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(bh_rs, (__attribute__((address_space(3))) void *)dst, 16, (int)b_off[q], uk, 0, 0);
                if constexpr (!ONE || K64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(bl_rs, (__attribute__((address_space(3))) void *)(dst + WTILEB), 16, (int)b_off[q], uk, 0, 0);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ah_rs, (__attribute__((address_space(3))) void *)(dst + XOFF), 16, (int)a_off[q], uk, 0, 0);
            if constexpr (!ONE || K64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(al_rs, (__attribute__((address_space(3))) void *)(dst + XOFF + G_XTILEB), 16, (int)a_off[q], uk, 0, 0);
        }
        ++islot;
        if (ikb + 1 < nkb) { ++ikb; return; }
        const int nid = next_tile(iid);                  // this was the tile's last stage
        if (nid >= n_ids) { more = false; return; }
        iid = nid; ikb = 0;
        bind_tile(iid);
    };

    const unsigned stg_a = (unsigned)(size_t)(gsm + 2 * STG) + (unsigned)(wave * 4096);      // this wave's 4 KB of epilogue staging
    // what the epilogue needs of the arguments, in registers from here on: a kernel argument re-read inside the epilogue is a scalar
    // load, and its s_waitcnt lgkmcnt(0) would also wait for the LDS read-backs in flight
    float e_unscale = p.unscale;
    if (p.amax_a) e_unscale *= 1.0f / pow2_prescale(p.amax_a[0]);
    if (p.amax_b) e_unscale *= 1.0f / pow2_prescale(p.amax_b[0]);
    long e_M = p.M;
    int e_ldc = p.ldc, e_ldch = p.ldch, e_ng = p.n_g;
    unsigned long e_c = (unsigned long)p.c, e_ch = (unsigned long)p.ch, e_cl = (unsigned long)p.cl, e_bias = (unsigned long)p.bias;   // (as integers:
    // a pointer that went through an asm operand would come back as a generic one, i.e. flat_store)
    asm volatile("" : "+v"(e_unscale), "+s"(e_M), "+s"(e_ldc), "+s"(e_ldch), "+s"(e_ng), "+s"(e_c), "+s"(e_ch), "+s"(e_cl), "+s"(e_bias));
    typedef __attribute__((address_space(1))) g_f4 gl_f4;
    typedef __attribute__((address_space(1))) half4v gl_h4;
    f32x16 acc[MT][G_NT];
    int w_adr[2], x_adr[2];                 // fragment addresses of the hi images within a stage, per k step (lo: + the tile size)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int ia = wm * (32 * MT) + li, ib = wn * (32 * G_NT) + li;
        w_adr[tt] = ia * G_ROWB + (((2 * tt + lh) ^ ((ia >> 2) & 3)) << 4);
        x_adr[tt] = XOFF + ib * G_ROWB + (((2 * tt + lh) ^ ((ib >> 2) & 3)) << 4);
    }
    half8 fw[MT], xa[G_NT], xb[G_NT], xl[G_NT];
    auto read_w = [&](int m, int slot, int tt, int lo) __attribute__((always_inline)) {
        fw[m] = *reinterpret_cast<const half8 *>(gsm + slot * STG + lo * WTILEB + w_adr[tt] + m * 32 * G_ROWB);
    };
    auto read_x = [&](half8 (&fx)[G_NT], int slot, int tt, int lo) __attribute__((always_inline)) {
#pragma unroll
        for (int nn = 0; nn < G_NT; ++nn) fx[nn] = *reinterpret_cast<const half8 *>(gsm + slot * STG + lo * G_XTILEB + x_adr[tt] + nn * 32 * G_ROWB);
    };
    // one group of MT x G_NT MFMAs: fw x fx; behind the MFMAs of a weight fragment that fragment is reloaded (w_next: 0 = keep)
    auto group = [&](const half8 (&fx)[G_NT], int w_next, int slot, int tt, int lo) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int nn = 0; nn < G_NT; ++nn)
                acc[m][nn] = gemm_mfma<BF>(fw[m], fx[nn], acc[m][nn]);
            if (w_next) read_w(m, slot, tt, lo);
        }
    };

    constexpr int LGKM0 = 0xC07F;           // s_waitcnt lgkmcnt(0) only
    stage();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more) stage();
    read_x(xa, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) read_w(m, 0, 0, 0);
    // the two waves of a SIMD issue their copies at different points of the stage (filter_f16.h): waves 0-3 right behind the barrier,
    // waves 4-7 two groups later
    const bool late = wave_s >= 4;
    if (late) __builtin_amdgcn_s_setprio(3);
    int gs = 0;                             // stages computed so far: its parity is the slot of the stage in the registers
    bool owe = false;                       // a late wave's copy of the stage after next is due
    for (int cid = first; cid < n_ids; cid = next_tile(cid)) {
        int grp, ft_c;
        group_feature_of(cid, grp, ft_c);
        const int f0 = ft_c * BM;
        const int cbase = grp * p.n_g;
        const int ec = lane & 7, er = lane >> 3;         // epilogue, after the transposition: 4-feature chunk and row % 8 of the lane
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < G_NT; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
        g_f4 bzs[MT];
        // One stage.  The tile's LAST stage is a second copy of the code: it also fetches the lane's bias quads -- at its top, waited
        // for behind its barrier, where vmcnt is 0 anyway.  Loads and stores share the counter and return out of order with each
        // other, so a load waited for inside the epilogue costs s_waitcnt vmcnt(0): the copies in flight, and between the stores
        // every store before it.  (Quads past the group's features: the last one is read instead -- no select on a loaded value,
        // which would be a wait; their stores are masked.)
        auto body = [&](auto last_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            const int sl = gs & 1, nx = sl ^ 1;
            if constexpr (LAST) {
                if (e_bias) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int f = f0 + wm * (32 * MT) + m * 32 + 4 * ec;
                        bzs[m] = *(const gl_f4 *)(e_bias + (unsigned long)(cbase + (f < e_ng ? f : e_ng - 4)) * 4);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) bzs[m] = (g_f4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if constexpr (ONE && K64) {
                // one pass over a 64-deep stage: block 0 (the hi slots) step 0, step 1, block 1 (the lo slots) step 0, step 1; the barrier
                // in front of the last group, whose operands are in registers by then (fw rolls: behind a group the next one's weights)
                read_x(xb, sl, 1, 0);
                group(xa, 1, sl, 1, 0);             gemm_weave<MT, G_NT, true, 0>();                // block 0 step 0; fw <- W(block 0, step 1)
                read_x(xa, sl, 0, 1);
                group(xb, 1, sl, 0, 1);             gemm_weave<MT, G_NT, true, 0>();                // block 0 step 1; fw <- W(block 1, step 0)
                if (owe) { stage(); owe = false; }
                read_x(xb, sl, 1, 1);
                group(xa, 1, sl, 1, 1);             gemm_weave<MT, G_NT, true, 2>();                // block 1 step 0; fw <- W(block 1, step 1)
                __builtin_amdgcn_s_waitcnt(LGKM0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (LAST) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(bzs[m]));
                }
                if (more) { if (late) owe = true; else stage(); }
                read_x(xa, nx, 0, 0);
                group(xb, 1, nx, 0, 0);             gemm_weave<MT, G_NT, true, 2>();                // block 1 step 1; fw <- W(next stage, block 0, step 0)
                ++gs;
                return;
            }
            if constexpr (ONE) {
                // one pass: W_hi . X_hi of k step 0 (operands in registers), the barrier in front of k step 1's group as below
                read_x(xb, sl, 1, 0);
                group(xa, 1, sl, 1, 0);             gemm_weave<MT, G_NT, true, 0>();                // fw <- W_hi(step 1)
                if (owe) { stage(); owe = false; }
                __builtin_amdgcn_s_waitcnt(LGKM0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (LAST) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(bzs[m]));
                }
                if (more) { if (late) owe = true; else stage(); }
                read_x(xa, nx, 0, 0);
                group(xb, 1, nx, 0, 0);             gemm_weave<MT, G_NT, true, 2>();                // fw <- W_hi(next stage, step 0)
                ++gs;
                return;
            }
            // k step 0
            read_x(xl, sl, 0, 1);
            group(xa, 0, 0, 0, 0);                  gemm_weave<MT, G_NT, false, 0>();
            group(xl, 1, sl, 0, 1);                 gemm_weave<MT, 0, true, 0>();                   // fw <- W_lo(step 0)
            if (owe) { stage(); owe = false; }
            read_x(xb, sl, 1, 0);
            group(xa, 1, sl, 1, 0);                 gemm_weave<MT, G_NT, true, 2>();                // fw <- W_hi(step 1)
            // k step 1
            read_x(xl, sl, 1, 1);
            group(xb, 0, 0, 0, 0);                  gemm_weave<MT, G_NT, false, 0>();
            group(xl, 1, sl, 1, 1);                 gemm_weave<MT, 0, true, 0>();                   // fw <- W_lo(step 1)
            __builtin_amdgcn_s_waitcnt(LGKM0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (LAST) {
#pragma unroll
                for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(bzs[m]));                       // (the bias quads have arrived)
            }
            if (more) { if (late) owe = true; else stage(); }
            read_x(xa, nx, 0, 0);
            group(xb, 1, nx, 0, 0);                 gemm_weave<MT, G_NT, true, 2>();                // fw <- W_hi(next stage, step 0)
            ++gs;
        };
        for (int s = 0; s + 1 < nkb; ++s) body(std::false_type{});
        body(std::true_type{});

        // ---- epilogue of tile cid.  A lane holds runs of four features of ONE row per MFMA tile: stored from there, a wave's store
        // instruction is 32 rows x 32 bytes (16 for the fp16 images), every piece its own request to the L2.  Each 32 x 32 tile
        // goes through the wave's own 4 KB of LDS behind the ring instead ([row][8 x 16 B], the pieces XOR-swizzled by the row) and
        // leaves as whole rows: 8 lanes x 16 B = one 128-byte line (fp32), 8 lanes x 8 B = 64 B per image.  Tile i + 1 is written
        // while the read-back of tile i is in flight (a wave's LDS instructions execute in order).  asm LDS instructions: hipcc
        // would put s_waitcnt vmcnt(0) -- the stores just issued -- in front of its own ds_write.
        const long row0 = row_tile_of(cid) * G_BN + wn * (32 * G_NT);
        const unsigned e_wr = stg_a + (unsigned)(li * 128), e_rd = stg_a + (unsigned)(er * 128 + ((ec ^ (er & 7)) << 4));
        int mv = (e_ng - f0 - wm * (32 * MT) + 31) / 32;                                  // this wave's MFMA tiles that hold features of the group
        mv = mv < 0 ? 0 : (mv > MT ? MT : mv);
        const int n_it = __builtin_amdgcn_readfirstlane(mv * G_NT);
        // (the multiplications are compiler-visible VALU work between the MFMA and the asm stores: hazard wait states.  The four
        // stores of a tile are ONE statement from sixteen different registers with two idle cycles behind them: a VALU write to the
        // data registers of a 128-bit LDS store needs two wait states on gfx950, which hipcc inserts for its own stores only.  The
        // read-back, the next tile's stores and the wait are one statement too: hipcc may copy an asm output anywhere behind the
        // statement that produces it -- with the wait in a later statement it copied the registers before the data had arrived.)
        auto scaled = [&](int i, g_f4 (&v)[4]) __attribute__((always_inline)) {
            const int m = i / G_NT, nn = i % G_NT;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                v[g] = (g_f4){acc[m][nn][4 * g] * e_unscale, acc[m][nn][4 * g + 1] * e_unscale, acc[m][nn][4 * g + 2] * e_unscale,
                              acc[m][nn][4 * g + 3] * e_unscale};
        };
        const unsigned sw = (unsigned)(li & 7);
        const unsigned wa0 = e_wr + (((0 + lh) ^ sw) << 4), wa1 = e_wr + (((2 + lh) ^ sw) << 4), wa2 = e_wr + (((4 + lh) ^ sw) << 4),
                       wa3 = e_wr + (((6 + lh) ^ sw) << 4);
        if (n_it > 0) {
            g_f4 v[4];
            scaled(0, v);
            asm volatile("ds_write_b128 %0, %4\n\tds_write_b128 %1, %5\n\tds_write_b128 %2, %6\n\tds_write_b128 %3, %7\n\ts_nop 1"
                         ::"v"(wa0), "v"(wa1), "v"(wa2), "v"(wa3), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
        }
#pragma unroll
        for (int i = 0; i < MT * G_NT; ++i) {
            if (i >= n_it) break;
            const int m = i / G_NT, nn = i % G_NT;
            g_f4 r0, r1, r2, r3;
            if (i + 1 < MT * G_NT && i + 1 < n_it) {
                g_f4 v[4];
                scaled(i + 1 < MT * G_NT ? i + 1 : i, v);
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                             "ds_read_b128 %3, %4 offset:3072\n\t"
                             "ds_write_b128 %5, %9\n\tds_write_b128 %6, %10\n\tds_write_b128 %7, %11\n\tds_write_b128 %8, %12\n\t"
                             "s_waitcnt lgkmcnt(4)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "v"(e_rd), "v"(wa0), "v"(wa1), "v"(wa2), "v"(wa3), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                             "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(e_rd) : "memory");
            }
            const int f = f0 + wm * (32 * MT) + m * 32 + 4 * ec;
            if (f >= e_ng) continue;
            const g_f4 rr[4] = {r0 + bzs[m], r1 + bzs[m], r2 + bzs[m], r3 + bzs[m]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long row = row0 + nn * 32 + 8 * j + er;
                if (row >= e_M) continue;
                if (e_c) *(gl_f4 *)(e_c + (unsigned long)(row * e_ldc + cbase + f) * 4) = rr[j];
                if (e_ch) {
                    half4v hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)rr[j][e]; lo[e] = (_Float16)(rr[j][e] - (float)hi[e]); }
                    *(gl_h4 *)(e_ch + (unsigned long)(row * e_ldch + cbase + f) * 2) = hi;
                    *(gl_h4 *)(e_cl + (unsigned long)(row * e_ldch + cbase + f) * 2) = lo;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// fp32 rows [n, d] -> (hi, lo) fp16 images [n, dp] (dp >= d, multiple of 8; columns past d are zero), scaled by `scale` first
// (an exact power of two; 1 for activations).  One thread per 8 elements, U of them in flight per thread (all loads of an
// iteration are issued before the first conversion).  With seg_len: the rows come in segments of seg_rows (a [B, L, d] batch) of
// which only the first seg_len[b] are converted -- padding tokens are never read as keys.  (Its 62 registers let one of its waves
// per SIMD run beside a resident split_gemm_kernel block: issued behind a dense product, the pass starts under it.)
template <int U>
__global__ __launch_bounds__(256) void split_half_kernel(const float *__restrict__ src, long n, int d, long src_stride, int dp, float scale,
                                                         _Float16 *__restrict__ hi, _Float16 *__restrict__ lo,
                                                         const int64_t *__restrict__ seg_len, int seg_rows, const float *__restrict__ amax = nullptr)
{
    if (amax) scale *= pow2_prescale(amax[0]);
    const int cpr = dp / 8;
    const long total = n * cpr, step = (long)gridDim.x * 256;
    for (long t0 = (long)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += step * U) {
        g_f4 va[U], vb[U];
        long r[U];
        int c[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long t = t0 + u * step;
            ok[u] = t < total;
            r[u] = ok[u] ? t / cpr : 0;
            c[u] = (int)(t - r[u] * cpr) * 8;
            if (seg_len && ok[u]) ok[u] = (r[u] % seg_rows) < seg_len[r[u] / seg_rows];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            va[u] = vb[u] = (g_f4){0.f, 0.f, 0.f, 0.f};
            if (ok[u] && c[u] < d) va[u] = *reinterpret_cast<const g_f4 *>(src + r[u] * src_stride + c[u]);
            if (ok[u] && c[u] + 4 < d) vb[u] = *reinterpret_cast<const g_f4 *>(src + r[u] * src_stride + c[u] + 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            half8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = (j < 4 ? va[u][j] : vb[u][j - 4]) * scale;
                h[j] = (_Float16)x;
                l[j] = (_Float16)(x - (float)h[j]);
            }
            *reinterpret_cast<half8 *>(hi + r[u] * dp + c[u]) = h;
            *reinterpret_cast<half8 *>(lo + r[u] * dp + c[u]) = l;
        }
    }
}

// |x|_max of a buffer into amax[0] (zeroed by the caller): non-negative floats order like their bit patterns, so the block maxima
// meet in an integer atomicMax.  NaN / inf propagate as "non-finite" (pow2_prescale then returns 1).
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, long count, float *__restrict__ amax)
{
    unsigned m = 0u;
    const long step = (long)gridDim.x * 256 * 4;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    for (; i + 3 * step + 4 <= count; i += 4 * step) {                 // four independent 16-byte loads in flight per thread
        g_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const g_f4 *>(x + i + u * step);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[u][j]) & 0x7fffffffu);
    }
    for (; i < count; i += step) {
        if (i + 4 <= count) {
            const g_f4 v = *reinterpret_cast<const g_f4 *>(x + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7fffffffu);
        } else {
            for (long j = i; j < count; ++j) m = max(m, __float_as_uint(x[j]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    // one atomic per BLOCK (thousands of wave-level atomics on one address serialise in the L2: 40 us of a 48 us launch)
    __shared__ unsigned s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(reinterpret_cast<unsigned *>(amax), m);
    }
}

// The (hi, lo) images of the TRANSPOSE: src [n, d] fp32 -> hi, lo [d, np] (np >= n, a multiple of 8; columns n .. np - 1 zero) --
// the operands of a weight-gradient product dW = dY^T X contract over the ROWS, so both need their row index contiguous.  64 x 64
// tiles through LDS: rows are read as float4 (coalesced), columns leave as 32-byte pieces of 16 consecutive rows.
__global__ __launch_bounds__(256) void split_half_t_kernel(const float *__restrict__ src, long n, int d, long src_stride, long np, long group_cols,
                                                           float scale, _Float16 *__restrict__ hi, _Float16 *__restrict__ lo, const float *__restrict__ amax)
{
    // group_cols (a multiple of 64 dividing np): the images are written as np / group_cols GROUPS of [d, group_cols] stacked along
    // the rows -- the layout a grouped product reads as "group g = k chunk g", i.e. a split-K weight gradient in ONE launch
    __shared__ float tile[64][65];
    if (amax) scale *= pow2_prescale(amax[0]);
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
        g_f4 v = (g_f4){0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n && c0 + c < d) v = *reinterpret_cast<const g_f4 *>(src + (r0 + r) * src_stride + c0 + c);      // (d % 4 == 0)
        tile[r][c] = v[0]; tile[r][c + 1] = v[1]; tile[r][c + 2] = v[2]; tile[r][c + 3] = v[3];
    }
    __syncthreads();
    const int c = t >> 2, rs = (t & 3) * 16;
    if (c0 + c >= d) return;
    const long grp = r0 / group_cols, gcol0 = r0 - grp * group_cols;      // (a 64-row tile lies inside one group)
    _Float16 *oh = hi + ((grp * d + c0 + c) * group_cols + gcol0), *ol = lo + ((grp * d + c0 + c) * group_cols + gcol0);
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
        if (r0 + rs + 8 * h8 >= np) continue;
        half8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = tile[rs + 8 * h8 + j][c] * scale;        // (rows past n were loaded as zeros)
            hh[j] = (_Float16)x;
            ll[j] = (_Float16)(x - (float)hh[j]);
        }
        *reinterpret_cast<half8 *>(oh + rs + 8 * h8) = hh;
        *reinterpret_cast<half8 *>(ol + rs + 8 * h8) = ll;
    }
}

// ---- single 16-bit images (fp16 or bf16, no lo part) of an fp32 matrix: the operands of the one-pass products under autocast.
// Row-major [n, dp] (zero columns past d), or the TRANSPOSE [d, np] with the same grouping as split_half_t_kernel -- the operands of
// a weight-gradient product contract over the rows.  (torch's strided copies made these transposes: 2.7 ms of a 22 ms train step.)
template <bool BF>
__device__ __forceinline__ unsigned short half_bits(float x)
{
    if constexpr (BF) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b); }
    else { const _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); }
}
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(256) void half_image_kernel(const float *__restrict__ src, long n, int d, long src_stride, int dp, unsigned short *__restrict__ out)
{
    const int cpr = dp / 8;
    const long total = n * cpr;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / cpr;
        const int c = (int)(t - r * cpr) * 8;
        g_f4 va = (g_f4){0.f, 0.f, 0.f, 0.f}, vb = va;
        if (c < d) va = *reinterpret_cast<const g_f4 *>(src + r * src_stride + c);
        if (c + 4 < d) vb = *reinterpret_cast<const g_f4 *>(src + r * src_stride + c + 4);
        u16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = half_bits<BF>(j < 4 ? va[j] : vb[j - 4]);
        *reinterpret_cast<u16x8 *>(out + r * dp + c) = h;
    }
}

// plain (optional): the row-major image [n, plain_dp] of the same matrix (half_image_kernel's output) from the same pass -- a
// training-mode product needs both images of its upstream gradient and of its input (medtok_half_image_pair_f32).
// colsum (optional): [row tiles, d] fp32 -- the block's sums over its 64 rows of every column of its tile, in a fixed order (16 rows per
// thread, then the four threads of a column): a bias gradient is the sum of these partials over the row tiles, taken from the pass that
// makes the gradient's images instead of from a pass of its own over it (104 us for a 131 072 x 768 gradient).
template <bool BF>
__global__ __launch_bounds__(256) void half_image_t_kernel(const float *__restrict__ src, long n, int d, long src_stride, long np, long group_cols,
                                                           unsigned short *__restrict__ out, unsigned short *__restrict__ plain = nullptr, int plain_dp = 0,
                                                           float *__restrict__ colsum = nullptr)
{
    __shared__ float tile[64][65];
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
        g_f4 v = (g_f4){0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n && c0 + c < d) v = *reinterpret_cast<const g_f4 *>(src + (r0 + r) * src_stride + c0 + c);
        tile[r][c] = v[0]; tile[r][c + 1] = v[1]; tile[r][c + 2] = v[2]; tile[r][c + 3] = v[3];
        if (plain && r0 + r < n && c0 + c < plain_dp) {
            u16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = half_bits<BF>(v[j]);
            *reinterpret_cast<u16x4 *>(plain + (r0 + r) * plain_dp + c0 + c) = h;
        }
    }
    __syncthreads();
    const int c = t >> 2, rs = (t & 3) * 16;
    if (colsum) {                                         // (before the early exit: the shuffles want all four threads of a column)
        float a = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) a += tile[rs + jj][c];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if ((t & 3) == 0 && c0 + c < d) colsum[(long)blockIdx.x * d + c0 + c] = a;
    }
    if (c0 + c >= d) return;
    const long grp = r0 / group_cols, gcol0 = r0 - grp * group_cols;
    unsigned short *o = out + ((grp * d + c0 + c) * group_cols + gcol0);
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
        if (r0 + rs + 8 * h8 >= np) continue;
        u16x8 hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) hh[j] = half_bits<BF>(tile[rs + 8 * h8 + j][c]);
        *reinterpret_cast<u16x8 *>(o + rs + 8 * h8) = hh;
    }
}
