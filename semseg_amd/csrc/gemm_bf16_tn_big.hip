// 256 x 256 x 64 tiles of the bf16 weight-gradient GEMM  slab[z][M,N] = sum_{r in split z} A[r,M]^T . B[r,N]  for the shapes
// whose output is at least a tile wide both ways (the 128-tile kernel of gemm_bf16.hip moves 64 flop per byte from L2 into LDS
// and stops where that path does; this tile halves the bytes per flop, as gemm_bf16_big.hip does for the NT product).
//
//   512 threads = 8 waves as 4 (m) x 2 (n), each wave a 64 x 128 block of the tile, one workgroup per CU, two stages of 64 KiB.
//   Memory delivers both operands k-major ([r][m] and [r][n] rows), which is how they stay in LDS: a stage holds 64 rows of
//   512 bytes per operand, filled global -> LDS without passing registers (global_load_lds_dwordx4: one wave instruction = 1 KiB
//   = two rows).  MFMA fragments (k-contiguous per lane) come out through ds_read_b64_tr_b16.  With an unpadded 512-byte pitch
//   the four rows of a transpose block would share their banks, so 16-byte slot s of row r lives at slot s ^ ((r & 3) << 2):
//   the XOR is applied to the SOURCE address of the fill (an LDS-DMA destination is lane-linear) and to the fragment reads.
//   Columns past M / N are filled from the tile's first column: they only reach accumulators that are never stored.
//   Splits are whole k-tiles (R % 64 == 0 is part of the eligibility test), so no row needs masking.
#include "common.h"
#include "gemm_fast.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int xcd_remap_tnb(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Top of k-iteration kt in a ring of S stages: this wave's fills of tile kt have landed (fills complete in order; the `ahead` newer
// tiles, PER fills each, may still be on their way), then the workgroup barrier: every wave's part of tile kt is in LDS and nobody still
// reads the stage about to be refilled.
template <int PER>
__device__ __forceinline__ void tn_wait_barrier(int ahead) {
    static_assert(3 * PER <= 63, "vmcnt is a 6-bit field");
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(3 * PER) : "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * PER) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// BK rows per stage, S stages (S - 1 k-tiles in flight).  Both operands of a weight gradient are streams from HBM (every k-tile is new
// data, ~2 us away), and a k-tile of 64 rows is multiplied in 0.85 us: with one tile in flight the k-iteration was that round trip.
template <int BK, int S>
__global__ __launch_bounds__(512) void gemm_tn_bf16_256_kernel(FastGemm p) {
    constexpr int BM = 256, BN = 256;
    constexpr int PITCH = BM * 2;                      // 512-byte rows
    constexpr int IMG = BK * PITCH;                    // one operand image
    constexpr int STAGE = 2 * IMG;
    constexpr int NP = BK / 16;                        // 1 KiB fill pieces per wave and operand
    static_assert(S >= 2 && S <= 5 && S * STAGE <= 160 * 1024 && BK % 16 == 0, "ring must fit the LDS");
    __shared__ __attribute__((aligned(1024))) char smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    // all output tiles of a run of consecutive splits on one XCD: they stream the same A / B rows through that XCD's L2
    const int vid = xcd_remap_tnb(blockIdx.x, gridDim.x);
    const int split = vid / p.ntiles, wg = vid % p.ntiles;
    const int m0 = (wg / p.ntn) * BM, n0 = (wg % p.ntn) * BN;
    const int64_t kbeg = (int64_t)split * p.k_per_split;
    const int64_t kend = kbeg + p.k_per_split < p.K ? kbeg + p.k_per_split : p.K;
    const int nk = (int)((kend - kbeg) / BK);

    // LDS-DMA sources: piece q = wave*NP + i covers rows 2q, 2q+1 of a stage image; this lane fills physical slot (lane & 31) of
    // row 2q + (lane >> 5), which holds logical slot (lane & 31) ^ ((row & 3) << 2)
    const bf16* pa[NP]; const bf16* pb[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = (wave * NP + i) * 2 + (lane >> 5);
        const int ls = (lane & 31) ^ ((row & 3) << 2);
        const int ca = m0 + ls * 8 < p.M ? m0 + ls * 8 : m0;
        const int cb = n0 + ls * 8 < p.N ? n0 + ls * 8 : n0;
        pa[i] = p.A + (kbeg + row) * p.lda + ca;
        pb[i] = p.B + (kbeg + row) * p.ldb + cb;
    }
    // fills by hand (dma_b128, gemm_fast.h): with the builtin the compiler made the transpose reads of tile kt wait for the fill of tile
    // kt + 1 issued just before them -- fill and products took turns (2.4 us per k-tile for 0.85 us of matrix work)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * NP * 1024);
    auto issue = [&](int kt, int stage) {
        const unsigned st = lds0 + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NP; ++i) dma_b128(pa[i] + (int64_t)kt * BK * p.lda, st + i * 1024);
#pragma unroll
        for (int i = 0; i < NP; ++i) dma_b128(pb[i] + (int64_t)kt * BK * p.ldb, st + IMG + i * 1024);
    };

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transpose-read addressing (see gemm_bf16.hip): lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of a
    // 4 x 16 block and receives column (lane & 15).  Row = 16 ks + 8 lh + q (+4 for the upper half of the fragment).
    const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3, gsel = (lane >> 4) & 1;
    const int row_off = (8 * lh + tq) * PITCH;
    const int sub_off = (gsel * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;          // inside an aligned group of four slots
    auto trfrag = [&](const char* img, int ks, int colbase) {                  // colbase: multiple of 32
        const char* a0 = img + ks * 16 * PITCH + row_off + ((((colbase >> 3) ^ (tq << 2))) << 4) + sub_off;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * PITCH));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        return f;
    };

    const bool do_cs = p.colsum_ws && n0 == 0;
    float cs = 0.f;
    // DropPath row keep (see gemm_tn_bf16_kernel): k-tiles of dropped samples are skipped, the kept ones share one multiplier.  The flags
    // of the samples this split touches (at most 64: ss_big_tn_ok) are read ONCE into a mask -- a load per k-tile would queue behind the
    // fills and its wait would be a wait for them
    unsigned long long keep_mask = ~0ull;
    float keep_mul = 1.f;
    int srow = 0, sidx = 0;
    if (p.rowscale) {
        const int64_t s0 = kbeg / p.rows_per_scale;
        const int64_t nsmp = nk > 0 ? (kend - 1) / p.rows_per_scale - s0 + 1 : 0;
        const float v = lane < nsmp ? p.rowscale[s0 + lane] : 0.f;
        keep_mask = __builtin_amdgcn_ballot_w64(v != 0.f);
        keep_mul = group_max(v, 64);
        srow = (int)(kbeg - s0 * p.rows_per_scale);
    }
    for (int t = 0; t < S - 1 && t < nk; ++t) issue(t, t);
    int cur = 0, nxt = S - 1;                          // stages of tile kt and of tile kt + S - 1
    for (int kt = 0; kt < nk; ++kt) {
        tn_wait_barrier<2 * NP>(min(S - 2, nk - 1 - kt));
        if (kt + S - 1 < nk) issue(kt + S - 1, nxt);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + IMG;
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
        const bool kept = (keep_mask >> sidx) & 1;
        srow += BK;
        if (p.rowscale && srow >= p.rows_per_scale) { srow -= (int)p.rows_per_scale; ++sidx; }
#pragma unroll
        for (int ks = 0; ks < (kept ? BK / 16 : 0); ++ks) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = trfrag(As, ks, wm * 64 + a * 32);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = trfrag(Bs, ks, wn * 128 + b * 32);
            if (do_cs) {                               // column sums of A (the bias gradient): wave (wm, wn) sums its a = wn fragment
                const bf16x8 f = wn ? fa[1] : fa[0];
                const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 pr = {f[2 * e], f[2 * e + 1]};
                    cs = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, cs, false);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
    if (do_cs) {                                       // lane = column (lane & 31) of the A fragment, k split over the lane halves
        const float t = (cs + __shfl_xor(cs, 32)) * keep_mul;
        const int m = m0 + wm * 64 + wn * 32 + li;
        if (lh == 0 && m < p.M) p.colsum_ws[(int64_t)split * p.M + m] = t;
    }
    // fp32 slab: lanes run along n (128-byte segments per register row)
    float* C = (float*)p.C + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = n0 + wn * 128 + b * 32 + li;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) C[(int64_t)row * p.N + col] = acc[a][b][r] * keep_mul;
            }
        }
}

// The same kernel for outputs one of whose sides is a multiple of 320 but not of 256 (MiT stage 3: 320 x 1280 and 1280 x 320 weight
// gradients, which the 256-tile plan refuses as 60 % padding): 256 x 320 tiles (waves 4 x 2, 64 x 160 each) or 320 x 256 tiles
// (waves 2 x 4, 160 x 64 each).  Operand rows in LDS are BM*2 / BN*2 bytes (512 / 640: multiples of 128, so the rows of a transpose
// block share their banks); 16-byte slot s of row r lives at s ^ ((r & 1) << 2) -- odd rows shifted by 64 bytes inside their 128-byte
// block, which is what the four-row transpose read needs -- applied to the fill's source address and to the fragment reads.
// The thin weight gradients of the Mix-FFN (MiT stages 1-2: 64 .. 512 x 64 .. 512 outputs over 2^18 .. 2^20 rows) use it with ONE tile =
// the whole output, so every operand row is read once and a workgroup's fill in flight is 32 .. 80 KiB (two workgroups per CU where two
// pairs of stages fit): these products are HBM streams, and the 128-tile kernel kept 24 .. 32 KiB per workgroup in flight through registers.
template <int TA, int TB, int WM, int WN, int BK = 64, int S = 2>
__global__ __launch_bounds__(512, (WM * TA + WN * TB) * 32 * BK * 2 * S <= 80 * 1024 ? 2 : 1) void gemm_tn_bf16_wide_kernel(FastGemm p) {
    constexpr int BM = WM * TA * 32, BN = WN * TB * 32;
    constexpr int PA = BM * 2, PB = BN * 2;            // row pitches (bytes)
    constexpr int IMGA = BK * PA, IMGB = BK * PB, STAGE = IMGA + IMGB;
    constexpr int QA = IMGA / 1024 / 8, QB = IMGB / 1024 / 8;          // 1 KiB fill pieces per wave
    static_assert(WM * WN == 8 && IMGA % 8192 == 0 && IMGB % 8192 == 0, "eight waves, whole pieces per wave");
    static_assert(S >= 2 && S <= 5 && S * STAGE <= 160 * 1024 && BK % 16 == 0, "ring must fit the LDS");
    __shared__ __attribute__((aligned(1024))) char smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int vid = xcd_remap_tnb(blockIdx.x, gridDim.x);
    const int split = vid / p.ntiles, wg = vid % p.ntiles;
    const int m0 = (wg / p.ntn) * BM, n0 = (wg % p.ntn) * BN;
    const int64_t kbeg = (int64_t)split * p.k_per_split;
    const int64_t kend = kbeg + p.k_per_split < p.K ? kbeg + p.k_per_split : p.K;
    const int nk = (int)((kend - kbeg) / BK);

    const bf16* pa[QA]; const bf16* pb[QB];
#pragma unroll
    for (int i = 0; i < QA; ++i) {                     // this lane fills physical slot g of the A image: row g / (BM/8), slot g % (BM/8)
        const int g = (wave * QA + i) * 64 + lane, row = g / (BM / 8), ls = (g % (BM / 8)) ^ ((row & 1) << 2);
        const int ca = m0 + ls * 8 < p.M ? m0 + ls * 8 : m0;
        pa[i] = p.A + (kbeg + row) * p.lda + ca;
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int g = (wave * QB + i) * 64 + lane, row = g / (BN / 8), ls = (g % (BN / 8)) ^ ((row & 1) << 2);
        const int cb = n0 + ls * 8 < p.N ? n0 + ls * 8 : n0;
        pb[i] = p.B + (kbeg + row) * p.ldb + cb;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int kt, int stage) {              // by hand, see gemm_tn_bf16_256_kernel
        const unsigned st = lds0 + stage * STAGE;
#pragma unroll
        for (int i = 0; i < QA; ++i) dma_b128(pa[i] + (int64_t)kt * BK * p.lda, st + (wave * QA + i) * 1024);
#pragma unroll
        for (int i = 0; i < QB; ++i) dma_b128(pb[i] + (int64_t)kt * BK * p.ldb, st + IMGA + (wave * QB + i) * 1024);
    };

    floatx16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transpose-read addressing as in gemm_tn_bf16_256_kernel; row = 16 ks + 8 lh + tq (+4 for the upper half: same parity, same shift)
    const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3, gsel = (lane >> 4) & 1;
    const int sub_off = (gsel * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
    auto trfrag = [&](const char* img, int pitch, int ks, int colbase) {       // colbase: multiple of 32
        const char* a0 = img + (ks * 16 + 8 * lh + tq) * pitch + (((colbase >> 3) ^ ((tq & 1) << 2)) << 4) + sub_off;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * pitch));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        return f;
    };

    const bool do_cs = p.colsum_ws && n0 == 0 && wn == 0;     // column sums of A (the bias gradient): the wn = 0 waves sum their fragments
    float cs[TA];
#pragma unroll
    for (int a = 0; a < TA; ++a) cs[a] = 0.f;
    // DropPath row keep (see gemm_tn_bf16_kernel): k-tiles of dropped samples are skipped, the kept ones share one multiplier.  The flags
    // of the samples this split touches (at most 64: ss_big_tn_ok) are read ONCE into a mask -- a load per k-tile would queue behind the
    // fills and its wait would be a wait for them
    unsigned long long keep_mask = ~0ull;
    float keep_mul = 1.f;
    int srow = 0, sidx = 0;
    if (p.rowscale) {
        const int64_t s0 = kbeg / p.rows_per_scale;
        const int64_t nsmp = nk > 0 ? (kend - 1) / p.rows_per_scale - s0 + 1 : 0;
        const float v = lane < nsmp ? p.rowscale[s0 + lane] : 0.f;
        keep_mask = __builtin_amdgcn_ballot_w64(v != 0.f);
        keep_mul = group_max(v, 64);
        srow = (int)(kbeg - s0 * p.rows_per_scale);
    }
    for (int t = 0; t < S - 1 && t < nk; ++t) issue(t, t);
    int cur = 0, nxt = S - 1;
    for (int kt = 0; kt < nk; ++kt) {
        tn_wait_barrier<QA + QB>(min(S - 2, nk - 1 - kt));
        if (kt + S - 1 < nk) issue(kt + S - 1, nxt);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + IMGA;
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
        const bool kept = (keep_mask >> sidx) & 1;
        srow += BK;
        if (p.rowscale && srow >= p.rows_per_scale) { srow -= (int)p.rows_per_scale; ++sidx; }
#pragma unroll
        for (int ks = 0; ks < (kept ? BK / 16 : 0); ++ks) {
            bf16x8 fa[TA], fb[TB];
#pragma unroll
            for (int a = 0; a < TA; ++a) fa[a] = trfrag(As, PA, ks, wm * TA * 32 + a * 32);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb[b] = trfrag(Bs, PB, ks, wn * TB * 32 + b * 32);
            if (do_cs) {
                const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
                for (int a = 0; a < TA; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2 pr = {fa[a][2 * e], fa[a][2 * e + 1]};
                        cs[a] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, cs[a], false);
                    }
            }
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
    if (do_cs) {
#pragma unroll
        for (int a = 0; a < TA; ++a) {
            const float t = (cs[a] + __shfl_xor(cs[a], 32)) * keep_mul;
            const int m = m0 + wm * TA * 32 + a * 32 + li;
            if (lh == 0 && m < p.M) p.colsum_ws[(int64_t)split * p.M + m] = t;
        }
    }
    float* C = (float*)p.C + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int col = n0 + wn * TB * 32 + b * 32 + li;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TA * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) C[(int64_t)row * p.N + col] = acc[a][b][r] * keep_mul;
            }
        }
}

// tile shape of the big TN kernels for an [M, N] output: 256 x 256, or a 320-wide side where that side is a multiple of 320 (and not
// of 256) and the other one suits 256 / 320 rows.  0 = none (the 128-tile kernel takes the shape).
static int tn_tile_shape(int64_t M, int64_t N, int* bm, int* bn) {
    auto fits = [](int64_t x, int t) { return x >= t && (double)(cdiv(x, t) * t) <= 1.25 * (double)x; };
    if (M % 8 || N % 8) return 0;
    // one tile = the whole output (thin shapes; at most 25 % of a side is padding)
    auto whole = [](int64_t x, int t) { return x <= t && 5 * x >= 4 * t; };
    static const int thin[5][2] = {{512, 128}, {128, 512}, {256, 64}, {64, 256}, {128, 128}};
    for (int i = 0; i < 5; ++i)
        if (whole(M, thin[i][0]) && whole(N, thin[i][1])) { *bm = thin[i][0]; *bn = thin[i][1]; return 4 + i; }
    if (M < 256 || N < 256) return 0;
    if ((double)(cdiv(M, 256) * 256) * (double)(cdiv(N, 256) * 256) <= 1.5 * (double)M * (double)N) { *bm = 256; *bn = 256; return 1; }
    if (N % 320 == 0 && fits(M, 256)) { *bm = 256; *bn = 320; return 2; }
    if (M % 320 == 0 && fits(N, 256)) { *bm = 320; *bn = 256; return 3; }
    return 0;
}

// Plan for the 256-tile kernel: whole-chip launches of one workgroup per CU (tiles x splits <= 256, or several full rounds when the
// output alone has more tiles), splits of whole k-tiles.  False when the shape does not suit the tile.
bool ss_big_tn_plan(int64_t M, int64_t N, int64_t R, int* nsplit, int64_t* rps) {
    int bm, bn;
    const int shape = R % 64 || R < 4096 ? 0 : tn_tile_shape(M, N, &bm, &bn);
    if (!shape) return false;
    const int64_t tiles = cdiv(M, bm) * cdiv(N, bn);
    const int64_t slots = (bm + bn) <= 320 ? 512 : 256;        // two workgroups per CU where their LDS allows (256 x 64, 64 x 256, 128 x 128)
    int64_t s = tiles >= slots ? 1 : slots / tiles;
    const int64_t maxs = R / (8 * 64);                 // >= 8 k-tiles per split
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    const int64_t per = cdiv(cdiv(R, s), 64) * 64;
    *rps = per;
    *nsplit = (int)cdiv(R, per);
    return true;
}

bool ss_big_tn_ok(const FastGemm& p, int nsplit) {
    int bm, bn;
    if (!aligned16(p.A) || !aligned16(p.B) || p.lda % 8 || p.ldb % 8 || !tn_tile_shape(p.M, p.N, &bm, &bn)) return false;
    if (p.K % 64 || p.k_per_split % 64) return false;
    if (p.rowscale && (p.rows_per_scale % 64 || p.rows_per_scale >= (1LL << 31) || cdiv(p.k_per_split, p.rows_per_scale) + 1 > 64)) return false;   // the keep mask
    return cdiv(p.M, bm) * cdiv(p.N, bn) * nsplit < (1LL << 30);
}

// SS_GEMM_TN_RING=0: two stages of 64 rows instead of five of 32 in the 256 x 256 kernel (A/B runs)
static int tn_ring_mode() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("SS_GEMM_TN_RING"); m = e && e[0] == '0' ? 0 : 1; }
    return m;
}

int ss_big_tn_launch(FastGemm p, int nsplit, hipStream_t st) {
    int bm = 256, bn = 256;
    const int shape = tn_tile_shape(p.M, p.N, &bm, &bn);
    p.ntn = (int)cdiv(p.N, bn);
    p.ntiles = (int)(p.ntn * cdiv(p.M, bm));
    const dim3 grid((unsigned)(p.ntiles * nsplit));
    if (shape == 2) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<2, 5, 4, 2>), grid, dim3(512), 0, st, p);
    else if (shape == 3) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<5, 2, 2, 4>), grid, dim3(512), 0, st, p);
    else if (shape == 4) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<4, 2, 4, 2, 32, 4>), grid, dim3(512), 0, st, p);      // 512 x 128
    else if (shape == 5) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<2, 4, 2, 4, 32, 4>), grid, dim3(512), 0, st, p);      // 128 x 512
    else if (shape == 6) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<2, 1, 4, 2>), grid, dim3(512), 0, st, p);      // 256 x 64
    else if (shape == 7) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<1, 2, 2, 4>), grid, dim3(512), 0, st, p);      // 64 x 256
    else if (shape == 8) hipLaunchKernelGGL((gemm_tn_bf16_wide_kernel<2, 1, 2, 4, 32, 4>), grid, dim3(512), 0, st, p);      // 128 x 128
    else if (tn_ring_mode() == 0) hipLaunchKernelGGL((gemm_tn_bf16_256_kernel<64, 2>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_bf16_256_kernel<32, 5>), grid, dim3(512), 0, st, p);
    SS_LAUNCH_CHECK();
    return 0;
}
