// bf16 fast paths of the projection GEMMs, written for the shapes this model produces: M (tokens) is
// huge, N and K are channel counts (64 .. 2560), so most launches sit on the HBM roofline and the rest on MFMA.
//
//   gemm_nt_bf16_kernel : C[M,N] = epilogue(A[M,K] . B[N,K]^T)        128 x BN x 64 tiles, BN in {64, 128}
//   gemm_tn_bf16_kernel : slab[z][M,N] = sum_{r in split z} A[r,M]^T . B[r,N]   (weight gradients; fp32 slabs)
//
// Both: 256 threads = 4 waves (2 x 2), v_mfma_f32_32x32x16_bf16, <= 168 registers so three workgroups share a CU and
// hide each other's barriers; operands are staged global -> registers -> LDS with the next tile's loads in flight
// under the current tile's MFMAs.
//   NT  LDS image: 128-byte rows of 64 k-values; 16-byte chunk c of row r lives at slot ((r&1)*8 + c) ^ ((r>>1)&15) of
//       the 256-byte bank row r>>1, which makes the fragment ds_read_b128 (32 rows, one chunk) conflict-free.
//       The MFMA runs "transposed" (weights as the A operand): each lane then owns 4 CONSECUTIVE output columns of one
//       output row per register quad, the tile is written to an LDS [m][n] image with ds_write_b64 and leaves the
//       workgroup as whole rows of 16-byte stores (bias / GELU / DropPath scale / residual fused, one rounding).
//   TN  LDS image: [r][m] rows (k-major, as memory delivers them); fragments come out k-contiguous through
//       ds_read_b64_tr_b16.  Rows are padded by 64 bytes so the four rows of a transpose block tile the 64 banks.
#include "common.h"
#include "gemm_fast.h"


__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks bid, bid+8, ... share an XCD (and its L2): hand each XCD a contiguous run of tiles (bijective form)
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ int nt_slot(int row, int chunk) {      // 16-byte slot index inside a tile image
    int R = row >> 1;
    return R * 16 + ((((row & 1) << 3) | chunk) ^ (R & 15));
}

template <int BN, bool PATCH = false>
__global__ __launch_bounds__(256, 3) void gemm_nt_bf16_kernel(FastGemm p) {
    constexpr int BM = 128, BK = 64;
    constexpr int NB = BN / 32;                       // 16-byte chunks of B per thread per k-tile
    constexpr int TN = BN / 64;                       // 32-wide n tiles per wave
    constexpr int EP = BN * 2 + 16;                   // epilogue image pitch (bytes)
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr int SMEM = STAGE > BM * EP ? STAGE : BM * EP;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    bf16* As = (bf16*)smem;
    bf16* Bs = (bf16*)(smem + BM * BK * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / p.ntn) * BM;
    const int n0 = (wg % p.ntn) * BN;
    const int K = (int)p.K;

    // this thread's staging chunks: c = tid + i*256 -> row c>>3, 16-byte chunk c&7 (8 lanes read one 128-byte row segment)
    const int srow = tid >> 3, sch = tid & 7;
    const bf16* pa[4]; const bf16* pb[NB];
    bool oka[4], okb[NB];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t r = m0 + srow + i * 32;
        oka[i] = r < p.M;
        if (!oka[i]) r = 0;
        pa[i] = p.A + (PATCH && (p.pm.side & 1) ? patch_row(p.pm, r, p.lda) : r * p.lda) + sch * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int r = n0 + srow + i * 32;
        okb[i] = r < p.N;
        pb[i] = p.B + (int64_t)(okb[i] ? r : 0) * p.ldb + sch * 8;
    }
    bf16x8 ra[4], rb[NB];
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int k0) {
        const bool kin = k0 + sch * 8 < K;
        const int64_t ka = PATCH && (p.pm.side & 1) ? patch_col(p.pm, k0) : k0;      // a k-tile lies inside one run (kc % 64 == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = (oka[i] && kin) ? *(const bf16x8*)(pa[i] + ka) : zero;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = (okb[i] && kin) ? *(const bf16x8*)(pb[i] + k0) : zero;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(bf16x8*)(As + nt_slot(srow + i * 32, sch) * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(bf16x8*)(Bs + nt_slot(srow + i * 32, sch) * 8) = rb[i];
    };

    floatx16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // the residual tile rides in the staging registers during the LAST k-iteration (they are free then): its HBM round trip
    // overlaps that iteration's MFMAs instead of standing between the k-loop and the epilogue
    constexpr int CPR = BN / 8;                        // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;                     // rows per pass
    static_assert(BM / RPP <= 4 + NB, "residual tile does not fit the staging registers");
    const int erow = tid / CPR, ech = tid % CPR;
    const bool ncol_ok = n0 + ech * 8 < p.N;
    auto rload = [&]() {
#pragma unroll
        for (int ps = 0; ps < BM / RPP; ++ps) {
            const int row = ps * RPP + erow;
            const bf16x8 v = (m0 + row < p.M && ncol_ok) ? *(const bf16x8*)(p.res + (m0 + row) * p.ldr + n0 + ech * 8) : zero;
            if (ps < 4) ra[ps & 3] = v; else rb[(ps - 4) & (NB - 1)] = v;
        }
    };

    gload(0);
    lstore();
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) gload(k0 + BK);
        else if (p.res) rload();
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[TN];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(As + nt_slot(wm * 64 + a * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = *(const bf16x8*)(Bs + nt_slot(wn * (BN / 2) + b * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)       // D[i = n][j = m]: the lane is an output ROW, its registers run along n
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) { lstore(); __syncthreads(); }
    }

    // ---- epilogue through an LDS [m][n] image (pitch EP bytes)
    bf16* C = (bf16*)p.C;
    if (p.res) {
#pragma unroll
        for (int ps = 0; ps < BM / RPP; ++ps) {
            const int row = ps * RPP + erow;
            *(bf16x8*)(smem + row * EP + ech * 16) = ps < 4 ? ra[ps & 3] : rb[(ps - 4) & (NB - 1)];
        }
        __syncthreads();
    }
    nt_epilogue_dispatch<2, TN>(acc, p, smem, EP, m0, n0, wm * 64, wn * (BN / 2), li, lh);
    __syncthreads();
    float sx[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sx[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < BM / RPP; ++ps) {
        int row = ps * RPP + erow;
        if (m0 + row < p.M && ncol_ok) {
            const bf16x8 v = *(const bf16x8*)(smem + row * EP + ech * 16);
            if (PATCH && (p.pm.side & 2))
                *(bf16x8*)(C + patch_row(p.pm, m0 + row, p.ldc) + patch_col(p.pm, n0 + ech * 8)) = v;
            else nt_store_chunk(p, C, m0 + row, n0 + ech * 8, v);
            if (!PATCH && p.stat_ws) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sx[e] += f; sq[e] = fmaf(f, f, sq[e]); }
            }
        }
    }
    if (!PATCH && p.stat_ws) nt_tile_stats<BN, RPP, 256>(p, smem, sx, sq, erow, ech, m0 / BM, n0, tid);
}

// ------------------------------------------------------------------ TN (weight gradient)
template <int BN, bool PATCH = false>
__global__ __launch_bounds__(256, 3) void gemm_tn_bf16_kernel(FastGemm p) {
    constexpr int BM = 128, BK = 64;
    constexpr int PA = BM * 2 + 64, PB = BN * 2 + 64;   // row pitches in bytes
    constexpr int CA = BM / 8, CB = BN / 8;             // 16-byte chunks per row
    constexpr int NA = BK * CA / 256, NBk = BK * CB / 256;
    constexpr int TN = BN / 64;
    __shared__ __attribute__((aligned(16))) char smem[BK * (PA + PB)];
    char* As = smem;
    char* Bs = smem + BK * PA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    // one XCD (= one L2) takes ALL output tiles of a run of consecutive splits: the tiles of a split stream the same A / B rows,
    // and spread over the 8 XCDs every L2 would fetch those rows from the fabric again
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vid / p.ntiles, wg = vid % p.ntiles;
    const int m0 = (wg / p.ntn) * BM, n0 = (wg % p.ntn) * BN;
    const int64_t kbeg = (int64_t)split * p.k_per_split;
    const int64_t kend = kbeg + p.k_per_split < p.K ? kbeg + p.k_per_split : p.K;

    const int arow = tid / CA, ach = tid % CA, brow = tid / CB, bch = tid % CB;
    const bool oka = m0 + ach * 8 < p.M, okb = n0 + bch * 8 < p.N;
    const bf16* pa = p.A + m0 + ach * 8;
    const bf16* pb = p.B + (PATCH ? (okb ? patch_col(p.pm, n0 + bch * 8) : 0) : n0 + bch * 8);     // a column tile lies inside one run (kc % BN == 0)
    bf16x8 ra[NA], rb[NBk];
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { int64_t r = r0 + arow + i * (256 / CA); ra[i] = (oka && r < kend) ? *(const bf16x8*)(pa + r * p.lda) : zero; }
#pragma unroll
        for (int i = 0; i < NBk; ++i) { int64_t r = r0 + brow + i * (256 / CB); const bool in = okb && r < kend;
            rb[i] = in ? *(const bf16x8*)(pb + (PATCH ? patch_row(p.pm, r, p.ldb) : r * p.ldb)) : zero; }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) *(bf16x8*)(As + (arow + i * (256 / CA)) * PA + ach * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NBk; ++i) *(bf16x8*)(Bs + (brow + i * (256 / CB)) * PB + bch * 16) = rb[i];
    };

    floatx16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transpose-read addressing: lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of a 4 x 16 block and
    // receives column (lane & 15); block rows = k, block columns = this lane group's 16 of the 32 tile columns.
    const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3, gsel = (lane >> 4) & 1;
    const int tr_row = 8 * lh + tq;                    // + 16*ks + 4*t
    const int tr_col = gsel * 16 + 4 * tp;             // + tile column base
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    auto trfrag = [&](const char* base, int pitch, int ks, int colbase) {
        bf16x8 f;
        const char* a0 = base + (ks * 16 + tr_row) * pitch + (colbase + tr_col) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * pitch));
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        return f;
    };

    const bool do_cs = p.colsum_ws && n0 == 0;
    float cs = 0.f, keep_mul = p.rowscale ? 0.f : 1.f;
    if (kbeg < kend) {
        gload(kbeg);
        lstore();
        __syncthreads();
        for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = k0 + BK < kend;
            if (more) gload(k0 + BK);
            // row keep (DropPath in the backward): a k-tile lies inside one sample; tiles of dropped samples contribute nothing and the
            // kept ones share ONE multiplier, applied to the slab at the end
            const float kv = p.rowscale ? p.rowscale[k0 / p.rows_per_scale] : 1.f;
            if (kv != 0.f) keep_mul = kv;
#pragma unroll
            for (int ks = 0; ks < (kv != 0.f ? BK / 16 : 0); ++ks) {
                bf16x8 fa[2], fb[TN];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = trfrag(As, PA, ks, wm * 64 + a * 32);
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[b] = trfrag(Bs, PB, ks, wn * (BN / 2) + b * 32);
                if (do_cs) {                              // column sums of A (the bias gradient) ride along in the first n tile:
                    const bf16x8 f = wn ? fa[1] : fa[0];  // wave (wm, wn) sums its a = wn fragment, 4 x v_dot2_f32_bf16 with (1, 1)
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
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
            if (more) { lstore(); __syncthreads(); }
        }
    }
    if (do_cs) {                                          // lane = column (lane & 31) of the A fragment, k split over lane halves
        const float t = (cs + __shfl_xor(cs, 32)) * keep_mul;
        const int m = m0 + wm * 64 + wn * 32 + li;
        if (lh == 0 && m < p.M) p.colsum_ws[(int64_t)split * p.M + m] = t;
    }
    // fp32 slab: lanes run along n (128-byte segments per register row)
    float* C = (float*)p.C + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * (BN / 2) + b * 32 + li;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) C[(int64_t)row * p.N + col] = acc[a][b][r] * keep_mul;
            }
        }
}

// ------------------------------------------------------------------ host side (called from gemm.hip)
bool ss_fast_nt_ok(const void* A, int64_t lda, const void* B, int64_t ldb, const void* C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, const void* res, int64_t ldr, const float* bias) {
    if (K % 8 || lda % 8 || ldb % 8 || N % 8 || ldc % 8 || K >= (1LL << 30)) return false;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C)) return false;
    if (res && (!aligned16(res) || ldr % 8)) return false;
    if (bias && !aligned16(bias)) return false;
    return cdiv(M, 128) * cdiv(N, 64) < (1LL << 30);
}

static int big_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SS_GEMM_BIG"); on = !(e && e[0] == '0'); }
    return on;
}

int64_t ss_fast_nt_stat_tiles(const FastGemm& p) { return cdiv(p.M, big_enabled() && ss_big_nt_ok(p) ? 256 : 128); }

int ss_fast_nt_launch(FastGemm p, hipStream_t st) {
    if (ss_stream_nt_ok(p)) return ss_stream_nt_launch(p, st);
    if (big_enabled() && ss_big_nt_ok(p)) return ss_big_nt_launch(p, st);
    if (big_enabled() && ss_wide_nt_ok(p)) return ss_wide_nt_launch(p, st);
    const bool narrow = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 1024);
    if (narrow) {
        p.ntn = (int)cdiv(p.N, 64);
        hipLaunchKernelGGL(gemm_nt_bf16_kernel<64>, dim3((unsigned)(p.ntn * cdiv(p.M, 128))), dim3(256), 0, st, p);
    } else {
        p.ntn = (int)cdiv(p.N, 128);
        hipLaunchKernelGGL(gemm_nt_bf16_kernel<128>, dim3((unsigned)(p.ntn * cdiv(p.M, 128))), dim3(256), 0, st, p);
    }
    SS_LAUNCH_CHECK();
    return 0;
}

bool ss_fast_tn_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N) {
    return M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B) && M < (1 << 24) && N < (1 << 24);
}

int ss_fast_tn_launch(FastGemm p, int nsplit, hipStream_t st) {
    if (p.N <= 64) {
        p.ntn = (int)cdiv(p.N, 64);
        p.ntiles = (int)(p.ntn * cdiv(p.M, 128));
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<64>, dim3((unsigned)(p.ntiles * nsplit)), dim3(256), 0, st, p);
    } else {
        p.ntn = (int)cdiv(p.N, 128);
        p.ntiles = (int)(p.ntn * cdiv(p.M, 128));
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<128>, dim3((unsigned)(p.ntiles * nsplit)), dim3(256), 0, st, p);
    }
    SS_LAUNCH_CHECK();
    return 0;
}

// The same tile kernels with one operand addressed as image patches (PatchMap): the convolutions with kernel = stride.
int ss_patch_nt_launch(FastGemm p, hipStream_t st) {
    // fewer 128-column tiles than CUs (the spatial-reduction convs: 16384 patches x 128 channels = 128 tiles, k-loops of 32-64
    // iterations): 64-column tiles put twice as many workgroups on the chip; the second read of the patch rows comes from L2
    const bool narrow = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 1024) || cdiv(p.N, 128) * cdiv(p.M, 128) < 256;
    if (narrow) {
        p.ntn = (int)cdiv(p.N, 64);
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<64, true>), dim3((unsigned)(p.ntn * cdiv(p.M, 128))), dim3(256), 0, st, p);
    } else {
        p.ntn = (int)cdiv(p.N, 128);
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, true>), dim3((unsigned)(p.ntn * cdiv(p.M, 128))), dim3(256), 0, st, p);
    }
    SS_LAUNCH_CHECK();
    return 0;
}

int ss_patch_tn_launch(FastGemm p, int nsplit, hipStream_t st) {
    if (p.pm.kc % 128) {
        p.ntn = (int)cdiv(p.N, 64);
        p.ntiles = (int)(p.ntn * cdiv(p.M, 128));
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<64, true>), dim3((unsigned)(p.ntiles * nsplit)), dim3(256), 0, st, p);
    } else {
        p.ntn = (int)cdiv(p.N, 128);
        p.ntiles = (int)(p.ntn * cdiv(p.M, 128));
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<128, true>), dim3((unsigned)(p.ntiles * nsplit)), dim3(256), 0, st, p);
    }
    SS_LAUNCH_CHECK();
    return 0;
}
