// 256 x 256 x 64 tiles of the bf16 projection GEMM for the MFMA-bound shapes of the step (K >= 256, N a multiple of 256).
//
// Why a second tile size: a 128 x 128 tile moves 64 flop per byte from L2 into LDS, and the step's launches of that kernel
// top out where the L2 -> LDS path does (about 12 TB/s chip-wide = 800 TFLOP/s); 256 x 256 halves the bytes per flop.
//
//   512 threads = 8 waves as 4 (m) x 2 (n), each wave a 64 x 128 block of the tile (8 accumulators of 32 x 32), one
//   workgroup per CU.  Both operands go global -> LDS without passing registers (global_load_lds_dwordx4): one wave
//   instruction fills 1 KiB = 8 tile rows of 64 k-values; the image is the nt_slot XOR swizzle of gemm_bf16.hip, realised on
//   the SOURCE address (the destination of an LDS-DMA is lane-linear) and on the fragment reads.  Two stages of 64 KiB:
//   tile k+1 lands while tile k is multiplied, one barrier per k-tile.  Epilogue as in the 128-tile kernel: an LDS [m][n]
//   image, then whole rows in 16-byte stores (bias / activation / DropPath scale / residual fused, one rounding).
#include "common.h"
#include "gemm_fast.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ int xcd_remap_big(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ int big_slot(int row, int chunk) {
    int R = row >> 1;
    return R * 16 + ((((row & 1) << 3) | chunk) ^ (R & 15));
}

// ---- epilogue of a 256 x 256 tile through an LDS [m][n] image (pitch 528 bytes); all 512 threads, the operand stages are dead
__device__ __forceinline__ void big_nt_epilogue(floatx16 (&acc)[2][4], const FastGemm& p, char* smem, int64_t m0, int n0, int tid, int wave, int wm, int wn,
                                                int li, int lh) {
    constexpr int BM = 256, BN = 256;
    constexpr int EP = BN * 2 + 16;
    // ---- epilogue through an LDS [m][n] image (pitch EP bytes)
    bf16* C = (bf16*)p.C;
    constexpr int CPR = BN / 8;                        // 16-byte chunks per tile row
    constexpr int RPP = 512 / CPR;                     // rows per pass
    const int erow = tid / CPR, ech = tid % CPR;
    const bool ncol_ok = n0 + ech * 8 < p.N;
    if (p.res) {
#pragma unroll 4
        for (int ps = 0; ps < BM / RPP; ++ps) {
            int row = ps * RPP + erow;
            if (m0 + row < p.M && ncol_ok)
                *(bf16x8*)(smem + row * EP + ech * 16) = *(const bf16x8*)(p.res + (m0 + row) * p.ldr + n0 + ech * 8);
        }
        __syncthreads();
    }
    nt_epilogue_dispatch<2, 4>(acc, p, smem, EP, m0, n0, wm * 64, wn * 128, li, lh);
    __syncthreads();
    if (p.w2) {
        // second product: the finished bf16 tile [256 rows][256 = all N columns] times w2^T; wave w takes rows 32w .. 32w+31, the lane is
        // a row again and its first register quads are the n2 <= 32 output columns.  The tile itself is not stored.
        const int row = wave * 32 + li;
        const bf16* wp = p.w2 + (int64_t)(li < p.n2 ? li : 0) * p.ldw2 + lh * 8;
        const char* tp = smem + row * EP + lh * 16;
        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        floatx16 a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll 4
        for (int ks = 0; ks < BN / 16; ++ks)
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(li < p.n2 ? *(const bf16x8*)(wp + ks * 16) : zero, *(const bf16x8*)(tp + ks * 32), a2, 0, 0, 0);
        if (m0 + row < p.M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 8 * g + 4 * lh;
                if (c0 < p.n2) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)a2[4 * g + e];
                    *(bf16x4*)(p.out2 + (m0 + row) * p.ldo2 + c0) = o;
                }
            }
        }
        return;
    }
    float sx[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sx[e] = 0.f; sq[e] = 0.f; }
#pragma unroll 4
    for (int ps = 0; ps < BM / RPP; ++ps) {
        int row = ps * RPP + erow;
        if (m0 + row < p.M && ncol_ok && (!(p.dbg & 1) || acc[0][0][0] == 12345.f)) {
            const bf16x8 v = *(const bf16x8*)(smem + row * EP + ech * 16);
            nt_store_chunk(p, C, m0 + row, n0 + ech * 8, v);
            if (p.stat_ws) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sx[e] += f; sq[e] = fmaf(f, f, sq[e]); }
            }
        }
    }
    if (p.stat_ws) nt_tile_stats<BN, RPP, 512>(p, smem, sx, sq, erow, ech, m0 / BM, n0, tid);
}

__global__ __launch_bounds__(512) void gemm_nt_bf16_256_kernel(FastGemm p) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int IMG = BM * BK * 2;                  // one operand image: 32 KiB
    constexpr int STAGE = 2 * IMG;
    constexpr int EP = BN * 2 + 16;                   // epilogue image pitch (bytes)
    constexpr int SMEM = 2 * STAGE > BM * EP ? 2 * STAGE : BM * EP;
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap_big(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / p.ntn) * BM;
    const int n0 = (wg % p.ntn) * BN;
    const int nk = (int)(p.K / BK);

    // LDS-DMA sources: piece q = wave*4 + i covers image slots q*64 .. q*64+63; this lane fills slot q*64 + lane
    const bf16* pa[4]; const bf16* pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (wave * 4 + i) * 4 + (lane >> 4);
        const int x = (lane & 15) ^ (R & 15);
        const int row = 2 * R + (x >> 3), chunk = x & 7;
        int64_t ra = m0 + row; if (ra >= p.M) ra = p.M - 1;
        int rb = n0 + row; if (rb >= p.N) rb = p.N - 1;
        pa[i] = p.A + ra * p.lda + chunk * 8;
        pb[i] = p.B + (int64_t)rb * p.ldb + chunk * 8;
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(pa[i] + kt * BK), (lds_void*)(st + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(pb[i] + kt * BK), (lds_void*)(st + IMG + i * 1024), 16, 0, 0);
    };

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // tile kt has landed; nobody still reads the other stage
        if (kt + 1 < nk && !(p.dbg & 2)) issue(kt + 1);
        const bf16* As = (const bf16*)(smem + (kt & 1) * STAGE);
        const bf16* Bs = (const bf16*)(smem + (kt & 1) * STAGE + IMG);
        if (p.dbg & 4) continue;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(As + big_slot(wm * 64 + a * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = *(const bf16x8*)(Bs + big_slot(wn * 128 + b * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)        // D[i = n][j = m]: the lane is an output ROW, its registers run along n
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
    big_nt_epilogue(acc, p, smem, m0, n0, tid, wave, wm, wn, li, lh);
}

// Wide form for outputs whose width is a multiple of 320 but not of 256 (the N = 320 projections of MiT stage 3): 256 x 320 tiles, the
// 8 waves as 4 (m) x 2 (n) with 64 x 160 each (10 accumulators of 32 x 32).  The whole output row panel is ONE tile, so the activation
// operand is read once (the 128-column tile kernel these shapes ran on read it three times and moves 64 flop per byte into LDS; this
// tile moves 142).  Two stages of 72 KiB; the epilogue is per wave through 2 KiB strips (an [m][n] image of the tile would not fit).
template <int TB>
__global__ __launch_bounds__(512) void gemm_nt_bf16_256w_kernel(FastGemm p) {
    constexpr int BM = 256, BN = 64 * TB, BK = 64;
    constexpr int IMGA = BM * BK * 2, IMGB = BN * BK * 2, STAGE = IMGA + IMGB;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap_big(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / p.ntn) * BM;
    const int n0 = (wg % p.ntn) * BN;
    const int nk = (int)(p.K / BK);

    const bf16* pa[4]; const bf16* pb[TB];
    auto src = [&](int q, int& row, int& chunk) {      // piece q of an image: this lane's slot -> (row, 16-byte chunk)
        const int R = q * 4 + (lane >> 4);
        const int x = (lane & 15) ^ (R & 15);
        row = 2 * R + (x >> 3); chunk = x & 7;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, chunk; src(wave * 4 + i, row, chunk);
        int64_t ra = m0 + row; if (ra >= p.M) ra = p.M - 1;
        pa[i] = p.A + ra * p.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        int row, chunk; src(wave * TB + i, row, chunk);
        pb[i] = p.B + (int64_t)(n0 + row) * p.ldb + chunk * 8;
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(pa[i] + kt * BK), (lds_void*)(st + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < TB; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(pb[i] + kt * BK), (lds_void*)(st + IMGA + (wave * TB + i) * 1024), 16, 0, 0);
    };

    floatx16 acc[2][TB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // tile kt has landed; nobody still reads the other stage
        if (kt + 1 < nk) issue(kt + 1);
        const bf16* As = (const bf16*)(smem + (kt & 1) * STAGE);
        const bf16* Bs = (const bf16*)(smem + (kt & 1) * STAGE + IMGA);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[TB];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(As + big_slot(wm * 64 + a * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb[b] = *(const bf16x8*)(Bs + big_slot(wn * 32 * TB + b * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();                                   // the stages are free: every wave takes 2 KiB of them for its epilogue strip
    nt_epilogue_wave<2, TB>(acc, p, smem + wave * 2048, m0, n0, wm * 64, wn * 32 * TB, lane);
}

bool ss_wide_nt_ok(const FastGemm& p) {
    if (p.N % 320 || p.N % 256 == 0 || p.K % 64 || p.K < 256 || p.stat_ws || p.w2 || p.pm.side) return false;
    return cdiv(p.M, 256) * (p.N / 320) >= 128;
}

int ss_wide_nt_launch(FastGemm p, hipStream_t st) {
    p.ntn = p.N / 320; p.dbg = 0;
    hipLaunchKernelGGL(gemm_nt_bf16_256w_kernel<5>, dim3((unsigned)(p.ntn * cdiv(p.M, 256))), dim3(512), 0, st, p);
    SS_LAUNCH_CHECK();
    return 0;
}

// Persistent form: one workgroup per CU walks a run of tiles of its XCD.  The last k-iteration of a tile already issues the first
// k-tile of the NEXT tile into the free stage, and the epilogue is per wave through 4 KiB of LDS of its own (nt_epilogue_wave), so
// a tile's drain (bias / activation / residual, 128 KiB of stores) overlaps the next tile's fill instead of an idle chip; no
// workgroup ever waits for the launch of its successor.
__global__ __launch_bounds__(512) void gemm_nt_bf16_256p_kernel(FastGemm p, int ntiles, int pfd) {
    constexpr int BK = 64;
    constexpr int IMG = 256 * BK * 2;
    constexpr int STAGE = 2 * IMG;
    // three arrays, so that alias analysis keeps the compiler's own LDS-DMA waits (vmcnt(0) before an LDS access that may alias a
    // fill in flight) away from the epilogue and from the prefetch target
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    __shared__ __attribute__((aligned(1024))) char epi[8 * 2048];
    __shared__ __attribute__((aligned(256))) char pfl[8 * 256];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int nk = (int)(p.K / BK);
    // tiles of XCD x (blocks x, x + 8, ... share its L2): a contiguous run, walked by the XCD's workgroups side by side
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3, S = gridDim.x >> 3;
    const int q = ntiles >> 3, r8 = ntiles & 7;
    const int cnt = q + (x < r8 ? 1 : 0), start = x < r8 ? x * (q + 1) : r8 * (q + 1) + (x - r8) * q;
    if (s >= cnt) return;

    const bf16* pa[4]; const bf16* pb[4];
    auto setptr = [&](int tile) {
        const int64_t m0 = (int64_t)(tile / p.ntn) * 256;
        const int n0 = (tile % p.ntn) * 256;
        int ln = lane;
        asm volatile("" : "+v"(ln));                   // recomputed per tile: hoisted, the eight 64-bit row offsets were spilled
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = (wave * 4 + i) * 4 + (ln >> 4);
            const int xx = (ln & 15) ^ (R & 15);
            const int row = 2 * R + (xx >> 3), ch = (xx & 7) * 8;
            int64_t ra = m0 + row; if (ra >= p.M) ra = p.M - 1;
            pa[i] = p.A + ra * p.lda + ch;
            pb[i] = p.B + (int64_t)(n0 + row) * p.ldb + ch;
        }
    };
    const unsigned stage_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 4096);
    const unsigned pf_lds = __builtin_amdgcn_readfirstlane(lds_addr(pfl) + wave * 256);
    auto issue = [&](int stage, int kt) {
        const unsigned st = stage_lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b128(pa[i] + kt * BK, st + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b128(pb[i] + kt * BK, st + IMG + i * 1024);
    };

    // L2 prefetch of the activation operand, PFD k-tiles ahead of the fills: one 4-byte load per lane and k-iteration touches both
    // 64-byte halves of the 256 rows' next 128-byte segments.  A fill that finds its lines in L2 completes in ~0.6 us instead of the
    // ~2.5 us of an HBM round trip, and with ONE k-tile in flight per workgroup that round trip was the k-iteration (39 % of the MFMA
    // rate at any K).  The value goes nowhere (an LDS-DMA dword per lane into 256 bytes nobody reads: no register to keep alive); loads
    // return in order and the iteration top waits for all but the newest vector-memory operation -- that prefetch.
    int pf_t = s, pf_kt = 0;
    const int pf_row = tid >> 1, pf_half = (tid & 1) * 32;
    const bf16* pf_base;
    auto pf_settile = [&](int t) {
        int64_t ra = (int64_t)((start + t) / p.ntn) * 256 + pf_row; if (ra >= p.M) ra = p.M - 1;
        pf_base = p.A + ra * p.lda + pf_half;
    };
    auto prefetch = [&]() {
        dma_b32(pf_base + pf_kt * BK, pf_lds);
    };
    auto pf_advance = [&]() {
        if (pf_kt + 1 < nk) ++pf_kt;
        else if (pf_t + S < cnt) { pf_t += S; pf_kt = 0; pf_settile(pf_t); }
    };
    // (tried: starting the workgroups of an XCD 0 .. 7 x 3.4 us out of phase, so that drains and k-loops of different CUs interleave
    // instead of all draining at once -- no gain beyond the start delay itself on any shape: lockstep is not what the tile overhead is)
    pf_settile(s);
    for (int d = 0; d < pfd; ++d) { pf_advance(); prefetch(); }
    setptr(start + s);
    issue(0, 0);
    prefetch();
    int it = 0;
    for (int t = s; t < cnt; t += S) {
        const int tile = start + t;
        const int64_t m0 = (int64_t)(tile / p.ntn) * 256;
        const int n0 = (tile % p.ntn) * 256;
        floatx16 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++it) {
            // k-tile `it` has landed (everything but the newest prefetch is complete) and nobody still reads the other stage.  A bare
            // s_barrier: __syncthreads() carries a release fence, i.e. vmcnt(0), which would wait for the prefetch's HBM round trip
            asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            const bool last = kt + 1 == nk;
            if (!last) issue((it + 1) & 1, kt + 1);
            else if (t + S < cnt) { setptr(tile + S); issue((it + 1) & 1, 0); }
            if (!last || t + S < cnt) { pf_advance(); prefetch(); }
            const bf16* As = (const bf16*)(smem + (it & 1) * STAGE);
            const bf16* Bs = (const bf16*)(smem + (it & 1) * STAGE + IMG);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[2], fb[4];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(As + big_slot(wm * 64 + a * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
                for (int b = 0; b < 4; ++b) fb[b] = *(const bf16x8*)(Bs + big_slot(wn * 128 + b * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
            }
        }
        nt_epilogue_wave<2, 4>(acc, p, epi + wave * 2048, m0, n0, wm * 64, wn * 128, lane);
    }
}

// Deeper activation pipeline: the loop above keeps ONE k-tile in flight, and the activation operand streams from HBM (every A k-tile is new
// data, ~2 us away under load) while a k-tile is multiplied in ~1.3 us: the k-iteration was the round trip, fills, products and stores
// simply added up (tools/bench_gemm_parts.py: 262144 x 512 x 512 = 44 us of launch + first fill + drain, 55 us of fills, 83 us of
// products, 29 us of stores, 198 us in total).  Here the A operand has THREE 32 KiB stages (two k-tiles in flight: 64 KiB per CU, 16 MB
// chip-wide, what 8 TB/s x 2 us needs) and the weight operand, which comes from L2, keeps two: 5 x 32 KiB = all 160 KiB of LDS.
// Fills are issued by hand (asm, so the compiler adds no waits of its own) with counted waits: at the top of iteration kt only the four
// newest fills of each wave -- A(kt+1) -- may still be in flight.
__global__ __launch_bounds__(512) void gemm_nt_bf16_256a_kernel(FastGemm p) {
    constexpr int BK = 64;
    constexpr int IMG = 256 * BK * 2;
    constexpr int SMEM = 5 * IMG;                      // A0 A1 A2 B0 B1; the epilogue image (256 x 528 bytes) reuses it
    static_assert(256 * (256 * 2 + 16) <= SMEM, "epilogue image must fit");
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap_big(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / p.ntn) * 256;
    const int n0 = (wg % p.ntn) * 256;
    const int nk = (int)(p.K / BK);

    const bf16* pa[4]; const bf16* pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (wave * 4 + i) * 4 + (lane >> 4);
        const int x = (lane & 15) ^ (R & 15);
        const int row = 2 * R + (x >> 3), chunk = x & 7;
        int64_t ra = m0 + row; if (ra >= p.M) ra = p.M - 1;
        int rb = n0 + row; if (rb >= p.N) rb = p.N - 1;
        pa[i] = p.A + ra * p.lda + chunk * 8;
        pb[i] = p.B + (int64_t)rb * p.ldb + chunk * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 4096);
    auto issue_a = [&](int kt, int sa) {
        const unsigned st = lds0 + sa * IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b128(pa[i] + kt * BK, st + i * 1024);
    };
    auto issue_b = [&](int kt) {
        const unsigned st = lds0 + (3 + (kt & 1)) * IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b128(pb[i] + kt * BK, st + i * 1024);
    };

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue_a(0, 0); issue_b(0);
    if (nk > 1) issue_a(1, 1);
    int sa = 0;                                        // A stage of k-tile kt
    for (int kt = 0; kt < nk; ++kt) {
        // A(kt) and B(kt) have landed (fills complete in order; A(kt+1), the wave's four newest, may still be on its way), and
        // nobody still reads the stages about to be refilled.  A bare s_barrier: __syncthreads() would wait for vmcnt(0)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) issue_b(kt + 1);
        if (kt + 2 < nk) issue_a(kt + 2, sa == 0 ? 2 : sa - 1);          // (kt + 2) % 3
        const bf16* As = (const bf16*)(smem + sa * IMG);
        const bf16* Bs = (const bf16*)(smem + (3 + (kt & 1)) * IMG);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(As + big_slot(wm * 64 + a * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = *(const bf16x8*)(Bs + big_slot(wn * 128 + b * 32 + li, ks * 2 + lh) * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
        sa = sa == 2 ? 0 : sa + 1;
    }
    __syncthreads();
    big_nt_epilogue(acc, p, smem, m0, n0, tid, wave, wm, wn, li, lh);
}

static int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        n = v & ~7;
    }
    return n;
}

bool ss_big_nt_ok(const FastGemm& p) {
    // MFMA-bound shapes only: enough k-tiles to amortise the tile's fill and drain, whole 256-column tiles, a full chip
    if (p.K % 64 || p.K < 128 || p.N % 256) return false;
    // K = 128 (two k-tiles) only in the persistent form, where the next tile's fill overlaps the drain: 262144 x 512 x 128 runs in
    // 84 us there against 105 us on the 128-tile kernel (half the bytes through L1 per flop); with N = 256 or K = 64 it does not pay
    if (p.K < 256 && !(cdiv(p.M, 256) * (p.N / 256) >= 4 * cu_count() && p.N >= 512)) return false;
    return cdiv(p.M, 256) * (p.N / 256) >= 256;
}

int ss_big_nt2_launch(FastGemm p, hipStream_t st) {
    p.ntn = 1; p.dbg = 0;
    hipLaunchKernelGGL(gemm_nt_bf16_256_kernel, dim3((unsigned)cdiv(p.M, 256)), dim3(512), 0, st, p);
    SS_LAUNCH_CHECK();
    return 0;
}

int ss_big_nt_launch(FastGemm p, hipStream_t st) {
    // SS_GEMM_BIG: 1 = one tile per workgroup with two stages, 2 = persistent, default 3 = one tile per workgroup with the deep A pipeline.
    // Whole training step on one box, back to back: 58.0 ms persistent, 57.6 ms two stages, 57.5 ms deep pipeline (tools/bench_gemm_modes.py
    // has the shapes in isolation: the persistent form loses 5 - 10 % there on N >= 512, K = 512)
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("SS_GEMM_BIG"); mode = e && e[0] == '1' ? 1 : (e && e[0] == '2' ? 2 : 3); }
    { const char* e = getenv("SS_GEMM_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.ntn = p.N / 256;
    const int ntiles = (int)(p.ntn * cdiv(p.M, 256));
    // measured inside the training step (not in isolation): the persistent form wins 4-11 % where a CU walks >= 4 tiles of a
    // multi-column-tile output with a short k-loop, and loses up to 10 % on single-column-tile, one-tile-per-CU or long-K launches
    if (mode == 2 && ntiles >= 4 * cu_count() && p.N >= 512 && p.K <= 1024 && !p.stat_ws) {      // tile statistics: one tile per workgroup
        static int pfd = -1;                               // prefetch distance in k-tiles (SS_GEMM_PFD; measured: 0 .. 5 all within 1 %)
        if (pfd < 0) { const char* e = getenv("SS_GEMM_PFD"); pfd = e ? atoi(e) : 2; if (pfd < 0 || pfd > 16) pfd = 2; }
        hipLaunchKernelGGL(gemm_nt_bf16_256p_kernel, dim3((unsigned)cu_count()), dim3(512), 0, st, p, ntiles, pfd);
        SS_LAUNCH_CHECK();
        return 0;
    }
    if (mode == 3 && p.K >= 192) {
        hipLaunchKernelGGL(gemm_nt_bf16_256a_kernel, dim3((unsigned)(p.ntn * cdiv(p.M, 256))), dim3(512), 0, st, p);
        SS_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(gemm_nt_bf16_256_kernel, dim3((unsigned)(p.ntn * cdiv(p.M, 256))), dim3(512), 0, st, p);
    SS_LAUNCH_CHECK();
    return 0;
}
