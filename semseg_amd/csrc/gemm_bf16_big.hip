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
#pragma unroll 4
    for (int ps = 0; ps < BM / RPP; ++ps) {
        int row = ps * RPP + erow;
        if (m0 + row < p.M && ncol_ok && (!(p.dbg & 1) || acc[0][0][0] == 12345.f))
            nt_store_chunk(p, C, m0 + row, n0 + ech * 8, *(const bf16x8*)(smem + row * EP + ech * 16));
    }
}

bool ss_big_nt_ok(const FastGemm& p) {
    // MFMA-bound shapes only: enough k-tiles to amortise the tile's fill and drain, whole 256-column tiles, a full chip
    if (p.K % 64 || p.K < 256 || p.N % 256) return false;
    return cdiv(p.M, 256) * (p.N / 256) >= 256;
}

int ss_big_nt_launch(FastGemm p, hipStream_t st) {
    { const char* e = getenv("SS_GEMM_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.ntn = p.N / 256;
    hipLaunchKernelGGL(gemm_nt_bf16_256_kernel, dim3((unsigned)(p.ntn * cdiv(p.M, 256))), dim3(512), 0, st, p);
    SS_LAUNCH_CHECK();
    return 0;
}
