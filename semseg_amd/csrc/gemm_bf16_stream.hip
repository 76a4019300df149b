// Streaming NT GEMM for the thin projections of the token path: C[M, N] = epilogue(A[M, K] . B[N, K]^T) with K = 64 or 128 and
// N <= 256 per workgroup (MiT stage-1 fc1 / the data gradient of fc2: M = 1048576, N = 256, K = 64 -- 134 MB in, 537 MB out, 34 flop per byte).
//
// Such a launch is a copy with a matrix product in the middle, and the tile kernels run it as load -> barrier -> MFMA -> barrier ->
// LDS image -> barrier -> store per workgroup, at the speed of the slowest phase of each tile.  Here there is no workgroup barrier
// after the weights have been staged: the whole B operand (N x K, <= 64 KB) sits in LDS for the lifetime of the workgroup, and every
// WAVE is a pipeline of its own over 32-row tiles of A -- the 16-byte fragment loads of tile t+1 are in flight (they go straight
// from global memory into the MFMA operand registers: a lane is a row, its 16 bytes are one k-chunk, four loads cover the row's
// 128-byte line) while tile t is multiplied and its 32 x N result leaves through the wave's own LDS strip (32 rows x 128 columns
// at a time; bias and row scale only: other epilogues stay on the tile kernels).  Stores are asynchronous, so a wave's stores of
// tile t overlap its MFMAs of tile t+1 and every other wave's loads.
#include "gemm_fast.h"

namespace {

__device__ __forceinline__ int st_slot(int row, int chunk) {      // 16-byte slot of (row, k-chunk) inside a [rows][64] image (gemm_bf16.hip's nt_slot)
    const int R = row >> 1;
    return R * 16 + ((((row & 1) << 3) | chunk) ^ (R & 15));
}

// Plain epilogue (bias, row scale) through a wave-private [32 rows][HB * 32 columns] LDS strip: whole 128-byte lines leave per row
// (nt_epilogue_wave writes 64-byte halves of a line from two different instructions).
template <int NT, int HB>
__device__ __forceinline__ void stream_epilogue(floatx16 (&acc)[1][NT], const FastGemm& p, char* wl, int64_t m0, int n0, int lane) {
    constexpr int PITCH = HB * 64 + 16, CPR = HB * 4, RPP = 64 / CPR;      // bytes per strip row; 16-byte chunks per row; rows per pass
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m = m0 + li;
    float rs = 1.f;
    if (p.rowscale && m < p.M) rs = p.rowscale[m / p.rows_per_scale];
    bf16* C = (bf16*)p.C;
    const int rr = lane / CPR, cc = lane % CPR;
#pragma unroll
    for (int half = 0; half < NT / HB; ++half) {
#pragma unroll
        for (int bl = 0; bl < HB; ++bl) {
            const int b = half * HB + bl;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = bl * 32 + 8 * g + 4 * lh;
                floatx4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *(const floatx4*)(p.bias + n0 + half * HB * 32 + nl);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)((acc[0][b][4 * g + e] + bv[e]) * rs);
                *(bf16x4*)(wl + li * PITCH + nl * 2) = o;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int r = ps * RPP + rr;
            const bf16x8 v = *(const bf16x8*)(wl + r * PITCH + cc * 16);
            if (m0 + r < p.M) *(bf16x8*)(C + (m0 + r) * p.ldc + n0 + half * HB * 32 + cc * 8) = v;
        }
        asm volatile("" ::: "memory");
    }
}

// N = 32 * NT columns per workgroup (blockIdx.y selects the column block), K = 64 * KT, WAVES waves per workgroup
template <int NT, int KT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void gemm_nt_stream_kernel(FastGemm p, int ntiles) {
    constexpr int N = 32 * NT, IMG = N * 64;       // elements of one k-tile image of B
    constexpr int HB = NT < 4 ? NT : 4, STRIP = 32 * (HB * 64 + 16);
    __shared__ __attribute__((aligned(16))) bf16 Bs[KT * IMG];
    __shared__ __attribute__((aligned(16))) char epi[WAVES * STRIP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * N;
    for (int c = tid; c < KT * N * 8; c += 64 * WAVES) {
        const int kt = c / (N * 8), rc = c - kt * (N * 8), row = rc >> 3, ch = rc & 7;
        *(bf16x8*)(Bs + kt * IMG + st_slot(row, ch) * 8) = *(const bf16x8*)(p.B + (int64_t)(n0 + row) * p.ldb + kt * 64 + ch * 8);
    }
    __syncthreads();
    const int stride = gridDim.x * WAVES;
    int t = blockIdx.x * WAVES + wave;
    if (t >= ntiles) return;
    bf16x8 fn[KT * 4];
    auto aload = [&](int tile) {
        int64_t r = (int64_t)tile * 32 + li; if (r >= p.M) r = p.M - 1;
        const bf16* src = p.A + r * p.lda + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KT * 4; ++ks) fn[ks] = *(const bf16x8*)(src + ks * 16);
    };
    aload(t);
    for (; t < ntiles; t += stride) {
        bf16x8 fa[KT * 4];
#pragma unroll
        for (int ks = 0; ks < KT * 4; ++ks) fa[ks] = fn[ks];
        if (t + stride < ntiles) aload(t + stride);
        floatx16 acc[1][NT];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT * 4; ++ks) {
            const bf16* img = Bs + (ks >> 2) * IMG;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const bf16x8 fb = *(const bf16x8*)(img + st_slot(b * 32 + li, (ks & 3) * 2 + lh) * 8);
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa[ks], acc[0][b], 0, 0, 0);
            }
        }
        stream_epilogue<NT, HB>(acc, p, epi + wave * STRIP, (int64_t)t * 32, n0, lane);
    }
}

int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        n = v;
    }
    return n;
}

}  // namespace

// shapes: K = 64 (the kernel itself also takes K = 128), N in {64, 128, 256} (whole operand in LDS), enough 32-row tiles to give every
// wave of the chip several;
// plain epilogues only (bias, row scale: no activation, residual, tile statistics, second product, patch addressing, second output)
bool ss_stream_nt_ok(const FastGemm& p) {
    static int mode = -1;                                  // SS_GEMM_STREAM=0 switches the kernel off (A/B measurements)
    if (mode < 0) { const char* e = getenv("SS_GEMM_STREAM"); mode = e ? atoi(e) : 1; }
    if (!mode || p.stat_ws || p.w2 || p.C2 || p.pm.side || p.res || p.act) return false;
    // measured inside the training step: M = 1048576, N = 256, K = 64: 156 us against 202 us on the 128-row tile kernel (4.3 TB/s of
    // in + out); N = 64: 58 against 66 us; K = 128 at N = 128 (M = 262144): 38.6 against 36.6 us, so that shape stays on the tile kernels.
    // N = 512, K = 128 (MiT stage-2 fc1 and the data gradient of fc2, M = 262144): two column blocks of 256 (blockIdx.y), 8 waves per
    // workgroup around one 64 KB weight block: 95.7 against 100.7 us on the persistent 256-tile kernel (same box)
    const bool wide = p.K == 128 && p.N == 512;
    if (!wide && (p.K != 64 || !(p.N == 64 || p.N == 128 || p.N == 256))) return false;
    return p.M >= 32 * 8 * 4 * (int64_t)cu_count();
}

int ss_stream_nt_launch(FastGemm p, hipStream_t st) {
    const int ntiles = (int)cdiv(p.M, 32);
    // workgroups per CU by register use: 240 VGPRs at N = 256 (2), ~164 at N = 128 (3), 110 at N = 64 (4)
    const int per_cu = p.N == 256 ? 2 : (p.N == 64 ? 4 : 3);
    const dim3 grid((unsigned)(per_cu * cu_count())), block(256);
#define SS_STREAM(NT, KT) hipLaunchKernelGGL((gemm_nt_stream_kernel<NT, KT, 4>), grid, block, 0, st, p, ntiles)
    if (p.K == 128 && p.N == 512)
        hipLaunchKernelGGL((gemm_nt_stream_kernel<8, 2, 8>), dim3((unsigned)cu_count(), 2), dim3(512), 0, st, p, ntiles);
    else if (p.N == 64) SS_STREAM(2, 1);
    else if (p.N == 128) SS_STREAM(4, 1);
    else SS_STREAM(8, 1);
#undef SS_STREAM
    SS_LAUNCH_CHECK();
    return 0;
}
