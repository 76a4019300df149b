// Skinny bf16 GEMMs of the 6-class classifier: one side of the product is <= 8 wide, so there is nothing for
// the MFMA tiles to chew on and the launch is bound by streaming the wide operand through HBM once.
//
//   gemm_nt_smallk_kernel : C[M,N] = A[M,K] . B[N,K]^T (+ bias), K <= 8      (input gradient dX = dY . W: writes M x N)
//   gemm_tn_smallm_kernel : slab[z][M,N] = sum_r A[r,M]^T . B[r,N], M <= 8   (weight gradient dW = dY^T . X: reads R x N)
//
// Both: a lane owns 8 consecutive columns of the wide side (one 16-byte access per row), a wave walks rows; the <= 8 values of
// the narrow operand's row are fetched by 8 lanes each for 8 rows at a time and broadcast with v_readlane.
#include "common.h"
#include <stdlib.h>
#include "gemm_fast.h"

__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__global__ __launch_bounds__(256) void gemm_nt_smallk_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                                             int64_t ldb, bf16* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                             const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.y * 64 + lane) * 8;
    const bool active = col < N;
    float w[8][8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bv[j] = (active && bias) ? bias[col + j] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k][j] = (active && k < K) ? (float)B[(int64_t)(col + j) * ldb + k] : 0.f;
    }
    const int64_t groups = cdiv(M, 8);
    const int rr = lane >> 3, kk = lane & 7;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < groups; g += (int64_t)gridDim.x * 4) {
        const int64_t r0 = g * 8;
        const float av = (r0 + rr < M && kk < K) ? (float)A[(r0 + rr) * lda + kk] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (r0 + i >= M) break;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = bv[j];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float a = lane_bcast(av, i * 8 + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(a, w[k][j], acc[j]);
            }
            if (active) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
                *(bf16x8*)(C + (r0 + i) * ldc + col) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_smallm_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                                             int64_t ldb, float* __restrict__ slabs, int M, int N, int64_t R,
                                                             int64_t rps) {
    __shared__ float red[4][64][65];                   // [wave][m*8+j][lane] (+1: the gather below walks the middle index)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.y * 64 + lane) * 8;
    const bool active = col < N;
    const int64_t rbeg = (int64_t)blockIdx.x * rps;
    const int64_t rend = rbeg + rps < R ? rbeg + rps : R;
    float acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    const int rr = lane >> 3, mm = lane & 7;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r0 = rbeg + wave * 8; r0 < rend; r0 += 32) {
        const float av = (r0 + rr < rend && mm < M) ? (float)A[(r0 + rr) * lda + mm] : 0.f;
        bf16x8 b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (active && r0 + i < rend) ? *(const bf16x8*)(B + (r0 + i) * ldb + col) : zero;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float bf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = (float)b[i][j];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float a = lane_bcast(av, i * 8 + m);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m][j] = fmaf(a, bf[j], acc[m][j]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][m * 8 + j][lane] = acc[m][j];
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * M * N;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int mj = e & 63, ln = e >> 6;                // consecutive threads -> consecutive j, m of one lane's 8 columns
        const int m = mj >> 3, j = mj & 7;
        const int c = (blockIdx.y * 64 + ln) * 8 + j;
        if (m < M && c < N) slab[(int64_t)m * N + c] = ((red[0][mj][ln] + red[1][mj][ln]) + red[2][mj][ln]) + red[3][mj][ln];
    }
}

// ------------------------------------------------------------------ both sides narrow (K, N in {8, 16, 32}; the 8-channel maps of the
// RefinerHead at 256 x 256, modules/heads.py:173-269): a THREAD owns a row.  The kernels above give a lane eight columns of the wide side
// and walk rows one at a time, which with N = 8 leaves 63 of 64 lanes idle (1048576 x 8 x 8: 125 us for 34 MB).  Weights sit in LDS as
// fp32 and are read as broadcasts.
template <typename T, int K, int N>
__global__ __launch_bounds__(256) void gemm_nt_rows_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                           T* __restrict__ C, int64_t ldc, int64_t M, const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) float w[N][K];
    __shared__ float bs[N];
    for (int i = threadIdx.x; i < N * K; i += 256) w[i / K][i % K] = (float)B[(int64_t)(i / K) * ldb + i % K];
    for (int i = threadIdx.x; i < N; i += 256) bs[i] = bias ? bias[i] : 0.f;
    __syncthreads();
    // RPT rows per thread (256 apart, so every load and store instruction is still row-contiguous across the wave): a weight read from
    // LDS (a broadcast of four) serves RPT rows -- one row per thread made the kernel LDS-bound at K = 32 (K N / 4 reads per row)
    constexpr int RPT = K >= 16 ? 2 : 1;
    for (int64_t r0 = (int64_t)blockIdx.x * 256 * RPT + threadIdx.x; r0 < M; r0 += (int64_t)gridDim.x * 256 * RPT) {
        float a[RPT][K];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = r0 + u * 256 < M ? r0 + u * 256 : M - 1;
#pragma unroll
            for (int c = 0; c < K / 8; ++c) {
                const F8 v = load8(A + row * lda + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[u][c * 8 + e] = v.v[e];
            }
        }
#pragma unroll
        for (int c = 0; c < N / 8; ++c) {
            F8 o[RPT];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc[RPT];
#pragma unroll
                for (int u = 0; u < RPT; ++u) acc[u] = bs[c * 8 + j];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float wk = w[c * 8 + j][k];
#pragma unroll
                    for (int u = 0; u < RPT; ++u) acc[u] = fmaf(a[u][k], wk, acc[u]);
                }
#pragma unroll
                for (int u = 0; u < RPT; ++u) o[u].v[j] = acc[u];
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u)
                if (r0 + u * 256 < M) store8(C + (r0 + u * 256) * ldc + c * 8, o[u]);
        }
    }
}

// dtype: SS_BF16 or SS_F32 (operands and output alike)
// SS_GEMM_ROWS=0 (read per call): the narrow kernels off (A/B runs)
static bool rows_enabled() { const char* e = getenv("SS_GEMM_ROWS"); return !(e && e[0] == '0'); }

bool ss_rows_nt_ok(int dtype, const void* A, int64_t lda, const void* C, int64_t ldc, int64_t M, int64_t N, int64_t K) {
    if (!rows_enabled()) return false;
    auto narrow = [](int64_t v) { return v == 8 || v == 16 || v == 32; };
    const int al = dtype == SS_F32 ? 4 : 8;            // elements per 16 bytes
    // K N <= 256: beyond that the compiler keeps the whole weight matrix in registers and spills (16 x 32: 118 .. 300 registers).
    // fp32 16 x 16 stays on the generic kernel: only the fp32 parity runs reach it (TopFormer's 16-channel layers are bf16 in production), and
    // the TopFormer training goldens are chaotic enough that the rounding of this one product decides whether the B step lands 0.5 x or 3 x
    // the reference's own fp32 deviation from float64 -- reversing this kernel's summation order alone moved it from 3 x back to 0.7 x
    // (tools/topformer_train_gradnorm_spread.py; both kernels are within 1.6e-7 of float64, tools/narrow_gemm_precision.py)
    if (dtype == SS_F32 && K == 16 && N == 16) return false;
    return (dtype == SS_BF16 || dtype == SS_F32) && narrow(K) && narrow(N) && K * N <= 256 && M >= 4096 && lda % al == 0 && ldc % al == 0 && aligned16(A) && aligned16(C);
}

int ss_rows_nt_launch(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      const float* bias, hipStream_t st) {
    const int64_t need = cdiv(M, 256 * (K >= 16 ? 2 : 1));
    const dim3 grid((unsigned)(need < 8192 ? need : 8192)), block(256);
#define SS_ROWS_NT(K, N) SS_DISPATCH(dtype, hipLaunchKernelGGL((gemm_nt_rows_kernel<T, K, N>), grid, block, 0, st, (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, bias))
    switch ((int)K * 100 + (int)N) {
        case 808: SS_ROWS_NT(8, 8); break;    case 816: SS_ROWS_NT(8, 16); break;   case 832: SS_ROWS_NT(8, 32); break;
        case 1608: SS_ROWS_NT(16, 8); break;  case 1616: SS_ROWS_NT(16, 16); break; case 3208: SS_ROWS_NT(32, 8); break;
        default: return ss_fail(-1, "ss_rows_nt_launch: K=%d N=%d not a narrow pair with K N <= 256", (int)K, (int)N);
    }
#undef SS_ROWS_NT
    SS_LAUNCH_CHECK();
    return 0;
}

// slab[z][8, N] = sum_r A[r, 8]^T . B[r, N] for N = 8 * CPL <= 32: a lane owns one ROW's eight-column chunk of B (256 / CPL rows per
// workgroup step) and that row's eight A values; the lanes of a workgroup that share a chunk are added up through LDS in a fixed order.
template <typename T, int CPL>
__global__ __launch_bounds__(256) void gemm_tn_smallmn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ slabs, int64_t R, int64_t rps) {
    constexpr int RPB = 256 / CPL, N = 8 * CPL;        // rows per workgroup step
    __shared__ float red[256][65];
    const int tid = threadIdx.x, chunk = tid % CPL, rsub = tid / CPL;
    const int64_t rbeg = (int64_t)blockIdx.x * rps;
    const int64_t rend = rbeg + rps < R ? rbeg + rps : R;
    float acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    for (int64_t r0 = rbeg; r0 < rend; r0 += 2 * RPB) {           // two rows per lane in flight
        F8 av[2], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = r0 + u * RPB + rsub;
            const bool ok = r < rend;
            av[u] = ok ? load8(A + r * lda) : zero8();
            bv[u] = ok ? load8(B + r * ldb + chunk * 8) : zero8();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m][j] = fmaf(av[u].v[m], bv[u].v[j], acc[m][j]);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][m * 8 + j] = acc[m][j];
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * 8 * N;
    for (int e = tid; e < CPL * 64; e += 256) {
        const int ch = e / 64, mj = e % 64, m = mj >> 3, j = mj & 7;
        float sum = 0.f;
        for (int t = 0; t < RPB; ++t) sum += red[t * CPL + ch][mj];
        slab[(int64_t)m * N + ch * 8 + j] = sum;
    }
}

// both sides narrow (M = 8, N in {8, 16, 32}), bf16 or fp32 operands
bool ss_smallmn_tn_ok(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N) {
    if (!rows_enabled()) return false;
    const int al = dtype == SS_F32 ? 4 : 8;
    return (dtype == SS_BF16 || dtype == SS_F32) && M == 8 && (N == 8 || N == 16 || N == 32) && lda % al == 0 && ldb % al == 0 && aligned16(A) && aligned16(B);
}

int ss_smallmn_tn_launch(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* slabs, int64_t N, int64_t R, int64_t rps, int nsplit,
                         hipStream_t st) {
    const dim3 grid((unsigned)nsplit), block(256);
#define SS_SMN(CPL) SS_DISPATCH(dtype, hipLaunchKernelGGL((gemm_tn_smallmn_kernel<T, CPL>), grid, block, 0, st, (const T*)A, lda, (const T*)B, ldb, slabs, R, rps))
    if (N == 8) SS_SMN(1); else if (N == 16) SS_SMN(2); else SS_SMN(4);
#undef SS_SMN
    SS_LAUNCH_CHECK();
    return 0;
}

bool ss_skinny_nt_ok(const FastGemm& p) {
    return p.K <= 8 && p.N % 8 == 0 && p.ldc % 8 == 0 && aligned16(p.C) && !p.res && !p.rowscale && p.act == 0;
}

int ss_skinny_nt_launch(const FastGemm& p, hipStream_t st) {
    const int64_t need = cdiv(cdiv(p.M, 8), 4);
    dim3 grid((unsigned)(need < 4096 ? need : 4096), (unsigned)cdiv(p.N, 512));
    hipLaunchKernelGGL(gemm_nt_smallk_kernel, grid, dim3(256), 0, st, p.A, p.lda, p.B, p.ldb, (bf16*)p.C, p.ldc, p.M, p.N, (int)p.K, p.bias);
    SS_LAUNCH_CHECK();
    return 0;
}

bool ss_skinny_tn_ok(const void* B, int64_t ldb, int64_t M, int64_t N) {
    return M <= 8 && N % 8 == 0 && ldb % 8 == 0 && aligned16(B);
}

int ss_skinny_tn_launch(const FastGemm& p, int nsplit, hipStream_t st) {
    dim3 grid((unsigned)nsplit, (unsigned)cdiv(p.N, 512));
    hipLaunchKernelGGL(gemm_tn_smallm_kernel, grid, dim3(256), 0, st, p.A, p.lda, p.B, p.ldb, (float*)p.C, (int)p.M, p.N, p.K, p.k_per_split);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ NT with M <= 16 rows (the per-image vectors of LawinHead's image-pool
// branch: 16 x 512 x 512): a wave per output column n keeps that weight row in registers (a lane holds 16-byte k-chunks), multiplies
// it with the M rows of A (16 KB, served by L1 / L2) and reduces across its lanes in a fixed order.  The 128-row tile kernel spends
// 20 us on such a launch (one row tile, 8 k-iterations of latency), the generic fp32-output kernel 40 us.
template <typename TO>
__global__ __launch_bounds__(256) void gemm_nt_tinym_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
        TO* __restrict__ C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const bf16x8 w = *(const bf16x8*)(B + (int64_t)n * ldb + k);
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if (m < M) {
                const bf16x8 a = *(const bf16x8*)(A + (int64_t)m * lda + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[m] = fmaf((float)a[e], (float)w[e], acc[m]);
            }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (m < M) {
            float v = acc[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) C[(int64_t)m * ldc + n] = (TO)(v + (bias ? bias[n] : 0.f));
        }
}

bool ss_tinym_nt_ok(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb) {
    return M <= 16 && N >= 64 && K >= 64 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B);
}

int ss_tinym_nt_launch(const bf16* A, int64_t lda, const bf16* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int64_t M, int64_t N,
                       int64_t K, const float* bias, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(N, 4)), block(256);
    if (out_f32) hipLaunchKernelGGL(gemm_nt_tinym_kernel<float>, grid, block, 0, st, A, lda, B, ldb, (float*)C, ldc, (int)M, (int)N, (int)K, bias);
    else hipLaunchKernelGGL(gemm_nt_tinym_kernel<bf16>, grid, block, 0, st, A, lda, B, ldb, (bf16*)C, ldc, (int)M, (int)N, (int)K, bias);
    SS_LAUNCH_CHECK();
    return 0;
}
