// MFMA GEMM family for the 1x1 / linear projections.
//
//   ss_gemm_nt : C[M,N] = epilogue(A[M,K] . B[N,K]^T)            (forward, and dgrad with a transposed weight copy)
//   ss_gemm_tn : C[M,N] (+)= sum_r A[r,M]^T . B[r,N]   (fp32 out)  (weight gradient; split over r, deterministic slab reduce)
//
// One kernel template serves both: 128 x BN x (128 bytes of K) tiles, 4 waves in a 2x2 grid, each wave
// 2 x (BN/64) MFMA tiles of 32x32.  bf16 uses v_mfma_f32_32x32x16_bf16; fp32 uses eight v_mfma_f32_32x32x2_f32
// per 16-deep k-step with the SAME fragment layout (lane (i,h) holds k = 8h..8h+7), so both dtypes share every
// line of tiling code.  Operands are staged global -> registers -> LDS (rows padded by 16 B: conflict-free
// ds_read_b128), with the next tile's global loads in flight under the current tile's MFMAs.
#include "common.h"
#include "gemm_fast.h"
#include "colreduce.h"

__device__ __forceinline__ floatx16 mma32(bf16x8 a, bf16x8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mma32(floatx8 a, floatx8 b, floatx16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ bf16x8 ldfrag(const bf16* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ floatx8 ldfrag(const float* p) {
    floatx4 a = *(const floatx4*)p, b = *(const floatx4*)(p + 4);
    floatx8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}

template <typename T> struct GemmT;
template <> struct GemmT<bf16> { typedef bf16x8 frag; typedef bf16x8 vec; static constexpr int BK = 64, VEC = 8; };
template <> struct GemmT<float> { typedef floatx8 frag; typedef floatx4 vec; static constexpr int BK = 32, VEC = 4; };

struct GemmParams {
    const void* A; const void* B; void* C;
    int64_t M; int N; int64_t K;          // TRANS: K = number of reduction rows R
    int64_t lda, ldb, ldc;
    const float* bias;
    const void* res; int64_t ldr;
    const float* rowscale; int64_t rows_per_scale;
    int act;                               // 0 none, 1 GELU(erf)
    int vecA, vecB;                        // 16-byte vector loads allowed
    int64_t k_per_split;                   // TRANS: reduction rows per blockIdx.z
};

template <typename T, typename TC, int BN, bool TRANS>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    typedef GemmT<T> G;
    typedef typename G::frag frag;
    typedef typename G::vec vec;
    constexpr int BM = 128, BK = G::BK, VEC = G::VEC;
    constexpr int LDK = BK + VEC;                 // NT image: [row][k], rows padded by 16 B
    constexpr int LDMA = BM + VEC, LDMB = BN + VEC;   // TRANS image: [k][row]
    constexpr int A_ELEMS = TRANS ? BK * LDMA : BM * LDK;
    constexpr int B_ELEMS = TRANS ? BK * LDMB : BN * LDK;
    constexpr int NA = BM * BK / VEC / 256, NB = BN * BK / VEC / 256;   // 16-B chunks per thread
    constexpr int TM = 2, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) T As[A_ELEMS];
    __shared__ __attribute__((aligned(16))) T Bs[B_ELEMS];

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntiles_n = (p.N + BN - 1) / BN;            // n-tile fastest: the blocks sharing an A panel run together
    const int64_t m0 = (int64_t)(blockIdx.x / ntiles_n) * BM;
    const int n0 = (int)(blockIdx.x % ntiles_n) * BN;
    int64_t kbeg = 0, kend = p.K;
    if (TRANS) { kbeg = (int64_t)blockIdx.z * p.k_per_split; kend = min(p.K, kbeg + p.k_per_split); }

    vec ra[NA], rb[NB];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int c = tid + i * 256;
            int64_t row, col; bool ok;            // row: leading index in memory, col: contiguous index
            if (TRANS) { int r = c / (BM / VEC), mc = c % (BM / VEC); row = k0 + r; col = m0 + mc * VEC; ok = row < kend && col < p.M; }
            else { int r = c / (BK / VEC), kc = c % (BK / VEC); row = m0 + r; col = k0 + kc * VEC; ok = row < p.M && col < kend; }
            vec v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
            if (ok) {
                const T* src = A + row * p.lda + col;
                if (p.vecA) v = *(const vec*)src;
                else { int64_t lim = TRANS ? p.M : kend;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (col + e < lim) v[e] = src[e]; }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int c = tid + i * 256;
            int64_t row, col; bool ok;
            if (TRANS) { int r = c / (BN / VEC), nc = c % (BN / VEC); row = k0 + r; col = n0 + nc * VEC; ok = row < kend && col < p.N; }
            else { int r = c / (BK / VEC), kc = c % (BK / VEC); row = n0 + r; col = k0 + kc * VEC; ok = row < p.N && col < kend; }
            vec v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
            if (ok) {
                const T* src = B + row * p.ldb + col;
                if (p.vecB) v = *(const vec*)src;
                else { int64_t lim = TRANS ? (int64_t)p.N : kend;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (col + e < lim) v[e] = src[e]; }
            }
            rb[i] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int c = tid + i * 256;
            if (TRANS) { int r = c / (BM / VEC), mc = c % (BM / VEC); *(vec*)&As[r * LDMA + mc * VEC] = ra[i]; }
            else { int r = c / (BK / VEC), kc = c % (BK / VEC); *(vec*)&As[r * LDK + kc * VEC] = ra[i]; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int c = tid + i * 256;
            if (TRANS) { int r = c / (BN / VEC), nc = c % (BN / VEC); *(vec*)&Bs[r * LDMB + nc * VEC] = rb[i]; }
            else { int r = c / (BK / VEC), kc = c % (BK / VEC); *(vec*)&Bs[r * LDK + kc * VEC] = rb[i]; }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (kbeg < kend) {
        gload(kbeg);
        lstore();
        __syncthreads();
        for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = k0 + BK < kend;
            if (more) gload(k0 + BK);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                frag fa[TM], fb[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    int row = wm * 64 + a * 32 + li;
                    if (TRANS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) fa[a][e] = As[(ks * 16 + lh * 8 + e) * LDMA + row];
                    } else fa[a] = ldfrag(&As[row * LDK + ks * 16 + lh * 8]);
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    int row = wn * (BN / 2) + b * 32 + li;
                    if (TRANS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) fb[b][e] = Bs[(ks * 16 + lh * 8 + e) * LDMB + row];
                    } else fb[b] = ldfrag(&Bs[row * LDK + ks * 16 + lh * 8]);
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = mma32(fa[a], fb[b], acc[a][b]);
            }
            __syncthreads();
            if (more) { lstore(); __syncthreads(); }
        }
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    TC* __restrict__ C = (TC*)p.C;
    if (TRANS) C += (int64_t)blockIdx.z * p.M * p.N;
    const TC* __restrict__ R = (const TC*)p.res;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            int col = n0 + wn * (BN / 2) + b * 32 + li;
            if (col >= p.N) continue;
            float bv = (!TRANS && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int64_t row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[a][b][r];
                if (!TRANS) {
                    v += bv;
                    if (p.act == 6) {                     // gradient through a GELU: times GELU'(res), res = the pre-activation
                        if (p.rowscale) v *= p.rowscale[row / p.rows_per_scale];
                        v *= gelu_erf_grad(to_f(R[row * p.ldr + col]));
                    } else {
                        v = apply_act(v, p.act);
                        if (p.rowscale) v *= p.rowscale[row / p.rows_per_scale];
                        if (R) v += to_f(R[row * p.ldr + col]);
                    }
                }
                C[row * p.ldc + col] = from_f<TC>(v);
            }
        }
}

// sum nsplit dense [n] fp32 slabs into out (row-major MxN with leading dimension ldc)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, int64_t M, int N, float* __restrict__ out,
                                    int64_t ldc, int accumulate) {
    int64_t n = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += slabs[(int64_t)z * n + i];
        int64_t o = (i / N) * ldc + (i % N);
        out[o] = accumulate ? out[o] + s : s;
    }
}

// bf16 fast paths (gemm_bf16.hip)

__global__ void reduce_slabs4_kernel(const float* __restrict__ slabs, int nsplit, int zstride, int64_t M, int N, float* __restrict__ out,
                                     int64_t ldc, int accumulate) {
    const int64_t n4 = M * N / 4, n = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        floatx4 s = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nsplit; z += zstride) s += *(const floatx4*)(slabs + (int64_t)z * n + i * 4);
        int64_t e = i * 4;
        float* o = out + (e / N) * ldc + (e % N);
        if (accumulate) s += *(const floatx4*)o;
        *(floatx4*)o = s;
    }
}

// One launch for the whole split reduction: out = sum_z slab[z] (row-major MxN, leading dimension ldc) and, when the GEMM kernel
// also left per-split column sums (cs_ws [nsplit][M], the fused bias gradient), colsum = sum_z cs_ws[z].  A block is COLS float4
// columns x PARTS slab parts: thread (c, p) adds slabs p, p + PARTS, ... with independent loads, the parts meet in LDS and are
// added in a fixed order (bitwise reproducible; PARTS depends only on the shape).  Replaces the two-level slab tree + the
// separate column-sum reduce: three tiny launches per weight gradient became one.
template <int PARTS>
__global__ __launch_bounds__(256) void reduce_slabs_fused_kernel(const float* __restrict__ slabs, int nsplit, int64_t M, int N, float* __restrict__ out,
                                                                 int64_t ldc, int accumulate, const float* __restrict__ cs_ws,
                                                                 float* __restrict__ colsum, int cs_accumulate) {
    constexpr int COLS = 256 / PARTS;
    __shared__ floatx4 sh[PARTS][COLS + 1];
    const int c = threadIdx.x % COLS, part = threadIdx.x / COLS;
    const int64_t n = M * N, n4 = n / 4, m4 = cs_ws ? M / 4 : 0;
    const int64_t col = (int64_t)blockIdx.x * COLS + c;
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4) {
        const float* p = slabs + col * 4;
#pragma unroll 4
        for (int z = part; z < nsplit; z += PARTS) s += *(const floatx4*)(p + (int64_t)z * n);
    } else if (col < n4 + m4) {
        const float* p = cs_ws + (col - n4) * 4;
#pragma unroll 4
        for (int z = part; z < nsplit; z += PARTS) s += *(const floatx4*)(p + (int64_t)z * M);
    }
    sh[part][c] = s;
    __syncthreads();
    if (part != 0 || col >= n4 + m4) return;
    floatx4 t = sh[0][c];
#pragma unroll
    for (int k = 1; k < PARTS; ++k) t += sh[k][c];
    if (col < n4) {
        const int64_t e = col * 4;
        float* o = out + (e / N) * ldc + (e % N);
        if (accumulate) t += *(const floatx4*)o;
        *(floatx4*)o = t;
    } else {
        float* o = colsum + (col - n4) * 4;
        if (cs_accumulate) { o[0] += t[0]; o[1] += t[1]; o[2] += t[2]; o[3] += t[3]; }
        else { o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
    }
}

// in-place first level of the slab tree: slab[g*group] = sum of slabs g*group .. g*group+group-1 (fixed order)
__global__ void reduce_slabs_group_kernel(float* __restrict__ slabs, int nsplit, int group, int64_t n4) {
    const int g = blockIdx.y;
    const int z0 = g * group, z1 = min(nsplit, z0 + group);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        floatx4 s = *(const floatx4*)(slabs + ((int64_t)z0 * n4 + i) * 4);
        for (int z = z0 + 1; z < z1; ++z) s += *(const floatx4*)(slabs + ((int64_t)z * n4 + i) * 4);
        *(floatx4*)(slabs + ((int64_t)z0 * n4 + i) * 4) = s;
    }
}

template <typename T, typename TC>
static int launch_nt(const GemmParams& p, hipStream_t st) {
    dim3 block(256);
    if (p.N <= 64) {
        dim3 grid((unsigned)(cdiv(p.N, 64) * cdiv(p.M, 128)), 1, 1);
        hipLaunchKernelGGL((gemm_kernel<T, TC, 64, false>), grid, block, 0, st, p);
    } else {
        dim3 grid((unsigned)(cdiv(p.N, 128) * cdiv(p.M, 128)), 1, 1);
        hipLaunchKernelGGL((gemm_kernel<T, TC, 128, false>), grid, block, 0, st, p);
    }
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_gemm_nt(int dtype, int out_f32, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldr,
                          const float* rowscale, int64_t rows_per_scale, int act, hipStream_t stream) {
    SS_CHECK(A && B && C, "ss_gemm_nt: null operand");
    SS_CHECK(M > 0 && N > 0 && K > 0 && N < (1 << 30), "ss_gemm_nt: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    SS_CHECK(act >= 0 && act <= 6 && (act != 6 || res), "ss_gemm_nt: act must be 0..6, and 6 (times GELU'(res)) needs res");
    SS_CHECK(lda >= K && ldb >= K && ldc >= N, "ss_gemm_nt: leading dimension smaller than the row");
    SS_CHECK(!res || ldr >= N, "ss_gemm_nt: bad ldr");
    SS_CHECK(!rowscale || rows_per_scale > 0, "ss_gemm_nt: rows_per_scale must be > 0");
    SS_CHECK(cdiv(M, 128) * cdiv(N, 64) < 2147483647LL, "ss_gemm_nt: too many tiles");
    if (!out_f32 && !res && !rowscale && act == 0 && ss_rows_nt_ok(dtype, A, lda, C, ldc, M, N, K))
        return ss_rows_nt_launch(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, stream);
    if (dtype == SS_BF16 && !res && !rowscale && act == 0 && ss_tinym_nt_ok(M, N, K, A, lda, B, ldb))
        return ss_tinym_nt_launch((const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, out_f32, M, N, K, bias, stream);
    if (dtype == SS_BF16 && !out_f32) {
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = C; f.M = M; f.N = (int)N; f.K = K; f.lda = lda; f.ldb = ldb; f.ldc = ldc;
        f.bias = bias; f.res = (const bf16*)res; f.ldr = ldr; f.rowscale = rowscale; f.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
        f.act = act; f.k_per_split = K;
        if (ss_skinny_nt_ok(f)) return ss_skinny_nt_launch(f, stream);
        if (ss_fast_nt_ok(A, lda, B, ldb, C, ldc, M, N, K, res, ldr, bias)) return ss_fast_nt_launch(f, stream);
    }
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = (int)N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.res = res; p.ldr = ldr; p.rowscale = rowscale; p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    p.act = act; p.k_per_split = K;
    int vec = dtype == SS_BF16 ? 8 : 4;
    p.vecA = aligned16(A) && lda % vec == 0 && K % vec == 0;
    p.vecB = aligned16(B) && ldb % vec == 0 && K % vec == 0;
    if (dtype == SS_F32) return launch_nt<float, float>(p, stream);
    if (dtype == SS_BF16) return out_f32 ? launch_nt<bf16, float>(p, stream) : launch_nt<bf16, bf16>(p, stream);
    return ss_fail(-1, "ss_gemm_nt: unsupported dtype %d", dtype);
}

extern "C" int ss_gelu_fwd(int dtype, const void* z, void* a, int64_t n, hipStream_t stream);

// Linear + GELU with BOTH tensors kept (modules/convnext.py:43-45: pw_conv1 -> act; the backward needs the pre-activation for
// GELU' and the activation for pw_conv2's weight gradient): Z = A . B^T + bias, G = GELU(Z as stored).  On the bf16 tile kernels
// the two stores share one epilogue (no second pass over Z); other shapes / fp32 run the GEMM and then ss_gelu_fwd.
extern "C" int ss_gemm_nt_gelu2(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* Z, int64_t ldz, void* G, int64_t ldg,
                                int64_t M, int64_t N, int64_t K, const float* bias, hipStream_t stream) {
    SS_CHECK(A && B && Z && G && M > 0 && N > 0 && K > 0 && N < (1 << 30) && lda >= K && ldb >= K && ldz >= N && ldg >= N,
             "ss_gemm_nt_gelu2: bad arguments");
    SS_CHECK(cdiv(M, 128) * cdiv(N, 64) < 2147483647LL, "ss_gemm_nt_gelu2: too many tiles");
    if (dtype == SS_BF16 && ldg % 8 == 0 && aligned16(G)) {
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = G; f.M = M; f.N = (int)N; f.K = K; f.lda = lda; f.ldb = ldb; f.ldc = ldg;
        f.bias = bias; f.rows_per_scale = 1; f.act = 0; f.k_per_split = K; f.C2 = (bf16*)Z; f.ldc2 = ldz;
        if (!ss_skinny_nt_ok(f) && ss_fast_nt_ok(A, lda, B, ldb, Z, ldz, M, N, K, nullptr, 0, bias)) return ss_fast_nt_launch(f, stream);
    }
    SS_CHECK(ldz == N && ldg == N && (M * N) % 8 == 0, "ss_gemm_nt_gelu2: outside the bf16 tile kernels Z and G must be dense and M*N a multiple of 8");
    int rc = ss_gemm_nt(dtype, 0, A, lda, B, ldb, Z, ldz, M, N, K, bias, nullptr, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    return ss_gelu_fwd(dtype, Z, G, M * N, stream);
}

// Two chained 1x1 convolutions where the second is narrow (a segmentation head's conv + BN + act followed by its classifier,
// modules/heads.py:120-170 + core/models.py:103-115): out2[M, N2] = act(A . B^T + bias) . W2^T with the N = 256 wide intermediate kept in
// the GEMM's LDS tile (it is neither written nor read back).  ss_gemm_nt_nt_ok says whether the tile kernel takes the shape.
extern "C" int ss_gemm_nt_nt_ok(int dtype, int64_t M, int64_t N, int64_t K, int64_t N2) {
    return dtype == SS_BF16 && M >= 256 && N == 256 && K % 64 == 0 && K >= 128 && K < (1LL << 30) && N2 > 0 && N2 <= 32 && N2 % 4 == 0 && cdiv(M, 256) < 2147483647LL;
}
extern "C" int ss_gemm_nt_nt(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, int act, const void* W2,
                             int64_t ldw2, void* out2, int64_t ldo2, int64_t M, int64_t N, int64_t K, int64_t N2, hipStream_t stream) {
    SS_CHECK(A && B && W2 && out2 && ss_gemm_nt_nt_ok(dtype, M, N, K, N2), "ss_gemm_nt_nt: needs bf16, N == 256, K a multiple of 64 >= 128, N2 <= 32 (multiple of 4)");
    SS_CHECK(act >= 0 && act <= 5, "ss_gemm_nt_nt: act must be 0..5");
    SS_CHECK(lda >= K && ldb >= K && ldw2 >= N && ldo2 >= N2 && lda % 8 == 0 && ldb % 8 == 0 && ldw2 % 8 == 0 && ldo2 % 4 == 0 && aligned16(A) && aligned16(B)
             && aligned16(W2) && (((uintptr_t)out2) & 7) == 0 && (!bias || aligned16(bias)), "ss_gemm_nt_nt: operands must be aligned (16 bytes; out2 rows 8 bytes)");
    FastGemm f{};
    f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = nullptr; f.M = M; f.N = (int)N; f.K = K; f.lda = lda; f.ldb = ldb; f.ldc = N;
    f.bias = bias; f.rows_per_scale = 1; f.act = act; f.k_per_split = K;
    f.w2 = (const bf16*)W2; f.ldw2 = ldw2; f.n2 = (int)N2; f.out2 = (bf16*)out2; f.ldo2 = ldo2;
    return ss_big_nt2_launch(f, stream);
}

static int tn_big_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SS_GEMM_TN_BIG"); on = !(e && e[0] == '0'); }
    return on;
}

// returns true when the split was planned for the 256-tile bf16 kernel (gemm_bf16_tn_big.hip)
static bool tn_plan(int64_t M, int64_t N, int64_t R, int bk, int* nsplit, int64_t* rps, bool allow_big = true) {
    if (allow_big && bk == 64 && tn_big_enabled() && ss_big_tn_plan(M, N, R, nsplit, rps)) return true;
    // split the reduction rows so that ~8 workgroups per CU exist (tail quantisation), every split has >= 4 k-tiles, and the fp32 slabs
    // (written once, read once) stay below ~a quarter of the operand bytes
    int64_t tiles = cdiv(M, 128) * cdiv(N, N <= 64 ? 64 : 128);
    int64_t want = cdiv(2048, tiles);
    int64_t maxs = cdiv(R, 4 * bk);
    int64_t cap = (R * (M + N)) / (8 * M * N);
    int64_t s = want;
    if (s > maxs) s = maxs;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    int64_t per = cdiv(cdiv(R, s), bk) * bk;
    *rps = per;
    *nsplit = (int)cdiv(R, per);
    return false;
}

extern "C" int ss_colsum(int dtype, const void* x, int64_t ld, int64_t rows, int C, int groups, float scale, float* out,
                         int accumulate, float* workspace, int64_t ws_bytes, hipStream_t stream);
static const int64_t kColsumWsRows = SS_COLRED_MAX_BLOCKS;             // ss_colsum's partial rows (COLRED_BLOCKS) when the sums cannot be fused

extern "C" int64_t ss_gemm_tn_workspace(int dtype, int64_t M, int64_t N, int64_t R) {
    int ns; int64_t rps;
    tn_plan(M, N, R, dtype == SS_BF16 ? 64 : 32, &ns, &rps);
    return ((int64_t)ns * M * N + kColsumWsRows * M) * (int64_t)sizeof(float);
}

static int gemm_tn_impl(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum, int colsum_accumulate,
                        hipStream_t stream, const PatchMap* pm = nullptr, const float* rowkeep = nullptr, int64_t rows_per_keep = 0);

// The weight gradient of a layer inside a DropPath branch (modules/segformer.py:96-97 with modules/helpers.py:9-24): rows of sample
// b carry the multiplier rowkeep[b], which is 0 (dropped) or ONE common positive value.  The bf16 tile kernels skip the k-tiles of
// dropped samples and scale their slabs, so no scaled copy of the gradient is made; ss_gemm_tn_keep_ok says whether they take the call.
extern "C" int ss_gemm_tn_keep_ok(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t R,
                                  int64_t rows_per_keep) {
    if (dtype != SS_BF16 || rows_per_keep <= 0 || rows_per_keep % 64 || R % rows_per_keep || M % 8 || lda % 8) return 0;
    if (ss_skinny_tn_ok(B, ldb, M, N) || !ss_fast_tn_ok(A, lda, B, ldb, M, N)) return 0;
    int ns; int64_t rps;
    tn_plan(M, N, R, 64, &ns, &rps);
    return ns <= kColsumWsRows && N % 4 == 0;
}
extern "C" int ss_gemm_tn_colsum_keep(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                      int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum,
                                      int colsum_accumulate, const float* rowkeep, int64_t rows_per_keep, hipStream_t stream) {
    SS_CHECK(colsum && rowkeep && ldc % 4 == 0 && aligned16(C) && ss_gemm_tn_keep_ok(dtype, A, lda, B, ldb, M, N, R, rows_per_keep),
             "ss_gemm_tn_colsum_keep: shape / layout outside the bf16 tile kernels (ask ss_gemm_tn_keep_ok first)");
    return gemm_tn_impl(dtype, A, lda, B, ldb, C, ldc, M, N, R, workspace, ws_bytes, accumulate, colsum, colsum_accumulate, stream, nullptr,
                        rowkeep, rows_per_keep);
}

extern "C" int ss_gemm_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, hipStream_t stream) {
    return gemm_tn_impl(dtype, A, lda, B, ldb, C, ldc, M, N, R, workspace, ws_bytes, accumulate, nullptr, 0, stream);
}

extern "C" int ss_gemm_tn_colsum(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                 int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum,
                                 int colsum_accumulate, hipStream_t stream) {
    SS_CHECK(colsum && M % 8 == 0 && lda % 8 == 0, "ss_gemm_tn_colsum: null colsum, or M / lda not a multiple of 8");
    return gemm_tn_impl(dtype, A, lda, B, ldb, C, ldc, M, N, R, workspace, ws_bytes, accumulate, colsum, colsum_accumulate, stream);
}

static int gemm_tn_impl(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum, int colsum_accumulate,
                        hipStream_t stream, const PatchMap* pm, const float* rowkeep, int64_t rows_per_keep) {
    SS_CHECK(A && B && C && workspace, "ss_gemm_tn: null operand");
    SS_CHECK(M > 0 && N > 0 && R > 0 && N < (1 << 30) && M < (1 << 30), "ss_gemm_tn: bad shape");
    SS_CHECK(lda >= M && (pm || ldb >= N) && ldc >= N, "ss_gemm_tn: leading dimension smaller than the row");
    int ns; int64_t rps;
    int bk = dtype == SS_BF16 ? 64 : 32;
    const bool big = tn_plan(M, N, R, bk, &ns, &rps, !pm);
    const int64_t ws_need = ((int64_t)ns * M * N + kColsumWsRows * M) * 4;
    SS_CHECK(ws_bytes >= ws_need, "ss_gemm_tn: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ws_need);
    float* cs_ws = (float*)workspace + (int64_t)ns * M * N;
    bool cs_fused = false;
    GemmParams p{};
    p.A = A; p.B = B; p.C = workspace; p.M = M; p.N = (int)N; p.K = R; p.lda = lda; p.ldb = ldb; p.ldc = N;
    p.k_per_split = rps;
    int vec = dtype == SS_BF16 ? 8 : 4;
    p.vecA = aligned16(A) && lda % vec == 0 && M % vec == 0;
    p.vecB = aligned16(B) && ldb % vec == 0 && N % vec == 0;
    dim3 block(256);
    if (pm) {                                              // B is addressed as image patches: the 128-row tile kernel only
        SS_CHECK(dtype == SS_BF16 && ss_fast_tn_ok(A, lda, B, ldb, M, N), "ss_conv_patch_wgrad: operands must be bf16, 16-byte aligned, multiples of 8");
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = workspace; f.M = M; f.N = (int)N; f.K = R; f.lda = lda; f.ldb = ldb; f.ldc = N;
        f.k_per_split = rps; f.pm = *pm;
        if (colsum && ns <= kColsumWsRows) { f.colsum_ws = cs_ws; cs_fused = true; }
        int rc = ss_patch_tn_launch(f, ns, stream);
        if (rc) return rc;
    } else if (!rowkeep && ss_smallmn_tn_ok(dtype, A, lda, B, ldb, M, N)) {
        int rc = ss_smallmn_tn_launch(dtype, A, lda, B, ldb, (float*)workspace, N, R, rps, ns, stream);
        if (rc) return rc;
    } else if (dtype == SS_BF16 && ss_skinny_tn_ok(B, ldb, M, N)) {
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = workspace; f.M = M; f.N = (int)N; f.K = R; f.lda = lda; f.ldb = ldb; f.ldc = N;
        f.k_per_split = rps;
        int rc = ss_skinny_tn_launch(f, ns, stream);
        if (rc) return rc;
    } else if (dtype == SS_BF16 && ss_fast_tn_ok(A, lda, B, ldb, M, N)) {
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = workspace; f.M = M; f.N = (int)N; f.K = R; f.lda = lda; f.ldb = ldb; f.ldc = N;
        f.k_per_split = rps; f.rowscale = rowkeep; f.rows_per_scale = rows_per_keep > 0 ? rows_per_keep : 1;
        if (colsum && ns <= kColsumWsRows) { f.colsum_ws = cs_ws; cs_fused = true; }
        int rc = big && ss_big_tn_ok(f, ns) ? ss_big_tn_launch(f, ns, stream) : ss_fast_tn_launch(f, ns, stream);
        if (rc) return rc;
    } else if (N <= 64) {
        dim3 grid((unsigned)(cdiv(N, 64) * cdiv(M, 128)), 1, ns);
        if (dtype == SS_F32) hipLaunchKernelGGL((gemm_kernel<float, float, 64, true>), grid, block, 0, stream, p);
        else if (dtype == SS_BF16) hipLaunchKernelGGL((gemm_kernel<bf16, float, 64, true>), grid, block, 0, stream, p);
        else return ss_fail(-1, "ss_gemm_tn: unsupported dtype %d", dtype);
    } else {
        dim3 grid((unsigned)(cdiv(N, 128) * cdiv(M, 128)), 1, ns);
        if (dtype == SS_F32) hipLaunchKernelGGL((gemm_kernel<float, float, 128, true>), grid, block, 0, stream, p);
        else if (dtype == SS_BF16) hipLaunchKernelGGL((gemm_kernel<bf16, float, 128, true>), grid, block, 0, stream, p);
        else return ss_fail(-1, "ss_gemm_tn: unsupported dtype %d", dtype);
    }
    SS_LAUNCH_CHECK();
    const bool vec4 = N % 4 == 0 && ldc % 4 == 0 && aligned16(C);
    if (vec4) {                                           // the common case: one launch reduces the slabs and the fused column sums
        if (colsum && !cs_fused) {
            int rc = ss_colsum(dtype, A, lda, R, (int)M, 1, 1.f, colsum, colsum_accumulate, cs_ws, kColsumWsRows * M * 4, stream);
            if (rc) return rc;
        }
        const float* cw = colsum && cs_fused ? cs_ws : nullptr;
        const int64_t cols = M * N / 4 + (cw ? M / 4 : 0);
        if (ns > 16 || cols < 32768)
            hipLaunchKernelGGL(reduce_slabs_fused_kernel<32>, dim3((unsigned)cdiv(cols, 8)), dim3(256), 0, stream, (const float*)workspace, ns, M, (int)N, C,
                               ldc, accumulate, cw, colsum, colsum_accumulate);
        else                                              // few slabs of a large output: a thread per float4 column adds them all
            hipLaunchKernelGGL(reduce_slabs_fused_kernel<1>, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, stream, (const float*)workspace, ns, M, (int)N, C,
                               ldc, accumulate, cw, colsum, colsum_accumulate);
        SS_LAUNCH_CHECK();
        return 0;
    }
    if (colsum) {
        if (cs_fused) {
            hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(M, 8), 1), dim3(256), 0, stream, (const float*)cs_ws, ns, (int)M, 1.f,
                               colsum, colsum_accumulate);
            SS_LAUNCH_CHECK();
        } else {
            int rc = ss_colsum(dtype, A, lda, R, (int)M, 1, 1.f, colsum, colsum_accumulate, cs_ws, kColsumWsRows * M * 4, stream);
            if (rc) return rc;
        }
    }
    int64_t n = M * N;
    if (N % 4 == 0 && ldc % 4 == 0 && aligned16(C)) {
        int64_t n4 = n / 4;
        int rb4 = (int)(cdiv(n4, 256) < 4096 ? cdiv(n4, 256) : 4096);
        int zstride = 1;
        if (ns > 32) {
            zstride = 32;
            hipLaunchKernelGGL(reduce_slabs_group_kernel, dim3(rb4, (unsigned)cdiv(ns, 32)), dim3(256), 0, stream, (float*)workspace, ns, 32, n4);
            SS_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(reduce_slabs4_kernel, dim3(rb4), dim3(256), 0, stream, (const float*)workspace, ns, zstride, M, (int)N, C, ldc, accumulate);
        SS_LAUNCH_CHECK();
        return 0;
    }
    int rb = (int)(cdiv(n, 256) < 4096 ? cdiv(n, 256) : 4096);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(rb), dim3(256), 0, stream, (const float*)workspace, ns, M, (int)N, C, ldc, accumulate);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ convolutions with kernel = stride, without the patch matrix
// nn.Conv2d(C, N, s, s) on a channels-last image [B, H, W, C] (modules/segformer.py:27 `sr`, modules/convnext.py:96-104 down-sampling)
// is a GEMM whose patch operand is the image itself, read (or written) s runs of s*C elements per patch row (PatchMap).
static bool patch_map(PatchMap* pm, int side, int B, int H, int W, int C, int s) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || s <= 0 || H % s || W % s || C % 8) return false;
    const int64_t kc = (int64_t)s * C;
    if (kc % 64 || (int64_t)B * H * W >= (1LL << 31)) return false;
    pm->side = side; pm->kc = (int)kc; pm->cs = (int64_t)W * C; pm->outer = (int64_t)s * W * C; pm->inner = fast_div((unsigned)(W / s));
    return true;
}

extern "C" int ss_conv_patch_ok(int dtype, int B, int H, int W, int C, int s, int N) {
    PatchMap pm;
    return dtype == SS_BF16 && N > 0 && N % 8 == 0 && patch_map(&pm, 1, B, H, W, C, s);
}

extern "C" int ss_conv_patch_fwd(int dtype, const void* x, int B, int H, int W, int C, int s, const void* w, int64_t ldw, const float* bias,
                                 void* y, int64_t ldy, int N, hipStream_t stream) {
    FastGemm f{};
    SS_CHECK(x && w && y && ss_conv_patch_ok(dtype, B, H, W, C, s, N), "ss_conv_patch_fwd: needs bf16, H and W multiples of s, s*C a multiple of 64, N of 8");
    const int64_t K = (int64_t)s * s * C, M = (int64_t)B * (H / s) * (W / s);
    SS_CHECK(ldw >= K && ldy >= N && ss_fast_nt_ok(x, s * C, w, ldw, y, ldy, M, N, K, nullptr, 0, bias), "ss_conv_patch_fwd: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
    patch_map(&f.pm, 1, B, H, W, C, s);
    f.A = (const bf16*)x; f.B = (const bf16*)w; f.C = y; f.M = M; f.N = N; f.K = K; f.lda = (int64_t)s * C; f.ldb = ldw; f.ldc = ldy;
    f.bias = bias; f.rows_per_scale = 1; f.k_per_split = K;
    return ss_patch_nt_launch(f, stream);
}

// dx [B, H, W, C] = the patches dy [M, N] . wt [s*s*C, N]^T, each written to its place in the image (every pixel belongs to one patch)
extern "C" int ss_conv_patch_dgrad(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldw, void* dx, int B, int H, int W, int C,
                                   int s, int N, hipStream_t stream) {
    FastGemm f{};
    SS_CHECK(dy && wt && dx && ss_conv_patch_ok(dtype, B, H, W, C, s, N), "ss_conv_patch_dgrad: needs bf16, H and W multiples of s, s*C a multiple of 64, N of 8");
    const int64_t K = (int64_t)s * s * C, M = (int64_t)B * (H / s) * (W / s);
    SS_CHECK(ldw >= N && lddy >= N && ss_fast_nt_ok(dy, lddy, wt, ldw, dx, s * C, M, K, N, nullptr, 0, nullptr), "ss_conv_patch_dgrad: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
    patch_map(&f.pm, 2, B, H, W, C, s);
    f.A = (const bf16*)dy; f.B = (const bf16*)wt; f.C = dx; f.M = M; f.N = (int)K; f.K = N; f.lda = lddy; f.ldb = ldw; f.ldc = (int64_t)s * C;
    f.rows_per_scale = 1; f.k_per_split = N;
    return ss_patch_nt_launch(f, stream);
}

extern "C" int64_t ss_conv_patch_wgrad_workspace(int dtype, int B, int H, int W, int C, int s, int N) {
    if (s <= 0) return 0;
    int ns; int64_t rps;
    const int64_t K = (int64_t)s * s * C, M = (int64_t)B * (H / s) * (W / s);
    tn_plan(N, K, M, 64, &ns, &rps, false);
    return ((int64_t)ns * N * K + kColsumWsRows * N) * (int64_t)sizeof(float);
}

// dw [N, s*s*C] fp32 = dy [M, N]^T . patches(x), dbias [N] = column sums of dy (or null)
extern "C" int ss_conv_patch_wgrad(int dtype, const void* dy, int64_t lddy, const void* x, int B, int H, int W, int C, int s, int N, float* dw,
                                   int64_t lddw, float* dbias, void* workspace, int64_t ws_bytes, hipStream_t stream) {
    PatchMap pm;
    SS_CHECK(dy && x && dw && ss_conv_patch_ok(dtype, B, H, W, C, s, N), "ss_conv_patch_wgrad: needs bf16, H and W multiples of s, s*C a multiple of 64, N of 8");
    patch_map(&pm, 4, B, H, W, C, s);
    const int64_t K = (int64_t)s * s * C, M = (int64_t)B * (H / s) * (W / s);
    SS_CHECK(lddy % 8 == 0, "ss_conv_patch_wgrad: lddy must be a multiple of 8");
    return gemm_tn_impl(dtype, dy, lddy, x, (int64_t)s * C, dw, lddw, N, K, M, workspace, ws_bytes, 0, dbias, 0, stream, &pm);
}
