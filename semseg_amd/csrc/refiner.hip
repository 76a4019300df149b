// TopFormer RefinerHead, inference (reference modules/heads.py:22-39 SimpleAttention, :42-60 AttentionUniRes, :173-269
// RefinerHead; core/models.py:52-72,99-115).  Channels-last token matrices [images * rows_per_image, C].
//
// The head is, per image, two data-dependent normalisations around small linear maps, so it runs as
//   ss_image_moments  per (image, channel) mean and centred sum of squares  (SimAM's statistics; GroupNorm's, combined per group)
//   ss_simam_apply    x * sigmoid((x - mu)^2 / (4 (k + M2 / (n - 1))) + 0.5)
//   [grouped 1x1 conv = one skinny GEMM per stage at the stage's own resolution, then bilinear to the finest map: both linear]
//   ss_refiner_fold   GroupNorm statistics + affine + the linear map that follows it -> one [Cout x Cin] matrix and offset per image
//   ss_image_affine   out[p] = M[image] * f(in[p]) + c[image]   (f = sigmoid for the 'CAN' group conv, identity after it)
// Everything between the sigmoid and the logits is fp32.  Fixed-order reductions: bitwise reproducible.
#include "common.h"
#include "colreduce.h"

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// Shifted sums per image: d = f(x) - f(x[first row of the image]) keeps E[d^2] - E[d]^2 from cancelling (see BnStatF in norm.hip).
template <typename T, bool SIG> struct ImageMomentF {
    static constexpr int K = 2;
    const T* x; int64_t ld; int64_t rows;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        F8 v = load8(x + row * ld + c0), s = load8(x + (int64_t)blockIdx.y * rows * ld + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = v.v[e], b = s.v[e];
            if (SIG) { a = sigmoid_f(a); b = sigmoid_f(b); }
            const float d = a - b; acc[0][e] += d; acc[1][e] += d * d;
        }
    }
};

// moments[img][0][c] = mean, moments[img][1][c] = sum (f(x) - mean)^2; double accumulation in a fixed order
template <typename T, bool SIG>
__global__ __launch_bounds__(256) void image_moments_final_kernel(const float* __restrict__ partial, int nb, int C, int64_t n, const T* __restrict__ x,
                                                                  int64_t ld, float* __restrict__ moments) {
    __shared__ double sh[2][32][9];
    const int cx = threadIdx.x & 7, part = threadIdx.x >> 3, img = blockIdx.y;
    const int c = blockIdx.x * 8 + cx;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const float* p = partial + (int64_t)img * nb * 2 * C;
#pragma unroll 4
        for (int b = part; b < nb; b += 32) { s += p[((int64_t)b * 2 + 0) * C + c]; q += p[((int64_t)b * 2 + 1) * C + c]; }
    }
    sh[0][part][cx] = s; sh[1][part][cx] = q;
    __syncthreads();
    if (part != 0 || c >= C) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < 32; ++k) { s += sh[0][k][cx]; q += sh[1][k][cx]; }
    float f0 = to_f(x[(int64_t)img * n * ld + c]);
    if (SIG) f0 = sigmoid_f(f0);
    const double md = s / (double)n;
    double m2 = q - md * s;
    if (m2 < 0) m2 = 0;
    moments[((int64_t)img * 2 + 0) * C + c] = (float)((double)f0 + md);
    moments[((int64_t)img * 2 + 1) * C + c] = (float)m2;
}

extern "C" int ss_image_moments(int dtype, const void* x, int64_t ld, int64_t rows_per_image, int C, int images, int pre_act, float* moments,
                                float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(x && moments && workspace && rows_per_image > 0 && images > 0 && images <= 65535 && C > 0 && C % 8 == 0 && ld >= C && ld % 8 == 0 &&
             aligned16(x) && (pre_act == 0 || pre_act == 5), "ss_image_moments: bad arguments (C=%d images=%d pre_act=%d)", C, images, pre_act);
    ColGeom g = col_geom(rows_per_image, C);
    SS_CHECK(ws_bytes >= (int64_t)images * g.nb * 2 * C * 4, "ss_image_moments: workspace too small");
#define SS_MOM(SIG) do { ImageMomentF<T, SIG> f{(const T*)x, ld, rows_per_image}; \
        hipLaunchKernelGGL((colreduce_partial_kernel<ImageMomentF<T, SIG>>), dim3(g.nb, images), dim3(256), 0, stream, f, rows_per_image, C, g.cpb, g.rpb, workspace); \
        hipLaunchKernelGGL((image_moments_final_kernel<T, SIG>), dim3((unsigned)cdiv(C, 8), images), dim3(256), 0, stream, workspace, g.nb, C, \
                           rows_per_image, (const T*)x, ld, moments); } while (0)
    SS_DISPATCH(dtype, { if (pre_act == 5) SS_MOM(true); else SS_MOM(false); });
#undef SS_MOM
    SS_LAUNCH_CHECK();
    return 0;
}

// SimAM (modules/heads.py:27-39): y = x * sigmoid((x - mu)^2 / (4 (k + sum (x - mu)^2 / (h w - 1))) + 0.5); k is the learnable scalar
template <typename T>
__global__ __launch_bounds__(256) void simam_apply_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ moments, const float* __restrict__ kp,
                                                          T* __restrict__ y, int64_t ldy, int64_t rows, int C, int blocks_per_image) {
    const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    const int chunks = C / 8;
    const float k = kp[0], inv_nm1 = 1.f / (float)(rows - 1);
    const int64_t total = rows * chunks;
    const int64_t i0 = (int64_t)blk * blockDim.x + threadIdx.x, step = (int64_t)blocks_per_image * blockDim.x;
    // consecutive threads walk consecutive chunks, so a thread's channel chunk advances by (step % chunks) per iteration; with
    // step a multiple of chunks (C <= 2048) it is fixed and the per-channel terms are hoisted
    const bool fixed = step % chunks == 0;
    float mu[8], inv[8];
    int cc = (int)(i0 % chunks) * 8;
    auto params = [&](int c) {
        F8 m = load8(moments + ((int64_t)img * 2 + 0) * C + c), q = load8(moments + ((int64_t)img * 2 + 1) * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = m.v[e]; inv[e] = 1.f / (4.f * (k + q.v[e] * inv_nm1)); }
    };
    if (fixed && i0 < total) params(cc);
    for (int64_t i = i0; i < total; i += step) {
        const int c = (int)(i % chunks) * 8; const int64_t r = (int64_t)img * rows + i / chunks;
        if (!fixed) params(c);
        F8 v = load8(x + r * ldx + c), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v.v[e] - mu[e]; o.v[e] = v.v[e] * sigmoid_f(d * d * inv[e] + 0.5f); }
        store8(y + r * ldy + c, o);
    }
}

extern "C" int ss_simam_apply(int dtype, const void* x, int64_t ldx, const float* moments, const float* k, void* y, int64_t ldy,
                              int64_t rows_per_image, int C, int images, hipStream_t stream) {
    SS_CHECK(x && moments && k && y && rows_per_image > 0 && images > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0 &&
             aligned16(x) && aligned16(y) && aligned16(moments), "ss_simam_apply: bad arguments (C=%d images=%d)", C, images);
    int64_t bpi = cdiv(rows_per_image * (C / 8), 256 * 4);
    bpi = bpi < 1 ? 1 : (bpi > 1024 ? 1024 : bpi);
    SS_CHECK(bpi * images < (1ll << 31), "ss_simam_apply: grid too large");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((simam_apply_kernel<T>), dim3((unsigned)(bpi * images)), dim3(256), 0, stream, (const T*)x, ldx, moments, k,
                                          (T*)y, ldy, rows_per_image, C, (int)bpi));
    SS_LAUNCH_CHECK();
    return 0;
}

// SimAM and the grouped 1x1 conv that consumes it (modules/heads.py:27-39 + :200-222) in one kernel: out[p][n] = sum_c W[n][c] att[p][c],
// att = x * sigmoid(...) as above.  As two launches the attention map is written and read back (1.07 GB at the finest stage of TopFormer-B,
// 4 x 2048^2, for 32 MB of output).  A workgroup owns 64 rows of one image: a thread keeps one 8-channel chunk (its mu / inv stay in
// registers), att is rounded to the compute dtype into an LDS tile -- the value the two-launch form stored -- and waves 0 and 1 multiply
// the tile by W on the matrix cores (weights as the A operand: the lane is a row, its first register quads the N <= 32 outputs).
__global__ __launch_bounds__(256) void simam_gemm_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ moments,
        const float* __restrict__ kp, const bf16* __restrict__ w, int64_t ldw, float* __restrict__ out, int64_t ldo, int64_t rows, int C, int N,
        int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) char tile[];           // [64][C] bf16, pitch C*2 + 16
    const int PT = C * 2 + 16;
    const int img = blockIdx.x / tiles_per_image, t = blockIdx.x - img * tiles_per_image;
    const int64_t r0 = (int64_t)t * 64;                                    // first row of the tile inside the image
    const int chunks = C >> 3, tid = threadIdx.x;
    const float k = kp[0], inv_nm1 = 1.f / (float)(rows - 1);
    const int rpp = 256 / chunks;                                          // rows per pass (chunks divides 256: C in 64 .. 2048, power of two)
    const int ch = tid % chunks, rl = tid / chunks, c = ch * 8;
    float mu[8], inv[8];
    { const F8 m = load8(moments + ((int64_t)img * 2 + 0) * C + c), q = load8(moments + ((int64_t)img * 2 + 1) * C + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { mu[e] = m.v[e]; inv[e] = 1.f / (4.f * (k + q.v[e] * inv_nm1)); } }
    const bf16* xp = x + ((int64_t)img * rows + r0) * ldx + c;
#pragma unroll 4
    for (int r = rl; r < 64; r += rpp) {
        F8 o = zero8();
        if (r0 + r < rows) {
            const F8 v = load8(xp + (int64_t)r * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v.v[e] - mu[e]; o.v[e] = v.v[e] * sigmoid_f(d * d * inv[e] + 0.5f); }
        }
        store8((bf16*)(tile + r * PT) + c, o);
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    if (wave >= 2) return;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16* wp = w + (int64_t)(li < N ? li : 0) * ldw + lh * 8;
    const char* tp = tile + (wave * 32 + li) * PT + lh * 16;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int ks = 0; ks < (C >> 4); ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(li < N ? *(const bf16x8*)(wp + ks * 16) : zero, *(const bf16x8*)(tp + ks * 32), acc, 0, 0, 0);
    const int64_t row = r0 + wave * 32 + li;
    if (row < rows) {
        float* op = out + ((int64_t)img * rows + row) * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = 8 * g + 4 * lh;
            if (n0 < N) *(floatx4*)(op + n0) = floatx4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
    }
}

extern "C" int ss_simam_gemm_ok(int dtype, int C, int N) {
    // the tile needs 64 * (2C + 16) bytes of LDS: 129 KiB at C = 1024, more than the CU's 160 KiB at C = 2048
    return dtype == SS_BF16 && C >= 64 && C <= 1024 && (C & (C - 1)) == 0 && N > 0 && N <= 32 && N % 4 == 0;
}

extern "C" int ss_simam_gemm(int dtype, const void* x, int64_t ldx, const float* moments, const float* k, const void* w, int64_t ldw, float* out,
                             int64_t ldo, int64_t rows_per_image, int C, int images, int N, hipStream_t stream) {
    SS_CHECK(x && moments && k && w && out && rows_per_image > 1 && images > 0 && ss_simam_gemm_ok(dtype, C, N),
             "ss_simam_gemm: needs bf16, C a power of two in 64..1024, N <= 32 and a multiple of 4 (C=%d N=%d)", C, N);
    SS_CHECK(ldx >= C && ldw >= C && ldo >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(w) && aligned16(out) && aligned16(moments),
             "ss_simam_gemm: operands must be 16-byte aligned");
    const int64_t tpi = cdiv(rows_per_image, 64);
    SS_CHECK(tpi * images < (1LL << 31), "ss_simam_gemm: grid too large");
    const size_t lds = (size_t)64 * (C * 2 + 16);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)simam_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ss_fail(-2, "ss_simam_gemm: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(simam_gemm_kernel, dim3((unsigned)(tpi * images)), dim3(256), lds, stream, (const bf16*)x, ldx, moments, k, (const bf16*)w, ldw, out, ldo,
                       rows_per_image, C, N, (int)tpi);
    SS_LAUNCH_CHECK();
    return 0;
}

// GroupNorm folded into the linear map that follows it.  Input columns: `groups` blocks of `pitch` columns, the first `cg` of each
// valid (input channel o = g*cg + j sits in column g*pitch + j; the rest, up to `ncol` columns, is padding).  GroupNorm(groups, groups*cg) statistics per
// image from the per-column moments (mean_g = avg of the column means, var_g = (sum M2_j + n (mean_j - mean_g)^2) / (cg n), biased),
// t[o] = (s[o] - mean_g) rstd_g gamma[o] + beta[o], then  out[r] = sum_o L[r][o] t[o]  with the fixed matrix L [lrows][groups*cg]:
//   M[img][r][col(o)] = L[r][o] rstd_g gamma[o]   (M: [images][mrows][ncol]),   cvec[img][r] = sum_o L[r][o] (beta[o] - mean_g rstd_g gamma[o]);
// rows r >= lrows of M (up to mrows) and padding columns are zero.  One block per image.
__global__ __launch_bounds__(256) void refiner_fold_kernel(const float* __restrict__ moments, int64_t n, int groups, int cg, int pitch, int ncol, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ L,
                                                           int lrows, int mrows, float* __restrict__ M, float* __restrict__ cvec) {
    extern __shared__ float sm[];                     // a[groups*cg], b[groups*cg]
    const int img = blockIdx.x, nch = groups * cg;
    float* a = sm; float* b = sm + nch;
    const float* mean = moments + (int64_t)img * 2 * ncol; const float* m2 = mean + ncol;
    for (int o = threadIdx.x; o < nch; o += blockDim.x) {
        const int g = o / cg;
        double mg = 0.0;
        for (int j = 0; j < cg; ++j) mg += (double)mean[g * pitch + j];
        mg /= (double)cg;
        double ss = 0.0;
        for (int j = 0; j < cg; ++j) { const double d = (double)mean[g * pitch + j] - mg; ss += (double)m2[g * pitch + j] + (double)n * d * d; }
        const double rstd = 1.0 / sqrt(ss / ((double)cg * (double)n) + (double)eps);
        const double ao = rstd * (double)gamma[o];
        a[o] = (float)ao; b[o] = (float)((double)beta[o] - mg * ao);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < mrows * ncol; i += blockDim.x) {
        const int r = i / ncol, col = i % ncol, g = col / pitch, j = col % pitch;
        float v = 0.f;
        if (r < lrows && g < groups && j < cg) { const int o = g * cg + j; v = L[r * nch + o] * a[o]; }
        M[(int64_t)img * mrows * ncol + i] = v;
    }
    for (int r = threadIdx.x; r < mrows; r += blockDim.x) {
        float s = 0.f;
        if (r < lrows) for (int o = 0; o < nch; ++o) s += L[r * nch + o] * b[o];
        cvec[(int64_t)img * mrows + r] = s;
    }
}

extern "C" int ss_refiner_fold(const float* moments, int64_t rows_per_image, int images, int groups, int cg, int pitch, int ncol, float eps, const float* gamma,
                               const float* beta, const float* L, int lrows, int mrows, float* M, float* cvec, hipStream_t stream) {
    SS_CHECK(moments && gamma && beta && L && M && cvec && rows_per_image > 0 && images > 0 && groups > 0 && cg > 0 && pitch >= cg && lrows > 0 &&
             mrows >= lrows && ncol >= (groups - 1) * pitch + cg && (int64_t)groups * cg * 2 * 4 <= 48 * 1024, "ss_refiner_fold: bad arguments (groups=%d cg=%d pitch=%d)", groups, cg, pitch);
    hipLaunchKernelGGL(refiner_fold_kernel, dim3(images), dim3(256), (size_t)groups * cg * 2 * sizeof(float), stream, moments, rows_per_image, groups, cg,
                       pitch, ncol, eps, gamma, beta, L, lrows, mrows, M, cvec);
    SS_LAUNCH_CHECK();
    return 0;
}

// out[p][r] = sum_c M[img][r][c] f(in[p][c]) + cvec[img][r]; in fp32 [P, CIN], out TO [P, mrows]; the image's matrix sits in LDS,
// a thread owns one pixel and 8 output columns.
template <typename TO, bool SIG>
__global__ __launch_bounds__(256) void image_affine_kernel(const float* __restrict__ in, int64_t ldi, const float* __restrict__ M, const float* __restrict__ cvec,
                                                           TO* __restrict__ out, int64_t ldo, int64_t rows, int cin, int mrows, int blocks_per_image) {
    extern __shared__ float sm[];                     // M[mrows][cin], c[mrows]
    const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    for (int i = threadIdx.x; i < mrows * cin; i += blockDim.x) sm[i] = M[(int64_t)img * mrows * cin + i];
    for (int i = threadIdx.x; i < mrows; i += blockDim.x) sm[mrows * cin + i] = cvec[(int64_t)img * mrows + i];
    __syncthreads();
    const int oc = mrows / 8;
    const int64_t total = rows * oc;
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total; i += (int64_t)blocks_per_image * blockDim.x) {
        const int r0 = (int)(i % oc) * 8; const int64_t p = (int64_t)img * rows + i / oc;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = sm[mrows * cin + r0 + e];
        const float* src = in + p * ldi;
        for (int c = 0; c < cin; c += 4) {
            floatx4 v = *(const floatx4*)(src + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float f = SIG ? sigmoid_f(v[u]) : v[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += sm[(r0 + e) * cin + c + u] * f;
            }
        }
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = acc[e];
        store8(out + p * ldo + r0, o);
    }
}

extern "C" int ss_image_affine(int out_dtype, const float* in, int64_t ldi, const float* M, const float* cvec, void* out, int64_t ldo,
                               int64_t rows_per_image, int images, int cin, int mrows, int pre_act, hipStream_t stream) {
    SS_CHECK(in && M && cvec && out && rows_per_image > 0 && images > 0 && cin > 0 && cin % 4 == 0 && mrows > 0 && mrows % 8 == 0 && ldi >= cin &&
             ldi % 4 == 0 && ldo >= mrows && ldo % 8 == 0 && aligned16(in) && aligned16(out) && (pre_act == 0 || pre_act == 5) &&
             (int64_t)(mrows + 1) * cin * 4 + mrows * 4 <= 64 * 1024, "ss_image_affine: bad arguments (cin=%d mrows=%d pre_act=%d)", cin, mrows, pre_act);
    int64_t bpi = cdiv(rows_per_image * (mrows / 8), 256);
    bpi = bpi < 1 ? 1 : (bpi > 2048 ? 2048 : bpi);
    SS_CHECK(bpi * images < (1ll << 31), "ss_image_affine: grid too large");
    const size_t shm = ((size_t)mrows * cin + mrows) * sizeof(float);
#define SS_AFF(TO, SIG) hipLaunchKernelGGL((image_affine_kernel<TO, SIG>), dim3((unsigned)(bpi * images)), dim3(256), shm, stream, in, ldi, M, cvec, \
                                           (TO*)out, ldo, rows_per_image, cin, mrows, (int)bpi)
    if (out_dtype == SS_F32) { if (pre_act == 5) SS_AFF(float, true); else SS_AFF(float, false); }
    else if (out_dtype == SS_BF16) { if (pre_act == 5) SS_AFF(bf16, true); else SS_AFF(bf16, false); }
    else return ss_fail(-1, "ss_image_affine: unknown dtype %d", out_dtype);
#undef SS_AFF
    SS_LAUNCH_CHECK();
    return 0;
}
