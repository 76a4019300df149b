// Backward-side kernels of the TopFormer training path (modules/topformer.py, modules/heads.py, modules/blocks.py in train mode;
// train.py:45-52 trains TopFormerModel('B', head refiner)).  The forward of these layers already exists (dwconvk.hip, topformer.hip,
// refiner.hip); what a training step adds is
//   * the data and weight gradients of the depthwise convolutions with stride 1 / 2 and the asymmetric AutoPadding
//     (modules/layers.py:146-164): ss_dwconvk_dgrad, ss_dwconvk_wgrad_s;
//   * the gradient of the pyramid pooling's exact-window average (modules/topformer.py:527-552): ss_avgpool_bwd;
//   * activation gradients from the activation's OUTPUT (ReLU / ReLU6 / sigmoid: the mask is a function of y): ss_act_bwd;
//   * element-wise products (InjectionDotSum, modules/topformer.py:640-673): ss_mul;
//   * SimAM (modules/heads.py:22-39) forward with saved statistics and its backward: ss_simam_fwd / ss_simam_bwd;
//   * GroupNorm over channels-last maps with a fused pre-activation (sigmoid) in front of it (modules/heads.py:200-238):
//     ss_groupnorm_stats / ss_groupnorm_apply / ss_groupnorm_bwd.
// All bandwidth-bound and small next to the inference path they sit beside: plain channels-last kernels, 8 channels per thread,
// reductions in fixed order (no floating-point atomics).
#include "common.h"
#include "colreduce.h"
#include "bilinear.h"

namespace {

__device__ __forceinline__ float tt_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

inline int tt_blocks(int64_t items, int cap = 8192) { int64_t b = cdiv(items, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// ------------------------------------------------------------------ depthwise k x k, stride s: data gradient
// dx[b, y, x, c] = sum_{ky, kx} w[ky, kx, c] * g[b, (y + pt - ky) / s, (x + pl - kx) / s, c]   over taps whose quotient is exact and in range
// Compile-time K and S: the taps that can contribute to an input pixel are ky = ky0, ky0 + S, ... with ky0 = (y + pt) % S, i.e. ceil(K / S)
// per axis; every one is an unconditional load from a clamped position, masked where the tap or its output pixel does not exist.
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwk_dgrad_kernel(const T* __restrict__ g, const float* __restrict__ w, T* __restrict__ dx, int B, int H, int W,
                                                        int C, int pt, int pl, int Ho, int Wo) {
    constexpr int NT = (K + S - 1) / S;
    const int nch = C / 8;
    const int64_t items = (int64_t)B * H * W * nch;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(it % nch);
        int64_t p = it / nch;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        const int ky0 = (y + pt) % S, kx0 = (x + pl) % S;
        const T* gb = g + (int64_t)b * Ho * Wo * C + ch * 8;
        const float* wb = w + ch * 8;
        F8 gv[NT][NT], wv[NT][NT]; float m[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int ky = ky0 + a * S, oy = (y + pt - ky) / S;                   // exact: y + pt - ky is a multiple of S (or negative)
            const bool yok = ky < K && y + pt - ky >= 0 && oy < Ho;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int kx = kx0 + c * S, ox = (x + pl - kx) / S;
                const bool ok = yok && kx < K && x + pl - kx >= 0 && ox < Wo;
                m[a][c] = ok ? 1.f : 0.f;
                gv[a][c] = load8(gb + ((int64_t)(ok ? oy : 0) * Wo + (ok ? ox : 0)) * C);
                wv[a][c] = load8(wb + (int64_t)(ok ? ky * K + kx : 0) * C);
            }
        }
        F8 acc = zero8();
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] = fmaf(gv[a][c].v[e] * m[a][c], wv[a][c].v[e], acc.v[e]);
        store8(dx + (((int64_t)b * H + y) * W + x) * C + ch * 8, acc);
    }
}

// ------------------------------------------------------------------ depthwise k x k, stride s: weight + bias gradient
// dw[ky, kx, c] = sum_{b, oy, ox} g[b, oy, ox, c] * x[b, oy*s - pt + ky, ox*s - pl + kx, c];  db[c] = sum g.
// A thread owns CH consecutive channels of one pixel lane and all K*K + 1 sums for them (CH = 8 for 3x3, 4 for 5x5: 80 / 104
// accumulators); consecutive threads take consecutive channel chunks of one pixel, so a wave reads whole pixel rows; the pixel lanes of
// a workgroup are combined through LDS in a fixed order; partial [nb][(K*K+1)][C], reduced by colreduce_final_kernel.
template <typename T, int CH> __device__ __forceinline__ void load_n(const T* p, float (&v)[CH]);
template <> __device__ __forceinline__ void load_n<float, 8>(const float* p, float (&v)[8]) { const F8 r = load8(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r.v[e]; }
template <> __device__ __forceinline__ void load_n<bf16, 8>(const bf16* p, float (&v)[8]) { const F8 r = load8(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r.v[e]; }
template <> __device__ __forceinline__ void load_n<float, 4>(const float* p, float (&v)[4]) { const floatx4 r = *(const floatx4*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e]; }
template <> __device__ __forceinline__ void load_n<bf16, 4>(const bf16* p, float (&v)[4]) { const bf16x4 r = *(const bf16x4*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)r[e]; }

template <typename T, int K, int CH>
__global__ __launch_bounds__(256) void dwk_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ x, float* __restrict__ partial, int B, int H,
                                                        int W, int C, int s, int pt, int pl, int Ho, int Wo, int cpb) {
    constexpr int KK = K * K;
    __shared__ float sh[256 * CH];
    const int tid = threadIdx.x, cl = tid % cpb, lane = tid / cpb, nlanes = 256 / cpb;
    const int c = (blockIdx.y * cpb + cl) * CH;
    const bool cok = c < C;
    float acc[KK + 1][CH];
#pragma unroll
    for (int t = 0; t <= KK; ++t)
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[t][e] = 0.f;
    const int64_t npix = (int64_t)B * Ho * Wo;
    if (cok) {
        for (int64_t p = (int64_t)blockIdx.x * nlanes + lane; p < npix; p += (int64_t)gridDim.x * nlanes) {
            const int ox = (int)(p % Wo); const int64_t q = p / Wo;
            const int oy = (int)(q % Ho);
            const int b = (int)(q / Ho);
            float gv[CH];
            load_n<T, CH>(g + p * C + c, gv);
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[KK][e] += gv[e];
            // every tap is an unconditional load from a clamped position (a load under a branch is waited for at the join: K * K round
            // trips per pixel); taps outside the image take the gradient times zero
            const T* xb = x + (int64_t)b * H * W * C + c;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy * s - pt + ky;
                const bool yok = iy >= 0 && iy < H;
                const int64_t rowoff = (int64_t)min(max(iy, 0), H - 1) * W;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ox * s - pl + kx;
                    const float m = (yok && ix >= 0 && ix < W) ? 1.f : 0.f;
                    float xv[CH];
                    load_n<T, CH>(xb + (rowoff + min(max(ix, 0), W - 1)) * C, xv);
#pragma unroll
                    for (int e = 0; e < CH; ++e) acc[ky * K + kx][e] = fmaf(gv[e] * m, xv[e], acc[ky * K + kx][e]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t <= KK; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < CH; ++e) sh[tid * CH + e] = acc[t][e];
        __syncthreads();
        // CH * cpb sums of nlanes terms each: spread over the workgroup's threads (thread u sums column u)
        for (int u = tid; u < cpb * CH; u += 256) {
            const int ucl = u / CH, ue = u % CH, uc = (blockIdx.y * cpb + ucl) * CH + ue;
            if (uc < C) {
                float sum = 0.f;
                for (int j = 0; j < nlanes; ++j) sum += sh[(j * cpb + ucl) * CH + ue];
                partial[((int64_t)blockIdx.x * (KK + 1) + t) * C + uc] = sum;
            }
        }
    }
}

// ------------------------------------------------------------------ exact-window average pooling: gradient
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ g, int64_t ldg, T* __restrict__ dx, int64_t ldx, int B, int H, int W,
                                                          int C, int win) {
    const int nch = C / 8, ho = H / win, wo = W / win;
    const float inv = 1.f / (float)(win * win);
    const int64_t items = (int64_t)B * H * W * nch;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(it % nch);
        int64_t p = it / nch;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        F8 v = load8(g + (((int64_t)b * ho + y / win) * wo + x / win) * ldg + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] *= inv;
        store8(dx + (((int64_t)b * H + y) * W + x) * ldx + ch * 8, v);
    }
}

// ------------------------------------------------------------------ element-wise: activation gradient from the output, product
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, int64_t n8, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const F8 d = load8(dy + i * 8), v = load8(y + i * 8);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float yy = v.v[e];
            float m;
            if (act == 2) m = yy > 0.f ? 1.f : 0.f;
            else if (act == 3) m = (yy > 0.f && yy < 6.f) ? 1.f : 0.f;
            else if (act == 5) m = yy * (1.f - yy);
            else m = 1.f;
            o.v[e] = d.v[e] * m;
        }
        store8(dx + i * 8, o);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void mul_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const F8 u = load8(a + i * 8), v = load8(b + i * 8);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = u.v[e] * v.v[e];
        store8(out + i * 8, o);
    }
}


// ------------------------------------------------------------------ SimAM backward (modules/heads.py:27-39)
// y = x s, s = sigmoid(e), e = d^2 / D + 0.5, d = x - mu, D = 4 (k + M2 / (n - 1)).  With ge = dy x s (1 - s):
//   T1 = sum ge d^2, T2 = sum ge d  per (image, channel);  gD = -T1 / D^2;  dk += 4 gD;
//   dx = dy s + 2 ge d / D + 8 gD d / (n - 1) - 2 T2 / (n D)
template <typename T> struct SimamBwdSumF {
    static constexpr int K = 2;
    const T* dy; const T* x; int64_t ldd, ldx; const float* moments; const float* kp; int64_t rows; int C;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        const int img = blockIdx.y;
        const F8 g = load8(dy + row * ldd + c0), v = load8(x + row * ldx + c0);
        const F8 mu = load8(moments + ((int64_t)img * 2 + 0) * C + c0), m2 = load8(moments + ((int64_t)img * 2 + 1) * C + c0);
        const float k = kp[0], inv_nm1 = 1.f / (float)(rows - 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v.v[e] - mu.v[e], inv = 1.f / (4.f * (k + m2.v[e] * inv_nm1));
            const float s = tt_sigmoid(d * d * inv + 0.5f), ge = g.v[e] * v.v[e] * s * (1.f - s);
            acc[0][e] += ge * d * d; acc[1][e] += ge * d;
        }
    }
};
template <typename T>
__global__ __launch_bounds__(256) void simam_bwd_dx_kernel(const T* __restrict__ dy, int64_t ldd, const T* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ moments, const float* __restrict__ kp, const float* __restrict__ sums,
                                                           T* __restrict__ dx, int64_t ldo, int64_t rows, int C, int blocks_per_image) {
    const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    const int chunks = C / 8;
    const float k = kp[0], inv_nm1 = 1.f / (float)(rows - 1), inv_n = 1.f / (float)rows;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total; i += (int64_t)blocks_per_image * blockDim.x) {
        const int c = (int)(i % chunks) * 8; const int64_t r = (int64_t)img * rows + i / chunks;
        const F8 g = load8(dy + r * ldd + c), v = load8(x + r * ldx + c);
        const F8 mu = load8(moments + ((int64_t)img * 2 + 0) * C + c), m2 = load8(moments + ((int64_t)img * 2 + 1) * C + c);
        const F8 t1 = load8(sums + ((int64_t)img * 2 + 0) * C + c), t2 = load8(sums + ((int64_t)img * 2 + 1) * C + c);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v.v[e] - mu.v[e], invD = 1.f / (4.f * (k + m2.v[e] * inv_nm1));
            const float s = tt_sigmoid(d * d * invD + 0.5f), ge = g.v[e] * v.v[e] * s * (1.f - s);
            const float gD = -t1.v[e] * invD * invD;
            o.v[e] = g.v[e] * s + 2.f * ge * d * invD + 8.f * gD * d * inv_nm1 - 2.f * t2.v[e] * invD * inv_n;
        }
        store8(dx + r * ldo + c, o);
    }
}
// dk = sum over (image, channel) of 4 gD = -4 T1 / D^2: one block, fixed order, double accumulation
__global__ __launch_bounds__(256) void simam_dk_kernel(const float* __restrict__ sums, const float* __restrict__ moments, const float* __restrict__ kp,
                                                       int64_t rows, int C, int images, float* __restrict__ dk, int accumulate) {
    __shared__ double sh[256];
    const float k = kp[0], inv_nm1 = 1.f / (float)(rows - 1);
    double s = 0.0;
    for (int i = threadIdx.x; i < images * C; i += 256) {
        const int img = i / C, c = i % C;
        const float invD = 1.f / (4.f * (k + moments[((int64_t)img * 2 + 1) * C + c] * inv_nm1));
        s += -4.0 * (double)sums[((int64_t)img * 2 + 0) * C + c] * (double)invD * (double)invD;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 256; ++i) t += sh[i]; dk[0] = accumulate ? dk[0] + (float)t : (float)t; }
}

// ------------------------------------------------------------------ GroupNorm over channels-last fp32 maps with an optional sigmoid in front
// Column g*pitch + j (j < cg) holds channel g*cg + j; everything else up to ncol is padding (reads ignored, writes zero).
// coef[img][0][col] = A = gamma rstd_g, [1] = B = beta - mean_g A;  stats[img][g] = (mean_g, rstd_g).  One thread per (image, group).
__global__ __launch_bounds__(64) void gn_coef_kernel(const float* __restrict__ moments, int64_t n, int images, int groups, int cg, int pitch, int ncol, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ coef,
                                                     float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= images * groups) return;
    const int img = i / groups, g = i % groups;
    const float* mean = moments + (int64_t)img * 2 * ncol; const float* m2 = mean + ncol;
    double mg = 0.0;
    for (int j = 0; j < cg; ++j) mg += (double)mean[g * pitch + j];
    mg /= (double)cg;
    double ss = 0.0;
    for (int j = 0; j < cg; ++j) { const double d = (double)mean[g * pitch + j] - mg; ss += (double)m2[g * pitch + j] + (double)n * d * d; }
    const double rstd = 1.0 / sqrt(ss / ((double)cg * (double)n) + (double)eps);
    stats[((int64_t)img * groups + g) * 2 + 0] = (float)mg; stats[((int64_t)img * groups + g) * 2 + 1] = (float)rstd;
    float* A = coef + (int64_t)img * 2 * ncol; float* Bc = A + ncol;
    const int cend = g == groups - 1 ? ncol : (g + 1) * pitch;
    for (int col = g * pitch; col < cend; ++col) {
        const int j = col - g * pitch;
        if (j < cg) { const double a = rstd * (double)gamma[g * cg + j]; A[col] = (float)a; Bc[col] = (float)((double)beta[g * cg + j] - mg * a); }
        else { A[col] = 0.f; Bc[col] = 0.f; }
    }
}
// y = A[img][c] f(x) + B[img][c]
template <bool SIG>
__global__ __launch_bounds__(256) void image_scale_shift_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ coef, float* __restrict__ y,
                                                                int64_t ldy, int64_t rows, int C, int blocks_per_image) {
    const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    const int chunks = C / 8;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total; i += (int64_t)blocks_per_image * blockDim.x) {
        const int c = (int)(i % chunks) * 8; const int64_t r = (int64_t)img * rows + i / chunks;
        const F8 v = load8(x + r * ldx + c), a = load8(coef + ((int64_t)img * 2 + 0) * C + c), b = load8(coef + ((int64_t)img * 2 + 1) * C + c);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = fmaf(a.v[e], SIG ? tt_sigmoid(v.v[e]) : v.v[e], b.v[e]);
        store8(y + r * ldy + c, o);
    }
}
// S[img][0][c] = sum dy, [1][c] = sum dy f(x)
template <bool SIG> struct GnBwdSumF {
    static constexpr int K = 2;
    const float* dy; const float* x; int64_t ldd, ldx;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        const F8 g = load8(dy + row * ldd + c0), v = load8(x + row * ldx + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = SIG ? tt_sigmoid(v.v[e]) : v.v[e]; acc[0][e] += g.v[e]; acc[1][e] += g.v[e] * f; }
    }
};
// With xhat = (f - mean) rstd and dxhat = gamma dy:  m1 = mean_g(dxhat), m2 = mean_g(dxhat xhat) over the group's cg n elements;
//   d f = rstd (gamma dy - m1 - xhat m2) = P dy + Q f + R,  P = rstd gamma, Q = -rstd^2 m2, R = -rstd m1 + rstd^2 mean m2
//   dgamma[o] = sum_img rstd (S2 - mean S1), dbeta[o] = sum_img S1.   One block; thread per channel o, images in order.
__global__ __launch_bounds__(256) void gn_bwd_coef_kernel(const float* __restrict__ S, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          int64_t n, int images, int groups, int cg, int pitch, int ncol, float* __restrict__ pqr,
                                                          float* __restrict__ dgb) {
    const int nch = groups * cg;
    for (int i = threadIdx.x; i < images * 3 * ncol; i += blockDim.x) pqr[i] = 0.f;
    __syncthreads();
    for (int o = threadIdx.x; o < nch; o += blockDim.x) {
        const int g = o / cg, j = o % cg, col = g * pitch + j;
        double dg = 0.0, db = 0.0;
        for (int img = 0; img < images; ++img) {
            const float* s1 = S + (int64_t)img * 2 * ncol; const float* s2 = s1 + ncol;
            const double mean = stats[((int64_t)img * groups + g) * 2], rstd = stats[((int64_t)img * groups + g) * 2 + 1];
            double a1 = 0.0, a2 = 0.0;
            for (int jj = 0; jj < cg; ++jj) {
                const double gm = gamma[g * cg + jj], t1 = s1[g * pitch + jj], t2 = s2[g * pitch + jj];
                a1 += gm * t1; a2 += gm * rstd * (t2 - mean * t1);
            }
            const double cnt = (double)cg * (double)n, m1 = a1 / cnt, m2 = a2 / cnt;
            float* o3 = pqr + (int64_t)img * 3 * ncol;
            o3[col] = (float)(rstd * (double)gamma[o]); o3[ncol + col] = (float)(-rstd * rstd * m2); o3[2 * ncol + col] = (float)(-rstd * m1 + rstd * rstd * mean * m2);
            dg += rstd * ((double)s2[col] - mean * (double)s1[col]); db += (double)s1[col];
        }
        dgb[o] = (float)dg; dgb[nch + o] = (float)db;
    }
}
template <bool SIG>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const float* __restrict__ dy, int64_t ldd, const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ pqr, float* __restrict__ dx, int64_t ldo, int64_t rows, int C,
                                                        int blocks_per_image) {
    const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    const int chunks = C / 8;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total; i += (int64_t)blocks_per_image * blockDim.x) {
        const int c = (int)(i % chunks) * 8; const int64_t r = (int64_t)img * rows + i / chunks;
        const F8 g = load8(dy + r * ldd + c), v = load8(x + r * ldx + c);
        const F8 P = load8(pqr + ((int64_t)img * 3 + 0) * C + c), Q = load8(pqr + ((int64_t)img * 3 + 1) * C + c), R = load8(pqr + ((int64_t)img * 3 + 2) * C + c);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = SIG ? tt_sigmoid(v.v[e]) : v.v[e];
            const float df = fmaf(P.v[e], g.v[e], fmaf(Q.v[e], f, R.v[e]));
            o.v[e] = SIG ? df * f * (1.f - f) : df;
        }
        store8(dx + r * ldo + c, o);
    }
}

inline int64_t tt_bpi(int64_t rows, int C) { int64_t b = cdiv(rows * (C / 8), 256 * 4); return b < 1 ? 1 : (b > 1024 ? 1024 : b); }


// ------------------------------------------------------------------ InjectionDotSum backward in one pass (modules/topformer.py:640-673)
// out = local * up(gact) + up(gemb)  =>  dlocal = g * up(gact),  dgact = up^T(g * local),  dgemb = up^T(g),  up = bilinear resize by an
// even integer factor r.  The transposed resize is a weighted sum over 2r x 2r output pixels per coarse pixel; done as a gather per
// coarse pixel (ss_bilinear_bwd) every g element is fetched four times, twice over (two maps) and the product g * local makes a round
// trip through memory.  Here the fine map is cut into cells of r x r pixels shifted by r/2 (cell (jb, kb): rows [r jb - r/2, r jb + r/2)):
// all pixels of a cell share their four coarse taps (rows jb-1 / jb, columns kb-1 / kb, clamped), so a workgroup reads its cell of g and
// local ONCE, writes dlocal, and leaves eight partial sums per channel -- (g, g * local) x four taps; a second small kernel gives each
// coarse pixel the four cells around it, in a fixed order.  Weights come from lin_tap, the taps ss_bilinear_fwd uses.
template <typename T>
__global__ __launch_bounds__(256) void inject_bwd_kernel(const T* __restrict__ g, const T* __restrict__ local, const T* __restrict__ gact, int64_t lda,
                                                         T* __restrict__ dlocal, float* __restrict__ partial, int hl, int wl, int hg, int wg, int C, int r,
                                                         int nl) {
    __shared__ float red[256 * 8];
    const int cell = blockIdx.x, kb = cell % (wg + 1), jb = (cell / (wg + 1)) % (hg + 1), b = cell / ((wg + 1) * (hg + 1));
    const int chunks = C / 8, tid = threadIdx.x, ch = tid % chunks, pl = tid / chunks;
    const bool active = pl < nl;
    const int oy0 = max(0, r * jb - r / 2), oy1 = min(hl, r * jb + r / 2), ox0 = max(0, r * kb - r / 2), ox1 = min(wl, r * kb + r / 2);
    const int rlo = max(jb - 1, 0), rhi = min(jb, hg - 1), clo = max(kb - 1, 0), chi = min(kb, wg - 1);
    const float sh = (float)hg / hl, sw = (float)wg / wl;
    float acc[2][4][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][s][e] = 0.f;
    if (active) {
        const LinTap ty0 = lin_tap(oy0, sh, hg), tx0 = lin_tap(ox0, sw, wg);          // the cell's taps (weights differ per pixel, indices do not)
        const T* gb = gact + (int64_t)b * hg * wg * lda + ch * 8;
        const F8 c00 = load8(gb + ((int64_t)ty0.i0 * wg + tx0.i0) * lda), c01 = load8(gb + ((int64_t)ty0.i0 * wg + tx0.i1) * lda);
        const F8 c10 = load8(gb + ((int64_t)ty0.i1 * wg + tx0.i0) * lda), c11 = load8(gb + ((int64_t)ty0.i1 * wg + tx0.i1) * lda);
        const int bw = ox1 - ox0, npx = (oy1 - oy0) * bw;
        for (int p = pl; p < npx; p += nl) {
            const int oy = oy0 + p / bw, ox = ox0 + p % bw;
            const LinTap ty = lin_tap(oy, sh, hg), tx = lin_tap(ox, sw, wg);
            const float wyl = (ty.i0 == rlo ? ty.l0 : 0.f) + (ty.i1 == rlo ? ty.l1 : 0.f);
            const float wyh = rhi != rlo ? (ty.i0 == rhi ? ty.l0 : 0.f) + (ty.i1 == rhi ? ty.l1 : 0.f) : 0.f;
            const float wxl = (tx.i0 == clo ? tx.l0 : 0.f) + (tx.i1 == clo ? tx.l1 : 0.f);
            const float wxh = chi != clo ? (tx.i0 == chi ? tx.l0 : 0.f) + (tx.i1 == chi ? tx.l1 : 0.f) : 0.f;
            const float w[4] = {wyl * wxl, wyl * wxh, wyh * wxl, wyh * wxh};
            const int64_t idx = (((int64_t)b * hl + oy) * wl + ox) * C + ch * 8;
            const F8 gv = load8(g + idx), lv = load8(local + idx);
            F8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = to_f(from_f<T>(bilerp(ty, tx, c00.v[e], c01.v[e], c10.v[e], c11.v[e])));      // the value ss_bilinear_fwd stores
                o.v[e] = gv.v[e] * u;
                const float q = gv.v[e] * lv.v[e];
#pragma unroll
                for (int s = 0; s < 4; ++s) { acc[0][s][e] = fmaf(w[s], gv.v[e], acc[0][s][e]); acc[1][s][e] = fmaf(w[s], q, acc[1][s][e]); }
            }
            store8(dlocal + idx, o);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[q][s][e];
            __syncthreads();
            if (pl == 0) {
                float* dst = partial + (((int64_t)cell * 2 + q) * 4 + s) * C + ch * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float sum = 0.f;
                    for (int j = 0; j < nl; ++j) sum += red[(j * chunks + ch) * 8 + e];
                    dst[e] = sum;
                }
            }
        }
}
template <typename T>
__global__ __launch_bounds__(256) void inject_bwd_final_kernel(const float* __restrict__ partial, T* __restrict__ dgact, int64_t lda, T* __restrict__ dgemb,
                                                               int64_t lde, int B, int hg, int wg, int C) {
    const int chunks = C / 8;
    const int64_t items = (int64_t)B * hg * wg * chunks;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(it % chunks);
        int64_t p = it / chunks;
        const int ix = (int)(p % wg); p /= wg;
        const int iy = (int)(p % hg);
        const int b = (int)(p / hg);
        F8 s0 = zero8(), s1 = zero8();
        for (int jb = iy; jb <= iy + 1; ++jb) {
            const int rlo = max(jb - 1, 0), rhi = min(jb, hg - 1);
            for (int vs = 0; vs < 2; ++vs) {
                if (vs == 0 ? rlo != iy : (rhi != iy || rhi == rlo)) continue;
                for (int kb = ix; kb <= ix + 1; ++kb) {
                    const int clo = max(kb - 1, 0), chi = min(kb, wg - 1);
                    for (int hs = 0; hs < 2; ++hs) {
                        if (hs == 0 ? clo != ix : (chi != ix || chi == clo)) continue;
                        const int64_t cell = ((int64_t)b * (hg + 1) + jb) * (wg + 1) + kb;
                        const F8 a = load8(partial + ((cell * 2 + 0) * 4 + vs * 2 + hs) * C + ch * 8);
                        const F8 q = load8(partial + ((cell * 2 + 1) * 4 + vs * 2 + hs) * C + ch * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s0.v[e] += a.v[e]; s1.v[e] += q.v[e]; }
                    }
                }
            }
        }
        const int64_t row = ((int64_t)b * hg + iy) * wg + ix;
        store8(dgemb + row * lde + ch * 8, s0);
        store8(dgact + row * lda + ch * 8, s1);
    }
}

}  // namespace

// g [B*Ho*Wo, C] -> dx [B*H*W, C]; wkkc: tap-major [k*k, C] fp32 (NOT flipped: the kernel walks the taps of the forward)
extern "C" int ss_dwconvk_dgrad(int dtype, const void* g, const float* wkkc, void* dx, int B, int H, int W, int C, int k, int stride,
                                int pad_top, int pad_left, int Ho, int Wo, hipStream_t stream) {
    SS_CHECK(g && wkkc && dx && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && k >= 1 && k <= 7 && (k & 1) && (stride == 1 || stride == 2) &&
             pad_top >= 0 && pad_left >= 0, "ss_dwconvk_dgrad: bad arguments (C=%d k=%d stride=%d; k must be 1, 3, 5 or 7)", C, k, stride);
    SS_CHECK(aligned16(g) && aligned16(dx) && aligned16(wkkc), "ss_dwconvk_dgrad: pointers must be 16-byte aligned");
    const int64_t items = (int64_t)B * H * W * (C / 8);
#define SS_DWK_DGRAD(K, S) SS_DISPATCH(dtype, hipLaunchKernelGGL((dwk_dgrad_kernel<T, K, S>), dim3(tt_blocks(items, 16384)), dim3(256), 0, stream, (const T*)g, wkkc, \
                                                        (T*)dx, B, H, W, C, pad_top, pad_left, Ho, Wo))
    switch (k * 10 + stride) {
        case 11: SS_DWK_DGRAD(1, 1); break; case 12: SS_DWK_DGRAD(1, 2); break;
        case 31: SS_DWK_DGRAD(3, 1); break; case 32: SS_DWK_DGRAD(3, 2); break;
        case 51: SS_DWK_DGRAD(5, 1); break; case 52: SS_DWK_DGRAD(5, 2); break;
        case 71: SS_DWK_DGRAD(7, 1); break; case 72: SS_DWK_DGRAD(7, 2); break;
        default: return ss_fail(-1, "ss_dwconvk_dgrad: k must be 1, 3, 5 or 7 (k=%d)", k);
    }
#undef SS_DWK_DGRAD
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db [(k*k + 1)][C] fp32 (taps tap-major, then the bias gradient) of y = dw_{k, stride, pads}(x); workspace >= 512 * (k*k+1) * C floats
extern "C" int ss_dwconvk_wgrad_s(int dtype, const void* g, const void* x, float* dw_db, int B, int H, int W, int C, int k, int stride,
                                  int pad_top, int pad_left, int Ho, int Wo, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(g && x && dw_db && workspace && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && (k == 3 || k == 5) &&
             (stride == 1 || stride == 2) && pad_top >= 0 && pad_left >= 0 && aligned16(g) && aligned16(x),
             "ss_dwconvk_wgrad_s: bad arguments (C=%d k=%d stride=%d; k must be 3 or 5, C a multiple of 8)", C, k, stride);
    const int CH = k == 3 ? 8 : 4, chunks = C / CH;
    int cpb = 1; while (cpb < chunks && cpb < 64) cpb <<= 1;
    const int groups = (int)cdiv(chunks, cpb), nlanes = 256 / cpb;
    const int64_t npix = (int64_t)B * Ho * Wo;
    int nb = (int)cdiv(npix, (int64_t)nlanes * 4);
    const int cap = groups >= 4 ? 256 : 512;
    if (nb > cap) nb = cap; if (nb < 1) nb = 1;
    const int rows = k * k + 1;
    SS_CHECK(ws_bytes >= (int64_t)nb * rows * C * 4, "ss_dwconvk_wgrad_s: workspace too small");
    dim3 grid((unsigned)nb, (unsigned)groups);
    if (k == 3) SS_DISPATCH(dtype, hipLaunchKernelGGL((dwk_wgrad_kernel<T, 3, 8>), grid, dim3(256), 0, stream, (const T*)g, (const T*)x, workspace, B, H, W, C,
                                                      stride, pad_top, pad_left, Ho, Wo, cpb));
    else SS_DISPATCH(dtype, hipLaunchKernelGGL((dwk_wgrad_kernel<T, 5, 4>), grid, dim3(256), 0, stream, (const T*)g, (const T*)x, workspace, B, H, W, C,
                                               stride, pad_top, pad_left, Ho, Wo, cpb));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(rows * C, 8), 1), dim3(256), 0, stream, (const float*)workspace, nb, rows * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// g [B*(H/win)*(W/win), C] (row pitch ldg) -> dx [B*H*W, C] (row pitch ldx): dx = g of the pixel's window / win^2
extern "C" int ss_avgpool_bwd(int dtype, const void* g, int64_t ldg, void* dx, int64_t ldx, int B, int H, int W, int C, int win, hipStream_t stream) {
    SS_CHECK(g && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && win >= 1 && H % win == 0 && W % win == 0 && ldg >= C && ldx >= C &&
             ldg % 8 == 0 && ldx % 8 == 0 && aligned16(g) && aligned16(dx), "ss_avgpool_bwd: bad arguments");
    const int64_t items = (int64_t)B * H * W * (C / 8);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(tt_blocks(items, 16384)), dim3(256), 0, stream, (const T*)g, ldg, (T*)dx, ldx, B, H, W,
                                          C, win));
    SS_LAUNCH_CHECK();
    return 0;
}

// dx = dy * act'(.) expressed through the activation's output y: act 2 ReLU, 3 ReLU6, 5 sigmoid (0: copy)
extern "C" int ss_act_bwd(int dtype, const void* dy, const void* y, void* dx, int64_t n, int act, hipStream_t stream) {
    SS_CHECK(dy && y && dx && n > 0 && n % 8 == 0 && aligned16(dy) && aligned16(y) && aligned16(dx) && (act == 0 || act == 2 || act == 3 || act == 5),
             "ss_act_bwd: need n %% 8 == 0, 16-byte aligned pointers and act in {0, 2, 3, 5}");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(tt_blocks(n / 8, 16384)), dim3(256), 0, stream, (const T*)dy, (const T*)y, (T*)dx, n / 8, act));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_mul(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t stream) {
    SS_CHECK(a && b && out && n > 0 && n % 8 == 0 && aligned16(a) && aligned16(b) && aligned16(out), "ss_mul: need n %% 8 == 0 and 16-byte aligned pointers");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((mul_kernel<T>), dim3(tt_blocks(n / 8, 16384)), dim3(256), 0, stream, (const T*)a, (const T*)b, (T*)out, n / 8));
    SS_LAUNCH_CHECK();
    return 0;
}

// ---- SimAM backward.  sums [images][2][C] fp32 (T1, T2 of the header comment); workspace >= images * 1024 * 2 * C floats.
extern "C" int ss_simam_bwd(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* moments, const float* k, void* dx,
                            int64_t ldo, float* dk, int accumulate_dk, float* sums, int64_t rows_per_image, int C, int images, float* workspace,
                            int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && moments && k && dx && dk && sums && workspace && rows_per_image > 1 && images > 0 && images <= 65535 && C > 0 && C % 8 == 0 &&
             ldd >= C && ldx >= C && ldo >= C && ldd % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx) &&
             aligned16(moments) && aligned16(sums), "ss_simam_bwd: bad arguments (C=%d images=%d)", C, images);
    ColGeom g = col_geom(rows_per_image, C);
    SS_CHECK(ws_bytes >= (int64_t)images * g.nb * 2 * C * 4, "ss_simam_bwd: workspace too small");
    const int64_t bpi = tt_bpi(rows_per_image, C);
    SS_CHECK(bpi * images < (1ll << 31), "ss_simam_bwd: grid too large");
    SS_DISPATCH(dtype, {
        SimamBwdSumF<T> f{(const T*)dy, (const T*)x, ldd, ldx, moments, k, rows_per_image, C};
        hipLaunchKernelGGL((colreduce_partial_kernel<SimamBwdSumF<T>>), dim3(g.nb, images), dim3(256), 0, stream, f, rows_per_image, C, g.cpb, g.rpb, workspace);
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(2 * C, 8), images), dim3(256), 0, stream, (const float*)workspace, g.nb, 2 * C, 1.f, sums, 0);
        hipLaunchKernelGGL((simam_bwd_dx_kernel<T>), dim3((unsigned)(bpi * images)), dim3(256), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, moments, k,
                           (const float*)sums, (T*)dx, ldo, rows_per_image, C, (int)bpi);
    });
    hipLaunchKernelGGL(simam_dk_kernel, dim3(1), dim3(256), 0, stream, (const float*)sums, moments, k, rows_per_image, C, images, dk, accumulate_dk);
    SS_LAUNCH_CHECK();
    return 0;
}

// ---- GroupNorm (modules/heads.py:200-222 'CAN' group conv: sigmoid then GroupNorm(n_heads); :251-268 the final GroupNorm) on fp32
// channels-last maps x [images * rows_per_image][ncol]; layout of the groups as ss_refiner_fold.  pre_act 0 / 5 (sigmoid before the norm).
// ss_groupnorm_fwd: moments = ss_image_moments(x, pre_act); writes coef [images][2][ncol], stats [images][groups][2] and y.
extern "C" int ss_groupnorm_fwd(const float* x, int64_t ldx, const float* moments, int64_t rows_per_image, int images, int groups, int cg, int pitch, int ncol,
                                float eps, const float* gamma, const float* beta, int pre_act, float* coef, float* stats, float* y, int64_t ldy,
                                hipStream_t stream) {
    SS_CHECK(x && moments && gamma && beta && coef && stats && y && rows_per_image > 0 && images > 0 && groups > 0 && cg > 0 && pitch >= cg &&
             ncol >= (groups - 1) * pitch + cg && ncol % 8 == 0 && ldx >= ncol && ldy >= ncol && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) &&
             aligned16(coef) && (pre_act == 0 || pre_act == 5), "ss_groupnorm_fwd: bad arguments (groups=%d cg=%d pitch=%d ncol=%d)", groups, cg, pitch, ncol);
    hipLaunchKernelGGL(gn_coef_kernel, dim3((unsigned)cdiv((int64_t)images * groups, 64)), dim3(64), 0, stream, moments, rows_per_image, images, groups, cg, pitch,
                       ncol, eps, gamma, beta, coef, stats);
    const int64_t bpi = tt_bpi(rows_per_image, ncol);
    SS_CHECK(bpi * images < (1ll << 31), "ss_groupnorm_fwd: grid too large");
    if (pre_act == 5) hipLaunchKernelGGL((image_scale_shift_kernel<true>), dim3((unsigned)(bpi * images)), dim3(256), 0, stream, x, ldx, (const float*)coef, y, ldy,
                                         rows_per_image, ncol, (int)bpi);
    else hipLaunchKernelGGL((image_scale_shift_kernel<false>), dim3((unsigned)(bpi * images)), dim3(256), 0, stream, x, ldx, (const float*)coef, y, ldy,
                            rows_per_image, ncol, (int)bpi);
    SS_LAUNCH_CHECK();
    return 0;
}
// dgamma_dbeta [2][groups*cg]; scratch: images * 5 * ncol floats (sums then P, Q, R); workspace >= images * 1024 * 2 * ncol floats
extern "C" int ss_groupnorm_bwd(const float* dy, int64_t ldd, const float* x, int64_t ldx, const float* stats, const float* gamma, int64_t rows_per_image,
                                int images, int groups, int cg, int pitch, int ncol, int pre_act, float* dx, int64_t ldo, float* dgamma_dbeta, float* scratch,
                                float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && stats && gamma && dx && dgamma_dbeta && scratch && workspace && rows_per_image > 0 && images > 0 && images <= 65535 && groups > 0 && cg > 0 &&
             pitch >= cg && ncol >= (groups - 1) * pitch + cg && ncol % 8 == 0 && ldd >= ncol && ldx >= ncol && ldo >= ncol && ldd % 8 == 0 && ldx % 8 == 0 &&
             ldo % 8 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(scratch) && (pre_act == 0 || pre_act == 5),
             "ss_groupnorm_bwd: bad arguments (groups=%d cg=%d pitch=%d ncol=%d)", groups, cg, pitch, ncol);
    ColGeom g = col_geom(rows_per_image, ncol);
    SS_CHECK(ws_bytes >= (int64_t)images * g.nb * 2 * ncol * 4, "ss_groupnorm_bwd: workspace too small");
    float* S = scratch; float* pqr = scratch + (int64_t)images * 2 * ncol;
    const int64_t bpi = tt_bpi(rows_per_image, ncol);
    SS_CHECK(bpi * images < (1ll << 31), "ss_groupnorm_bwd: grid too large");
#define SS_GNB(SIG) do { GnBwdSumF<SIG> f{dy, x, ldd, ldx}; \
        hipLaunchKernelGGL((colreduce_partial_kernel<GnBwdSumF<SIG>>), dim3(g.nb, images), dim3(256), 0, stream, f, rows_per_image, ncol, g.cpb, g.rpb, workspace); \
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(2 * ncol, 8), images), dim3(256), 0, stream, (const float*)workspace, g.nb, 2 * ncol, 1.f, S, 0); \
        hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(1), dim3(256), 0, stream, (const float*)S, stats, gamma, rows_per_image, images, groups, cg, pitch, ncol, pqr, \
                           dgamma_dbeta); \
        hipLaunchKernelGGL((gn_bwd_dx_kernel<SIG>), dim3((unsigned)(bpi * images)), dim3(256), 0, stream, dy, ldd, x, ldx, (const float*)pqr, dx, ldo, \
                           rows_per_image, ncol, (int)bpi); } while (0)
    if (pre_act == 5) SS_GNB(true); else SS_GNB(false);
#undef SS_GNB
    SS_LAUNCH_CHECK();
    return 0;
}

// ---- InjectionDotSum backward (see inject_bwd_kernel).  g, local, dlocal: dense [B*hl*wl][C]; gact, dgact, dgemb: [B*hg*wg][C] with row
// pitches lda / ldda / ldde.  hl = r hg, wl = r wg with r even; C % 8 == 0, C <= 2048.  workspace >= B (hg+1) (wg+1) 8 C floats.
extern "C" int ss_inject_bwd_ok(int hl, int wl, int hg, int wg, int C) {
    if (hg <= 0 || wg <= 0 || hl % hg || wl % wg || hl / hg != wl / wg) return 0;
    const int r = hl / hg;
    return r >= 2 && r % 2 == 0 && C % 8 == 0 && C >= 8 && C <= 2048;
}
extern "C" int ss_inject_bwd(int dtype, const void* g, const void* local, const void* gact, int64_t lda, void* dlocal, void* dgact, int64_t ldda,
                             void* dgemb, int64_t ldde, int B, int hl, int wl, int hg, int wg, int C, float* workspace, int64_t ws_bytes,
                             hipStream_t stream) {
    SS_CHECK(g && local && gact && dlocal && dgact && dgemb && workspace && B > 0 && ss_inject_bwd_ok(hl, wl, hg, wg, C),
             "ss_inject_bwd: need hl = r hg, wl = r wg with an even r >= 2 and C %% 8 == 0 (hl=%d wl=%d hg=%d wg=%d C=%d)", hl, wl, hg, wg, C);
    SS_CHECK(lda >= C && ldda >= C && ldde >= C && lda % 8 == 0 && ldda % 8 == 0 && ldde % 8 == 0 && aligned16(g) && aligned16(local) && aligned16(gact) &&
             aligned16(dlocal) && aligned16(dgact) && aligned16(dgemb), "ss_inject_bwd: rows must be 16-byte aligned");
    const int64_t cells = (int64_t)B * (hg + 1) * (wg + 1);
    SS_CHECK(cells < (1ll << 31) && ws_bytes >= cells * 8 * C * 4, "ss_inject_bwd: workspace too small");
    const int r = hl / hg, chunks = C / 8;
    int nl = 256 / chunks; if (nl < 1) nl = 1; if (nl > r * r) nl = r * r;
    SS_CHECK(chunks <= 256, "ss_inject_bwd: C too large");
    const int64_t items = (int64_t)B * hg * wg * chunks;
    SS_DISPATCH(dtype, {
        hipLaunchKernelGGL((inject_bwd_kernel<T>), dim3((unsigned)cells), dim3(256), 0, stream, (const T*)g, (const T*)local, (const T*)gact, lda, (T*)dlocal,
                           workspace, hl, wl, hg, wg, C, r, nl);
        hipLaunchKernelGGL((inject_bwd_final_kernel<T>), dim3(tt_blocks(items, 4096)), dim3(256), 0, stream, (const float*)workspace, (T*)dgact, ldda, (T*)dgemb,
                           ldde, B, hg, wg, C);
    });
    SS_LAUNCH_CHECK();
    return 0;
}
