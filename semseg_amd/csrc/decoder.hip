// core/decoder.py:49-59 + core/dnet.py:45-51 + modules/layers.py:25-43 on channels-last maps with a handful of channels.
//
// Decoder tail.  x = LawinHead logits [B,H4,W4,2oc].  ConvTranspose2d(2oc -> oc, k = 3, s = 4) has stride > kernel, so each
// source pixel stamps a DISJOINT 3x3 block of the (4H4-1) x (4W4-1) map and the remaining positions hold the bias only:
//   a_pre[b, 4i+dy, 4j+dx, o] = bias[o] + z9[(b,i,j)][o*9 + dy*3 + dx]   (dy, dx < 3),    bias[o] elsewhere,
// with z9 = x . Wd one [rows, 2oc] x [2oc, 9 oc] GEMM (done by ss_gemm_nt).  a = ReLU(BN(a_pre)); the BatchNorm statistics run
// over ALL positions: with d = a_pre - bias the gaps contribute zeros, so sum d / sum d^2 are column sums of z9.
// b = conv1x1(nearest-up(x)) = nearest-up(conv1x1(x)) =: nearest-up(bs); for in/out = H4/(4H4-1) torch's nearest index
// floor(y * in/out) equals y >> 2 (checked on the host: H4, W4 <= 1024).  prod = a * b.
// DNet tail: nearest resize (Hd,Wd) -> (H,W) with torch's float index rule, 1x1 conv oc -> nc, Softmax2d -> NCHW probabilities.
// All reductions are per-workgroup partials + a fixed-order final pass (deterministic).
#include "common.h"
#include "colreduce.h"

#define DEC_MAXC 32

template <typename T> struct SumSqF {
    static constexpr int K = 2;
    const T* x; int64_t ld;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        F8 v = load8(x + row * ld + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[0][e] += v.v[e]; acc[1][e] += v.v[e] * v.v[e]; }
    }
};

// stats[o] = mean, stats[oc + o] = rstd over the n positions of the deconvolved map; running buffers updated like nn.BatchNorm2d
__global__ void deconv_bn_finalize_kernel(const float* __restrict__ partial, int nb, int ncols, int oc, int64_t n, const float* __restrict__ dbias,
                                          float eps, float momentum, float* __restrict__ stats, float* running_mean, float* running_var,
                                          int64_t* nbt) {
    // 256 threads = 32 channels x 8 parts: part p adds partial rows p, p + 8, ...; the parts meet in LDS in a fixed order
    // (was one thread per channel walking all nb rows: 345 us of dependent loads)
    __shared__ double sh[2][8][33];
    const int o = threadIdx.x & 31, part = threadIdx.x >> 5;
    if (threadIdx.x == 0 && nbt) *nbt += 1;
    double s = 0.0, q = 0.0;
    if (o < oc)
        for (int b = part; b < nb; b += 8)
            for (int t = 0; t < 9; ++t) { s += partial[((int64_t)b * 2 + 0) * ncols + o * 9 + t]; q += partial[((int64_t)b * 2 + 1) * ncols + o * 9 + t]; }
    sh[0][part][o] = s; sh[1][part][o] = q;
    __syncthreads();
    if (part != 0 || o >= oc) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < 8; ++k) { s += sh[0][k][o]; q += sh[1][k][o]; }
    const double md = s / (double)n, mu = (double)dbias[o] + md;
    double var = q / (double)n - md * md;
    if (var < 0) var = 0;
    stats[o] = (float)mu; stats[oc + o] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_mean[o] = (1.f - momentum) * running_mean[o] + momentum * (float)mu;
        running_var[o] = (1.f - momentum) * running_var[o] + momentum * (float)unb;
    }
}

struct DecGeom { int B, H4, W4, Hd, Wd, oc, ld9, ldb; };

template <typename T>
__device__ __forceinline__ float dec_apre(const T* __restrict__ z9, const DecGeom& g, int64_t b, int y, int x, int o, float bias) {
    const int dy = y & 3, dx = x & 3;
    if (dy == 3 || dx == 3) return bias;
    const int64_t p = (b * g.H4 + (y >> 2)) * g.W4 + (x >> 2);
    return bias + to_f(z9[p * g.ld9 + o * 9 + dy * 3 + dx]);
}

template <typename T>
__global__ __launch_bounds__(256) void decoder_combine_fwd_kernel(const T* __restrict__ z9, const T* __restrict__ bs, const float* __restrict__ dbias,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ prod, DecGeom g) {
    const int64_t total = (int64_t)g.B * g.Hd * g.Wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.Wd); int64_t q = i / g.Wd; const int y = (int)(q % g.Hd); const int64_t b = q / g.Hd;
        const int64_t ps = (b * g.H4 + (y >> 2)) * g.W4 + (x >> 2);
        for (int o = 0; o < g.oc; ++o) {
            const float ap = dec_apre(z9, g, b, y, x, o, dbias[o]);
            const float a = fmaxf((ap - stats[o]) * stats[g.oc + o] * gamma[o] + beta[o], 0.f);
            prod[i * g.oc + o] = from_f<T>(a * to_f(bs[ps * g.ldb + o]));
        }
    }
}

// the same with the channel count known at compile time (DNet: oc = 12, or 6): the BatchNorm affine is folded into one scale and
// shift per channel held in registers, and a pixel's oc products leave as 8-byte stores instead of oc 2-byte ones
template <typename T, int OC>
__global__ __launch_bounds__(256) void decoder_combine_fwd_fixed_kernel(const T* __restrict__ z9, const T* __restrict__ bs, const float* __restrict__ dbias,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ prod, DecGeom g) {
    float db[OC], mean[OC], rstd[OC], gm[OC], bt[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) { db[o] = dbias[o]; mean[o] = stats[o]; rstd[o] = stats[OC + o]; gm[o] = gamma[o]; bt[o] = beta[o]; }
    const int64_t total = (int64_t)g.B * g.Hd * g.Wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.Wd); int64_t q = i / g.Wd; const int y = (int)(q % g.Hd); const int64_t b = q / g.Hd;
        const int64_t ps = (b * g.H4 + (y >> 2)) * g.W4 + (x >> 2);
        const int dy = y & 3, dx = x & 3;
        const bool tap = dy != 3 && dx != 3;                          // stride 4 > kernel 3: the gaps hold the bias only
        const T* zp = z9 + ps * g.ld9 + dy * 3 + dx;
        const T* bp = bs + ps * g.ldb;
        union { T v[OC]; unsigned long long w[(OC * sizeof(T)) / 8]; } out;
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            const float ap = tap ? db[o] + to_f(zp[o * 9]) : db[o];
            const float a = fmaxf((ap - mean[o]) * rstd[o] * gm[o] + bt[o], 0.f);
            out.v[o] = from_f<T>(a * to_f(bp[o]));
        }
        T* dst = prod + i * OC;
        if ((((uintptr_t)dst) & 7) == 0) {
#pragma unroll
            for (int k = 0; k < (int)((OC * sizeof(T)) / 8); ++k) ((unsigned long long*)dst)[k] = out.w[k];
        } else {
#pragma unroll
            for (int o = 0; o < OC; ++o) dst[o] = out.v[o];
        }
    }
}

// pass 1 of the backward: per channel sum g and sum g * xhat, g = dprod * b * [a > 0];  partial[blk][2*oc]
template <typename T>
__global__ __launch_bounds__(256) void decoder_combine_bwd_stats_kernel(const T* __restrict__ dprod, const T* __restrict__ z9, const T* __restrict__ bs,
        const float* __restrict__ dbias, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
        float* __restrict__ partial, DecGeom g) {
    __shared__ float sh[4][2 * DEC_MAXC];
    float sg[DEC_MAXC], sx[DEC_MAXC];
#pragma unroll
    for (int o = 0; o < DEC_MAXC; ++o) { sg[o] = 0.f; sx[o] = 0.f; }
    const int64_t total = (int64_t)g.B * g.Hd * g.Wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.Wd); int64_t q = i / g.Wd; const int y = (int)(q % g.Hd); const int64_t b = q / g.Hd;
        const int64_t ps = (b * g.H4 + (y >> 2)) * g.W4 + (x >> 2);
#pragma unroll
        for (int o = 0; o < DEC_MAXC; ++o)
            if (o < g.oc) {
                const float xh = (dec_apre(z9, g, b, y, x, o, dbias[o]) - stats[o]) * stats[g.oc + o];
                const float gg = (xh * gamma[o] + beta[o] > 0.f) ? to_f(dprod[i * g.oc + o]) * to_f(bs[ps * g.ldb + o]) : 0.f;
                sg[o] += gg; sx[o] += gg * xh;
            }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 0; o < DEC_MAXC; ++o)
        if (o < g.oc) { float a = wave_sum(sg[o]), c = wave_sum(sx[o]); if (lane == 0) { sh[wave][o] = a; sh[wave][g.oc + o] = c; } }
    __syncthreads();
    if (threadIdx.x < 2 * g.oc)
        partial[(int64_t)blockIdx.x * 2 * g.oc + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// red[o] = sum g, red[oc+o] = sum g*xhat  ->  dgb = (dgamma, dbeta), coef = (mean g, mean g*xhat) (0 in eval), dbias_grad
__global__ void decoder_bwd_finalize_kernel(const float* __restrict__ red, int oc, int64_t n, int training, const float* __restrict__ stats,
                                            const float* __restrict__ gamma, float* __restrict__ dgb, float* __restrict__ coef,
                                            float* __restrict__ dbias_grad) {
    const int o = threadIdx.x;
    if (o >= oc) return;
    const float s = red[o], q = red[oc + o];
    dgb[o] = q; dgb[oc + o] = s;
    coef[o] = training ? s / (float)n : 0.f;
    coef[oc + o] = training ? q / (float)n : 0.f;
    dbias_grad[o] = training ? 0.f : gamma[o] * stats[oc + o] * s;      // training-mode BN removes a constant shift
}

// pass 2: one thread per SOURCE pixel p = (b,i,j): dz9[p][o*9 + t] and dbs[p][o] from the 4x4 block of destination pixels it feeds
template <typename T>
__global__ __launch_bounds__(256) void decoder_combine_bwd_kernel(const T* __restrict__ dprod, const T* __restrict__ z9, const T* __restrict__ bs,
        const float* __restrict__ dbias, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
        const float* __restrict__ coef, T* __restrict__ dz9, T* __restrict__ dbs, DecGeom g) {
    const int64_t total = (int64_t)g.B * g.H4 * g.W4;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(p % g.W4); int64_t q = p / g.W4; const int i = (int)(q % g.H4); const int64_t b = q / g.H4;
        for (int o = 0; o < g.oc; ++o) {
            const float mu = stats[o], rs = stats[g.oc + o], ga = gamma[o], be = beta[o], bi = dbias[o], c1 = coef[o], c2 = coef[g.oc + o];
            const float bv = to_f(bs[p * g.ldb + o]);
            float db = 0.f;
            for (int dy = 0; dy < 4; ++dy) {
                const int y = 4 * i + dy;
                if (y >= g.Hd) continue;
                for (int dx = 0; dx < 4; ++dx) {
                    const int x = 4 * j + dx;
                    if (x >= g.Wd) continue;
                    const bool st = dy < 3 && dx < 3;
                    const float ap = st ? bi + to_f(z9[p * g.ld9 + o * 9 + dy * 3 + dx]) : bi;
                    const float xh = (ap - mu) * rs, u = xh * ga + be, a = fmaxf(u, 0.f);
                    const float dp = to_f(dprod[((b * g.Hd + y) * g.Wd + x) * g.oc + o]);
                    db += dp * a;
                    if (st) {
                        const float gg = u > 0.f ? dp * bv : 0.f;
                        dz9[p * g.ld9 + o * 9 + dy * 3 + dx] = from_f<T>(ga * rs * (gg - c1 - xh * c2));
                    }
                }
            }
            dbs[p * g.ldb + o] = from_f<T>(db);
        }
        for (int c = g.oc * 9; c < g.ld9; ++c) dz9[p * g.ld9 + c] = from_f<T>(0.f);        // padding columns of the GEMM operand
    }
}

// Pass 2 with the channel count known at compile time.  The generic kernel moves every element with its own 2-byte access (about
// 430 scattered accesses per source pixel: 5.6 ms per DNet step); here a thread loads its source pixel's whole z9 row as 16-byte
// vectors, each destination pixel's OC gradients as one contiguous vector, rewrites the row IN PLACE (every z9 entry is used by
// exactly one destination position) and stores it back as 16-byte vectors; the per-channel constants are loaded once.
template <int V> struct IntC { static constexpr int value = V; };
template <typename T, int OC> struct OcVec;
template <int OC> struct OcVec<bf16, OC> {             // OC % 4 == 0: 8-byte pieces
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[OC]) {
#pragma unroll
        for (int i = 0; i < OC / 4; ++i) { const bf16x4 t = *(const bf16x4*)(p + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)t[e]; }
    }
};
template <int OC> struct OcVec<float, OC> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[OC]) {
#pragma unroll
        for (int i = 0; i < OC / 4; ++i) { const floatx4 t = *(const floatx4*)(p + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * i + e] = t[e]; }
    }
};

template <typename T, int OC>
__global__ __launch_bounds__(256) void decoder_combine_bwd_fixed_kernel(const T* __restrict__ dprod, const T* __restrict__ z9, const T* __restrict__ bs,
        const float* __restrict__ dbias, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
        const float* __restrict__ coef, T* __restrict__ dz9, T* __restrict__ dbs, DecGeom g) {
    constexpr int N9 = OC * 9, NV = (N9 + 7) / 8;
    float mu[OC], rs[OC], ga[OC], be[OC], bi[OC], c1[OC], c2[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) { mu[o] = stats[o]; rs[o] = stats[OC + o]; ga[o] = gamma[o]; be[o] = beta[o]; bi[o] = dbias[o]; c1[o] = coef[o]; c2[o] = coef[OC + o]; }
    const int64_t total = (int64_t)g.B * g.H4 * g.W4;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(p % g.W4); const int64_t q = p / g.W4; const int i = (int)(q % g.H4); const int64_t b = q / g.H4;
        F8 zr[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) zr[v] = load8(z9 + p * g.ld9 + v * 8);
        float bv[OC], db[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) { bv[o] = to_f(bs[p * g.ldb + o]); db[o] = 0.f; }
        auto cell = [&](auto DYc, auto DXc) {
            constexpr int DY = decltype(DYc)::value, DX = decltype(DXc)::value;
            const int y = 4 * i + DY, x = 4 * j + DX;
            if (y < g.Hd && x < g.Wd) {
                float dp[OC];
                OcVec<T, OC>::load(dprod + ((b * g.Hd + y) * g.Wd + x) * OC, dp);
                constexpr bool st = DY < 3 && DX < 3;
#pragma unroll
                for (int o = 0; o < OC; ++o) {
                    const int idx = st ? o * 9 + DY * 3 + DX : 0;
                    const float ap = st ? bi[o] + zr[idx / 8].v[idx % 8] : bi[o];
                    const float xh = (ap - mu[o]) * rs[o], u = xh * ga[o] + be[o], a = fmaxf(u, 0.f);
                    db[o] = fmaf(dp[o], a, db[o]);
                    if (st) {
                        const float gg = u > 0.f ? dp[o] * bv[o] : 0.f;
                        zr[idx / 8].v[idx % 8] = ga[o] * rs[o] * (gg - c1[o] - xh * c2[o]);
                    }
                }
            }
        };
#define SS_ROW(DY) cell(IntC<DY>{}, IntC<0>{}); cell(IntC<DY>{}, IntC<1>{}); cell(IntC<DY>{}, IntC<2>{}); cell(IntC<DY>{}, IntC<3>{})
        SS_ROW(0); SS_ROW(1); SS_ROW(2); SS_ROW(3);
#undef SS_ROW
        if (N9 % 8) {
#pragma unroll
            for (int e = N9 % 8; e < 8; ++e) zr[NV - 1].v[e] = 0.f;                  // padding columns of the GEMM operand
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) store8(dz9 + p * g.ld9 + v * 8, zr[v]);
        for (int c = NV * 8; c < g.ld9; ++c) dz9[p * g.ld9 + c] = from_f<T>(0.f);
#pragma unroll
        for (int o = 0; o < OC; ++o) dbs[p * g.ldb + o] = from_f<T>(db[o]);
    }
}

// ------------------------------------------------------------------ DNet terminator
struct TermGeom { int B, Hd, Wd, H, W, oc, nc; float sy, sx; };
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) { int s = (int)floorf(dst * scale); return s < in_size - 1 ? s : in_size - 1; }

template <typename T, typename TP>
__global__ __launch_bounds__(256) void terminator_fwd_kernel(const T* __restrict__ prod, const float* __restrict__ wt, const float* __restrict__ bt,
        TP* __restrict__ probs, TermGeom g) {
    const int64_t total = (int64_t)g.B * g.H * g.W, hw = (int64_t)g.H * g.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % g.W); int64_t q = i / g.W; const int Y = (int)(q % g.H); const int64_t b = q / g.H;
        const int sy = nearest_src(Y, g.sy, g.Hd), sx = nearest_src(X, g.sx, g.Wd);
        const T* pr = prod + ((b * g.Hd + sy) * g.Wd + sx) * g.oc;
        float pv[DEC_MAXC], lg[DEC_MAXC];
#pragma unroll
        for (int o = 0; o < DEC_MAXC; ++o) pv[o] = o < g.oc ? to_f(pr[o]) : 0.f;
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < DEC_MAXC; ++c)
            if (c < g.nc) { float s = bt[c];
                for (int o = 0; o < g.oc; ++o) s += wt[c * g.oc + o] * pv[o];
                lg[c] = s; mx = fmaxf(mx, s); }
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) { lg[c] = __expf(lg[c] - mx); den += lg[c]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) probs[(b * g.nc + c) * hw + (int64_t)Y * g.W + X] = from_f<TP>(lg[c] * inv);
    }
}

// DNet's own channel counts known at compile time (nc = 6 classes, oc = 6 or 12): the generic kernel above walks DEC_MAXC
// predicated slots per pixel; here the product row is read as 4-byte pairs and everything unrolls to nc * oc fmas.
template <typename T, typename TP, int NC, int OC>
__global__ __launch_bounds__(256) void terminator_fwd_fixed_kernel(const T* __restrict__ prod, const float* __restrict__ wt, const float* __restrict__ bt,
        TP* __restrict__ probs, TermGeom g) {
    __shared__ float ws[NC * OC + NC];
    for (int t = threadIdx.x; t < NC * OC + NC; t += blockDim.x) ws[t] = t < NC * OC ? wt[t] : bt[t - NC * OC];
    __syncthreads();
    const int64_t total = (int64_t)g.B * g.H * g.W, hw = (int64_t)g.H * g.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % g.W); int64_t q = i / g.W; const int Y = (int)(q % g.H); const int64_t b = q / g.H;
        const int sy = nearest_src(Y, g.sy, g.Hd), sx = nearest_src(X, g.sx, g.Wd);
        const T* pr = prod + ((b * g.Hd + sy) * g.Wd + sx) * OC;
        float pv[OC], lg[NC];
#pragma unroll
        for (int o = 0; o < OC; ++o) pv[o] = to_f(pr[o]);
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float s = ws[NC * OC + c];
#pragma unroll
            for (int o = 0; o < OC; ++o) s += ws[c * OC + o] * pv[o];
            lg[c] = s; mx = fmaxf(mx, s);
        }
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { lg[c] = __expf(lg[c] - mx); den += lg[c]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int c = 0; c < NC; ++c) probs[(b * NC + c) * hw + (int64_t)Y * g.W + X] = from_f<TP>(lg[c] * inv);
    }
}

// one thread per SOURCE pixel of the (Hd,Wd) map: softmax backward is linear in dprobs, so the destination pixels that share the
// source are summed first.  dprod written; dW (nc*oc) and db (nc) as per-workgroup partials [blk][nc*oc + nc].
template <typename T, typename TP>
__global__ __launch_bounds__(256) void terminator_bwd_kernel(const TP* __restrict__ dprobs, const T* __restrict__ prod, const float* __restrict__ wt,
        const float* __restrict__ bt, T* __restrict__ dprod, float* __restrict__ partial, TermGeom g) {
    extern __shared__ float shp[];                                   // [4][nc*oc + nc]
    const int ncols = g.nc * g.oc + g.nc;
    for (int t = threadIdx.x; t < 4 * ncols; t += blockDim.x) shp[t] = 0.f;
    __syncthreads();
    const int64_t total = (int64_t)g.B * g.Hd * g.Wd, hw = (int64_t)g.H * g.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t padded = cdiv(total, 64) * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (int64_t)gridDim.x * blockDim.x) {
        const bool ok = i < total;
        float pv[DEC_MAXC], pr[DEC_MAXC], dl[DEC_MAXC];
#pragma unroll
        for (int c = 0; c < DEC_MAXC; ++c) { pv[c] = 0.f; pr[c] = 0.f; dl[c] = 0.f; }
        if (ok) {
            const int sx = (int)(i % g.Wd); int64_t q = i / g.Wd; const int sy = (int)(q % g.Hd); const int64_t b = q / g.Hd;
#pragma unroll
            for (int o = 0; o < DEC_MAXC; ++o) if (o < g.oc) pv[o] = to_f(prod[i * g.oc + o]);
            float mx = -INFINITY, den = 0.f;
#pragma unroll
            for (int c = 0; c < DEC_MAXC; ++c)
                if (c < g.nc) { float s = bt[c];
                    for (int o = 0; o < g.oc; ++o) s += wt[c * g.oc + o] * pv[o];
                    pr[c] = s; mx = fmaxf(mx, s); }
#pragma unroll
            for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) { pr[c] = __expf(pr[c] - mx); den += pr[c]; }
            // destination rows / columns whose nearest source is (sy, sx), by the same float rule as the forward
            int y0 = (int)((int64_t)sy * g.H / g.Hd) - 1, y1 = (int)(((int64_t)sy + 1) * g.H / g.Hd) + 1;
            int x0 = (int)((int64_t)sx * g.W / g.Wd) - 1, x1 = (int)(((int64_t)sx + 1) * g.W / g.Wd) + 1;
            if (y0 < 0) y0 = 0; if (x0 < 0) x0 = 0; if (y1 > g.H - 1) y1 = g.H - 1; if (x1 > g.W - 1) x1 = g.W - 1;
            for (int Y = y0; Y <= y1; ++Y) {
                if (nearest_src(Y, g.sy, g.Hd) != sy) continue;
                for (int X = x0; X <= x1; ++X) {
                    if (nearest_src(X, g.sx, g.Wd) != sx) continue;
#pragma unroll
                    for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) dl[c] += to_f(dprobs[(b * g.nc + c) * hw + (int64_t)Y * g.W + X]);
                }
            }
            float dot = 0.f; const float inv = 1.f / den;
#pragma unroll
            for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) { pr[c] *= inv; dot += pr[c] * dl[c]; }
#pragma unroll
            for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) dl[c] = pr[c] * (dl[c] - dot);           // d(logits)
            for (int o = 0; o < g.oc; ++o) { float s = 0.f;
#pragma unroll
                for (int c = 0; c < DEC_MAXC; ++c) if (c < g.nc) s += wt[c * g.oc + o] * dl[c];
                dprod[i * g.oc + o] = from_f<T>(s); }
        }
        // weight / bias gradient: wave reduce, accumulate per wave in LDS (each wave owns its row: no races, fixed order)
#pragma unroll
        for (int c = 0; c < DEC_MAXC; ++c)
            if (c < g.nc) {
                for (int o = 0; o < g.oc; ++o) { float v = wave_sum(dl[c] * pv[o]); if (lane == 0) shp[wave * ncols + c * g.oc + o] += v; }
                float v = wave_sum(dl[c]); if (lane == 0) shp[wave * ncols + g.nc * g.oc + c] += v;
            }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ncols; t += blockDim.x)
        partial[(int64_t)blockIdx.x * ncols + t] = shp[t] + shp[ncols + t] + shp[2 * ncols + t] + shp[3 * ncols + t];
}

// The same with the channel counts known at compile time (DNet: nc = oc): every thread keeps its share of dW / db in NC*OC + NC
// registers over the whole grid-stride loop and the wave reductions run once per workgroup instead of NC*OC + NC times per pixel
// (9.5 ms per DNet step at 1024^2 went into those shuffles).
template <typename T, typename TP, int NC, int OC>
__global__ __launch_bounds__(256) void terminator_bwd_fixed_kernel(const TP* __restrict__ dprobs, const T* __restrict__ prod, const float* __restrict__ wt,
        const float* __restrict__ bt, T* __restrict__ dprod, float* __restrict__ partial, TermGeom g) {
    constexpr int NCOLS = NC * OC + NC;
    __shared__ float shp[4][NCOLS];
    const int64_t total = (int64_t)g.B * g.Hd * g.Wd, hw = (int64_t)g.H * g.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float w[NC][OC], bias[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { bias[c] = bt[c];
#pragma unroll
        for (int o = 0; o < OC; ++o) w[c][o] = wt[c * OC + o]; }
    float aw[NC][OC], ab[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ab[c] = 0.f;
#pragma unroll
        for (int o = 0; o < OC; ++o) aw[c][o] = 0.f; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int sx = (int)(i % g.Wd); const int64_t q = i / g.Wd; const int sy = (int)(q % g.Hd); const int64_t b = q / g.Hd;
        float pv[OC], pr[NC], dl[NC];
#pragma unroll
        for (int o = 0; o < OC; ++o) pv[o] = to_f(prod[i * OC + o]);
        float mx = -INFINITY, den = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { float s = bias[c];
#pragma unroll
            for (int o = 0; o < OC; ++o) s = fmaf(w[c][o], pv[o], s);
            pr[c] = s; mx = fmaxf(mx, s); dl[c] = 0.f; }
#pragma unroll
        for (int c = 0; c < NC; ++c) { pr[c] = __expf(pr[c] - mx); den += pr[c]; }
        int y0 = (int)((int64_t)sy * g.H / g.Hd) - 1, y1 = (int)(((int64_t)sy + 1) * g.H / g.Hd) + 1;
        int x0 = (int)((int64_t)sx * g.W / g.Wd) - 1, x1 = (int)(((int64_t)sx + 1) * g.W / g.Wd) + 1;
        if (y0 < 0) y0 = 0; if (x0 < 0) x0 = 0; if (y1 > g.H - 1) y1 = g.H - 1; if (x1 > g.W - 1) x1 = g.W - 1;
        for (int Y = y0; Y <= y1; ++Y) {
            if (nearest_src(Y, g.sy, g.Hd) != sy) continue;
            for (int X = x0; X <= x1; ++X) {
                if (nearest_src(X, g.sx, g.Wd) != sx) continue;
#pragma unroll
                for (int c = 0; c < NC; ++c) dl[c] += to_f(dprobs[(b * NC + c) * hw + (int64_t)Y * g.W + X]);
            }
        }
        float dot = 0.f; const float inv = 1.f / den;
#pragma unroll
        for (int c = 0; c < NC; ++c) { pr[c] *= inv; dot = fmaf(pr[c], dl[c], dot); }
#pragma unroll
        for (int c = 0; c < NC; ++c) dl[c] = pr[c] * (dl[c] - dot);            // d(logits)
#pragma unroll
        for (int o = 0; o < OC; ++o) { float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) s = fmaf(w[c][o], dl[c], s);
            dprod[i * OC + o] = from_f<T>(s); }
#pragma unroll
        for (int c = 0; c < NC; ++c) { ab[c] += dl[c];
#pragma unroll
            for (int o = 0; o < OC; ++o) aw[c][o] = fmaf(dl[c], pv[o], aw[c][o]); }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int o = 0; o < OC; ++o) { const float v = wave_sum(aw[c][o]); if (lane == 0) shp[wave][c * OC + o] = v; }
        const float v = wave_sum(ab[c]); if (lane == 0) shp[wave][NC * OC + c] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NCOLS; t += blockDim.x)
        partial[(int64_t)blockIdx.x * NCOLS + t] = shp[0][t] + shp[1][t] + shp[2][t] + shp[3][t];
}

// ------------------------------------------------------------------ host entry points
static int dec_geom(DecGeom* g, int B, int H4, int W4, int oc, int64_t ld9, int64_t ldb, const char* who) {
    if (!(B > 0 && H4 > 0 && W4 > 0 && H4 <= 1024 && W4 <= 1024 && oc > 0 && oc <= DEC_MAXC && ld9 >= 9 * oc && ld9 % 8 == 0 && ldb >= oc))
        return ss_fail(-1, "%s: need 0 < H4,W4 <= 1024, 0 < oc <= %d, ld9 >= 9*oc and a multiple of 8 (B=%d H4=%d W4=%d oc=%d ld9=%lld)", who,
                       DEC_MAXC, B, H4, W4, oc, (long long)ld9);
    g->B = B; g->H4 = H4; g->W4 = W4; g->Hd = 4 * H4 - 1; g->Wd = 4 * W4 - 1; g->oc = oc; g->ld9 = (int)ld9; g->ldb = (int)ldb;
    return 0;
}
static inline int dec_blocks(int64_t n, int cap = 8192) { int64_t b = cdiv(n, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

extern "C" int ss_decoder_stats(int dtype, const void* z9, int64_t ld9, int B, int H4, int W4, int oc, const float* dbias, float eps,
                                float momentum, float* stats, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(z9 && dbias && stats && workspace && aligned16(z9), "ss_decoder_stats: bad pointers");
    DecGeom g; if (dec_geom(&g, B, H4, W4, oc, ld9, oc, "ss_decoder_stats")) return -1;
    const int64_t rows = (int64_t)B * H4 * W4;
    ColGeom cg = col_geom(rows, (int)ld9);
    SS_CHECK(ws_bytes >= (int64_t)cg.nb * 2 * ld9 * 4, "ss_decoder_stats: workspace too small");
    SS_DISPATCH(dtype, {
        SumSqF<T> f{(const T*)z9, ld9};
        hipLaunchKernelGGL((colreduce_partial_kernel<SumSqF<T>>), dim3(cg.nb, 1), dim3(256), 0, stream, f, rows, (int)ld9, cg.cpb, cg.rpb, workspace);
    });
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(deconv_bn_finalize_kernel, dim3(1), dim3(256), 0, stream, workspace, cg.nb, (int)ld9, oc, (int64_t)B * g.Hd * g.Wd, dbias, eps,
                       momentum, stats, running_mean, running_var, num_batches_tracked);
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_decoder_combine_fwd(int dtype, const void* z9, int64_t ld9, const void* bs, int64_t ldb, const float* dbias, const float* stats,
                                      const float* gamma, const float* beta, void* prod, int B, int H4, int W4, int oc, hipStream_t stream) {
    SS_CHECK(z9 && bs && dbias && stats && gamma && beta && prod, "ss_decoder_combine_fwd: null pointer");
    DecGeom g; if (dec_geom(&g, B, H4, W4, oc, ld9, ldb, "ss_decoder_combine_fwd")) return -1;
    dim3 grid(dec_blocks((int64_t)B * g.Hd * g.Wd, 32768));
    if (oc == 12)
        SS_DISPATCH(dtype, hipLaunchKernelGGL((decoder_combine_fwd_fixed_kernel<T, 12>), grid, dim3(256), 0, stream, (const T*)z9, (const T*)bs, dbias, stats,
                                              gamma, beta, (T*)prod, g));
    else if (oc == 8)
        SS_DISPATCH(dtype, hipLaunchKernelGGL((decoder_combine_fwd_fixed_kernel<T, 8>), grid, dim3(256), 0, stream, (const T*)z9, (const T*)bs, dbias, stats,
                                              gamma, beta, (T*)prod, g));
    else
        SS_DISPATCH(dtype, hipLaunchKernelGGL((decoder_combine_fwd_kernel<T>), grid, dim3(256), 0, stream, (const T*)z9, (const T*)bs, dbias, stats, gamma,
                                              beta, (T*)prod, g));
    SS_LAUNCH_CHECK();
    return 0;
}

// dgb: [2*oc] (dgamma, dbeta); dbias_grad: [oc]; workspace >= (1024*2*oc + 4*oc) floats.  dz9 has ld9 columns (padding columns zeroed).
extern "C" int ss_decoder_combine_bwd(int dtype, const void* dprod, const void* z9, int64_t ld9, const void* bs, int64_t ldb, const float* dbias,
                                      const float* stats, const float* gamma, const float* beta, int training, void* dz9, void* dbs, float* dgb,
                                      float* dbias_grad, int B, int H4, int W4, int oc, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dprod && z9 && bs && dbias && stats && gamma && beta && dz9 && dbs && dgb && dbias_grad && workspace, "ss_decoder_combine_bwd: null pointer");
    DecGeom g; if (dec_geom(&g, B, H4, W4, oc, ld9, ldb, "ss_decoder_combine_bwd")) return -1;
    const int64_t n = (int64_t)B * g.Hd * g.Wd;
    const int nb = dec_blocks(n, 1024);
    SS_CHECK(ws_bytes >= ((int64_t)nb * 2 * oc + 4 * oc) * 4, "ss_decoder_combine_bwd: workspace too small");
    float* red = workspace + (int64_t)nb * 2 * oc; float* coef = red + 2 * oc;
    SS_DISPATCH(dtype, {
        hipLaunchKernelGGL((decoder_combine_bwd_stats_kernel<T>), dim3(nb), dim3(256), 0, stream, (const T*)dprod, (const T*)z9, (const T*)bs, dbias,
                           stats, gamma, beta, workspace, g);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(2 * oc, 8), 1), dim3(256), 0, stream, workspace, nb, 2 * oc, 1.f, red, 0);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(decoder_bwd_finalize_kernel, dim3(1), dim3(64), 0, stream, red, oc, n, training, stats, gamma, dgb, coef, dbias_grad);
        SS_LAUNCH_CHECK();
        if (oc == 12 && ld9 % 8 == 0 && ld9 >= 112 && aligned16(z9) && aligned16(dz9) && aligned16(dprod))
            hipLaunchKernelGGL((decoder_combine_bwd_fixed_kernel<T, 12>), dim3(dec_blocks((int64_t)B * H4 * W4, 16384)), dim3(256), 0, stream,
                               (const T*)dprod, (const T*)z9, (const T*)bs, dbias, stats, gamma, beta, coef, (T*)dz9, (T*)dbs, g);
        else
            hipLaunchKernelGGL((decoder_combine_bwd_kernel<T>), dim3(dec_blocks((int64_t)B * H4 * W4, 16384)), dim3(256), 0, stream, (const T*)dprod,
                               (const T*)z9, (const T*)bs, dbias, stats, gamma, beta, coef, (T*)dz9, (T*)dbs, g);
    });
    SS_LAUNCH_CHECK();
    return 0;
}

static int term_geom(TermGeom* g, int B, int Hd, int Wd, int H, int W, int oc, int nc, const char* who) {
    if (!(B > 0 && Hd > 0 && Wd > 0 && H > 0 && W > 0 && oc > 0 && oc <= DEC_MAXC && nc > 0 && nc <= DEC_MAXC))
        return ss_fail(-1, "%s: bad geometry (oc, nc <= %d)", who, DEC_MAXC);
    g->B = B; g->Hd = Hd; g->Wd = Wd; g->H = H; g->W = W; g->oc = oc; g->nc = nc;
    g->sy = H == Hd ? 1.f : (float)Hd / (float)H; g->sx = W == Wd ? 1.f : (float)Wd / (float)W;
    return 0;
}

// probs_f32 != 0: the probabilities (and their gradient) are fp32 whatever the compute dtype -- what torch.autocast does with softmax
#define SS_TERM_DISPATCH(dtype, probs_f32, ...) \
    do { if ((dtype) == SS_F32) { typedef float T; typedef float TP; __VA_ARGS__; } \
         else if ((dtype) == SS_BF16 && (probs_f32)) { typedef bf16 T; typedef float TP; __VA_ARGS__; } \
         else if ((dtype) == SS_BF16) { typedef bf16 T; typedef bf16 TP; __VA_ARGS__; } \
         else return ss_fail(-1, "unsupported dtype code %d", (int)(dtype)); } while (0)

extern "C" int ss_terminator_fwd(int dtype, int probs_f32, const void* prod, const float* wt, const float* bt, void* probs, int B, int Hd, int Wd,
                                 int H, int W, int oc, int nc, hipStream_t stream) {
    SS_CHECK(prod && wt && bt && probs, "ss_terminator_fwd: null pointer");
    TermGeom g; if (term_geom(&g, B, Hd, Wd, H, W, oc, nc, "ss_terminator_fwd")) return -1;
    dim3 grid(dec_blocks((int64_t)B * H * W, 32768));
    if (nc == 6 && oc == 12)
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_fwd_fixed_kernel<T, TP, 6, 12>), grid, dim3(256), 0, stream, (const T*)prod, wt, bt, (TP*)probs, g));
    else if (nc == 6 && oc == 6)
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_fwd_fixed_kernel<T, TP, 6, 6>), grid, dim3(256), 0, stream, (const T*)prod, wt, bt, (TP*)probs, g));
    else
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_fwd_kernel<T, TP>), grid, dim3(256), 0, stream, (const T*)prod, wt, bt, (TP*)probs, g));
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db: [nc*oc + nc] fp32 (weight gradient row-major [nc][oc], then the bias gradient); workspace >= 1024 * (nc*oc + nc) floats
extern "C" int ss_terminator_bwd(int dtype, int probs_f32, const void* dprobs, const void* prod, const float* wt, const float* bt, void* dprod,
                                 float* dw_db, int B, int Hd, int Wd, int H, int W, int oc, int nc, float* workspace, int64_t ws_bytes,
                                 hipStream_t stream) {
    SS_CHECK(dprobs && prod && wt && bt && dprod && dw_db && workspace, "ss_terminator_bwd: null pointer");
    TermGeom g; if (term_geom(&g, B, Hd, Wd, H, W, oc, nc, "ss_terminator_bwd")) return -1;
    const int ncols = nc * oc + nc;
    const int nb = dec_blocks((int64_t)B * Hd * Wd, 1024);
    SS_CHECK(ws_bytes >= (int64_t)nb * ncols * 4, "ss_terminator_bwd: workspace too small");
    if (nc == 6 && oc == 12) {                            // DNet(n_classes = 6): the decoder hands 2 * n_classes channels to the classifier
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_bwd_fixed_kernel<T, TP, 6, 12>), dim3(nb), dim3(256), 0, stream, (const TP*)dprobs,
                                                               (const T*)prod, wt, bt, (T*)dprod, workspace, g));
    } else if (nc == 6 && oc == 6) {
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_bwd_fixed_kernel<T, TP, 6, 6>), dim3(nb), dim3(256), 0, stream, (const TP*)dprobs,
                                                               (const T*)prod, wt, bt, (T*)dprod, workspace, g));
    } else {
        SS_TERM_DISPATCH(dtype, probs_f32, hipLaunchKernelGGL((terminator_bwd_kernel<T, TP>), dim3(nb), dim3(256), 4 * ncols * sizeof(float), stream,
                                                               (const TP*)dprobs, (const T*)prod, wt, bt, (T*)dprod, workspace, g));
    }
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(ncols, 8), 1), dim3(256), 0, stream, workspace, nb, ncols, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}
