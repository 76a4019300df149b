// LayerNorm (token layout) and BatchNorm2d (channels-last) forward/backward, bias-gradient
// column sums and per-image average pooling (K14).  All bandwidth-bound: 16-byte channel-contiguous accesses,
// fp32 statistics, deterministic reductions (colreduce.h).
#include "common.h"
#include "colreduce.h"
#include "bilinear.h"
#include "gemm_fast.h"

// ------------------------------------------------------------------ LayerNorm
// One row is handled by `lpr` lanes (power of two), NCH chunks of 8 channels per lane.
// 8 consecutive elements in storage dtype (converted where used: four rows in flight cost 4 registers each, not 8)
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *(const bf16x8*)p; }
    __device__ __forceinline__ float get(int e) const { return (float)v[e]; }
};
template <> struct Raw8<float> {
    floatx4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const floatx4*)p; b = *(const floatx4*)(p + 4); }
    __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e & 3] : b[e & 3]; }
};                       // chunk lanes x row lanes of a 256-thread workgroup (col_geom)

// A BatchNorm input that is the bilinear resize (align_corners = False) of a coarser channels-last map, read through its four
// taps instead of from a resized copy: row r of the virtual [B*ho*wo, C] matrix is output pixel r, its value the very bits
// ss_bilinear_fwd would have stored (same taps, same fma order, rounded to T).  The resized tensor -- 4x the map for the x2
// up-sampling in front of LawinHead's `cat` BatchNorm -- is then never written nor read; the taps come from L1 / L2.
struct UpGeom { int hi, wi, ho, wo; float sh, sw; FastDiv dwo, dho; };
template <typename T> struct UpTaps { Raw8<T> a, b, c, d; float y0, y1, x0, x1; };
template <typename T>
__device__ __forceinline__ void up_issue(const T* __restrict__ x, int64_t ld, const UpGeom& g, int64_t r, int c, UpTaps<T>& t) {
    const unsigned q = fdiv((unsigned)r, g.dwo), ox = (unsigned)r - q * (unsigned)g.wo;
    const unsigned b = fdiv(q, g.dho), oy = q - b * (unsigned)g.ho;
    const LinTap ty = lin_tap((int)oy, g.sh, g.hi), tx = lin_tap((int)ox, g.sw, g.wi);
    const T* r0 = x + (((int64_t)b * g.hi + ty.i0) * g.wi) * ld + c;
    const T* r1 = x + (((int64_t)b * g.hi + ty.i1) * g.wi) * ld + c;
    t.a.load(r0 + tx.i0 * ld); t.b.load(r0 + tx.i1 * ld); t.c.load(r1 + tx.i0 * ld); t.d.load(r1 + tx.i1 * ld);
    t.y0 = ty.l0; t.y1 = ty.l1; t.x0 = tx.l0; t.x1 = tx.l1;
}
template <typename T>
__device__ __forceinline__ float up_value(const UpTaps<T>& t, int e) {
    LinTap ty, tx; ty.l0 = t.y0; ty.l1 = t.y1; tx.l0 = t.x0; tx.l1 = t.x1;
    return to_f(from_f<T>(bilerp(ty, tx, t.a.get(e), t.b.get(e), t.c.get(e), t.d.get(e))));
}
static inline UpGeom up_geom(int hi, int wi, int ho, int wo) {
    UpGeom g; g.hi = hi; g.wi = wi; g.ho = ho; g.wo = wo; g.sh = (float)hi / ho; g.sw = (float)wi / wo;
    g.dwo = fast_div((unsigned)wo); g.dho = fast_div((unsigned)ho);
    return g;
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
        int64_t rows, int C, int lpr, float eps) {
    const int lane = threadIdx.x & 63, lig = lane & (lpr - 1), rpw = 64 / lpr;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    F8 gm[NCH], bt[NCH];                                   // this lane's channels never change: load the affine once
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (lig + j * lpr) * 8;
        gm[j] = c < C ? load8(gamma + c) : zero8(); bt[j] = c < C ? load8(beta + c) : zero8();
    }
    const float invC = 1.f / C;
    for (int64_t r0 = wave * rpw; r0 < rows; r0 += 2 * nwaves * rpw) {      // two rows per lane group in flight
        int64_t r[2]; bool rok[2];
        Raw8<T> raw[2][NCH];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            r[u] = r0 + u * nwaves * rpw + lane / lpr;
            rok[u] = r[u] < rows;
            const int64_t rr = rok[u] ? r[u] : 0;             // clamped: the loads stay unconditional
#pragma unroll
            for (int j = 0; j < NCH; ++j) { const int c = (lig + j * lpr) * 8; raw[u][j].load(x + rr * C + (c < C ? c : 0)); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            F8 v[NCH]; float s = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const float live = (lig + j * lpr) * 8 < C ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[j].v[e] = raw[u][j].get(e) * live; s += v[j].v[e]; }
            }
            const float mu = group_sum(s, lpr) * invC;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const float live = (lig + j * lpr) * 8 < C ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (v[j].v[e] - mu) * live; v[j].v[e] = d; q = fmaf(d, d, q); }
            }
            const float rs = rsqrtf(group_sum(q, lpr) * invC + eps);
            if (rok[u]) {
                if (lig == 0) { mean[r[u]] = mu; rstd[r[u]] = rs; }
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int c = (lig + j * lpr) * 8;
                    if (c < C) {
                        F8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] = fmaf(v[j].v[e] * rs, gm[j].v[e], bt[j].v[e]);
                        store8(y + r[u] * C + c, o);
                    }
                }
            }
        }
    }
}

// dx = res + rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat));  per-block partials of dgamma = sum dy*xhat, dbeta = sum dy.
// `res` (optional) is the gradient arriving over the residual connection around the normalised branch: adding it here saves
// the separate elementwise pass.  Two rows per lane group are in flight per step; gamma is loaded once per lane.
// `dy2` (optional) is a second gradient of the same output (the output feeds two consumers: a stage's feature map goes to the head
// AND to the next stage): summed here, in fp32, instead of by a separate elementwise pass.
template <typename T, int NCH, bool RES, bool DY2>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ x,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma, const T* __restrict__ res,
        T* __restrict__ dx, float* __restrict__ partial, int64_t rows, int C, int lpr) {
    extern __shared__ float sh[];                 // [groups][2][C]
    const int lane = threadIdx.x & 63, lig = lane & (lpr - 1), rpw = 64 / lpr;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float dg[NCH][8], db[NCH][8];
    F8 gm[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (lig + j * lpr) * 8;
        gm[j] = c < C ? load8(gamma + c) : zero8();
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; }
    }
    for (int64_t r0 = wave * rpw; r0 < rows; r0 += 2 * nwaves * rpw) {
        int64_t r[2]; bool rok[2]; float mu[2], rs[2];
        Raw8<T> xv[2][NCH], dv[2][NCH], qv[2][NCH], ev[2][NCH];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            r[u] = r0 + u * nwaves * rpw + lane / lpr;
            rok[u] = r[u] < rows;
            const int64_t rr = rok[u] ? r[u] : 0;                 // clamped: the loads stay unconditional
            mu[u] = mean[rr]; rs[u] = rstd[rr];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = (lig + j * lpr) * 8, cc = c < C ? c : 0;
                xv[u][j].load(x + rr * C + cc); dv[u][j].load(dy + rr * C + cc);
                if (RES) qv[u][j].load(res + rr * C + cc);
                if (DY2) ev[u][j].load(dy2 + rr * C + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            F8 xh[NCH], gy[NCH]; float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = (lig + j * lpr) * 8;
                const float live = (rok[u] && c < C) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float h = (xv[u][j].get(e) - mu[u]) * rs[u], d = (DY2 ? dv[u][j].get(e) + ev[u][j].get(e) : dv[u][j].get(e)) * live;
                    dg[j][e] = fmaf(d, h, dg[j][e]); db[j][e] += d;
                    const float t = d * gm[j].v[e];
                    xh[j].v[e] = h; gy[j].v[e] = t; s1 += t; s2 = fmaf(t, h, s2);
                }
            }
            s1 = group_sum(s1, lpr) / C; s2 = group_sum(s2, lpr) / C;
            if (rok[u]) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int c = (lig + j * lpr) * 8;
                    if (c < C) { F8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            o.v[e] = rs[u] * (gy[j].v[e] - s1 - xh[j].v[e] * s2);
                            if (RES) o.v[e] += qv[u][j].get(e);
                        }
                        store8(dx + r[u] * C + c, o); }
                }
            }
        }
    }
    // fixed-order block reduction of the per-lane column partials
    const int groups = (blockDim.x >> 6) * rpw, grp = (threadIdx.x >> 6) * rpw + lane / lpr;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        int c = (lig + j * lpr) * 8;
        if (c < C)
#pragma unroll
            for (int e = 0; e < 8; ++e) { sh[(grp * 2 + 0) * C + c + e] = dg[j][e]; sh[(grp * 2 + 1) * C + c + e] = db[j][e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += sh[g * 2 * C + i];
        partial[(int64_t)blockIdx.x * 2 * C + i] = s;
    }
}

static int ln_geom(int C, int* lpr, int* nch) {
    int chunks = C / 8, l = 1;
    while (l < chunks && l < 64) l <<= 1;
    *lpr = l; *nch = (int)cdiv(chunks, l);
    return (C % 8 == 0 && *nch <= 2) ? 0 : -1;
}

extern "C" int ss_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int64_t rows, int C, float eps, hipStream_t stream) {
    int lpr, nch;
    SS_CHECK(x && gamma && beta && y && mean && rstd && rows > 0, "ss_layernorm_fwd: bad arguments");
    SS_CHECK(ln_geom(C, &lpr, &nch) == 0, "ss_layernorm_fwd: C=%d must be a multiple of 8 and <= 1024", C);
    int64_t waves = cdiv(rows, 64 / lpr);
    int blocks = (int)(cdiv(waves, 4) < 4096 ? cdiv(waves, 4) : 4096);
    SS_DISPATCH(dtype, {
        if (nch == 1) hipLaunchKernelGGL((layernorm_fwd_kernel<T, 1>), dim3(blocks), dim3(256), 0, stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, lpr, eps);
        else hipLaunchKernelGGL((layernorm_fwd_kernel<T, 2>), dim3(blocks), dim3(256), 0, stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, lpr, eps);
    });
    SS_LAUNCH_CHECK();
    return 0;
}

// workspace: SS_LN_BWD_BLOCKS * 2 * C floats
#define SS_LN_BWD_BLOCKS 1024
extern "C" int ss_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const void* res, void* dx, float* dgamma, float* dbeta, int64_t rows, int C, float* workspace,
                                int64_t ws_bytes, hipStream_t stream) {
    int lpr, nch;
    SS_CHECK(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace && rows > 0, "ss_layernorm_bwd: bad arguments");
    SS_CHECK(ln_geom(C, &lpr, &nch) == 0, "ss_layernorm_bwd: C=%d must be a multiple of 8 and <= 1024", C);
    SS_CHECK(dbeta == dgamma + C, "ss_layernorm_bwd: dgamma/dbeta must be one [2,C] buffer");
    int64_t waves = cdiv(rows, 64 / lpr);
    int blocks = (int)(cdiv(waves, 4) < SS_LN_BWD_BLOCKS ? cdiv(waves, 4) : SS_LN_BWD_BLOCKS);
    SS_CHECK(ws_bytes >= (int64_t)blocks * 2 * C * 4, "ss_layernorm_bwd: workspace too small");
    size_t shm = (size_t)4 * (64 / lpr) * 2 * C * sizeof(float);
#define SS_LN_BWD(NCH, RES) do { if (dy2) SS_LN_BWD_(NCH, RES, true); else SS_LN_BWD_(NCH, RES, false); } while (0)
#define SS_LN_BWD_(NCH, RES, DY2) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH, RES, DY2>), dim3(blocks), dim3(256), shm, stream, (const T*)dy, (const T*)dy2, (const T*)x, \
        mean, rstd, gamma, (const T*)res, (T*)dx, workspace, rows, C, lpr)
    SS_DISPATCH(dtype, {
        if (nch == 1) { if (res) SS_LN_BWD(1, true); else SS_LN_BWD(1, false); }
        else { if (res) SS_LN_BWD(2, true); else SS_LN_BWD(2, false); }
    });
#undef SS_LN_BWD
#undef SS_LN_BWD_
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(2 * C, 8), 1), dim3(256), 0, stream, workspace, blocks, 2 * C, 1.f, dgamma, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ column sums (bias gradients, per-image pooling)
template <typename T> struct ColSumF {
    static constexpr int K = 1;
    const T* x; int64_t ld;
    __device__ void operator()(int64_t row, int c0, float (&acc)[1][8]) const {
        F8 v = load8(x + row * ld + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[0][e] += v.v[e];
    }
};

// out[g][c] (+)= scale * sum_{r < rows} x[g*rows + r][c]      (groups = images for average pooling, 1 for bias gradients)
extern "C" int ss_colsum(int dtype, const void* x, int64_t ld, int64_t rows, int C, int groups, float scale, float* out,
                         int accumulate, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(x && out && workspace && rows > 0 && groups > 0 && C > 0 && C % 8 == 0 && ld >= C && ld % 8 == 0, "ss_colsum: bad arguments (C=%d)", C);
    ColGeom g = col_geom(rows, C);
    SS_CHECK(ws_bytes >= (int64_t)groups * g.nb * C * 4, "ss_colsum: workspace too small");
    SS_DISPATCH(dtype, {
        ColSumF<T> f{(const T*)x, ld};
        hipLaunchKernelGGL((colreduce_partial_kernel<ColSumF<T>>), dim3(g.nb, groups), dim3(256), 0, stream, f, rows, C, g.cpb, g.rpb, workspace);
    });
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(C, 8), groups), dim3(256), 0, stream, workspace, g.nb, C, scale, out, accumulate);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ BatchNorm2d on [rows, C]
// Shifted sums: d = x - x[row 0] per channel, so that var = E[d^2] - E[d]^2 does not cancel when |mean| >> std
// (sum x^2 - n mean^2 in fp32 loses the variance of a channel whose values nearly agree, e.g. the (B,C,1,1) image-pool BN).
template <typename T> struct BnStatF {
    static constexpr int K = 2;
    const T* x; int64_t ld;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        F8 v = load8(x + row * ld + c0), s = load8(x + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = v.v[e] - s.v[e]; acc[0][e] += d; acc[1][e] += d * d; }
    }
};

// the same sums over the resized map (row 0 of the virtual matrix is pixel (0, 0, 0) = x[0] exactly: its taps weigh 1 and 0)
template <typename T> struct BnStatUpF {
    static constexpr int K = 2;
    const T* x; int64_t ld; UpGeom g;
    __device__ void operator()(int64_t row, int c0, float (&acc)[2][8]) const {
        UpTaps<T> t; up_issue<T>(x, ld, g, row, c0, t);
        F8 s = load8(x + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = up_value<T>(t, e) - s.v[e]; acc[0][e] += d; acc[1][e] += d * d; }
    }
};

// stats[0][c] = mean, stats[1][c] = rstd (biased variance); running stats updated like nn.BatchNorm2d (unbiased var)
// CX channels x 256 / CX row-parts per block (double accumulation, fixed order).  CX = 8 normally; CX = 1 where few channels meet many
// partial rows (the GEMM leaves one row per 128-row tile: 32768 rows for the 16-channel 512 x 512 layers of TopFormer, which two
// workgroups of the CX = 8 form took 147 us to add up)
template <int CX>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nb, int C, int64_t n, float eps, float momentum,
                                   float* __restrict__ stats, float* running_mean, float* running_var, int64_t* nbt,
                                   const void* __restrict__ x, int x_is_bf16) {
    constexpr int PARTS = 256 / CX;
    __shared__ double sh[2][PARTS][CX + 1];
    const int cx = threadIdx.x % CX, part = threadIdx.x / CX;
    const int c = blockIdx.x * CX + cx;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double s = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = part; b < nb; b += PARTS) { s += partial[((int64_t)b * 2 + 0) * C + c]; q += partial[((int64_t)b * 2 + 1) * C + c]; }
    }
    sh[0][part][cx] = s; sh[1][part][cx] = q;
    __syncthreads();
    if (part != 0 || c >= C) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < PARTS; ++k) { s += sh[0][k][cx]; q += sh[1][k][cx]; }
    const double shift = !x ? 0.0 : x_is_bf16 ? (double)(float)((const bf16*)x)[c] : (double)((const float*)x)[c];
    double md = s / (double)n, var = q / (double)n - md * md, mu = shift + md;
    if (var < 0) var = 0;
    stats[c] = (float)mu;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// (channels, partial rows, stream, kernel arguments ...)
#define SS_BN_FINALIZE(CH, NB, STREAM, ...) do { \
        if (cdiv((CH), 8) < 32 && (NB) > 1024) hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((unsigned)(CH)), dim3(256), 0, STREAM, __VA_ARGS__); \
        else hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3((unsigned)cdiv((CH), 8)), dim3(256), 0, STREAM, __VA_ARGS__); \
    } while (0)

extern "C" int ss_bn_stats(int dtype, const void* x, int64_t ld, int64_t rows, int C, float eps, float momentum, float* stats,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                           int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(x && stats && workspace && rows > 0 && C > 0 && C % 8 == 0 && ld >= C && ld % 8 == 0, "ss_bn_stats: bad arguments (C=%d)", C);
    ColGeom g = col_geom(rows, C);
    SS_CHECK(ws_bytes >= (int64_t)g.nb * 2 * C * 4, "ss_bn_stats: workspace too small");
    SS_DISPATCH(dtype, {
        BnStatF<T> f{(const T*)x, ld};
        hipLaunchKernelGGL((colreduce_partial_kernel<BnStatF<T>>), dim3(g.nb, 1), dim3(256), 0, stream, f, rows, C, g.cpb, g.rpb, workspace);
    });
    SS_LAUNCH_CHECK();
    SS_BN_FINALIZE(C, g.nb, stream, workspace, g.nb, C, rows, eps, momentum,
                       stats, running_mean, running_var, num_batches_tracked, x, dtype == SS_BF16 ? 1 : 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ the x2 case (ho = 2 hi, wo = 2 wi) as a column walker
// The generic kernels above pay four 16-byte taps, two index divisions and a full bilerp per output chunk.  For the exact x2
// up-sampling of LawinHead's `cat` a thread instead owns ONE input column j and 8 channels and walks down input rows: output
// columns 2j, 2j+1 take taps (j-1, j) and (j, j+1), output rows 2p, 2p+1 take input rows (p-1, p) and (p, p+1), all clamped at
// the borders exactly as lin_tap clamps them (a clamped duplicate meets weight 0, or both weights of the same value -- the same
// bits as the generic formula).  The horizontal lerps of an input row are computed once and serve the four output rows that use
// it; three lerped rows (p-1, p, p+1) rotate through registers and rows p+2 .. p+4 are in flight: one new input row (4 loads, 2
// of them the same line) and 2 x 8 x 2 horizontal + 4 x 8 x 2 vertical flops per FOUR output chunks.
// UN channels per thread: with 8 the window rows, the rows in flight and the per-channel constants take 250-350 registers (one
// wave per SIMD), which only the store-heavy apply kernel tolerates (16-byte stores: 350 us against 460 with 8-byte ones); the
// statistics and backward kernels are bound by their vector arithmetic (Mish') and run with 4 (140-200 registers).
#define UN 8
namespace w8 {
#include "bn_up2_inc.h"
}
#undef UN
#define UN 4
namespace w4 {
#include "bn_up2_inc.h"
}
#undef UN
static inline bool up2_ok(int hi, int wi, int ho, int wo) { return ho == 2 * hi && wo == 2 * wi; }

// partial rows the first pass of a resize-BatchNorm kernel writes: the x2 walker's block count can exceed the column geometry's on
// thin maps (C = 64, 64 x 2 -> 128 x 4: 11 against 4), so the workspace check and the offset of `coef` use the larger of the two
static inline int up_partial_rows(int nb_cols, int B, int hi, int wi, int ho, int wo, int C) {
    if (!up2_ok(hi, wi, ho, wo)) return nb_cols;
    const int nb2 = w4::up2_blocks(w4::up2_geom(B, hi, wi, C), B);
    return nb2 > nb_cols ? nb2 : nb_cols;
}

#define SS_UP_CHECK(name) \
    SS_CHECK(B > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && (int64_t)B * ho * wo < (1LL << 31) && (int64_t)B * hi * wi * ldx < (1LL << 40), \
             name ": bad resize geometry")
extern "C" int ss_bn_stats_up(int dtype, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo, int C, float eps, float momentum,
                              float* stats, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                              int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(x && stats && workspace && C > 0 && C % 8 == 0 && ldx >= C && ldx % 8 == 0 && aligned16(x), "ss_bn_stats_up: bad arguments (C=%d)", C);
    SS_UP_CHECK("ss_bn_stats_up");
    SS_CHECK(!bvec || up2_ok(hi, wi, ho, wo), "ss_bn_stats_up: a per-image constant is only taken by the x2 resize");
    const int64_t rows = (int64_t)B * ho * wo;
    ColGeom g = col_geom(rows, C);
    SS_CHECK(ws_bytes >= (int64_t)up_partial_rows(g.nb, B, hi, wi, ho, wo, C) * 2 * C * 4, "ss_bn_stats_up: workspace too small");
    int nb = g.nb;
    if (up2_ok(hi, wi, ho, wo)) {
        using namespace w4;
        const Up2Geom u = up2_geom(B, hi, wi, C);
        nb = up2_blocks(u, B);
        SS_DISPATCH(dtype, hipLaunchKernelGGL((bn_stats_up2_kernel<T>), dim3(nb, u.groups), dim3(256), 0, stream, (const T*)x, ldx, bvec, B, hi, wi, C, u, workspace));
    } else SS_DISPATCH(dtype, {
        BnStatUpF<T> f{(const T*)x, ldx, up_geom(hi, wi, ho, wo)};
        hipLaunchKernelGGL((colreduce_partial_kernel<BnStatUpF<T>>), dim3(g.nb, 1), dim3(256), 0, stream, f, rows, C, g.cpb, g.rpb, workspace);
    });
    SS_LAUNCH_CHECK();
    SS_BN_FINALIZE(C, nb, stream, workspace, nb, C, rows, eps, momentum,
                       stats, running_mean, running_var, num_batches_tracked, x, dtype == SS_BF16 ? 1 : 0);
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_gemm_nt(int dtype, int out_f32, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldr,
                          const float* rowscale, int64_t rows_per_scale, int act, hipStream_t stream);
// A bias-free 1x1 conv and the batch statistics of its output in one pass (ConvModule / conv_out of modules/lawin.py:49-54,97-105:
// conv -> BatchNorm2d): C = A . B^T (+ res) as ss_gemm_nt, stats as ss_bn_stats(C).  On the bf16 tile kernels the column sums of the stored
// (rounded) tile ride in the GEMM's copy-out -- one partial row per row tile, same fixed-order double-precision finalize -- and the
// separate pass over C disappears (plain sums per 128 / 256 rows instead of sums shifted by row 0: the finalize subtracts in double).
// Other shapes / fp32: the two calls.  workspace >= max(cdiv(M, 128), 1024) * 2 * N floats.
extern "C" int ss_gemm_nt_bn_stats(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                                   int64_t K, const void* res, int64_t ldr, float eps, float momentum, float* stats, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(A && B && C && stats && workspace && M > 0 && N > 0 && K > 0 && N < (1 << 30) && lda >= K && ldb >= K && ldc >= N,
             "ss_gemm_nt_bn_stats: bad arguments");
    SS_CHECK(!res || ldr >= N, "ss_gemm_nt_bn_stats: bad ldr");
    if (dtype == SS_BF16 && N % 8 == 0 && cdiv(M, 128) * cdiv(N, 64) < 2147483647LL) {
        FastGemm f{};
        f.A = (const bf16*)A; f.B = (const bf16*)B; f.C = C; f.M = M; f.N = (int)N; f.K = K; f.lda = lda; f.ldb = ldb; f.ldc = ldc;
        f.rows_per_scale = 1; f.k_per_split = K; f.res = (const bf16*)res; f.ldr = ldr;
        // both sides narrow (K, N <= 32: TopFormer's 16-channel layers at 512 x 512): the thread-per-row product and a statistics pass over
        // its 2 N bytes per row take 130 us where 32768 tiles of the fused kernel take 280
        const bool narrow = !res && ss_rows_nt_ok(dtype, A, lda, C, ldc, M, N, K);
        if (!narrow && !ss_skinny_nt_ok(f) && ss_fast_nt_ok(A, lda, B, ldb, C, ldc, M, N, K, res, ldr, nullptr)) {
            f.stat_ws = workspace;
            const int64_t nb = ss_fast_nt_stat_tiles(f);
            SS_CHECK(ws_bytes >= nb * 2 * N * 4, "ss_gemm_nt_bn_stats: workspace too small");
            int rc = ss_fast_nt_launch(f, stream);
            if (rc) return rc;
            SS_BN_FINALIZE(N, (int)nb, stream, workspace, (int)nb, (int)N, M, eps, momentum,
                               stats, running_mean, running_var, num_batches_tracked, (const void*)nullptr, 0);
            SS_LAUNCH_CHECK();
            return 0;
        }
    }
    int rc = ss_gemm_nt(dtype, 0, A, lda, B, ldb, C, ldc, M, N, K, nullptr, res, ldr, nullptr, 0, 0, stream);
    if (rc) return rc;
    return ss_bn_stats(dtype, C, ldc, M, (int)N, eps, momentum, stats, running_mean, running_var, num_batches_tracked, workspace, ws_bytes, stream);
}

// eval mode: stats from running buffers
__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps, float* stats) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { stats[c] = rm[c]; stats[C + c] = rsqrtf(rv[c] + eps); }
}
extern "C" int ss_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* stats, hipStream_t stream) {
    SS_CHECK(running_mean && running_var && stats && C > 0, "ss_bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var, C, eps, stats);
    SS_LAUNCH_CHECK();
    return 0;
}

// Row walkers of the BatchNorm kernels: a thread keeps ONE 8-channel chunk (so mean / rstd / gamma / beta / coefficients are
// loaded once, folded into scale + shift) and walks rows with a fixed stride, 4 rows of loads in flight per step.
struct BnWalk { int cpb, rpb; };
__device__ __forceinline__ int64_t bn_image_of(int64_t row, int64_t rows_per_image, int rpi_shift) {
    return rpi_shift >= 0 ? row >> rpi_shift : row / rows_per_image;
}
static inline int pow2_shift(int64_t v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctzll((unsigned long long)v) : -1; }

// y[r] = res[r] + mask[b,c] * act(gamma * (z - mean) * rstd + beta);   act: 0 none, 1 Mish, 2 ReLU, 3 ReLU6 (the last two: ss_bn_apply / ss_bn_bwd only)
// UP: z is the coarse map and `up` its resize geometry (rows = B*ho*wo virtual rows, see UpGeom)
template <typename T, int ACT, bool MASK, bool RES, bool UP = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, int64_t ldz, const float* __restrict__ stats,
        const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res, int64_t ldr,
        const float* __restrict__ mask, int64_t rows_per_image, int rpi_shift, T* __restrict__ y, int64_t ldy, int64_t rows, int C, BnWalk w,
        UpGeom up) {
    const int chunks = C / 8, t = threadIdx.x, ch = t % w.cpb, rl = t / w.cpb;
    if (rl >= w.rpb) return;
    for (int cbase = 0; cbase < chunks; cbase += w.cpb) {
        const int chunk = cbase + ch;
        if (chunk >= chunks) continue;
        const int c = chunk * 8;
        F8 sc, sh;
        { F8 mu = load8(stats + c), rs = load8(stats + C + c), g = load8(gamma + c), b = load8(beta + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc.v[e] = rs.v[e] * g.v[e]; sh.v[e] = b.v[e] - mu.v[e] * sc.v[e]; } }
        const int64_t stride = (int64_t)gridDim.x * w.rpb;
        for (int64_t r0 = (int64_t)blockIdx.x * w.rpb + rl; r0 < rows; r0 += 4 * stride) {
            Raw8<T> v[4], q[4]; F8 m[4]; UpTaps<T> ut[UP ? 4 : 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = r0 + j * stride;
                if (r < rows) {
                    if (UP) up_issue<T>(z, ldz, up, r, c, ut[UP ? j : 0]); else v[j].load(z + r * ldz + c);
                    if (RES) q[j].load(res + r * ldr + c);
                    if (MASK) m[j] = load8(mask + bn_image_of(r, rows_per_image, rpi_shift) * C + c);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = r0 + j * stride;
                if (r < rows) {
                    F8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float u = fmaf(UP ? up_value<T>(ut[UP ? j : 0], e) : v[j].get(e), sc.v[e], sh.v[e]);
                        if (ACT == 1) u = mish_f(u);
                        if (ACT == 2) u = fmaxf(u, 0.f);
                        if (ACT == 3) u = fminf(fmaxf(u, 0.f), 6.f);
                        if (MASK) u *= m[j].v[e];
                        if (RES) u += q[j].get(e);
                        o.v[e] = u;
                    }
                    store8(y + r * ldy + c, o);
                }
            }
        }
    }
}

static inline int bn_walk_blocks(int64_t rows, int rpb) {
    int64_t b = cdiv(rows, (int64_t)rpb * 4);
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int ss_bn_apply(int dtype, const void* z, int64_t ldz, const float* stats, const float* gamma, const float* beta,
                           const void* res, int64_t ldr, const float* mask, int64_t rows_per_image, void* y, int64_t ldy,
                           int64_t rows, int C, int act, hipStream_t stream) {
    SS_CHECK(z && stats && gamma && beta && y && rows > 0 && C > 0 && C % 8 == 0, "ss_bn_apply: bad arguments (C=%d)", C);
    SS_CHECK(ldz >= C && ldy >= C && ldz % 8 == 0 && ldy % 8 == 0 && (!res || (ldr >= C && ldr % 8 == 0)), "ss_bn_apply: bad leading dimensions");
    SS_CHECK(!mask || rows_per_image > 0, "ss_bn_apply: rows_per_image required with a mask");
    SS_CHECK(act >= 0 && act <= 3, "ss_bn_apply: act must be 0 (none), 1 (Mish), 2 (ReLU) or 3 (ReLU6)");
    ColGeom g = col_geom(rows, C);
    BnWalk w{g.cpb, g.rpb};
    const int blocks = bn_walk_blocks(rows, g.rpb), sh = pow2_shift(rows_per_image);
#define SS_BN_APPLY(ACT, MASK, RES) hipLaunchKernelGGL((bn_apply_kernel<T, ACT, MASK, RES>), dim3(blocks), dim3(256), 0, stream, (const T*)z, ldz, \
        stats, gamma, beta, (const T*)res, ldr, mask, rows_per_image, sh, (T*)y, ldy, rows, C, w, UpGeom{})
#define SS_BN_APPLY_ACT(ACT) do { if (mask) { if (res) SS_BN_APPLY(ACT, true, true); else SS_BN_APPLY(ACT, true, false); } \
                                  else { if (res) SS_BN_APPLY(ACT, false, true); else SS_BN_APPLY(ACT, false, false); } } while (0)
    SS_DISPATCH(dtype, {
        if (act == 1) SS_BN_APPLY_ACT(1); else if (act == 2) SS_BN_APPLY_ACT(2); else if (act == 3) SS_BN_APPLY_ACT(3); else SS_BN_APPLY_ACT(0);
    });
#undef SS_BN_APPLY_ACT
#undef SS_BN_APPLY
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_bn_apply_up(int dtype, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo, const float* stats,
                              const float* gamma, const float* beta, void* y, int64_t ldy, int C, int act, hipStream_t stream) {
    SS_CHECK(x && stats && gamma && beta && y && C > 0 && C % 8 == 0 && aligned16(x) && aligned16(y), "ss_bn_apply_up: bad arguments (C=%d)", C);
    SS_CHECK(ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0, "ss_bn_apply_up: bad leading dimensions");
    SS_CHECK(act == 0 || act == 1, "ss_bn_apply_up: act must be 0 (none) or 1 (Mish)");
    SS_UP_CHECK("ss_bn_apply_up");
    SS_CHECK(!bvec || up2_ok(hi, wi, ho, wo), "ss_bn_apply_up: a per-image constant is only taken by the x2 resize");
    const int64_t rows = (int64_t)B * ho * wo;
    ColGeom g = col_geom(rows, C);
    BnWalk w{g.cpb, g.rpb};
    const int blocks = bn_walk_blocks(rows, g.rpb);
    const UpGeom up = up_geom(hi, wi, ho, wo);
#define SS_BN_APPLY_UP(ACT) hipLaunchKernelGGL((bn_apply_kernel<T, ACT, false, false, true>), dim3(blocks), dim3(256), 0, stream, (const T*)x, ldx, \
        stats, gamma, beta, (const T*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, -1, (T*)y, ldy, rows, C, w, up)
    if (up2_ok(hi, wi, ho, wo)) {
        using namespace w8;
        const Up2Geom u = up2_geom(B, hi, wi, C);
        dim3 grid((unsigned)(B * u.xtiles * u.ystrips), (unsigned)u.groups);
        SS_DISPATCH(dtype, {
            if (act) hipLaunchKernelGGL((bn_apply_up2_kernel<T, 1>), grid, dim3(256), 0, stream, (const T*)x, ldx, bvec, hi, wi, stats, gamma, beta, (T*)y, ldy, C, u);
            else hipLaunchKernelGGL((bn_apply_up2_kernel<T, 0>), grid, dim3(256), 0, stream, (const T*)x, ldx, bvec, hi, wi, stats, gamma, beta, (T*)y, ldy, C, u);
        });
    } else SS_DISPATCH(dtype, { if (act) SS_BN_APPLY_UP(1); else SS_BN_APPLY_UP(0); });
#undef SS_BN_APPLY_UP
    SS_LAUNCH_CHECK();
    return 0;
}

// backward.  g = dy * mask * act'(u), u = xhat * gamma + beta.
//   pass 1: partial[b][0][c] = sum g, partial[b][1][c] = sum g * xhat over the rows of workgroup b (fixed order)
//   pass 2: dz = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))
// PASS2 selects which; both recompute g from dy and z (10 bytes per element in total, nothing extra stored by the forward).
template <typename T, int ACT, bool MASK, bool PASS2, bool UP = false>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const T* __restrict__ dy, int64_t ldd, const T* __restrict__ z, int64_t ldz,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask,
        int64_t rows_per_image, int rpi_shift, const float* __restrict__ coef, T* __restrict__ dz, int64_t ldo,
        float* __restrict__ partial, int64_t rows, int C, BnWalk w, UpGeom up) {
    __shared__ float red[PASS2 ? 1 : 256 * 8];
    const int chunks = C / 8, t = threadIdx.x, ch = t % w.cpb, rl = t / w.cpb;
    const bool rowlane = rl < w.rpb;
    for (int cbase = 0; cbase < chunks; cbase += w.cpb) {
        const int chunk = cbase + ch, c = chunk * 8;
        const bool live = rowlane && chunk < chunks;
        float acc[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
        if (live) {
            F8 xs, xo, gm, bt, k0, k1, k2;             // xhat = z * xs + xo;  dz = k0 * g + k1 + xhat * k2
            { F8 mu = load8(stats + c), rs = load8(stats + C + c);
              gm = load8(gamma + c); bt = load8(beta + c);
#pragma unroll
              for (int e = 0; e < 8; ++e) { xs.v[e] = rs.v[e]; xo.v[e] = -mu.v[e] * rs.v[e]; }
              if (PASS2) { F8 c1 = load8(coef + c), c2 = load8(coef + C + c);
#pragma unroll
                  for (int e = 0; e < 8; ++e) { k0.v[e] = gm.v[e] * rs.v[e]; k1.v[e] = -k0.v[e] * c1.v[e]; k2.v[e] = -k0.v[e] * c2.v[e]; } } }
            const int64_t stride = (int64_t)gridDim.x * w.rpb;
            for (int64_t r0 = (int64_t)blockIdx.x * w.rpb + rl; r0 < rows; r0 += 4 * stride) {
                Raw8<T> d[4], v[4]; F8 m[4]; UpTaps<T> ut[UP ? 4 : 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r = r0 + j * stride;
                    if (r < rows) {
                        d[j].load(dy + r * ldd + c);
                        if (UP) up_issue<T>(z, ldz, up, r, c, ut[UP ? j : 0]); else v[j].load(z + r * ldz + c);
                        if (MASK) m[j] = load8(mask + bn_image_of(r, rows_per_image, rpi_shift) * C + c);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r = r0 + j * stride;
                    if (r < rows) {
                        F8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = fmaf(UP ? up_value<T>(ut[UP ? j : 0], e) : v[j].get(e), xs.v[e], xo.v[e]);
                            float gg = d[j].get(e);
                            if (MASK) gg *= m[j].v[e];
                            if (ACT == 1) gg *= mish_grad(fmaf(xh, gm.v[e], bt.v[e]));
                            if (ACT == 2) gg = fmaf(xh, gm.v[e], bt.v[e]) > 0.f ? gg : 0.f;
                            if (ACT == 3) { const float u = fmaf(xh, gm.v[e], bt.v[e]); gg = (u > 0.f && u < 6.f) ? gg : 0.f; }
                            if (PASS2) o.v[e] = fmaf(k0.v[e], gg, fmaf(xh, k2.v[e], k1.v[e]));
                            else { acc[0][e] += gg; acc[1][e] = fmaf(gg, xh, acc[1][e]); }
                        }
                        if (PASS2) store8(dz + r * ldo + c, o);
                    }
                }
            }
        }
        if (!PASS2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; ++e) red[t * 8 + e] = acc[k][e];
                __syncthreads();
                if (rl == 0 && chunk < chunks) {
                    float* dst = partial + ((int64_t)blockIdx.x * 2 + k) * C + c;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float s = 0.f;
                        for (int j = 0; j < w.rpb; ++j) s += red[(j * w.cpb + ch) * 8 + e];
                        dst[e] = s;
                    }
                }
            }
        }
    }
}

// dgb[0][c] = dgamma = sum g*xhat ; dgb[1][c] = dbeta = sum g ; coef[0][c] = mean g, coef[1][c] = mean g*xhat (0 in eval mode)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nb, int C, int64_t n, int training,
                                       float* __restrict__ dgb, float* __restrict__ coef) {
    __shared__ double sh[2][32][9];
    const int cx = threadIdx.x & 7, part = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    double s = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = part; b < nb; b += 32) { s += partial[((int64_t)b * 2 + 0) * C + c]; q += partial[((int64_t)b * 2 + 1) * C + c]; }
    }
    sh[0][part][cx] = s; sh[1][part][cx] = q;
    __syncthreads();
    if (part != 0 || c >= C) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < 32; ++k) { s += sh[0][k][cx]; q += sh[1][k][cx]; }
    dgb[c] = (float)q; dgb[C + c] = (float)s;
    coef[c] = training ? (float)(s / (double)n) : 0.f;
    coef[C + c] = training ? (float)(q / (double)n) : 0.f;
}

// workspace: nb*2*C floats (partials) followed by 2*C floats (coefficients)
extern "C" int ss_bn_bwd(int dtype, const void* dy, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
                         const float* beta, const float* mask, int64_t rows_per_image, int training, void* dz, int64_t ldo,
                         float* dgamma_dbeta, int64_t rows, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && z && stats && gamma && beta && dz && dgamma_dbeta && workspace && rows > 0 && C > 0 && C % 8 == 0, "ss_bn_bwd: bad arguments");
    SS_CHECK(ldd >= C && ldz >= C && ldo >= C && ldd % 8 == 0 && ldz % 8 == 0 && ldo % 8 == 0, "ss_bn_bwd: bad leading dimensions");
    SS_CHECK(!mask || rows_per_image > 0, "ss_bn_bwd: rows_per_image required with a mask");
    SS_CHECK(act >= 0 && act <= 3, "ss_bn_bwd: act must be 0 (none), 1 (Mish), 2 (ReLU) or 3 (ReLU6)");
    ColGeom g = col_geom(rows, C);
    SS_CHECK(ws_bytes >= ((int64_t)g.nb * 2 * C + 2 * C) * 4, "ss_bn_bwd: workspace too small");
    float* coef = workspace + (int64_t)g.nb * 2 * C;
    BnWalk w{g.cpb, g.rpb};
    const int blocks2 = bn_walk_blocks(rows, g.rpb), sh = pow2_shift(rows_per_image);
#define SS_BN_BWD(ACT, MASK, P2, GRID) hipLaunchKernelGGL((bn_bwd_kernel<T, ACT, MASK, P2>), dim3(GRID), dim3(256), 0, stream, (const T*)dy, ldd, \
        (const T*)z, ldz, stats, gamma, beta, mask, rows_per_image, sh, coef, (T*)dz, ldo, workspace, rows, C, w, UpGeom{})
#define SS_BN_BWD_ACT(ACT, P2, GRID) do { if (mask) SS_BN_BWD(ACT, true, P2, GRID); else SS_BN_BWD(ACT, false, P2, GRID); } while (0)
#define SS_BN_BWD_PASS(P2, GRID) do { \
        if (act == 1) SS_BN_BWD_ACT(1, P2, GRID); else if (act == 2) SS_BN_BWD_ACT(2, P2, GRID); \
        else if (act == 3) SS_BN_BWD_ACT(3, P2, GRID); else SS_BN_BWD_ACT(0, P2, GRID); } while (0)
    SS_DISPATCH(dtype, {
        SS_BN_BWD_PASS(false, g.nb);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(256), 0, stream, workspace, g.nb, C, rows, training, dgamma_dbeta, coef);
        SS_LAUNCH_CHECK();
        SS_BN_BWD_PASS(true, blocks2);
    });
#undef SS_BN_BWD_PASS
#undef SS_BN_BWD_ACT
#undef SS_BN_BWD
    SS_LAUNCH_CHECK();
    return 0;
}

// Second pass alone, for a BatchNorm without activation whose first-pass sums came from elsewhere (ss_lawin_branch_bn_sums):
// dz = gamma * rstd * (dy - coef[0] - xhat * coef[1]),  coef = [2, C] (mean dy, mean dy * xhat; zeros in eval mode)
extern "C" int ss_bn_bwd_coef(int dtype, const void* dy, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
                              const float* beta, const float* coef, void* dz, int64_t ldo, int64_t rows, int C, hipStream_t stream) {
    SS_CHECK(dy && z && stats && gamma && beta && coef && dz && rows > 0 && C > 0 && C % 8 == 0, "ss_bn_bwd_coef: bad arguments");
    SS_CHECK(ldd >= C && ldz >= C && ldo >= C && ldd % 8 == 0 && ldz % 8 == 0 && ldo % 8 == 0, "ss_bn_bwd_coef: bad leading dimensions");
    ColGeom g = col_geom(rows, C);
    BnWalk w{g.cpb, g.rpb};
    const int blocks2 = bn_walk_blocks(rows, g.rpb);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((bn_bwd_kernel<T, 0, false, true>), dim3(blocks2), dim3(256), 0, stream, (const T*)dy, ldd, (const T*)z, ldz, stats,
                                          gamma, beta, (const float*)nullptr, (int64_t)0, -1, coef, (T*)dz, ldo, (float*)nullptr, rows, C, w, UpGeom{}));
    SS_LAUNCH_CHECK();
    return 0;
}

// the same with z = resize(x) read through its taps (no mask): dz is the gradient of the RESIZED map, [B*ho*wo, C]
extern "C" int ss_bn_bwd_up(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo,
                            const float* stats, const float* gamma, const float* beta, int training, void* dz, int64_t ldo,
                            float* dgamma_dbeta, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && stats && gamma && beta && dz && dgamma_dbeta && workspace && C > 0 && C % 8 == 0, "ss_bn_bwd_up: bad arguments");
    SS_CHECK(ldd >= C && ldx >= C && ldo >= C && ldd % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && aligned16(dy) && aligned16(x) && aligned16(dz),
             "ss_bn_bwd_up: bad leading dimensions");
    SS_CHECK(act == 0 || act == 1, "ss_bn_bwd_up: act must be 0 (none) or 1 (Mish)");
    SS_UP_CHECK("ss_bn_bwd_up");
    SS_CHECK(!bvec || up2_ok(hi, wi, ho, wo), "ss_bn_bwd_up: a per-image constant is only taken by the x2 resize");
    const int64_t rows = (int64_t)B * ho * wo;
    ColGeom g = col_geom(rows, C);
    const int nbw = up_partial_rows(g.nb, B, hi, wi, ho, wo, C);
    SS_CHECK(ws_bytes >= ((int64_t)nbw * 2 * C + 2 * C) * 4, "ss_bn_bwd_up: workspace too small");
    float* coef = workspace + (int64_t)nbw * 2 * C;
    BnWalk w{g.cpb, g.rpb};
    const int blocks2 = bn_walk_blocks(rows, g.rpb);
    const UpGeom up = up_geom(hi, wi, ho, wo);
#define SS_BN_BWD_UP(ACT, P2, GRID) hipLaunchKernelGGL((bn_bwd_kernel<T, ACT, false, P2, true>), dim3(GRID), dim3(256), 0, stream, (const T*)dy, ldd, \
        (const T*)x, ldx, stats, gamma, beta, (const float*)nullptr, (int64_t)0, -1, coef, (T*)dz, ldo, workspace, rows, C, w, up)
    if (up2_ok(hi, wi, ho, wo)) {
        using namespace w4;
        const Up2Geom u = up2_geom(B, hi, wi, C);
        const int nb = up2_blocks(u, B);
        dim3 g1((unsigned)nb, (unsigned)u.groups), g2((unsigned)(B * u.xtiles * u.ystrips), (unsigned)u.groups);
#define SS_BN_BWD_UP2(ACT, P2, GRID) hipLaunchKernelGGL((bn_bwd_up2_kernel<T, ACT, P2>), GRID, dim3(256), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, \
        bvec, B, hi, wi, stats, gamma, beta, coef, P2 ? (T*)dz : (T*)nullptr, ldo, workspace, C, u)
        SS_DISPATCH(dtype, {
            if (act) SS_BN_BWD_UP2(1, false, g1); else SS_BN_BWD_UP2(0, false, g1);
            SS_LAUNCH_CHECK();
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(256), 0, stream, workspace, nb, C, rows, training, dgamma_dbeta, coef);
            SS_LAUNCH_CHECK();
            if (act) SS_BN_BWD_UP2(1, true, g2); else SS_BN_BWD_UP2(0, true, g2);
        });
#undef SS_BN_BWD_UP2
        SS_LAUNCH_CHECK();
        return 0;
    }
    SS_DISPATCH(dtype, {
        if (act) SS_BN_BWD_UP(1, false, g.nb); else SS_BN_BWD_UP(0, false, g.nb);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(256), 0, stream, workspace, g.nb, C, rows, training, dgamma_dbeta, coef);
        SS_LAUNCH_CHECK();
        if (act) SS_BN_BWD_UP(1, true, blocks2); else SS_BN_BWD_UP(0, true, blocks2);
    });
#undef SS_BN_BWD_UP
    SS_LAUNCH_CHECK();
    return 0;
}

// x2 only: pass 1 as ss_bn_bwd_up, pass 2 returns dx [B*hi*wi, C] = ss_bilinear_bwd(dz) directly (dz is never written)
extern "C" int ss_bn_bwd_up2_dx(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi,
                                const float* stats, const float* gamma, const float* beta, int training, void* dx, int64_t ldo,
                                float* dgamma_dbeta, int C, int act, float* workspace, int64_t ws_bytes, void* g_scratch, int64_t ldg,
                                hipStream_t stream) {
    SS_CHECK(dy && x && stats && gamma && beta && dx && dgamma_dbeta && workspace && C > 0 && C % 8 == 0, "ss_bn_bwd_up2_dx: bad arguments");
    SS_CHECK(!g_scratch || (ldg >= C && ldg % 8 == 0 && aligned16(g_scratch) && g_scratch != dx), "ss_bn_bwd_up2_dx: bad g_scratch");
    if (!act) g_scratch = nullptr;                         // nothing to keep: g is dy
    SS_CHECK(ldd >= C && ldx >= C && ldo >= C && ldd % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx),
             "ss_bn_bwd_up2_dx: bad leading dimensions");
    SS_CHECK(act == 0 || act == 1, "ss_bn_bwd_up2_dx: act must be 0 (none) or 1 (Mish)");
    const int ho = 2 * hi, wo = 2 * wi;
    SS_UP_CHECK("ss_bn_bwd_up2_dx");
    const int64_t rows = (int64_t)B * ho * wo;
    ColGeom g = col_geom(rows, C);
    const int nbw = up_partial_rows(g.nb, B, hi, wi, ho, wo, C);
    SS_CHECK(ws_bytes >= ((int64_t)nbw * 2 * C + 2 * C) * 4, "ss_bn_bwd_up2_dx: workspace too small");
    float* coef = workspace + (int64_t)nbw * 2 * C;
    using namespace w4;
    const Up2Geom u = up2_geom(B, hi, wi, C);
    const int nb = up2_blocks(u, B);
    const Up2DxGeom u2 = up2dx_geom(B, hi, wi, C);
    dim3 g1((unsigned)nb, (unsigned)u.groups), g2((unsigned)(B * u2.xtiles * u2.ystrips), (unsigned)u2.groups);
    SS_DISPATCH(dtype, {
        if (act) hipLaunchKernelGGL((bn_bwd_up2_kernel<T, 1, false>), g1, dim3(256), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, bvec, B, hi, wi, stats, gamma, beta,
                                    coef, (T*)g_scratch, ldg, workspace, C, u);
        else hipLaunchKernelGGL((bn_bwd_up2_kernel<T, 0, false>), g1, dim3(256), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, bvec, B, hi, wi, stats, gamma, beta,
                                coef, (T*)nullptr, (int64_t)0, workspace, C, u);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(256), 0, stream, workspace, nb, C, rows, training, dgamma_dbeta, coef);
        SS_LAUNCH_CHECK();
        if (act && g_scratch) hipLaunchKernelGGL((bn_bwd_up2_dx_kernel<T, 0>), g2, dim3(512), 0, stream, (const T*)g_scratch, ldg, (const T*)x, ldx, bvec, hi, wi, stats,
                                                 gamma, beta, coef, (T*)dx, ldo, C, u2);
        else if (act) hipLaunchKernelGGL((bn_bwd_up2_dx_kernel<T, 1>), g2, dim3(512), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, bvec, hi, wi, stats, gamma, beta, coef,
                                    (T*)dx, ldo, C, u2);
        else hipLaunchKernelGGL((bn_bwd_up2_dx_kernel<T, 0>), g2, dim3(512), 0, stream, (const T*)dy, ldd, (const T*)x, ldx, bvec, hi, wi, stats, gamma, beta, coef,
                                (T*)dx, ldo, C, u2);
    });
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ BatchNorm + Mish + Dropout2d + the nc-class 1x1 classifier in one pass
// LawinHead's tail (modules/lawin.py:227-228): seg = linear_pred(dropout(Mish(BN(zl)))).  The E-channel activation between the
// BatchNorm and the classifier is a full-resolution tensor that is written once and read twice (classifier forward, its weight
// gradient) and whose gradient is written once and read twice (both passes of the BatchNorm backward).  Here it never exists:
//   forward : a pixel's E channels sit in the cpb <= 64 lanes of one wave (8 channels per lane); every lane forms its share of the
//             NC dot products with the classifier rows it keeps in registers, the lanes are summed and one lane stores the logits;
//   backward: d(activation) = dseg . Wp is 6 fmas per element from the pixel's NC logit gradients (a broadcast load) instead of
//             a 2-byte read; pass 1 also accumulates the classifier's weight gradient from the recomputed activation.
// Mish and its derivative share one exp and their reciprocals.
__device__ __forceinline__ void mish_both(float x, float& val, float& grad) {
    const float e = __expf(fminf(x, 20.f)), n = e * (e + 2.f), d = __builtin_amdgcn_rcpf(n + 2.f);
    const float t = n * d, omt2 = 4.f * (n + 1.f) * d * d;
    val = x * t;
    grad = t + x * omt2 * (e * __builtin_amdgcn_rcpf(1.f + e));
}
template <typename T> __device__ __forceinline__ void load_row_nc(const T* p, int nc, float (&o)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = k < nc ? to_f(p[k]) : 0.f;
}

template <typename T, int NC, bool MASK>
__global__ __launch_bounds__(256) void bn_apply_cls_kernel(const T* __restrict__ z, int64_t ldz, const float* __restrict__ stats,
        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, int64_t rows_per_image, int rpi_shift,
        const float* __restrict__ wp, const float* __restrict__ bp, T* __restrict__ seg, int64_t lds_, int64_t rows, int C, BnWalk w) {
    const int t = threadIdx.x, ch = t % w.cpb, rl = t / w.cpb, c = ch * 8;
    F8 sc, sh; float wk[NC][8];
    { F8 mu = load8(stats + c), rs = load8(stats + C + c), g = load8(gamma + c), b = load8(beta + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc.v[e] = rs.v[e] * g.v[e]; sh.v[e] = b.v[e] - mu.v[e] * sc.v[e]; } }
#pragma unroll
    for (int k = 0; k < NC; ++k) { F8 r = load8(wp + (int64_t)k * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) wk[k][e] = r.v[e]; }
    const int64_t stride = (int64_t)gridDim.x * w.rpb;
    for (int64_t r0 = (int64_t)blockIdx.x * w.rpb + rl; r0 < rows; r0 += 4 * stride) {      // all lanes of a wave share r0 (cpb divides 64)
        Raw8<T> v[4]; F8 m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j * stride;
            if (r < rows) {
                v[j].load(z + r * ldz + c);
                if (MASK) m[j] = load8(mask + bn_image_of(r, rows_per_image, rpi_shift) * C + c);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j * stride;
            if (r < rows) {
                float dot[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) dot[k] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = mish_f(fmaf(v[j].get(e), sc.v[e], sh.v[e]));
                    if (MASK) u *= m[j].v[e];
#pragma unroll
                    for (int k = 0; k < NC; ++k) dot[k] = fmaf(u, wk[k][e], dot[k]);
                }
#pragma unroll
                for (int k = 0; k < NC; ++k) dot[k] = group_sum(dot[k], w.cpb);
                if (ch == 0) {
#pragma unroll
                    for (int k = 0; k < NC; ++k) seg[r * lds_ + k] = from_f<T>(dot[k] + bp[k]);
                }
            }
        }
    }
}

// PASS2 = false: partial[b][0][c] = sum g, [1] = sum g * xhat, partial2[b][k][c] = sum dseg[k] * activation[c]; true: dz
template <typename T, int NC, bool MASK, bool PASS2>
__global__ __launch_bounds__(256) void bn_bwd_cls_kernel(const T* __restrict__ dseg, int64_t ldd, const T* __restrict__ z, int64_t ldz,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask,
        int64_t rows_per_image, int rpi_shift, const float* __restrict__ wp, const float* __restrict__ coef, T* __restrict__ dz, int64_t ldo,
        float* __restrict__ partial, float* __restrict__ partial2, int64_t rows, int C, BnWalk w) {
    constexpr int RF = 2;                                  // rows in flight (the accumulators and classifier rows take the registers)
    __shared__ float red[PASS2 ? 1 : 256 * 8];
    const int t = threadIdx.x, ch = t % w.cpb, rl = t / w.cpb, c = ch * 8;
    float acc[2][8], aw[PASS2 ? 1 : NC][8], wk[NC][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    if (!PASS2) {
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) aw[k][e] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) { F8 r = load8(wp + (int64_t)k * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) wk[k][e] = r.v[e]; }
    F8 xs, xo, gm, bt, k0, k1, k2;                         // xhat = z * xs + xo;  dz = k0 * g + k1 + xhat * k2
    { F8 mu = load8(stats + c), rs = load8(stats + C + c);
      gm = load8(gamma + c); bt = load8(beta + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xs.v[e] = rs.v[e]; xo.v[e] = -mu.v[e] * rs.v[e]; }
      if (PASS2) { F8 c1 = load8(coef + c), c2 = load8(coef + C + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) { k0.v[e] = gm.v[e] * rs.v[e]; k1.v[e] = -k0.v[e] * c1.v[e]; k2.v[e] = -k0.v[e] * c2.v[e]; } } }
    const int64_t stride = (int64_t)gridDim.x * w.rpb;
    for (int64_t r0 = (int64_t)blockIdx.x * w.rpb + rl; r0 < rows; r0 += RF * stride) {
        Raw8<T> v[RF]; F8 m[RF]; float ds[RF][8];
#pragma unroll
        for (int j = 0; j < RF; ++j) {
            const int64_t r = r0 + j * stride;
            if (r < rows) {
                v[j].load(z + r * ldz + c);
                load_row_nc<T>(dseg + r * ldd, NC, ds[j]);
                if (MASK) m[j] = load8(mask + bn_image_of(r, rows_per_image, rpi_shift) * C + c);
            }
        }
#pragma unroll
        for (int j = 0; j < RF; ++j) {
            const int64_t r = r0 + j * stride;
            if (r < rows) {
                F8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = fmaf(v[j].get(e), xs.v[e], xo.v[e]);
                    float val, grd;
                    mish_both(fmaf(xh, gm.v[e], bt.v[e]), val, grd);
                    float gg = 0.f;
#pragma unroll
                    for (int k = 0; k < NC; ++k) gg = fmaf(ds[j][k], wk[k][e], gg);
                    if (MASK) { gg *= m[j].v[e]; val *= m[j].v[e]; }
                    gg *= grd;
                    if (PASS2) o.v[e] = fmaf(k0.v[e], gg, fmaf(xh, k2.v[e], k1.v[e]));
                    else {
                        acc[0][e] += gg; acc[1][e] = fmaf(gg, xh, acc[1][e]);
#pragma unroll
                        for (int k = 0; k < NC; ++k) aw[k][e] = fmaf(ds[j][k], val, aw[k][e]);
                    }
                }
                if (PASS2) store8(dz + r * ldo + c, o);
            }
        }
    }
    if (!PASS2) {
#pragma unroll
        for (int k = 0; k < 2 + NC; ++k) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[t * 8 + e] = k < 2 ? acc[k < 2 ? k : 0][e] : aw[k < 2 ? 0 : k - 2][e];
            __syncthreads();
            if (rl == 0) {
                float* dst = k < 2 ? partial + ((int64_t)blockIdx.x * 2 + k) * C + c : partial2 + ((int64_t)blockIdx.x * NC + (k - 2)) * C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = 0.f;
                    for (int j = 0; j < w.rpb; ++j) s += red[(j * w.cpb + ch) * 8 + e];
                    dst[e] = s;
                }
            }
        }
    }
}

// bf16 forward on the matrix cores: a wave owns 16 pixels at a time; its lanes (8 channels of one pixel each) leave the activation
// in a wave-private LDS tile [16][C], the tile is read back as v_mfma_f32_16x16x32_bf16 A fragments (lane = pixel, 8 consecutive
// channels) against the classifier rows held as B fragments in registers (classes >= NC are zero columns), and the lanes of the
// first NC columns store the logits.  Operands are bf16 with fp32 accumulation, which is what the reference's autocast conv does.
template <int NC, bool MASK, int CPB>
__global__ __launch_bounds__(256) void bn_apply_cls_mfma_kernel(const bf16* __restrict__ z, int64_t ldz, const float* __restrict__ stats,
        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, int64_t rows_per_image, int rpi_shift,
        const float* __restrict__ wp, const float* __restrict__ bp, bf16* __restrict__ seg, int64_t lds_, int64_t rows) {
    constexpr int C = CPB * 8, PITCH = C * 2 + 16, RPS = 64 / CPB, KS = C / 32 > 0 ? C / 32 : 1;      // RPS pixels per wave step
    __shared__ __attribute__((aligned(16))) char tile[4][16 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ch = lane % CPB, pr = lane / CPB, c = ch * 8;
    char* tl = tile[wave];
    F8 sc, sh;
    { F8 mu = load8(stats + c), rs = load8(stats + C + c), g = load8(gamma + c), b = load8(beta + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc.v[e] = rs.v[e] * g.v[e]; sh.v[e] = b.v[e] - mu.v[e] * sc.v[e]; } }
    const int col = lane & 15, kg = lane >> 4;
    bf16x8 wf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = 32 * s + 8 * kg + e; wf[s][e] = (col < NC && k < C) ? (bf16)wp[(int64_t)col * C + k] : (bf16)0.f; }
    }
    const float bias = col < NC ? bp[col] : 0.f;
    const int64_t ntile = (rows + 15) / 16;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntile; t += (int64_t)gridDim.x * 4) {
        const int64_t r0 = t * 16;
        bf16x8 v[16 / RPS];
#pragma unroll
        for (int i = 0; i < 16 / RPS; ++i) {
            const int64_t r = min(r0 + i * RPS + pr, rows - 1);
            v[i] = *(const bf16x8*)(z + r * ldz + c);
        }
#pragma unroll
        for (int i = 0; i < 16 / RPS; ++i) {
            const int64_t r = min(r0 + i * RPS + pr, rows - 1);
            F8 m;
            if (MASK) m = load8(mask + bn_image_of(r, rows_per_image, rpi_shift) * C + c);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = mish_f(fmaf((float)v[i][e], sc.v[e], sh.v[e]));
                if (MASK) u *= m.v[e];
                o[e] = (bf16)u;
            }
            *(bf16x8*)(tl + (i * RPS + pr) * PITCH + c * 2) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (32 * s + 8 * kg < C) a = *(const bf16x8*)(tl + col * PITCH + (32 * s + 8 * kg) * 2);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[s], acc, 0, 0, 0);
        }
        // D[i = pixel 4 kg + e][j = class col]
        if (col < NC) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = r0 + 4 * kg + e;
                if (r < rows) seg[r * lds_ + col] = (bf16)(acc[e] + bias);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static inline bool cls_ok(int C, int nc) { const int ch = C / 8; return C % 8 == 0 && ch >= 1 && ch <= 64 && (ch & (ch - 1)) == 0 && nc >= 1 && nc <= 8; }
#define SS_NC_SWITCH(nc, ...) do { switch (nc) { \
        case 1: { constexpr int NC = 1; __VA_ARGS__; } break; case 2: { constexpr int NC = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NC = 3; __VA_ARGS__; } break; case 4: { constexpr int NC = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int NC = 5; __VA_ARGS__; } break; case 6: { constexpr int NC = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int NC = 7; __VA_ARGS__; } break; default: { constexpr int NC = 8; __VA_ARGS__; } break; } } while (0)

extern "C" int ss_bn_apply_classify(int dtype, const void* z, int64_t ldz, const float* stats, const float* gamma, const float* beta,
                                    const float* mask, int64_t rows_per_image, const float* wp, const float* bp, int nc, void* seg,
                                    int64_t lds_, int64_t rows, int C, hipStream_t stream) {
    SS_CHECK(z && stats && gamma && beta && wp && bp && seg && rows > 0 && cls_ok(C, nc), "ss_bn_apply_classify: bad arguments (C=%d: 8 * power of two <= 512; nc=%d <= 8)", C, nc);
    SS_CHECK(ldz >= C && ldz % 8 == 0 && lds_ >= nc && aligned16(z), "ss_bn_apply_classify: bad leading dimensions");
    SS_CHECK(!mask || rows_per_image > 0, "ss_bn_apply_classify: rows_per_image required with a mask");
    ColGeom g = col_geom(rows, C);
    BnWalk w{g.cpb, g.rpb};
    const int blocks = bn_walk_blocks(rows, g.rpb), sh = pow2_shift(rows_per_image);
    if (dtype == SS_BF16 && C >= 32) {
        int64_t nt = cdiv(cdiv(rows, 16), 4);
        const int grid = (int)(nt < 4096 ? nt : 4096);
#define SS_CLS_MFMA(CPB) do { SS_NC_SWITCH(nc, { \
            if (mask) hipLaunchKernelGGL((bn_apply_cls_mfma_kernel<NC, true, CPB>), dim3(grid), dim3(256), 0, stream, (const bf16*)z, ldz, stats, gamma, beta, \
                                         mask, rows_per_image, sh, wp, bp, (bf16*)seg, lds_, rows); \
            else hipLaunchKernelGGL((bn_apply_cls_mfma_kernel<NC, false, CPB>), dim3(grid), dim3(256), 0, stream, (const bf16*)z, ldz, stats, gamma, beta, \
                                    mask, rows_per_image, sh, wp, bp, (bf16*)seg, lds_, rows); }); } while (0)
        switch (C / 8) { case 4: SS_CLS_MFMA(4); break; case 8: SS_CLS_MFMA(8); break; case 16: SS_CLS_MFMA(16); break;
                         case 32: SS_CLS_MFMA(32); break; default: SS_CLS_MFMA(64); break; }
#undef SS_CLS_MFMA
        SS_LAUNCH_CHECK();
        return 0;
    }
    SS_DISPATCH(dtype, SS_NC_SWITCH(nc, {
        if (mask) hipLaunchKernelGGL((bn_apply_cls_kernel<T, NC, true>), dim3(blocks), dim3(256), 0, stream, (const T*)z, ldz, stats, gamma, beta, mask,
                                     rows_per_image, sh, wp, bp, (T*)seg, lds_, rows, C, w);
        else hipLaunchKernelGGL((bn_apply_cls_kernel<T, NC, false>), dim3(blocks), dim3(256), 0, stream, (const T*)z, ldz, stats, gamma, beta, mask,
                                rows_per_image, sh, wp, bp, (T*)seg, lds_, rows, C, w);
    }));
    SS_LAUNCH_CHECK();
    return 0;
}

// workspace: nb*2*C (BatchNorm partials) + 2*C (coefficients) + nb*nc*C (classifier weight-gradient partials) floats, nb <= 1024
extern "C" int ss_bn_bwd_classify(int dtype, const void* dseg, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
                                  const float* beta, const float* mask, int64_t rows_per_image, int training, const float* wp, int nc, void* dz,
                                  int64_t ldo, float* dgamma_dbeta, float* dwp, int64_t rows, int C, float* workspace, int64_t ws_bytes,
                                  hipStream_t stream) {
    SS_CHECK(dseg && z && stats && gamma && beta && wp && dz && dgamma_dbeta && dwp && workspace && rows > 0 && cls_ok(C, nc),
             "ss_bn_bwd_classify: bad arguments (C=%d: 8 * power of two <= 512; nc=%d <= 8)", C, nc);
    SS_CHECK(ldd >= nc && ldz >= C && ldo >= C && ldz % 8 == 0 && ldo % 8 == 0 && aligned16(z) && aligned16(dz), "ss_bn_bwd_classify: bad leading dimensions");
    SS_CHECK(!mask || rows_per_image > 0, "ss_bn_bwd_classify: rows_per_image required with a mask");
    ColGeom g = col_geom(rows, C);
    SS_CHECK(ws_bytes >= ((int64_t)g.nb * (2 + nc) * C + 2 * C) * 4, "ss_bn_bwd_classify: workspace too small");
    float* coef = workspace + (int64_t)g.nb * 2 * C;
    float* part2 = coef + 2 * C;
    BnWalk w{g.cpb, g.rpb};
    const int blocks2 = bn_walk_blocks(rows, g.rpb), sh = pow2_shift(rows_per_image);
#define SS_BN_CLS(MASK, P2, GRID) hipLaunchKernelGGL((bn_bwd_cls_kernel<T, NC, MASK, P2>), dim3(GRID), dim3(256), 0, stream, (const T*)dseg, ldd, (const T*)z, ldz, \
        stats, gamma, beta, mask, rows_per_image, sh, wp, coef, (T*)dz, ldo, workspace, part2, rows, C, w)
    SS_DISPATCH(dtype, SS_NC_SWITCH(nc, {
        if (mask) SS_BN_CLS(true, false, g.nb); else SS_BN_CLS(false, false, g.nb);
    }));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(256), 0, stream, workspace, g.nb, C, rows, training, dgamma_dbeta, coef);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv((int64_t)nc * C, 8), 1), dim3(256), 0, stream, part2, g.nb, nc * C, 1.f, dwp, 0);
    SS_LAUNCH_CHECK();
    SS_DISPATCH(dtype, SS_NC_SWITCH(nc, {
        if (mask) SS_BN_CLS(true, true, blocks2); else SS_BN_CLS(false, true, blocks2);
    }));
#undef SS_BN_CLS
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ helpers for narrow / sliced tensors
// column sums of a narrow matrix (C <= 32, any C): out[c] = sum_r x[r][c]   (e.g. the bias gradient of the nc-class classifier)
template <typename T>
__global__ __launch_bounds__(256) void colsum_small_kernel(const T* __restrict__ x, int64_t ld, int64_t rows, int C, float* __restrict__ partial) {
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 32; ++c) if (c < C) acc[c] += to_f(x[r * ld + c]);
    }
    __shared__ float sh[4][32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { float v = wave_sum(acc[c]); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v; }
    __syncthreads();
    if (threadIdx.x < C) partial[(int64_t)blockIdx.x * C + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
extern "C" int ss_colsum_small(int dtype, const void* x, int64_t ld, int64_t rows, int C, float* out, float* workspace, int64_t ws_bytes,
                               hipStream_t stream) {
    SS_CHECK(x && out && workspace && rows > 0 && C > 0 && C <= 32 && ld >= C, "ss_colsum_small: need 0 < C <= 32");
    int nb = (int)(cdiv(rows, 256 * 8) < 512 ? cdiv(rows, 256 * 8) : 512);
    if (nb < 1) nb = 1;
    SS_CHECK(ws_bytes >= (int64_t)nb * C * 4, "ss_colsum_small: workspace too small");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((colsum_small_kernel<T>), dim3(nb), dim3(256), 0, stream, (const T*)x, ld, rows, C, workspace));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(C, 8), 1), dim3(256), 0, stream, workspace, nb, C, 1.f, out, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// out[r][c] = sum_{k < nslices} x[r][k*slice_stride + c] + bvec[r / rows_per_image][c]     (gradient fan-in of concat slices)
template <typename T>
__global__ __launch_bounds__(256) void sum_slices_kernel(const T* __restrict__ x, int64_t ld, int nslices, int slice_stride,
        const float* __restrict__ bvec, int64_t rows_per_image, T* __restrict__ out, int64_t rows, int C) {
    const int chunks = C / 8; const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks; const int c = (int)(i % chunks) * 8;
        F8 acc = bvec ? load8(bvec + (r / rows_per_image) * C + c) : zero8();
        for (int k = 0; k < nslices; ++k) { F8 v = load8(x + r * ld + (int64_t)k * slice_stride + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e]; }
        store8(out + r * C + c, acc);
    }
}
extern "C" int ss_sum_slices(int dtype, const void* x, int64_t ld, int nslices, int slice_stride, const float* bvec, int64_t rows_per_image,
                             void* out, int64_t rows, int C, hipStream_t stream) {
    SS_CHECK(x && out && rows > 0 && C > 0 && C % 8 == 0 && nslices > 0 && slice_stride % 8 == 0 && ld % 8 == 0 &&
             ld >= (int64_t)(nslices - 1) * slice_stride + C && (!bvec || rows_per_image > 0), "ss_sum_slices: bad arguments");
    int64_t total = rows * (C / 8);
    int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((sum_slices_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, ld, nslices, slice_stride, bvec,
                                          rows_per_image, (T*)out, rows, C));
    SS_LAUNCH_CHECK();
    return 0;
}
