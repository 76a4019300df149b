// Depthwise 3x3 (stride 1, pad 1) + bias (+ GELU) on the channels-last token layout of the MiT Mix-FFN
// (reference modules/segformer.py:44-64).  Bandwidth-bound: every thread owns 8 consecutive channels
// of one pixel, the 9 taps are 16-byte loads that hit L1/L2 (each input row is read by 3 output rows of the same
// block column), weights are repacked tap-major [9][C] fp32.
//   forward : y = act(dw(x) + b)
//   backward: g = dy * act'(dw(x) + b)  (u recomputed, not stored)  +  per-block partials of dW[9][C], db[C];
//             dx = dw_flipped(g) reuses the forward kernel with the flipped taps and no bias / activation.
#include "common.h"
#include "colreduce.h"

template <typename T>
__device__ __forceinline__ void dw_taps(const T* __restrict__ x, const float* __restrict__ w, int64_t img, int yy, int xx, int H, int W,
                                        int C, int c, float (&u)[8]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        int sy = yy + ky - 1;
        if (sy < 0 || sy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int sx = xx + kx - 1;
            if (sx < 0 || sx >= W) continue;
            F8 v = load8(x + ((img * H + sy) * W + sx) * C + c), ww = load8(w + (ky * 3 + kx) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] += v.v[e] * ww.v[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
        T* __restrict__ y, int B, int H, int W, int C, int act) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * H * W * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int xx = (int)(p % W); p /= W; const int yy = (int)(p % H); const int64_t img = p / H;
        float u[8];
        if (bias) { F8 b = load8(bias + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = b.v[e]; }
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f; }
        dw_taps(x, w, img, yy, xx, H, W, C, c, u);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = act == 1 ? gelu_erf(u[e]) : u[e];
        store8(y + ((img * H + yy) * W + xx) * C + c, o);
    }
}

// g = dy * act'(u);  partial[blk][10][C]: taps 0..8 = sum g * x(tap), 9 = sum g
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ g, float* __restrict__ partial, int B, int H, int W, int C, int act, int cpb, int rpb) {
    __shared__ float sh[256 * 8];
    const int t = threadIdx.x, ch = t % cpb, rl = t / cpb, nb = gridDim.x;
    const int chunks = C / 8;
    const int64_t pixels = (int64_t)B * H * W;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {
        const int chunk = cbase + ch, c = chunk * 8;
        float acc[10][8];
#pragma unroll
        for (int k = 0; k < 10; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
        if (rl < rpb && chunk < chunks)
            for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < pixels; p += (int64_t)nb * rpb) {
                int64_t q = p; const int xx = (int)(q % W); q /= W; const int yy = (int)(q % H); const int64_t img = q / H;
                float u[8];
                F8 b = load8(bias + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = b.v[e];
                F8 xv[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        int sy = yy + ky - 1, sx = xx + kx - 1;
                        if (sy < 0 || sy >= H || sx < 0 || sx >= W) { xv[ky * 3 + kx] = zero8(); continue; }
                        xv[ky * 3 + kx] = load8(x + ((img * H + sy) * W + sx) * C + c);
                        F8 ww = load8(w + (ky * 3 + kx) * C + c);
#pragma unroll
                        for (int e = 0; e < 8; ++e) u[e] += xv[ky * 3 + kx].v[e] * ww.v[e];
                    }
                F8 d = load8(dy + p * C + c), go;
#pragma unroll
                for (int e = 0; e < 8; ++e) { go.v[e] = act == 1 ? d.v[e] * gelu_erf_grad(u[e]) : d.v[e]; acc[9][e] += go.v[e]; }
                store8(g + p * C + c, go);
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[k][e] += go.v[e] * xv[k].v[e];
            }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[t * 8 + e] = acc[k][e];
            __syncthreads();
            if (rl == 0 && chunk < chunks) {
                float* dst = partial + ((int64_t)blockIdx.x * 10 + k) * C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) { float s = 0.f;
                    for (int j = 0; j < rpb; ++j) s += sh[(j * cpb + ch) * 8 + e];
                    dst[e] = s; }
            }
        }
    }
}

extern "C" int ss_dwconv3_fwd(int dtype, const void* x, const float* w9c, const float* bias, void* y, int B, int H, int W, int C,
                              int act, hipStream_t stream) {
    SS_CHECK(x && w9c && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_fwd: bad arguments (C=%d)", C);
    int64_t total = (int64_t)B * H * W * (C / 8);
    int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dwconv3_fwd_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, w9c, bias, (T*)y, B, H, W, C, act));
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db: [10][C] fp32 (9 taps tap-major, then bias).  workspace: nb * 10 * C floats.
extern "C" int ss_dwconv3_bwd(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* g, float* dw_db,
                              int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && w9c && bias && g && dw_db && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_bwd: bad arguments");
    ColGeom geo = col_geom((int64_t)B * H * W, C);
    SS_CHECK(ws_bytes >= (int64_t)geo.nb * 10 * C * 4, "ss_dwconv3_bwd: workspace too small");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dwconv3_bwd_kernel<T>), dim3(geo.nb), dim3(256), 0, stream, (const T*)dy, (const T*)x, w9c, bias,
                                          (T*)g, workspace, B, H, W, C, act, geo.cpb, geo.rpb));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, geo.nb, 10 * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}
