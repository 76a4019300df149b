// Depthwise 3x3 (stride 1, pad 1) + bias (+ GELU) on the channels-last token layout of the MiT Mix-FFN
// (reference modules/segformer.py:44-64).  Bandwidth-bound, so the kernels are built around L1 traffic:
// a thread owns 8 consecutive channels of ONE image column x and walks down a strip of TY rows keeping the 3x3 input window
// in registers (raw storage dtype), so every output costs 3 new 16-byte loads instead of 9, and the 9 x 8 weights stay in
// registers for the whole strip.  A workgroup is (chunk lanes) x (x lanes): lanes of a wave read consecutive 16-byte chunks
// of a pixel, then the next pixel -- contiguous kilobytes per wave instruction.
//   forward : y = act(dw(x) + b)
//   backward: g = dy * act'(dw(x) + b)  (u recomputed, not stored)  +  dW[9][C], db[C]: per-thread register accumulators over
//             all the strips a workgroup walks, one LDS combine per workgroup, fixed-order final reduce (deterministic);
//             dx = dw_flipped(g) reuses the forward kernel with the flipped taps and no bias / activation.
#include "common.h"
#include "colreduce.h"

#define DW_TY 16

template <typename T, int N> using vecn = T __attribute__((ext_vector_type(N)));

// NE channels per thread: 8 (16-byte bf16 accesses) in the forward, 4 in the backward (its 10 x NE accumulators + 9 x NE weights
// + window must leave room for several waves per SIMD) and for fp32 (16-byte accesses).
template <typename T, int NE> __device__ __forceinline__ vecn<T, NE> raw_load(const T* p) { return *(const vecn<T, NE>*)p; }
template <int NE> __device__ __forceinline__ void loadf(const float* p, float (&o)[NE]) {
#pragma unroll
    for (int i = 0; i < NE / 4; ++i) { floatx4 a = *(const floatx4*)(p + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * i + e] = a[e]; }
}

struct DwGeom { int cpb, xl, groups, xtiles, ystrips, chunks; };
static inline DwGeom dw_geom(int H, int W, int C, int ne) {
    DwGeom g; g.chunks = C / ne;
    const int cap = ne == 8 ? 32 : 64;
    int cpb = 1; while (cpb < g.chunks && cpb < cap) cpb <<= 1;
    g.cpb = cpb; g.xl = 256 / cpb; g.groups = (int)cdiv(g.chunks, cpb);
    g.xtiles = (int)cdiv(W, g.xl); g.ystrips = (int)cdiv(H, DW_TY);
    return g;
}

// one row of the window: columns x-1, x, x+1 of image row yy (zeros outside the image)
template <typename T, int NE>
__device__ __forceinline__ void dw_load_row(const T* __restrict__ x, int64_t img_base, int yy, int xx, int H, int W, int C, int c,
                                            vecn<T, NE> (&row)[3]) {
    const bool yok = yy >= 0 && yy < H;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int sx = xx + kx - 1;
        vecn<T, NE> v;
#pragma unroll
        for (int e = 0; e < NE; ++e) v[e] = (T)0.f;
        if (yok && sx >= 0 && sx < W) v = raw_load<T, NE>(x + (img_base + (int64_t)yy * W + sx) * C + c);
        row[kx] = v;
    }
}

template <typename T, int NE>
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
        T* __restrict__ y, int H, int W, int C, int act, DwGeom g) {
    typedef vecn<T, NE> V;
    const int tid = threadIdx.x, chunk = blockIdx.y * g.cpb + tid % g.cpb, xs = tid / g.cpb;
    const int xt = blockIdx.x % g.xtiles, ys = blockIdx.x / g.xtiles;
    const int xx = xt * g.xl + xs, y0 = ys * DW_TY, c = chunk * NE;
    if (chunk >= g.chunks || xx >= W) return;
    const int64_t img_base = (int64_t)blockIdx.z * H * W;
    float wt[9][NE], bv[NE];
#pragma unroll
    for (int k = 0; k < 9; ++k) loadf<NE>(w + k * C + c, wt[k]);
    if (bias) loadf<NE>(bias + c, bv);
    else {
#pragma unroll
        for (int e = 0; e < NE; ++e) bv[e] = 0.f; }
    V win[3][3];
    dw_load_row<T, NE>(x, img_base, y0 - 1, xx, H, W, C, c, win[0]);
    dw_load_row<T, NE>(x, img_base, y0, xx, H, W, C, c, win[1]);
    dw_load_row<T, NE>(x, img_base, y0 + 1, xx, H, W, C, c, win[2]);
    const int yend = min(H, y0 + DW_TY);
    for (int yy = y0; yy < yend; ++yy) {
        V nxt[3];                                          // row yy+2 travels while row yy is computed
        dw_load_row<T, NE>(x, img_base, yy + 2 <= yend ? yy + 2 : H, xx, H, W, C, c, nxt);
        float u[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) u[e] = bv[e];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < NE; ++e) u[e] += (float)win[ky][kx][e] * wt[ky * 3 + kx][e];
        V o;
#pragma unroll
        for (int e = 0; e < NE; ++e) o[e] = (T)(act == 1 ? gelu_erf(u[e]) : u[e]);
        *(V*)(y + (img_base + (int64_t)yy * W + xx) * C + c) = o;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { win[0][kx] = win[1][kx]; win[1][kx] = win[2][kx]; win[2][kx] = nxt[kx]; }
    }
}

// g = dy * act'(u);  partial[blockIdx.x][10][C]: taps 0..8 = sum g * x(tap), 9 = sum g   (this workgroup's channel group only)
template <typename T, int NE>
__global__ __launch_bounds__(256) void dwconv3_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ gout, float* __restrict__ partial, int B, int H, int W, int C, int act, DwGeom g) {
    typedef vecn<T, NE> V;
    __shared__ float sh[256 * NE];
    const int tid = threadIdx.x, cl = tid % g.cpb, chunk = blockIdx.y * g.cpb + cl, xs = tid / g.cpb, c = chunk * NE;
    const bool cok = chunk < g.chunks;
    float wt[9][NE], bv[NE], acc[10][NE];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[k][e] = 0.f;
    if (cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k) loadf<NE>(w + k * C + c, wt[k]);
        loadf<NE>(bias + c, bv);
        const int tiles_per_img = g.xtiles * g.ystrips, ntiles = B * tiles_per_img;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int img = tile / tiles_per_img, rem = tile % tiles_per_img;
            const int xx = (rem % g.xtiles) * g.xl + xs, y0 = (rem / g.xtiles) * DW_TY;
            if (xx >= W) continue;
            const int64_t img_base = (int64_t)img * H * W;
            V win[3][3];
            dw_load_row<T, NE>(x, img_base, y0 - 1, xx, H, W, C, c, win[0]);
            dw_load_row<T, NE>(x, img_base, y0, xx, H, W, C, c, win[1]);
            dw_load_row<T, NE>(x, img_base, y0 + 1, xx, H, W, C, c, win[2]);
            const int yend = min(H, y0 + DW_TY);
            V d = raw_load<T, NE>(dy + (img_base + (int64_t)y0 * W + xx) * C + c);
            for (int yy = y0; yy < yend; ++yy) {
                V nxt[3], dn = d;                              // row yy+2 of x and row yy+1 of dy travel while row yy is computed
                dw_load_row<T, NE>(x, img_base, yy + 2 <= yend ? yy + 2 : H, xx, H, W, C, c, nxt);
                const int64_t p = img_base + (int64_t)yy * W + xx;
                if (yy + 1 < yend) dn = raw_load<T, NE>(dy + (p + W) * C + c);
                float u[NE], go[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) u[e] = bv[e];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < NE; ++e) u[e] += (float)win[ky][kx][e] * wt[ky * 3 + kx][e];
                V gv;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    go[e] = act == 1 ? (float)d[e] * gelu_erf_grad(u[e]) : (float)d[e];
                    gv[e] = (T)go[e];
                    go[e] = (float)gv[e];                 // dW uses the stored (rounded) g, as a separate pass over g would
                    acc[9][e] += go[e];
                }
                *(V*)(gout + p * C + c) = gv;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[ky * 3 + kx][e] += go[e] * (float)win[ky][kx][e];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { win[0][kx] = win[1][kx]; win[1][kx] = win[2][kx]; win[2][kx] = nxt[kx]; }
                d = dn;
            }
        }
    }
    // combine the x lanes of each chunk lane through LDS in a fixed order
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) sh[tid * NE + e] = acc[k][e];
        __syncthreads();
        if (xs == 0 && cok) {
            float* dst = partial + ((int64_t)blockIdx.x * 10 + k) * C + c;
#pragma unroll
            for (int e = 0; e < NE; ++e) { float s = 0.f;
                for (int j = 0; j < g.xl; ++j) s += sh[(j * g.cpb + cl) * NE + e];
                dst[e] = s; }
        }
    }
}

extern "C" int ss_dwconv3_fwd(int dtype, const void* x, const float* w9c, const float* bias, void* y, int B, int H, int W, int C,
                              int act, hipStream_t stream) {
    SS_CHECK(x && w9c && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_fwd: bad arguments (C=%d)", C);
    SS_CHECK(aligned16(x) && aligned16(y) && aligned16(w9c) && (!bias || aligned16(bias)), "ss_dwconv3_fwd: pointers must be 16-byte aligned");
    DwGeom g = dw_geom(H, W, C, dtype == SS_BF16 ? 8 : 4);
    SS_CHECK((int64_t)g.xtiles * g.ystrips < 2147483647LL && B < 65536 && g.groups < 65536, "ss_dwconv3_fwd: grid too large");
    dim3 grid((unsigned)(g.xtiles * g.ystrips), g.groups, B);
    if (dtype == SS_BF16) hipLaunchKernelGGL((dwconv3_fwd_kernel<bf16, 8>), grid, dim3(256), 0, stream, (const bf16*)x, w9c, bias, (bf16*)y, H, W, C, act, g);
    else if (dtype == SS_F32) hipLaunchKernelGGL((dwconv3_fwd_kernel<float, 4>), grid, dim3(256), 0, stream, (const float*)x, w9c, bias, (float*)y, H, W, C, act, g);
    else return ss_fail(-1, "ss_dwconv3_fwd: unsupported dtype %d", dtype);
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db: [10][C] fp32 (9 taps tap-major, then bias).  workspace: 1024 * 10 * C floats.
extern "C" int ss_dwconv3_bwd(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* g, float* dw_db,
                              int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && w9c && bias && g && dw_db && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_bwd: bad arguments");
    SS_CHECK(aligned16(x) && aligned16(dy) && aligned16(g) && aligned16(w9c) && aligned16(bias), "ss_dwconv3_bwd: pointers must be 16-byte aligned");
    DwGeom geo = dw_geom(H, W, C, 4);
    const int64_t ntiles = (int64_t)B * geo.xtiles * geo.ystrips;
    SS_CHECK(ntiles < 2147483647LL && geo.groups < 65536, "ss_dwconv3_bwd: grid too large");
    int64_t nbg = 1024 / geo.groups; if (nbg < 1) nbg = 1; if (nbg > ntiles) nbg = ntiles;
    SS_CHECK(ws_bytes >= nbg * 10 * C * 4, "ss_dwconv3_bwd: workspace too small");
    dim3 grid((unsigned)nbg, geo.groups);
    if (dtype == SS_BF16) hipLaunchKernelGGL((dwconv3_bwd_kernel<bf16, 4>), grid, dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, w9c, bias, (bf16*)g, workspace, B, H, W, C, act, geo);
    else if (dtype == SS_F32) hipLaunchKernelGGL((dwconv3_bwd_kernel<float, 4>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, w9c, bias, (float*)g, workspace, B, H, W, C, act, geo);
    else return ss_fail(-1, "ss_dwconv3_bwd: unsupported dtype %d", dtype);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, (int)nbg, 10 * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}
