// General depthwise k x k convolution on channels-last maps: ConvNeXt's 7x7 (reference modules/convnext.py:27,39) and the
// TopFormer token pyramid's 3x3 / 5x5, stride 1 / 2 with the asymmetric AutoPadding of modules/layers.py:146-164
// (modules/topformer.py:166-184,70-88).  Weights are tap-major [k*k][C] fp32.  A thread owns 8 consecutive channels of one
// output pixel; taps are 16-byte loads served by L1/L2 (neighbouring pixels share them).
//   forward  : y = act(dw(x) + b)            act: 0 none, 1 GELU(erf), 2 ReLU, 3 ReLU6
//   data grad: the forward kernel with flipped taps and mirrored padding (stride 1)
//   wgrad    : dW[ky][kx][c] = sum_p g[p][c] x[p + (ky, kx) - pad][c], db = sum_p g; one launch row per ky so a thread keeps
//              k x 8 accumulators; per-workgroup partials, fixed-order final reduce (deterministic).
#include "common.h"
#include "colreduce.h"

template <typename T>
__global__ __launch_bounds__(256) void dwconvk_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
        T* __restrict__ y, int B, int H, int W, int C, int k, int stride, int pt, int pl, int Ho, int Wo, int act) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * Ho * Wo * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int ox = (int)(p % Wo); int64_t q = p / Wo; const int oy = (int)(q % Ho); const int64_t b = q / Ho;
        float u[8];
        if (bias) { F8 t = load8(bias + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = t.v[e]; }
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f; }
        const int y0 = oy * stride - pt, x0 = ox * stride - pl;
        for (int ky = 0; ky < k; ++ky) {
            const int sy = y0 + ky;
            if (sy < 0 || sy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int sx = x0 + kx;
                if (sx < 0 || sx >= W) continue;
                F8 v = load8(x + ((b * H + sy) * W + sx) * C + c), ww = load8(w + (ky * k + kx) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] += v.v[e] * ww.v[e];
            }
        }
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o.v[e] = apply_act(u[e], act);
        }
        store8(y + p * C + c, o);
    }
}

// partial[blk][(k*k + 1)][C]: rows ky*k + kx, last row = sum g (written by the ky == 0 launch row)
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconvk_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ x, float* __restrict__ partial,
        int B, int H, int W, int C, int pad, int cpb, int rpb) {
    __shared__ float sh[256 * 8];
    const int t = threadIdx.x, ch = t % cpb, rl = t / cpb, nb = gridDim.x, ky = blockIdx.y;
    const int chunks = C / 8;
    const int64_t pixels = (int64_t)B * H * W;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {
        const int chunk = cbase + ch, c = chunk * 8;
        float acc[K + 1][8];
#pragma unroll
        for (int kk = 0; kk <= K; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[kk][e] = 0.f;
        if (rl < rpb && chunk < chunks)
            for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < pixels; p += (int64_t)nb * rpb) {
                int64_t q = p; const int xx = (int)(q % W); q /= W; const int yy = (int)(q % H); const int64_t img = q / H;
                F8 gv = load8(g + p * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[K][e] += gv.v[e];
                const int sy = yy + ky - pad;
                if (sy < 0 || sy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int sx = xx + kx - pad;
                    if (sx < 0 || sx >= W) continue;
                    F8 xv = load8(x + ((img * H + sy) * W + sx) * C + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[kx][e] += gv.v[e] * xv.v[e];
                }
            }
#pragma unroll
        for (int kk = 0; kk <= K; ++kk) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[t * 8 + e] = acc[kk][e];
            __syncthreads();
            if (rl == 0 && chunk < chunks && (kk < K || ky == 0)) {
                const int row = kk < K ? ky * K + kk : K * K;
                float* dst = partial + ((int64_t)blockIdx.x * (K * K + 1) + row) * C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) { float s = 0.f;
                    for (int j = 0; j < rpb; ++j) s += sh[(j * cpb + ch) * 8 + e];
                    dst[e] = s; }
            }
        }
    }
}

extern "C" int ss_dwconvk_fwd(int dtype, const void* x, const float* wkkc, const float* bias, void* y, int B, int H, int W, int C, int k,
                              int stride, int pad_top, int pad_left, int Ho, int Wo, int act, hipStream_t stream) {
    SS_CHECK(x && wkkc && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && k >= 1 && k <= 15 && stride >= 1 && Ho > 0 && Wo > 0,
             "ss_dwconvk_fwd: bad arguments (C=%d k=%d stride=%d)", C, k, stride);
    SS_CHECK(aligned16(x) && aligned16(y) && aligned16(wkkc) && (!bias || aligned16(bias)), "ss_dwconvk_fwd: pointers must be 16-byte aligned");
    SS_CHECK((int64_t)(Ho - 1) * stride - pad_top < H && (int64_t)(Wo - 1) * stride - pad_left < W, "ss_dwconvk_fwd: output larger than the padded input");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
    const int blocks = (int)(cdiv(total, 256) < 32768 ? cdiv(total, 256) : 32768);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dwconvk_fwd_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, wkkc, bias, (T*)y, B, H, W, C, k,
                                          stride, pad_top, pad_left, Ho, Wo, act));
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db: [(k*k + 1)][C] fp32 (taps tap-major, then the bias gradient); stride 1, symmetric padding; workspace >= 1024*(k*k+1)*C floats
extern "C" int ss_dwconvk_wgrad(int dtype, const void* g, const void* x, float* dw_db, int B, int H, int W, int C, int k, int pad,
                                float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(g && x && dw_db && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (k == 3 || k == 5 || k == 7),
             "ss_dwconvk_wgrad: bad arguments (C=%d k=%d; k must be 3, 5 or 7)", C, k);
    SS_CHECK(aligned16(x) && aligned16(g), "ss_dwconvk_wgrad: pointers must be 16-byte aligned");
    ColGeom geo = col_geom((int64_t)B * H * W, C);
    const int rows = k * k + 1;
    SS_CHECK(ws_bytes >= (int64_t)geo.nb * rows * C * 4, "ss_dwconvk_wgrad: workspace too small");
    dim3 grid(geo.nb, k);
    SS_DISPATCH(dtype, {
        if (k == 3) hipLaunchKernelGGL((dwconvk_wgrad_kernel<T, 3>), grid, dim3(256), 0, stream, (const T*)g, (const T*)x, workspace, B, H, W, C, pad, geo.cpb, geo.rpb);
        else if (k == 5) hipLaunchKernelGGL((dwconvk_wgrad_kernel<T, 5>), grid, dim3(256), 0, stream, (const T*)g, (const T*)x, workspace, B, H, W, C, pad, geo.cpb, geo.rpb);
        else hipLaunchKernelGGL((dwconvk_wgrad_kernel<T, 7>), grid, dim3(256), 0, stream, (const T*)g, (const T*)x, workspace, B, H, W, C, pad, geo.cpb, geo.rpb);
    });
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(rows * C, 8), 1), dim3(256), 0, stream, workspace, geo.nb, rows * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// out[r] = sum_c a[r][c] * b[r][c]   (fp32; e.g. the layer-scale gradient of modules/convnext.py:46-47 from the folded weight gradient)
__global__ __launch_bounds__(256) void rowdot_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                         int rows, int C) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += a[(int64_t)wave * C + c] * b[(int64_t)wave * C + c];
    s = wave_sum(s);
    if (lane == 0) out[wave] = s;
}
extern "C" int ss_rowdot_f32(const float* a, const float* b, float* out, int rows, int C, hipStream_t stream) {
    SS_CHECK(a && b && out && rows > 0 && C > 0, "ss_rowdot_f32: bad arguments");
    hipLaunchKernelGGL(rowdot_f32_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, a, b, out, rows, C);
    SS_LAUNCH_CHECK();
    return 0;
}
