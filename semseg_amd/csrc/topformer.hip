// TopFormer token-pyramid pieces that are not GEMMs / depthwise convs (reference modules/topformer.py, modules/heads.py,
// core/models.py).  All bandwidth-bound, channels-last, 16-byte accesses where the channel count allows.
//   im2col_ex      3x3 / s2 stem patches with the ASYMMETRIC AutoPadding of modules/layers.py:146-164 (pad_top/left + explicit Ho, Wo)
//   avgpool        adaptive_avg_pool2d with exact windows (PyramidPoolAgg, modules/topformer.py:527-552) into a concat slice
//   act            stand-alone activation (the 'ACN' projection of the attention block applies ReLU6 BEFORE the conv, :394-409)
//   inject_fuse    InjectionDotSum.forward (:640-673): local * bilinear(act) + bilinear(global), both bilinear maps on the fly
//   bilinear_add   y += bilinear(x) (UniRes + Sum fusion of modules/heads.py:60-106,120-170)
#include "common.h"

struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap tap_of(int o, float scale, int in) {        // torch bilinear, align_corners = False
    float src = scale * (o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Tap t; t.i0 = (int)src; if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = src - t.i0; t.l0 = 1.f - t.l1;
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void im2col_ex_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C, int k, int s,
                                                        int pt, int pl, int ho, int wo, int Kp) {
    const int64_t total = (int64_t)B * ho * wo * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % Kp); int64_t p = i / Kp;
        float v = 0.f;
        if (kc < k * k * C) {
            const int c = kc % C, tap = kc / C;
            const int ox = (int)(p % wo); int64_t q = p / wo; const int oy = (int)(q % ho); const int64_t b = q / ho;
            const int iy = oy * s - pt + tap / k, ix = ox * s - pl + tap % k;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = to_f(x[((b * H + iy) * W + ix) * C + c]);
        }
        col[i] = from_f<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int B, int H, int W,
                                                      int C, int win) {
    const int chunks = C / 8, ho = H / win, wo = W / win;
    const int64_t total = (int64_t)B * ho * wo * chunks;
    const float inv = 1.f / (win * win);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int ox = (int)(p % wo); int64_t q = p / wo; const int oy = (int)(q % ho); const int64_t b = q / ho;
        F8 acc = zero8();
        for (int dy = 0; dy < win; ++dy)
            for (int dx = 0; dx < win; ++dx) {
                F8 v = load8(x + ((b * H + oy * win + dy) * W + ox * win + dx) * ldx + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc.v[e] *= inv;
        store8(y + p * ldy + c, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n8, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        F8 v = load8(x + i * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = apply_act(v.v[e], act);
        store8(y + i * 8, o);
    }
}

template <typename T>
__device__ __forceinline__ F8 bilerp8(const T* __restrict__ x, int64_t ld, int64_t b, int hi, int wi, const Tap& ty, const Tap& tx, int c) {
    const T* r0 = x + ((b * hi + ty.i0) * wi) * ld + c;
    const T* r1 = x + ((b * hi + ty.i1) * wi) * ld + c;
    F8 a = load8(r0 + tx.i0 * ld), bq = load8(r0 + tx.i1 * ld), cq = load8(r1 + tx.i0 * ld), d = load8(r1 + tx.i1 * ld), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = ty.l0 * (tx.l0 * a.v[e] + tx.l1 * bq.v[e]) + ty.l1 * (tx.l0 * cq.v[e] + tx.l1 * d.v[e]);
    return o;
}

// out = local * up(ga) + up(ge);  local/out at (hl, wl), ga/ge at (hg, wg)
template <typename T>
__global__ __launch_bounds__(256) void inject_fuse_kernel(const T* __restrict__ local, int64_t ldl, const T* __restrict__ ga, int64_t lda,
        const T* __restrict__ ge, int64_t lde, T* __restrict__ out, int64_t ldo, int B, int hl, int wl, int hg, int wg, int C) {
    const int chunks = C / 8;
    const float sh = (float)hg / hl, sw = (float)wg / wl;
    const int64_t total = (int64_t)B * hl * wl * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int ox = (int)(p % wl); int64_t q = p / wl; const int oy = (int)(q % hl); const int64_t b = q / hl;
        const Tap ty = tap_of(oy, sh, hg), tx = tap_of(ox, sw, wg);
        F8 l = load8(local + p * ldl + c), a = bilerp8(ga, lda, b, hg, wg, ty, tx, c), g = bilerp8(ge, lde, b, hg, wg, ty, tx, c), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = l.v[e] * a.v[e] + g.v[e];
        store8(out + p * ldo + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_add_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int B, int hi,
                                                           int wi, int ho, int wo, int C) {
    const int chunks = C / 8;
    const float sh = (float)hi / ho, sw = (float)wi / wo;
    const int64_t total = (int64_t)B * ho * wo * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int ox = (int)(p % wo); int64_t q = p / wo; const int oy = (int)(q % ho); const int64_t b = q / ho;
        const Tap ty = tap_of(oy, sh, hi), tx = tap_of(ox, sw, wi);
        F8 u = bilerp8(x, ldx, b, hi, wi, ty, tx, c), v = load8(y + p * ldy + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] += u.v[e];
        store8(y + p * ldy + c, v);
    }
}

static inline int tf_blocks(int64_t n, int cap = 32768) { int64_t b = cdiv(n, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }
static inline bool ok8(const void* p, int64_t ld) { return aligned16(p) && ld % 8 == 0; }

extern "C" int ss_im2col_ex(int dtype, const void* x, void* col, int B, int H, int W, int C, int k, int s, int pad_top, int pad_left, int Ho,
                            int Wo, int Kp, hipStream_t stream) {
    SS_CHECK(x && col && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && Ho > 0 && Wo > 0 && Kp >= k * k * C, "ss_im2col_ex: bad arguments");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((im2col_ex_kernel<T>), dim3(tf_blocks((int64_t)B * Ho * Wo * Kp)), dim3(256), 0, stream, (const T*)x,
                                          (T*)col, B, H, W, C, k, s, pad_top, pad_left, Ho, Wo, Kp));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_avgpool(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, int win, hipStream_t stream) {
    SS_CHECK(x && y && B > 0 && win > 0 && H % win == 0 && W % win == 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ok8(x, ldx) && ok8(y, ldy),
             "ss_avgpool: need H, W multiples of the window, C %% 8 == 0, 16-byte aligned rows (H=%d W=%d win=%d C=%d)", H, W, win, C);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((avgpool_kernel<T>), dim3(tf_blocks((int64_t)B * (H / win) * (W / win) * (C / 8))), dim3(256), 0, stream,
                                          (const T*)x, ldx, (T*)y, ldy, B, H, W, C, win));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_act(int dtype, const void* x, void* y, int64_t n, int act, hipStream_t stream) {
    SS_CHECK(x && y && n > 0 && n % 8 == 0 && aligned16(x) && aligned16(y), "ss_act: need n %% 8 == 0 and 16-byte aligned pointers");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((act_kernel<T>), dim3(tf_blocks(n / 8, 16384)), dim3(256), 0, stream, (const T*)x, (T*)y, n / 8, act));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_inject_fuse(int dtype, const void* local, int64_t ldl, const void* gact, int64_t lda, const void* gemb, int64_t lde, void* out,
                              int64_t ldo, int B, int hl, int wl, int hg, int wg, int C, hipStream_t stream) {
    SS_CHECK(local && gact && gemb && out && B > 0 && hl > 0 && wl > 0 && hg > 0 && wg > 0 && C > 0 && C % 8 == 0, "ss_inject_fuse: bad arguments");
    SS_CHECK(ok8(local, ldl) && ok8(gact, lda) && ok8(gemb, lde) && ok8(out, ldo) && ldl >= C && lda >= C && lde >= C && ldo >= C,
             "ss_inject_fuse: rows must be 16-byte aligned");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((inject_fuse_kernel<T>), dim3(tf_blocks((int64_t)B * hl * wl * (C / 8))), dim3(256), 0, stream,
                                          (const T*)local, ldl, (const T*)gact, lda, (const T*)gemb, lde, (T*)out, ldo, B, hl, wl, hg, wg, C));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_bilinear_add(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                               hipStream_t stream) {
    SS_CHECK(x && y && B > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ok8(x, ldx) && ok8(y, ldy),
             "ss_bilinear_add: bad arguments");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((bilinear_add_kernel<T>), dim3(tf_blocks((int64_t)B * ho * wo * (C / 8))), dim3(256), 0, stream,
                                          (const T*)x, ldx, (T*)y, ldy, B, hi, wi, ho, wo, C));
    SS_LAUNCH_CHECK();
    return 0;
}
