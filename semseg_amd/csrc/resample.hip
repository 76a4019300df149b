// Layout / resampling kernels on channels-last tensors: im2col + col2im for the overlapped PatchEmbed convolutions and
// the non-overlapping spatial-reduction conv, bilinear resize forward/backward writing into channel
// slices of concat buffers (K10), NCHW image -> NHWC, weight packing / casting, small elementwise helpers.
// All bandwidth-bound: 16-byte channel-contiguous accesses wherever C % 8 == 0, scalar fall-backs for C = 3 / 6.
#include "common.h"
#include "bilinear.h"

// ------------------------------------------------------------------ im2col / col2im   (column order: ky, kx, c)
// grid: (B*ho output rows, blocks over the wo * k*k * C/8 chunk items of a row)
template <typename T>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const T* __restrict__ x, T* __restrict__ col, int H, int W, int C,
                                                         int k, int s, int pad, int ho, int wo, FastDiv dchunks, FastDiv dper, FastDiv dk) {
    const unsigned chunks = dchunks.d, per = dper.d;                 // per = k*k*chunks items per output pixel
    const unsigned idx = blockIdx.y * 256u + threadIdx.x;
    if (idx >= (unsigned)wo * per) return;
    const unsigned row = blockIdx.x, b = row / (unsigned)ho, oy = row - b * ho;
    const unsigned ox = fdiv(idx, dper), rem = idx - ox * per, tap = fdiv(rem, dchunks), c = (rem - tap * chunks) * 8;
    const unsigned ky = fdiv(tap, dk), kx = tap - ky * k;
    const int iy = (int)(oy * s + ky) - pad, ix = (int)(ox * s + kx) - pad;
    F8 v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? load8(x + (((int64_t)b * H + iy) * W + ix) * C + c) : zero8();
    store8(col + ((int64_t)row * wo + ox) * ((int64_t)k * k * C) + (int64_t)tap * C + c, v);
}
template <typename T>
__global__ __launch_bounds__(256) void im2col_scalar_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C,
                                                            int k, int s, int pad, int ho, int wo, int Kp) {
    const int64_t total = (int64_t)B * ho * wo * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % Kp); int64_t p = i / Kp;
        float v = 0.f;
        if (kc < k * k * C) {
            const int c = kc % C, tap = kc / C;
            const int ox = (int)(p % wo); int64_t q = p / wo; const int oy = (int)(q % ho); const int64_t b = q / ho;
            const int iy = oy * s - pad + tap / k, ix = ox * s - pad + tap % k;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = to_f(x[((b * H + iy) * W + ix) * C + c]);
        }
        col[i] = from_f<T>(v);
    }
}
// narrow inputs (the 3-channel image under patch_embed1's 7x7 / 4): a thread gathers 8 consecutive k*k*C entries of one output
// pixel (2- or 4-byte loads, served by L1: neighbouring taps share lines) and writes them as one 16-byte store.
// grid: (B*ho output rows, blocks over wo * Kp/8)
template <typename T>
__global__ __launch_bounds__(256) void im2col_narrow_kernel(const T* __restrict__ x, T* __restrict__ col, int H, int W, int C,
                                                            int k, int s, int pad, int ho, int wo, int Kp, FastDiv dchunks, FastDiv dC, FastDiv dk) {
    const unsigned chunks = dchunks.d;
    const unsigned idx = blockIdx.y * 256u + threadIdx.x;
    if (idx >= (unsigned)wo * chunks) return;
    const unsigned row = blockIdx.x, b = row / (unsigned)ho, oy = row - b * ho;
    const unsigned ox = fdiv(idx, dchunks), ch = idx - ox * chunks;
    const int kkC = k * k * C;
    const int y0 = (int)oy * s - pad, x0 = (int)ox * s - pad;
    const T* img = x + (int64_t)b * H * W * C;
    F8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned kc = ch * 8 + e;
        const unsigned tap = fdiv(kc, dC), c = kc - tap * C;
        const unsigned ky = fdiv(tap, dk), kx = tap - ky * k;
        const int iy = y0 + (int)ky, ix = x0 + (int)kx;
        const bool ok = (int)kc < kkC && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int64_t off = ok ? ((int64_t)iy * W + ix) * C + c : 0;
        const float f = to_f(img[off]);
        v.v[e] = ok ? f : 0.f;
    }
    store8(col + ((int64_t)row * wo + ox) * Kp + ch * 8, v);
}
// gather form: every input pixel sums the taps that read it (deterministic).  grid: (B*H input rows, blocks over W * C/8)
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ col, T* __restrict__ dx, int H, int W, int C,
                                                     int k, int s, int pad, int ho, int wo, FastDiv dchunks, FastDiv ds) {
    const unsigned chunks = dchunks.d;
    const unsigned idx = blockIdx.y * 256u + threadIdx.x;
    if (idx >= (unsigned)W * chunks) return;
    const unsigned row = blockIdx.x, b = row / (unsigned)H, iy = row - b * H;
    const unsigned ix = fdiv(idx, dchunks), c = (idx - ix * chunks) * 8;
    const int64_t kkC = (int64_t)k * k * C;
    if (k == s && pad == 0) {                              // non-overlapping patches (the MiT spatial-reduction conv): a pure permutation
        const unsigned oy = fdiv(iy, ds), ox = fdiv(ix, ds), ky = iy - oy * s, kx = ix - ox * s;
        F8 v = zero8();
        if (oy < (unsigned)ho && ox < (unsigned)wo) v = load8(col + (((int64_t)b * ho + oy) * wo + ox) * kkC + (int64_t)(ky * k + kx) * C + c);
        store8(dx + ((int64_t)row * W + ix) * C + c, v);
        return;
    }
    F8 acc = zero8();
    for (int ky = 0; ky < k; ++ky) {
        const int ty = (int)iy + pad - ky;
        if (ty < 0) break;
        const unsigned oy = fdiv((unsigned)ty, ds);
        if (oy * s != (unsigned)ty || oy >= (unsigned)ho) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = (int)ix + pad - kx;
            if (tx < 0) break;
            const unsigned ox = fdiv((unsigned)tx, ds);
            if (ox * s != (unsigned)tx || ox >= (unsigned)wo) continue;
            F8 v = load8(col + (((int64_t)b * ho + oy) * wo + ox) * kkC + (int64_t)(ky * k + kx) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e];
        }
    }
    store8(dx + ((int64_t)row * W + ix) * C + c, acc);
}

extern "C" int ss_im2col(int dtype, const void* x, void* col, int B, int H, int W, int C, int k, int s, int pad, int Kp, hipStream_t stream) {
    SS_CHECK(x && col && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && pad >= 0, "ss_im2col: bad arguments");
    const int ho = (H + 2 * pad - k) / s + 1, wo = (W + 2 * pad - k) / s + 1;
    SS_CHECK(ho > 0 && wo > 0 && Kp >= k * k * C, "ss_im2col: bad geometry");
    if (C % 8 == 0) {
        SS_CHECK(Kp == k * k * C, "ss_im2col: Kp must equal k*k*C when C is a multiple of 8");
        const int64_t per_row = (int64_t)wo * k * k * (C / 8);
        SS_CHECK(per_row < (1LL << 31) && cdiv(per_row, 256) < 65536 && (int64_t)B * ho < (1LL << 31), "ss_im2col: row too long");
        dim3 grid((unsigned)(B * ho), (unsigned)cdiv(per_row, 256));
        SS_DISPATCH(dtype, hipLaunchKernelGGL((im2col_vec_kernel<T>), grid, dim3(256), 0, stream, (const T*)x, (T*)col, H, W, C, k, s, pad, ho, wo,
                                              fast_div(C / 8), fast_div(k * k * (C / 8)), fast_div(k)));
    } else if (Kp % 8 == 0 && aligned16(col) && (int64_t)wo * (Kp / 8) < (1LL << 31) && cdiv((int64_t)wo * (Kp / 8), 256) < 65536 &&
               (int64_t)B * ho < (1LL << 31)) {
        dim3 grid((unsigned)(B * ho), (unsigned)cdiv((int64_t)wo * (Kp / 8), 256));
        SS_DISPATCH(dtype, hipLaunchKernelGGL((im2col_narrow_kernel<T>), grid, dim3(256), 0, stream, (const T*)x, (T*)col, H, W, C, k, s, pad, ho, wo, Kp,
                                              fast_div(Kp / 8), fast_div(C), fast_div(k)));
    } else {
        int64_t total = (int64_t)B * ho * wo * Kp;
        int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
        SS_DISPATCH(dtype, hipLaunchKernelGGL((im2col_scalar_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, (T*)col, B, H, W, C, k, s, pad, ho, wo, Kp));
    }
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_col2im(int dtype, const void* col, void* dx, int B, int H, int W, int C, int k, int s, int pad, hipStream_t stream) {
    SS_CHECK(col && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && k > 0 && s > 0 && pad >= 0, "ss_col2im: bad arguments (C=%d)", C);
    const int ho = (H + 2 * pad - k) / s + 1, wo = (W + 2 * pad - k) / s + 1;
    SS_CHECK(ho > 0 && wo > 0, "ss_col2im: bad geometry");
    const int64_t per_row = (int64_t)W * (C / 8);
    SS_CHECK(per_row < (1LL << 31) && cdiv(per_row, 256) < 65536 && (int64_t)B * H < (1LL << 31), "ss_col2im: row too long");
    dim3 grid((unsigned)(B * H), (unsigned)cdiv(per_row, 256));
    SS_DISPATCH(dtype, hipLaunchKernelGGL((col2im_kernel<T>), grid, dim3(256), 0, stream, (const T*)col, (T*)dx, H, W, C, k, s, pad, ho, wo,
                                          fast_div(C / 8), fast_div(s)));
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ bilinear (align_corners = False), torch semantics
// grid: (B*ho output rows, blocks over the wo * chunks items of a row); the vertical tap is uniform per workgroup
// (x2: an optional second source of its own size; the two interpolated values are added in fp32 and rounded once)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
        int hi, int wi, int ho, int wo, FastDiv dchunks, int accumulate, const T* __restrict__ x2 = nullptr, int64_t ldx2 = 0,
        int hi2 = 0, int wi2 = 0) {
    constexpr int W8 = VEC ? 8 : 1;
    const unsigned chunks = dchunks.d;
    const unsigned idx = blockIdx.y * 256u + threadIdx.x;
    if (idx >= (unsigned)wo * chunks) return;
    const unsigned row = blockIdx.x, b = row / (unsigned)ho, oy = row - b * ho;
    const unsigned ox = fdiv(idx, dchunks), c = (idx - ox * chunks) * W8;
    const LinTap ty = lin_tap((int)oy, (float)hi / ho, hi), tx = lin_tap((int)ox, (float)wi / wo, wi);
    const T* r0 = x + (((int64_t)b * hi + ty.i0) * wi) * ldx + c;
    const T* r1 = x + (((int64_t)b * hi + ty.i1) * wi) * ldx + c;
    T* dst = y + ((int64_t)row * wo + ox) * ldy + c;
    if (VEC) {
        F8 a = load8(r0 + tx.i0 * ldx), bq = load8(r0 + tx.i1 * ldx), cq = load8(r1 + tx.i0 * ldx), d = load8(r1 + tx.i1 * ldx), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = bilerp(ty, tx, a.v[e], bq.v[e], cq.v[e], d.v[e]);
        if (x2) {
            const LinTap uy = lin_tap((int)oy, (float)hi2 / ho, hi2), ux = lin_tap((int)ox, (float)wi2 / wo, wi2);
            const T* s0 = x2 + (((int64_t)b * hi2 + uy.i0) * wi2) * ldx2 + c;
            const T* s1 = x2 + (((int64_t)b * hi2 + uy.i1) * wi2) * ldx2 + c;
            const F8 a2 = load8(s0 + ux.i0 * ldx2), b2 = load8(s0 + ux.i1 * ldx2), c2 = load8(s1 + ux.i0 * ldx2), d2 = load8(s1 + ux.i1 * ldx2);
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] += bilerp(uy, ux, a2.v[e], b2.v[e], c2.v[e], d2.v[e]);
        }
        if (accumulate) { const F8 p = load8(dst);
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] += p.v[e]; }
        store8(dst, o);
    } else {
        float o = bilerp(ty, tx, to_f(r0[tx.i0 * ldx]), to_f(r0[tx.i1 * ldx]), to_f(r1[tx.i0 * ldx]), to_f(r1[tx.i1 * ldx]));
        dst[0] = from_f<T>(accumulate ? o + to_f(dst[0]) : o);
    }
}

// Narrow tensors (the nc-class logits): a thread computes ALL channels of one output pixel into an LDS image of the
// workgroup's 256 pixels, which then leaves as one contiguous run (dense rows: ldy == C).  grid: (B*ho rows, blocks of 256 px)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_narrow_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
        int hi, int wi, int ho, int wo, int C) {
    extern __shared__ __attribute__((aligned(16))) char nsm[];
    T* img = (T*)nsm;
    const unsigned row = blockIdx.x, b = row / (unsigned)ho, oy = row - b * ho;
    const unsigned ox0 = blockIdx.y * 256u, ox = ox0 + threadIdx.x;
    const unsigned npx = min(256u, (unsigned)wo - ox0);
    if (ox < (unsigned)wo) {
        const LinTap ty = lin_tap((int)oy, (float)hi / ho, hi), tx = lin_tap((int)ox, (float)wi / wo, wi);
        const T* r0 = x + (((int64_t)b * hi + ty.i0) * wi) * ldx;
        const T* r1 = x + (((int64_t)b * hi + ty.i1) * wi) * ldx;
        const T *p00 = r0 + tx.i0 * ldx, *p01 = r0 + tx.i1 * ldx, *p10 = r1 + tx.i0 * ldx, *p11 = r1 + tx.i1 * ldx;
        for (int c = 0; c < C; ++c) {
            const float o = bilerp(ty, tx, to_f(p00[c]), to_f(p01[c]), to_f(p10[c]), to_f(p11[c]));
            img[threadIdx.x * C + c] = from_f<T>(o);
        }
    }
    __syncthreads();
    T* dst = y + ((int64_t)row * wo + ox0) * C;
    const unsigned bytes = npx * C * (unsigned)sizeof(T);
    if ((((uintptr_t)dst | bytes) & 15) == 0) {
        for (unsigned i = threadIdx.x * 16u; i < bytes; i += 256u * 16u) *(floatx4*)((char*)dst + i) = *(const floatx4*)(nsm + i);
    } else {
        for (unsigned i = threadIdx.x; i < npx * C; i += 256u) dst[i] = img[i];
    }
}

// Prediction writer (helper/callbacks.py:52-54) on logits that are still at the head's resolution: class map of the resized
// logits without ever writing them.  out[b, oy, ox] = first arg-max over c of the value ss_bilinear_fwd would have stored
// (same taps, same roundings, rounded to T), a pixel holding a NaN -> 0 (see class_map_kernel in loss.hip for the rule).
// A thread owns one output pixel (VEC: the channels of a tap are one 16-byte load), 4 neighbouring lanes share a 4-byte store.
// grid: (B*ho rows, blocks of 256 px)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bilinear_class_map_kernel(const T* __restrict__ x, int64_t ldx, unsigned char* __restrict__ out,
        int hi, int wi, int ho, int wo, int C) {
    const unsigned row = blockIdx.x, b = row / (unsigned)ho, oy = row - b * ho;
    const unsigned ox0 = blockIdx.y * 256u, oxr = ox0 + threadIdx.x, ox = min(oxr, (unsigned)wo - 1);
    const LinTap ty = lin_tap((int)oy, (float)hi / ho, hi), tx = lin_tap((int)ox, (float)wi / wo, wi);
    const T* r0 = x + (((int64_t)b * hi + ty.i0) * wi) * ldx;
    const T* r1 = x + (((int64_t)b * hi + ty.i1) * wi) * ldx;
    const T *p00 = r0 + tx.i0 * ldx, *p01 = r0 + tx.i1 * ldx, *p10 = r1 + tx.i0 * ldx, *p11 = r1 + tx.i1 * ldx;
    float best = 0.f; int arg = 0; bool nan = false;
    if (VEC) {                                               // bf16, C <= 8, 16-byte aligned rows
        const F8 a = load8(p00), bq = load8(p01), cq = load8(p10), d = load8(p11);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = to_f(from_f<T>(bilerp(ty, tx, a.v[c], bq.v[c], cq.v[c], d.v[c])));
            if (c < C) { nan |= v != v; if (c == 0 || v > best) { best = v; arg = c; } }
        }
    } else {
        for (int c = 0; c < C; ++c) {
            const float v = to_f(from_f<T>(bilerp(ty, tx, to_f(p00[c]), to_f(p01[c]), to_f(p10[c]), to_f(p11[c]))));
            nan |= v != v;
            if (c == 0 || v > best) { best = v; arg = c; }
        }
    }
    const unsigned mine = nan ? 0u : (unsigned)arg;
    const unsigned packed = mine | (__shfl_down(mine, 1) << 8) | (__shfl_down(mine, 2) << 16) | (__shfl_down(mine, 3) << 24);
    unsigned char* dst = out + (int64_t)row * wo + oxr;
    if ((wo & 3) == 0 && ((uintptr_t)out & 3) == 0) { if ((threadIdx.x & 3) == 0 && oxr < (unsigned)wo) *(unsigned*)dst = packed; }
    else if (oxr < (unsigned)wo) *dst = (unsigned char)mine;
}

// gather-form backward: dx[b,iy,ix,:] = sum over the output pixels whose taps touch (iy,ix).
// grid: (B*hi input rows, blocks over wi * chunks); the output-row range and its weights are uniform per workgroup.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, int64_t ldy, T* __restrict__ dx, int64_t ldx,
        int hi, int wi, int ho, int wo, FastDiv dchunks, int accumulate) {
    constexpr int W8 = VEC ? 8 : 1;
    const unsigned chunks = dchunks.d;
    const unsigned idx = blockIdx.y * 256u + threadIdx.x;
    if (idx >= (unsigned)wi * chunks) return;
    const unsigned row = blockIdx.x, b = row / (unsigned)hi;
    const int iy = (int)(row - b * hi);
    const int ix = (int)fdiv(idx, dchunks), c = (int)(idx - (unsigned)ix * chunks) * W8;
    const float sh = (float)hi / ho, sw = (float)wi / wo;
    int oy0 = (int)floorf((iy - 0.5f) / sh - 0.5f) - 1, oy1 = (int)ceilf((iy + 1.5f) / sh - 0.5f) + 1;
    int ox0 = (int)floorf((ix - 0.5f) / sw - 0.5f) - 1, ox1 = (int)ceilf((ix + 1.5f) / sw - 0.5f) + 1;
    oy0 = max(oy0, 0); ox0 = max(ox0, 0); oy1 = min(oy1, ho - 1); ox1 = min(ox1, wo - 1);
    float acc[W8];
#pragma unroll
    for (int e = 0; e < W8; ++e) acc[e] = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const LinTap ty = lin_tap(oy, sh, hi);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        const T* srow = dy + (((int64_t)b * ho + oy) * wo) * ldy + c;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const LinTap tx = lin_tap(ox, sw, wi);
            const float wx = (tx.i0 == ix ? tx.l0 : 0.f) + (tx.i1 == ix ? tx.l1 : 0.f);
            if (wx == 0.f) continue;
            const T* src = srow + (int64_t)ox * ldy;
            if (VEC) { F8 v = load8(src);
#pragma unroll
                for (int e = 0; e < W8; ++e) acc[e] += wy * wx * v.v[e]; }
            else acc[0] += wy * wx * to_f(src[0]);
        }
    }
    T* dst = dx + ((int64_t)row * wi + ix) * ldx + c;
    if (VEC) { F8 o;
        if (accumulate) { o = load8(dst);
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] += acc[e]; }
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = acc[e]; }
        store8(dst, o);
    } else dst[0] = from_f<T>(accumulate ? to_f(dst[0]) + acc[0] : acc[0]);
}

// Narrow tensors (the nc-class logits, C <= 8) up-sampled by an integer factor S.  A workgroup owns 256 input pixels of one input row.
// The transposed resize is separable: (1) the 2S output rows that touch the input row are summed vertically with their (workgroup-
// uniform) weights, read as whole contiguous row segments (a thread takes consecutive 4-byte units: bf16 channel pairs, or single
// elements), into an LDS image v[channel][output column]; (2) a thread gathers the 2S horizontal taps of ITS input pixel from LDS.
// Output ox = S*ix + j, j in [-S/2, 3S/2): its source coordinate sits t = (j + 0.5)/S - 0.5 from ix and weighs 1 - |t|; at the
// borders the clamped taps of torch's formula fold the overhang onto the edge pixel (weight 1).  (The first form of this kernel
// gathered all 2S x 2S taps per thread straight from memory: 48-byte lane stride, 24 cache lines per load instruction, 342 us for
// the 16 x 1024^2 x 6 logit gradient; a dy row is now read twice, by the workgroups of two neighbouring input rows on the same XCD.)
// grid: (B*hi rows, blocks of 256 px)
template <typename T, int S>
__global__ __launch_bounds__(256) void bilinear_bwd_narrow_kernel(const T* __restrict__ dy, int64_t ldy, T* __restrict__ dx, int64_t ldx,
        int hi, int wi, int C, int accumulate, int pairs) {
    constexpr int NL = S * 256 + S, NLP = NL + 4;                // output columns per workgroup; padded channel pitch
    __shared__ float v[8 * NLP];
    // rows of one XCD (blocks b, b + 8, ...) are a contiguous run, so both readers of an output row share an L2
    const unsigned nrow = gridDim.x, q8 = nrow >> 3, r8 = nrow & 7, xcd = blockIdx.x & 7;
    const unsigned row = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const unsigned b = row / (unsigned)hi;
    const int iy = (int)(row - b * hi);
    const int ix0 = (int)(blockIdx.y * 256u), tid = threadIdx.x;
    const int ho = hi * S, wo = wi * S;
    auto weight = [](int j, int i, int n) {              // tap j of input index i on an axis of n inputs
        const float t = (j + 0.5f) / S - 0.5f;
        float w = 1.f - fabsf(t);
        if ((i == 0 && t < 0.f) || (i == n - 1 && t > 0.f)) w = 1.f;
        const int o = S * i + j;
        return (o >= 0 && o < n * S) ? w : 0.f;
    };
    float wy[2 * S];
    const T* srow[2 * S];
#pragma unroll
    for (int k = 0; k < 2 * S; ++k) {
        wy[k] = weight(k - S / 2, iy, hi);
        srow[k] = dy + (((int64_t)b * ho + min(max(S * iy + k - S / 2, 0), ho - 1)) * wo) * ldy;
    }
    const int o0 = S * ix0 - S / 2;                                  // output column of LDS column 0
    const int U = pairs ? C / 2 : C;                                 // 4-byte units (bf16 pairs) or elements per pixel
    for (int u0 = 0; u0 < NL * U; u0 += 256) {
        const int u = u0 + tid;
        if (u >= NL * U) break;
        const int l = u / U, e = u - l * U;
        const int64_t off = (int64_t)min(max(o0 + l, 0), wo - 1) * ldy;
        if (pairs) {
            unsigned raw[2 * S];
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) raw[k] = ((const unsigned*)(srow[k] + off))[e];
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) {
                a0 = fmaf(wy[k], __uint_as_float(raw[k] << 16), a0);
                a1 = fmaf(wy[k], __uint_as_float(raw[k] & 0xffff0000u), a1);
            }
            v[(2 * e) * NLP + l] = a0; v[(2 * e + 1) * NLP + l] = a1;
        } else {
            float raw[2 * S];
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) raw[k] = to_f(srow[k][off + e]);
            float a0 = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) a0 = fmaf(wy[k], raw[k], a0);
            v[e * NLP + l] = a0;
        }
    }
    __syncthreads();
    const int ix = ix0 + tid;
    if (ix >= wi) return;
    float wx[2 * S];
#pragma unroll
    for (int k = 0; k < 2 * S; ++k) wx[k] = weight(k - S / 2, ix, wi);
    T* dst = dx + ((int64_t)row * wi + ix) * ldx;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < C) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * S; ++k) a = fmaf(wx[k], v[e * NLP + S * tid + k], a);
            dst[e] = from_f<T>(accumulate ? to_f(dst[e]) + a : a);
        }
}

static int bilinear_fwd_impl(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                             int accumulate, hipStream_t stream) {
    SS_CHECK(x && y && B > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && C > 0 && ldx >= C && ldy >= C, "ss_bilinear_fwd: bad arguments");
    const bool vec = C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y);
    const int chunks = vec ? C / 8 : C;
    const int64_t per_row = (int64_t)wo * chunks;
    SS_CHECK(per_row < (1LL << 31) && cdiv(per_row, 256) < 65536 && (int64_t)B * ho < (1LL << 31), "ss_bilinear_fwd: row too long");
    const int esz = dtype == SS_BF16 ? 2 : 4;
    if (!vec && ldy == C && C <= 32 && !accumulate) {
        dim3 grid((unsigned)(B * ho), (unsigned)cdiv(wo, 256));
        SS_DISPATCH(dtype, hipLaunchKernelGGL((bilinear_fwd_narrow_kernel<T>), grid, dim3(256), 256 * C * esz, stream, (const T*)x, ldx, (T*)y, hi, wi,
                                              ho, wo, C));
        SS_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)(B * ho), (unsigned)cdiv(per_row, 256));
    SS_DISPATCH(dtype, {
        if (vec) hipLaunchKernelGGL((bilinear_fwd_kernel<T, true>), grid, dim3(256), 0, stream, (const T*)x, ldx, (T*)y, ldy, hi, wi, ho, wo, fast_div(chunks), accumulate);
        else hipLaunchKernelGGL((bilinear_fwd_kernel<T, false>), grid, dim3(256), 0, stream, (const T*)x, ldx, (T*)y, ldy, hi, wi, ho, wo, fast_div(chunks), accumulate);
    });
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_bilinear_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                               hipStream_t stream) {
    return bilinear_fwd_impl(dtype, x, ldx, y, ldy, B, hi, wi, ho, wo, C, 0, stream);
}
// y += resize(x): the interpolated value is added in fp32 to what y holds, one rounding
extern "C" int ss_bilinear_fwd_add(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                                   hipStream_t stream) {
    return bilinear_fwd_impl(dtype, x, ldx, y, ldy, B, hi, wi, ho, wo, C, 1, stream);
}
// y = resize(x) + resize(x2): two maps of different sizes brought to one size and summed in one pass (one rounding)
extern "C" int ss_bilinear_fwd2(int dtype, const void* x, int64_t ldx, int hi, int wi, const void* x2, int64_t ldx2, int hi2, int wi2,
                                void* y, int64_t ldy, int B, int ho, int wo, int C, hipStream_t stream) {
    SS_CHECK(x && x2 && y && B > 0 && hi > 0 && wi > 0 && hi2 > 0 && wi2 > 0 && ho > 0 && wo > 0 && C > 0, "ss_bilinear_fwd2: bad arguments");
    SS_CHECK(C % 8 == 0 && ldx >= C && ldx2 >= C && ldy >= C && ldx % 8 == 0 && ldx2 % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(x2) && aligned16(y),
             "ss_bilinear_fwd2: C and the leading dimensions must be multiples of 8, pointers 16-byte aligned");
    const int64_t per_row = (int64_t)wo * (C / 8);
    SS_CHECK(per_row < (1LL << 31) && cdiv(per_row, 256) < 65536 && (int64_t)B * ho < (1LL << 31), "ss_bilinear_fwd2: row too long");
    dim3 grid((unsigned)(B * ho), (unsigned)cdiv(per_row, 256));
    SS_DISPATCH(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T, true>), grid, dim3(256), 0, stream, (const T*)x, ldx, (T*)y, ldy, hi, wi, ho, wo,
                                          fast_div(C / 8), 0, (const T*)x2, ldx2, hi2, wi2));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_bilinear_class_map_u8(int dtype, const void* x, int64_t ldx, uint8_t* out, int B, int hi, int wi, int ho, int wo, int C,
                                        hipStream_t stream) {
    SS_CHECK(x && out && B > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && C > 0 && C <= 256 && ldx >= C,
             "ss_bilinear_class_map_u8: bad arguments (1 <= C <= 256, ldx >= C)");
    SS_CHECK((int64_t)B * ho < (1LL << 31) && cdiv(wo, 256) < 65536, "ss_bilinear_class_map_u8: image too large");
    dim3 grid((unsigned)(B * ho), (unsigned)cdiv(wo, 256));
    if (dtype == SS_BF16 && C <= 8 && ldx % 8 == 0 && aligned16(x))
        hipLaunchKernelGGL((bilinear_class_map_kernel<bf16, true>), grid, dim3(256), 0, stream, (const bf16*)x, ldx, out, hi, wi, ho, wo, C);
    else SS_DISPATCH(dtype, hipLaunchKernelGGL((bilinear_class_map_kernel<T, false>), grid, dim3(256), 0, stream, (const T*)x, ldx, out, hi, wi, ho, wo, C));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_bilinear_bwd(int dtype, const void* dy, int64_t ldy, void* dx, int64_t ldx, int B, int hi, int wi, int ho, int wo, int C,
                               int accumulate, hipStream_t stream) {
    SS_CHECK(dy && dx && B > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && C > 0 && ldx >= C && ldy >= C, "ss_bilinear_bwd: bad arguments");
    const bool vec = C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(dx) && aligned16(dy);
    const int chunks = vec ? C / 8 : C;
    const int64_t per_row = (int64_t)wi * chunks;
    SS_CHECK(per_row < (1LL << 31) && cdiv(per_row, 256) < 65536 && (int64_t)B * hi < (1LL << 31), "ss_bilinear_bwd: row too long");
    if (!vec && C <= 8 && ho == wo / wi * hi && wo % wi == 0 && (wo / wi == 2 || wo / wi == 4)) {
        const int pairs = dtype == SS_BF16 && C % 2 == 0 && ldy % 2 == 0 && ((uintptr_t)dy & 3) == 0;
        dim3 grid((unsigned)(B * hi), (unsigned)cdiv(wi, 256));
        SS_DISPATCH(dtype, {
            if (wo / wi == 4) hipLaunchKernelGGL((bilinear_bwd_narrow_kernel<T, 4>), grid, dim3(256), 0, stream, (const T*)dy, ldy, (T*)dx, ldx, hi, wi, C, accumulate, pairs);
            else hipLaunchKernelGGL((bilinear_bwd_narrow_kernel<T, 2>), grid, dim3(256), 0, stream, (const T*)dy, ldy, (T*)dx, ldx, hi, wi, C, accumulate, pairs);
        });
        SS_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)(B * hi), (unsigned)cdiv(per_row, 256));
    SS_DISPATCH(dtype, {
        if (vec) hipLaunchKernelGGL((bilinear_bwd_kernel<T, true>), grid, dim3(256), 0, stream, (const T*)dy, ldy, (T*)dx, ldx, hi, wi, ho, wo, fast_div(chunks), accumulate);
        else hipLaunchKernelGGL((bilinear_bwd_kernel<T, false>), grid, dim3(256), 0, stream, (const T*)dy, ldy, (T*)dx, ldx, hi, wi, ho, wo, fast_div(chunks), accumulate);
    });
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ casts, packing, small elementwise
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = from_f<TO>(to_f(src[i]));
}
// dst[c][r] = src[r][reverse ? Cc-1-c : c]     (small weight matrices; fp32 source)
template <typename TO>
__global__ void transpose_cast_kernel(const float* __restrict__ src, TO* __restrict__ dst, int R, int Cc, int reverse) {
    const int64_t n = (int64_t)R * Cc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i / R), r = (int)(i % R);
        dst[i] = from_f<TO>(src[(int64_t)r * Cc + (reverse ? Cc - 1 - c : c)]);
    }
}
// conv weight [O][I][k][k] fp32 -> packed [O][Kp] (column order ky,kx,i; zero padded), optionally also transposed [Kp][O]
template <typename TO>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, TO* __restrict__ wp, TO* __restrict__ wpt, int O, int I, int k, int Kp) {
    const int64_t n = (int64_t)O * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int o = (int)(i / Kp), kc = (int)(i % Kp);
        float v = 0.f;
        if (kc < k * k * I) { int ci = kc % I, tap = kc / I; v = w[(((int64_t)o * I + ci) * k + tap / k) * k + tap % k]; }
        wp[i] = from_f<TO>(v);
        if (wpt) wpt[(int64_t)kc * O + o] = from_f<TO>(v);
    }
}
__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ gp, float* __restrict__ g, int O, int I, int k, int Kp) {
    const int64_t n = (int64_t)O * I * k * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int kx = (int)(i % k); int64_t t = i / k; int ky = (int)(t % k); t /= k; int ci = (int)(t % I); int o = (int)(t / I);
        g[i] = gp[(int64_t)o * Kp + (ky * k + kx) * I + ci];
    }
}
// image [B][C][H][W] fp32 -> [B][H][W][C] T
template <typename TO>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, TO* __restrict__ y, int B, int C, int64_t HW) {
    const int64_t n = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C); int64_t p = i / C; int64_t b = p / HW, hw = p % HW;
        y[i] = from_f<TO>(x[(b * C + c) * HW + hw]);
    }
}
// out = a + b  (8-wide)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        F8 u = load8(a + i * 8), v = load8(b + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) u.v[e] += v.v[e];
        store8(out + i * 8, u);
    }
}
// out[r][:] = x[r][:] * scale[r / rows_per_scale]
template <typename T>
__global__ void rowscale_kernel(const T* __restrict__ x, const float* __restrict__ scale, int64_t rows_per_scale, T* __restrict__ out,
                                int64_t rows, int C) {
    const int chunks = C / 8; const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = scale[(i / chunks) / rows_per_scale];
        F8 v = load8(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] *= s;
        store8(out + i * 8, v);
    }
}
// y[b*rows + r][off .. off+C) = v[b][:]   (broadcast of the pooled branch into its concat slice)
template <typename T>
__global__ void bcast_rows_kernel(const T* __restrict__ v, T* __restrict__ y, int64_t ldy, int64_t rows_per_image, int64_t rows, int C) {
    const int chunks = C / 8; const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks; const int c = (int)(i % chunks) * 8;
        store8(y + r * ldy + c, load8(v + (r / rows_per_image) * C + c));
    }
}

static inline int nblocks(int64_t n, int cap = 8192) { int64_t b = cdiv(n, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

extern "C" int ss_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t stream) {
    SS_CHECK(src && dst && n > 0, "ss_cast: bad arguments");
    dim3 g(nblocks(n)), b(256);
    if (src_dtype == SS_F32 && dst_dtype == SS_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, stream, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == SS_BF16 && dst_dtype == SS_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, stream, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == SS_F32 && dst_dtype == SS_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, stream, (const float*)src, (float*)dst, n);
    else if (src_dtype == SS_BF16 && dst_dtype == SS_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, stream, (const bf16*)src, (bf16*)dst, n);
    else return ss_fail(-1, "ss_cast: unsupported dtypes %d -> %d", src_dtype, dst_dtype);
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_transpose_cast(int dst_dtype, const float* src, void* dst, int R, int Cc, int reverse_cols, hipStream_t stream) {
    SS_CHECK(src && dst && R > 0 && Cc > 0, "ss_transpose_cast: bad arguments");
    dim3 g(nblocks((int64_t)R * Cc)), b(256);
    SS_DISPATCH(dst_dtype, hipLaunchKernelGGL((transpose_cast_kernel<T>), g, b, 0, stream, src, (T*)dst, R, Cc, reverse_cols));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_pack_conv_weight(int dst_dtype, const float* w, void* wp, void* wpt, int O, int I, int k, int Kp, hipStream_t stream) {
    SS_CHECK(w && wp && O > 0 && I > 0 && k > 0 && Kp >= k * k * I, "ss_pack_conv_weight: bad arguments");
    dim3 g(nblocks((int64_t)O * Kp)), b(256);
    SS_DISPATCH(dst_dtype, hipLaunchKernelGGL((pack_conv_weight_kernel<T>), g, b, 0, stream, w, (T*)wp, (T*)wpt, O, I, k, Kp));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_unpack_conv_wgrad(const float* gp, float* g, int O, int I, int k, int Kp, hipStream_t stream) {
    SS_CHECK(gp && g && O > 0 && I > 0 && k > 0 && Kp >= k * k * I, "ss_unpack_conv_wgrad: bad arguments");
    hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(nblocks((int64_t)O * I * k * k)), dim3(256), 0, stream, gp, g, O, I, k, Kp);
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_nchw_to_nhwc(int dst_dtype, const float* x, void* y, int B, int C, int64_t HW, hipStream_t stream) {
    SS_CHECK(x && y && B > 0 && C > 0 && HW > 0, "ss_nchw_to_nhwc: bad arguments");
    dim3 g(nblocks((int64_t)B * C * HW, 16384)), b(256);
    SS_DISPATCH(dst_dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), g, b, 0, stream, x, (T*)y, B, C, HW));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_add(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t stream) {
    SS_CHECK(a && b && out && n > 0 && n % 8 == 0, "ss_add: n must be a positive multiple of 8");
    dim3 g(nblocks(n / 8, 16384)), bl(256);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((add_kernel<T>), g, bl, 0, stream, (const T*)a, (const T*)b, (T*)out, n / 8));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_rowscale(int dtype, const void* x, const float* scale, int64_t rows_per_scale, void* out, int64_t rows, int C, hipStream_t stream) {
    SS_CHECK(x && scale && out && rows > 0 && C > 0 && C % 8 == 0 && rows_per_scale > 0, "ss_rowscale: bad arguments");
    dim3 g(nblocks(rows * (C / 8), 16384)), b(256);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((rowscale_kernel<T>), g, b, 0, stream, (const T*)x, scale, rows_per_scale, (T*)out, rows, C));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_bcast_rows(int dtype, const void* v, void* y, int64_t ldy, int64_t rows_per_image, int64_t rows, int C, hipStream_t stream) {
    SS_CHECK(v && y && rows > 0 && C > 0 && C % 8 == 0 && ldy >= C && ldy % 8 == 0 && rows_per_image > 0, "ss_bcast_rows: bad arguments");
    dim3 g(nblocks(rows * (C / 8), 16384)), b(256);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((bcast_rows_kernel<T>), g, b, 0, stream, (const T*)v, (T*)y, ldy, rows_per_image, rows, C));
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ all of a model's weight conversions in one launch
// The per-step casts / transposes / conv-weight packings of the fp32 master weights (ss_cast, ss_transpose_cast,
// ss_pack_conv_weight: ~280 launches of ~5 us each per MiT-B2 + Lawin step) replayed from a table that lives in device memory:
// entry e = { src, dst, n_out, kind | dst_dtype << 8 | reverse << 16, a, b, c, first_block } (8 x int64), element formulas
// identical to the single-tensor kernels above.  A block owns 1024 consecutive output elements of one entry, except for
// kind 1, where it owns one 32 x 32 tile (ceil(R / 32) * ceil(Cc / 32) blocks, row tiles fastest) and goes through LDS so
// that both the fp32 reads and the stores are contiguous.
//   kind 0 cast            dst[i] = src[i]
//   kind 1 transpose_cast  src [a = R][b = Cc] -> dst [Cc][R]
//   kind 2 pack (wp)       src [a = O][b = I][k][k] -> dst [O][Kp], c = k | Kp << 32
//   kind 3 pack (wpt)      same source -> dst [Kp][O]
template <typename TO>
__device__ __forceinline__ void prep_block(const int64_t* __restrict__ ent, int64_t base) {
    const float* __restrict__ src = (const float*)ent[0];
    TO* __restrict__ dst = (TO*)ent[1];
    const int64_t n = ent[2];
    const int kind = (int)(ent[3] & 0xff);                 // 0, 2 or 3 here: kind 1 is tiled in the kernel below
    const int a = (int)ent[4], b = (int)ent[5], k = (int)(ent[6] & 0xffffffff), Kp = (int)(ent[6] >> 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i >= n) break;
        float v;
        if (kind == 0) v = src[i];
        else {
            int o, kc;
            if (kind == 2) { o = (int)(i / Kp); kc = (int)(i % Kp); } else { kc = (int)(i / a); o = (int)(i % a); }
            v = 0.f;
            if (kc < k * k * b) { const int ci = kc % b, tap = kc / b; v = src[(((int64_t)o * b + ci) * k + tap / k) * k + tap % k]; }
        }
        dst[i] = from_f<TO>(v);
    }
}

__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const int64_t* __restrict__ table, int nent) {
    int lo = 0, hi = nent - 1;                         // last entry whose first_block <= blockIdx.x (wave-uniform search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 8 + 7] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* ent = table + (int64_t)lo * 8;
    if ((ent[3] & 0xff) == 1) {
        __shared__ float tile[32][33];
        const int R = (int)ent[4], Cc = (int)ent[5], reverse = (int)((ent[3] >> 16) & 1), bf = ((ent[3] >> 8) & 0xff) == SS_BF16;
        const int t = (int)((int64_t)blockIdx.x - ent[7]), tr = (R + 31) >> 5;
        const int r0 = (t % tr) * 32, c0 = (t / tr) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const float* __restrict__ src = (const float*)ent[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                       // tile[r][c] <- src[r0 + r][column of output row c0 + c]
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            if (r < R && c < Cc) tile[ty + 8 * j][tx] = src[(int64_t)r * Cc + (reverse ? Cc - 1 - c : c)];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < R && c < Cc) {
                const float v = tile[tx][ty + 8 * j];
                if (bf) ((bf16*)ent[1])[(int64_t)c * R + r] = from_f<bf16>(v); else ((float*)ent[1])[(int64_t)c * R + r] = v;
            }
        }
        return;
    }
    const int64_t base = ((int64_t)blockIdx.x - ent[7]) * 1024;
    if (((ent[3] >> 8) & 0xff) == SS_BF16) prep_block<bf16>(ent, base); else prep_block<float>(ent, base);
}

extern "C" int ss_weight_prep_batch(const int64_t* table_dev, int nent, int64_t total_blocks, hipStream_t stream) {
    SS_CHECK(table_dev && nent > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "ss_weight_prep_batch: bad arguments");
    hipLaunchKernelGGL(weight_prep_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table_dev, nent);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ input staging (data_factory/utils.py:6-37)
// image_to_tensor on the device: out[b, c, y, x] = (float(src[b, c, y, x]) - mu[c]) / sigma[c] in fp32 (correctly rounded
// division, as torch's), stored in the caller's dtype and strides -- NCHW fp32 reproduces the reference tensor bit for bit, the
// channels-last compute dtype is what the first layer of this engine reads without another pass.  The batch travels to the
// device as uint8 (a quarter of the float32 bytes).  A thread owns 4 neighbouring pixels of a row in every channel.
template <typename T, int CC>                       // CC = 3: RGB into a dense channels-last map, 4 pixels leave as one contiguous run
__global__ __launch_bounds__(256) void stage_image_u8_kernel(const unsigned char* __restrict__ src, const float* __restrict__ mu,
        const float* __restrict__ sigma, T* __restrict__ out, int64_t on, int64_t oc, int64_t oh, int64_t ow, int C, int H, int W, int64_t quads) {
    const int wq = (W + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(q % wq);
        const int64_t rowi = q / wq;                                 // b * H + y
        const int y = (int)(rowi % H);
        const int64_t b = rowi / H;
        const int x0 = xq * 4, nx = min(4, W - x0);
        if (CC > 0 && nx == 4) {
            union { T v[4 * CC]; unsigned long long w[(4 * CC * sizeof(T)) / 8]; } o;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const unsigned char* sp = src + ((b * CC + c) * H + y) * (int64_t)W + x0;
                unsigned char px[4];
                if ((((uintptr_t)sp) & 3) == 0) *(unsigned*)px = *(const unsigned*)sp;
                else { px[0] = sp[0]; px[1] = sp[1]; px[2] = sp[2]; px[3] = sp[3]; }
                const float m = mu[c], sg = sigma[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) o.v[j * CC + c] = from_f<T>(__fdiv_rn((float)px[j] - m, sg));
            }
            T* dp = out + b * on + y * oh + (int64_t)x0 * CC;
            if ((((uintptr_t)dp) & 7) == 0) {
#pragma unroll
                for (int k = 0; k < (int)((4 * CC * sizeof(T)) / 8); ++k) ((unsigned long long*)dp)[k] = o.w[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4 * CC; ++k) dp[k] = o.v[k];
            }
            continue;
        }
        for (int c = 0; c < C; ++c) {
            const unsigned char* sp = src + ((b * C + c) * H + y) * (int64_t)W + x0;
            unsigned char px[4] = {0, 0, 0, 0};
            if (nx == 4 && (((uintptr_t)sp) & 3) == 0) *(unsigned*)px = *(const unsigned*)sp;
            else for (int j = 0; j < nx; ++j) px[j] = sp[j];
            const float m = mu[c], sg = sigma[c];
            T* dp = out + b * on + c * oc + y * oh + x0 * ow;
            for (int j = 0; j < nx; ++j) dp[j * ow] = from_f<T>(__fdiv_rn((float)px[j] - m, sg));
        }
    }
}
__global__ void widen_u8_i64_kernel(const unsigned char* __restrict__ src, long long* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = src[i];
}
extern "C" int ss_stage_image_u8(int out_dtype, const uint8_t* src, const float* mu, const float* sigma, void* out, int64_t on, int64_t oc,
                                 int64_t oh, int64_t ow, int B, int C, int H, int W, hipStream_t stream) {
    SS_CHECK(src && mu && sigma && out && B > 0 && C > 0 && H > 0 && W > 0, "ss_stage_image_u8: bad arguments");
    const int64_t quads = (int64_t)B * H * ((W + 3) / 4);
    if (C == 3 && oc == 1 && ow == 3)
        SS_DISPATCH(out_dtype, hipLaunchKernelGGL((stage_image_u8_kernel<T, 3>), dim3(nblocks(quads, 16384)), dim3(256), 0, stream, src, mu, sigma,
                                                  (T*)out, on, oc, oh, ow, C, H, W, quads));
    else
        SS_DISPATCH(out_dtype, hipLaunchKernelGGL((stage_image_u8_kernel<T, 0>), dim3(nblocks(quads, 16384)), dim3(256), 0, stream, src, mu, sigma,
                                                  (T*)out, on, oc, oh, ow, C, H, W, quads));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_stage_label_u8(const uint8_t* src, int64_t* out, int64_t n, hipStream_t stream) {
    SS_CHECK(src && out && n > 0, "ss_stage_label_u8: bad arguments");
    hipLaunchKernelGGL(widen_u8_i64_kernel, dim3(nblocks(n, 16384)), dim3(256), 0, stream, src, (long long*)out, n);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ DropPath multiplier (modules/helpers.py:17-24)
// keep[i] = floor(kp + u[i]) / kp with u the torch.rand draw in the activation dtype: the sum is rounded to that dtype before the
// floor, as `kp + torch.rand(shape, dtype=x.dtype)` does, and the division is torch's scalar form (a multiply by 1 / kp in fp32).
// One launch instead of add, floor_, float and div.
template <typename T>
__global__ void droppath_keep_kernel(const T* __restrict__ u, float kp, float inv_kp, float* __restrict__ keep, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = floorf(to_f(from_f<T>(kp + to_f(u[i])))) * inv_kp;
}
extern "C" int ss_droppath_keep(int dtype, const void* u, float keep_prob, float* keep, int n, hipStream_t stream) {
    SS_CHECK(u && keep && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "ss_droppath_keep: bad arguments (keep_prob=%f)", (double)keep_prob);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((droppath_keep_kernel<T>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, (const T*)u, keep_prob,
                                          1.0f / keep_prob, keep, n));
    SS_LAUNCH_CHECK();
    return 0;
}
