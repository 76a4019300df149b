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

// Large windows of narrow maps (PyramidPoolAgg of the first stages: 16 x 16 windows of 32 channels): a thread per output chunk walks
// win^2 far-apart pixels and only B*ho*wo*C/8 threads exist.  Here a WAVE owns a window: lane = (row lane, 8-channel chunk), a lane sums its
// rows' pixels (consecutive chunk lanes read a pixel's contiguous channels), the row lanes are combined by a fixed butterfly.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_wave_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int B, int H, int W,
                                                           int C, int win) {
    const int chunks = C / 8, dyl = 64 / chunks, ho = H / win, wo = W / win;
    const int lane = threadIdx.x & 63, ch = lane % chunks, dl = lane / chunks;
    const int64_t nwin = (int64_t)B * ho * wo, wpb = blockDim.x >> 6;
    const float inv = 1.f / (win * win);
    for (int64_t p = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); p < nwin; p += (int64_t)gridDim.x * wpb) {
        const int ox = (int)(p % wo); const int64_t q = p / wo; const int oy = (int)(q % ho); const int64_t b = q / ho;
        F8 acc = zero8();
        for (int dy = dl; dy < win; dy += dyl) {
            const T* row = x + ((b * H + oy * win + dy) * W + (int64_t)ox * win) * ldx + ch * 8;
#pragma unroll 4
            for (int dx = 0; dx < win; ++dx) {
                const F8 v = load8(row + dx * ldx);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e];
            }
        }
        for (int off = chunks; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += __shfl_xor(acc.v[e], off, 64);
        if (dl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] *= inv;
            store8(y + p * ldy + ch * 8, acc);
        }
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

// out = local * up(ga) + up(ge) + sum_i up(extra_i): the injection of the finest stage and the SimpleHead's UniRes + Sum over the other
// stages in one pass over the finest map (the sum used to be three read-modify-write passes over it); fp32 until the one rounding.
// A workgroup walks output rows: the vertical taps of every source are per-row constants, a thread owns 8 channels of a pixel, the four
// taps enter with their products l_y * l_x as packed fp32 fmas (bf16 pairs widen with one shift and one mask), and the flat index is
// decomposed by multiply-shift (the 64-bit divisions of a flat grid cost more than the arithmetic).
struct UpSrc { const void* p; int64_t ld; int h, w; float sh, sw; };       // sh, sw: source / destination size
struct UpSrc3 { UpSrc s[3]; int n; };

template <typename T> struct Raw8;                               // 8 consecutive elements as loaded (the widening waits until they are used)
template <> struct Raw8<bf16> { bf16x8 v; __device__ __forceinline__ void load(const bf16* p) { v = *(const bf16x8*)p; }
    __device__ __forceinline__ void widen(floatx2 (&o)[4]) const { union { bf16x8 b; unsigned w[4]; } u; u.b = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = floatx2{__uint_as_float(u.w[e] << 16), __uint_as_float(u.w[e] & 0xffff0000u)}; } };
template <> struct Raw8<float> { floatx4 a, b; __device__ __forceinline__ void load(const float* p) { a = *(const floatx4*)p; b = *(const floatx4*)(p + 4); }
    __device__ __forceinline__ void widen(floatx2 (&o)[4]) const { o[0] = floatx2{a[0], a[1]}; o[1] = floatx2{a[2], a[3]}; o[2] = floatx2{b[0], b[1]}; o[3] = floatx2{b[2], b[3]}; } };
// the four taps of one bilinear sample: loads first (all samples of a pixel are requested before any is used), weights l_y * l_x
template <typename T> struct Quad {
    Raw8<T> t[4]; float w[4];
    __device__ __forceinline__ void load(const T* r0, const T* r1, int x0, int x1, float ly0, float ly1, float lx0, float lx1) {   // 32-bit in-row offsets: uniform base + lane offset
        t[0].load(r0 + x0); t[1].load(r0 + x1); t[2].load(r1 + x0); t[3].load(r1 + x1);
        w[0] = ly0 * lx0; w[1] = ly0 * lx1; w[2] = ly1 * lx0; w[3] = ly1 * lx1;
    }
    __device__ __forceinline__ void acc(floatx2 (&o)[4]) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            floatx2 v[4]; t[k].widen(v);
            const floatx2 wk = {w[k], w[k]};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_elementwise_fma(v[e], wk, o[e]);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void inject_fuse_sum_kernel(const T* __restrict__ local, int64_t ldl, const T* __restrict__ ga, int64_t lda,
        const T* __restrict__ ge, int64_t lde, UpSrc3 ex, T* __restrict__ out, int64_t ldo, int B, int hl, int wl, int hg, int wg, int C,
        FastDiv d_chunks, FastDiv d_hl) {
    const int chunks = C / 8, items = wl * chunks;
    const float sh = (float)hg / hl, sw = (float)wg / wl;
    // workgroups bid, bid + 8, ... share an XCD and its L2: each XCD takes a contiguous run of output rows, whose neighbours read the same
    // source rows (spread round-robin, every L2 fetched every source row)
    const int nrow = B * hl, per = (nrow + 7) >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = (gridDim.x + 7 - xcd) >> 3;
    for (int r = slot; r < per; r += slots) {
        const int row = xcd * per + r;
        if (row >= nrow) break;
        const int b = (int)fdiv((unsigned)row, d_hl), oy = row - b * hl;
        const Tap ty = tap_of(oy, sh, hg);
        const T* ga0 = ga + ((int64_t)(b * hg + ty.i0) * wg) * lda; const T* ga1 = ga + ((int64_t)(b * hg + ty.i1) * wg) * lda;
        const T* ge0 = ge + ((int64_t)(b * hg + ty.i0) * wg) * lde; const T* ge1 = ge + ((int64_t)(b * hg + ty.i1) * wg) * lde;
        // per-row constants of the extra sources: vertical taps, row pointers, horizontal scale
        const T* er0[3]; const T* er1[3]; float el0[3], el1[3], esw[3]; int eld[3];
        const int ilda = (int)lda, ilde = (int)lde, ildl = (int)ldl, ildo = (int)ldo;
        const T* lrow = local + (int64_t)row * wl * ldl; T* orow = out + (int64_t)row * wl * ldo;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const UpSrc& u = ex.s[k < ex.n ? k : 0];
            const Tap uy = tap_of(oy, ex.n ? u.sh : 1.f, ex.n ? u.h : 1);
            er0[k] = (const T*)u.p + ((int64_t)(b * u.h + uy.i0) * u.w) * u.ld; er1[k] = (const T*)u.p + ((int64_t)(b * u.h + uy.i1) * u.w) * u.ld;
            el0[k] = uy.l0; el1[k] = uy.l1; esw[k] = u.sw; eld[k] = (int)u.ld;
        }
        for (int it = threadIdx.x; it < items; it += 256) {
            const int ox = (int)fdiv((unsigned)it, d_chunks), c = (it - ox * chunks) * 8;
            const Tap tx = tap_of(ox, sw, wg);
            Quad<T> qa, qe, qx[3];
            Raw8<T> lr;
            qa.load(ga0, ga1, tx.i0 * ilda + c, tx.i1 * ilda + c, ty.l0, ty.l1, tx.l0, tx.l1);
            qe.load(ge0, ge1, tx.i0 * ilde + c, tx.i1 * ilde + c, ty.l0, ty.l1, tx.l0, tx.l1);
            lr.load(lrow + (ox * ildl + c));
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < ex.n) {
                    const UpSrc& u = ex.s[k];
                    const Tap ux = tap_of(ox, esw[k], u.w);
                    qx[k].load(er0[k], er1[k], ux.i0 * eld[k] + c, ux.i1 * eld[k] + c, el0[k], el1[k], ux.l0, ux.l1);
                }
            floatx2 act[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, o[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, l[4];
            qa.acc(act); qe.acc(o); lr.widen(l);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_elementwise_fma(l[e], act[e], o[e]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < ex.n) qx[k].acc(o);
            F8 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) { r.v[2 * e] = o[e][0]; r.v[2 * e + 1] = o[e][1]; }
            store8(orow + (ox * ildo + c), r);
        }
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

// Stem of the token pyramid (modules/topformer.py:1023-1033: Conv 3 -> 16, k3, s2, AutoPadding, folded BN, ReLU) straight from
// the image: no channels-last staging pass, no im2col matrix (at 2048 x 2048 those were 0.1 + 0.42 ms and 270 MB per batch of 4
// next to a 0.15 ms GEMM).  A thread owns one output pixel and its 16 channels; the 27 x 16 weights sit in LDS as fp32 and are
// read as broadcasts; inputs are rounded to the compute dtype on entry, exactly what the staging pass stored, and the packed
// weight is the GEMM's own operand, so only the fp32 summation order differs from the im2col + GEMM path.
template <typename T, typename TI, int CIN>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const TI* __restrict__ x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
        const T* __restrict__ wp, int Kp, const float* __restrict__ bias, T* __restrict__ y, int H, int W, int ho, int wo, int pt, int pl,
        int act, int64_t pairs) {
    constexpr int KK = 9 * CIN;
    __shared__ __attribute__((aligned(16))) float wsm[KK * 16];
    for (int i = threadIdx.x; i < KK * 16; i += 256) wsm[i] = to_f(wp[(int64_t)(i & 15) * Kp + (i >> 4)]);
    __syncthreads();
    // one thread = two neighbouring output pixels (they share an input column and every weight read); no loop over pixels and a
    // rolled loop over kernel rows: fully unrolled, the compiler loads all 432 weights up front (256 VGPRs + 1.5 KB of spills)
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= pairs) return;
    const int wo2 = (wo + 1) >> 1;
    const int oxp = (int)(p % wo2); const int64_t q = p / wo2; const int oy = (int)(q % ho); const int64_t b = q / ho;
    const int ox0 = oxp * 2;
    float acc[2][16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[0][o] = acc[1][o] = bias ? bias[o] : 0.f;
    const TI* xb = x + b * sn;
#pragma unroll 1                                            // a real loop: one kernel row's 9 x 16 weights in registers at a time
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - pt + ky;
        const bool yin = iy >= 0 && iy < H;
        float v[5][CIN];
#pragma unroll
        for (int jx = 0; jx < 5; ++jx) {
            const int ix = ox0 * 2 - pl + jx;
            const bool in = yin && ix >= 0 && ix < W;
            const TI* px = xb + (in ? iy * sh + ix * sw : 0);
#pragma unroll
            for (int c = 0; c < CIN; ++c) { const float t = to_f(from_f<T>((float)px[c * sc])); v[jx][c] = in ? t : 0.f; }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const floatx4* wr = (const floatx4*)(wsm + ((ky * 3 + kx) * CIN + c) * 16);
                const floatx4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][e] = fmaf(v[kx][c], w0[e], acc[0][e]);           acc[1][e] = fmaf(v[kx + 2][c], w0[e], acc[1][e]);
                    acc[0][4 + e] = fmaf(v[kx][c], w1[e], acc[0][4 + e]);   acc[1][4 + e] = fmaf(v[kx + 2][c], w1[e], acc[1][4 + e]);
                    acc[0][8 + e] = fmaf(v[kx][c], w2[e], acc[0][8 + e]);   acc[1][8 + e] = fmaf(v[kx + 2][c], w2[e], acc[1][8 + e]);
                    acc[0][12 + e] = fmaf(v[kx][c], w3[e], acc[0][12 + e]); acc[1][12 + e] = fmaf(v[kx + 2][c], w3[e], acc[1][12 + e]);
                }
            }
    }
    T* dst = y + ((q * wo) + ox0) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (ox0 + h >= wo) break;
        F8 lo, hi;
#pragma unroll
        for (int o = 0; o < 8; ++o) { lo.v[o] = apply_act(acc[h][o], act); hi.v[o] = apply_act(acc[h][8 + o], act); }
        store8(dst + h * 16, lo);
        store8(dst + h * 16 + 8, hi);
    }
}

extern "C" int ss_stem_conv3x3s2(int dtype, int in_dtype, const void* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const void* wp, int Kp,
                                 const float* bias, void* y, int B, int Cin, int H, int W, int pad_top, int pad_left, int Ho, int Wo, int act,
                                 hipStream_t stream) {
    SS_CHECK(x && wp && y && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Kp >= 9 * Cin, "ss_stem_conv3x3s2: bad arguments");
    SS_CHECK(Cin == 3, "ss_stem_conv3x3s2: built for 3 input channels (got %d)", Cin);
    SS_CHECK(in_dtype == SS_F32 || in_dtype == dtype, "ss_stem_conv3x3s2: the image is fp32 or already in the compute dtype");
    SS_CHECK(act >= 0 && act <= 5 && aligned16(y), "ss_stem_conv3x3s2: bad activation or unaligned output");
    SS_CHECK((Ho - 1) * 2 - pad_top < H && (Wo - 1) * 2 - pad_left < W, "ss_stem_conv3x3s2: output larger than the padded image allows");
    const int64_t total = (int64_t)B * Ho * ((Wo + 1) / 2);                 // pixel pairs, one per thread
    SS_CHECK(cdiv(total, 256) < 2147483647LL, "ss_stem_conv3x3s2: image too large");
    dim3 grid((unsigned)cdiv(total, 256));
    if (in_dtype == SS_F32) {
        SS_DISPATCH(dtype, hipLaunchKernelGGL((stem_conv_kernel<T, float, 3>), grid, dim3(256), 0, stream, (const float*)x, sn, sc, sh, sw, (const T*)wp,
                                              Kp, bias, (T*)y, H, W, Ho, Wo, pad_top, pad_left, act, total));
    } else {
        SS_DISPATCH(dtype, hipLaunchKernelGGL((stem_conv_kernel<T, T, 3>), grid, dim3(256), 0, stream, (const T*)x, sn, sc, sh, sw, (const T*)wp, Kp, bias,
                                              (T*)y, H, W, Ho, Wo, pad_top, pad_left, act, total));
    }
    SS_LAUNCH_CHECK();
    return 0;
}

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
    const int chunks = C / 8;
    if (win >= 4 && chunks <= 16 && (chunks & (chunks - 1)) == 0) {          // a wave per window
        SS_DISPATCH(dtype, hipLaunchKernelGGL((avgpool_wave_kernel<T>), dim3(tf_blocks((int64_t)B * (H / win) * (W / win) * 64)), dim3(256), 0, stream,
                                              (const T*)x, ldx, (T*)y, ldy, B, H, W, C, win));
        SS_LAUNCH_CHECK();
        return 0;
    }
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
extern "C" int ss_inject_fuse_sum(int dtype, const void* local, int64_t ldl, const void* gact, int64_t lda, const void* gemb, int64_t lde,
                                  int n_extra, const void* const* extra, const int64_t* extra_ld, const int* extra_h, const int* extra_w, void* out,
                                  int64_t ldo, int B, int hl, int wl, int hg, int wg, int C, hipStream_t stream) {
    SS_CHECK(local && gact && gemb && out && B > 0 && hl > 0 && wl > 0 && hg > 0 && wg > 0 && C > 0 && C % 8 == 0, "ss_inject_fuse_sum: bad arguments");
    SS_CHECK(ok8(local, ldl) && ok8(gact, lda) && ok8(gemb, lde) && ok8(out, ldo) && ldl >= C && lda >= C && lde >= C && ldo >= C,
             "ss_inject_fuse_sum: rows must be 16-byte aligned");
    SS_CHECK(n_extra >= 0 && n_extra <= 3 && (n_extra == 0 || (extra && extra_ld && extra_h && extra_w)), "ss_inject_fuse_sum: 0..3 extra sources");
    UpSrc3 ex{};
    ex.n = n_extra;
    for (int k = 0; k < n_extra; ++k) {
        SS_CHECK(extra[k] && ok8(extra[k], extra_ld[k]) && extra_ld[k] >= C && extra_h[k] > 0 && extra_w[k] > 0, "ss_inject_fuse_sum: bad extra source %d", k);
        ex.s[k].p = extra[k]; ex.s[k].ld = extra_ld[k]; ex.s[k].h = extra_h[k]; ex.s[k].w = extra_w[k];
        ex.s[k].sh = (float)extra_h[k] / hl; ex.s[k].sw = (float)extra_w[k] / wl;
    }
    SS_CHECK((int64_t)B * hl < (1LL << 31) && (int64_t)wl * (C / 8) < (1LL << 31), "ss_inject_fuse_sum: map too large");
    {   // in-row offsets are 32-bit
        int64_t mx = (int64_t)wl * (ldl > ldo ? ldl : ldo), g = (int64_t)wg * (lda > lde ? lda : lde);
        if (g > mx) mx = g;
        for (int k = 0; k < n_extra; ++k) if ((int64_t)extra_w[k] * extra_ld[k] > mx) mx = (int64_t)extra_w[k] * extra_ld[k];
        SS_CHECK(mx < (1LL << 31), "ss_inject_fuse_sum: a map row exceeds 2^31 elements");
    }
    const int64_t rows = (int64_t)B * hl;
    SS_DISPATCH(dtype, hipLaunchKernelGGL((inject_fuse_sum_kernel<T>), dim3((unsigned)(rows < 65536 ? rows : 65536)), dim3(256), 0, stream,
                                          (const T*)local, ldl, (const T*)gact, lda, (const T*)gemb, lde, ex, (T*)out, ldo, B, hl, wl, hg, wg, C,
                                          fast_div((unsigned)(C / 8)), fast_div((unsigned)hl)));
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
