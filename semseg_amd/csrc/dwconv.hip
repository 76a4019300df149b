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
#include "gemm_fast.h"

#define DW_UNROLL 9                                   // rows per loop iteration: divides the strip height, a multiple of the window height 3
// forward kernel: rows in flight per thread (the ring) and its loop unroll, a multiple of both the ring depth and the window height
#ifndef DW_FWD_RING
#define DW_FWD_RING 3                                 // 4 rows in flight (246 VGPRs, strips in multiples of 12) measured 7 % slower
#endif
#define DW_FWD_UNROLL (3 * DW_FWD_RING)

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

struct DwGeom { int cpb, xl, groups, xtiles, ystrips, chunks, ty; };
static inline DwGeom dw_geom(int B, int H, int W, int C, int ne, int unr = DW_UNROLL) {
    DwGeom g; g.chunks = C / ne;
    const int cap = ne == 8 ? 16 : 64;                  // 128 channels = two 128-byte lines per pixel, 16 columns per workgroup (32 / 8 columns: 3 % slower)
    int cpb = 1; while (cpb < g.chunks && cpb < cap) cpb <<= 1;
    g.cpb = cpb; g.xl = 256 / cpb; g.groups = (int)cdiv(g.chunks, cpb);
    g.xtiles = (int)cdiv(W, g.xl);
    // rows per strip: a multiple of DW_UNROLL.  Taller strips amortise the fill of window + ring and the 2 halo rows (about 6 rows
    // of work per strip); rows past the image bottom are computed and dropped; and the launch wants >= 8 workgroups per CU.
    int best = 2 * unr; int64_t best_cost = -1;
    for (int ty = 2 * unr; ty <= 8 * unr; ty += unr) {
        const int64_t strips = cdiv(H, ty), cost = strips * (ty + 6);
        const int64_t blocks = strips * g.xtiles * g.groups * B;
        if (ty > 2 * unr && blocks < 2048) break;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = ty; }
    }
    g.ty = best; g.ystrips = (int)cdiv(H, g.ty);
    return g;
}

// One row of the window in storage dtype: columns x-1, x, x+1 of a row that EXISTS (the caller clamps the row, offL / offR are
// -C / +C or 0 at the image border).  Always three unconditional loads: a load under a branch cannot stay in flight across
// the steps of the row loop, the compiler waits for it at the join.  What a clamped load fetched is discarded where it is used:
// border columns through zeroed taps (forward) or masked gradients (backward), the row below the image through `keep`.
template <typename T, int NE>
__device__ __forceinline__ void dw_issue_row(const T* __restrict__ p, int offL, int offR, vecn<T, NE> (&row)[3]) {
    row[1] = raw_load<T, NE>(p);
    row[0] = raw_load<T, NE>(p + offL);
    row[2] = raw_load<T, NE>(p + offR);
}
// the window itself is kept in fp32: every input is converted once, when it enters, not once per tap it takes part in.
// keep[kx] = ~0u, or 0 where column kx of this row lies outside the image (AND on the raw bits: bf16 / fp32 +0)
template <typename T, int NE>
__device__ __forceinline__ void dw_cvt_row(const vecn<T, NE> (&raw)[3], unsigned keepL, unsigned keepC, unsigned keepR, float (&row)[3][NE]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned keep = kx == 0 ? keepL : kx == 1 ? keepC : keepR;
        union { vecn<T, NE> v; unsigned u[sizeof(vecn<T, NE>) / 4]; } m;
        m.v = raw[kx];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(vecn<T, NE>) / 4); ++i) m.u[i] &= keep;
#pragma unroll
        for (int e = 0; e < NE; ++e) row[kx][e] = (float)m.v[e];
    }
}

// The scheduler may move memory and scalar instructions across these points, vector arithmetic stays in source order: hipcc
// otherwise emits each accumulator's whole dependent chain back to back (a stall per instruction at two waves per SIMD).
#define DW_PIN_VALU() __builtin_amdgcn_sched_barrier(0x0004 | 0x0010 | 0x0080)

// A ring of prefetched rows enters its loop with the prologue's loads in flight, issued in the prologue's order.  hipcc places ONE wait per
// use in the loop body, which must also hold on the first iteration, i.e. for the prologue's issue order -- there the rows needed first are
// among the ones issued last, so every iteration waited for all but the newest load (s_waitcnt vmcnt(9) / (4) / (1) over the three unrolled
// steps where the steady state would allow ~14 in flight).  A wait the compiler's counter pass sees (the builtin, not asm) between prologue
// and loop changes that entry state: the waits become vmcnt(8) / (8) / (7), fused backward 5.08 -> 4.90 ms per step.  (The prologue's
// loads are invariant loads and still sink below the builtin, nothing orders them: what is gained is the merged state, not a drain.)
// Encoding (gfx9): vmcnt = 0, expcnt and lgkmcnt left at their maxima.
#define DW_DRAIN_PROLOGUE() __builtin_amdgcn_s_waitcnt(0x0F70)

// tap-major: the NE accumulators advance together, so consecutive instructions are independent
template <int NE>
__device__ __forceinline__ void dw_taps(const float (&top)[3][NE], const float (&mid)[3][NE], const float (&bot)[3][NE],
                                        const float (&wt)[9][NE], const float (&bv)[NE], float (&u)[NE]) {
    float a[NE], b[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { u[e] = fmaf(top[0][e], wt[0][e], bv[e]); a[e] = mid[0][e] * wt[3][e]; b[e] = bot[0][e] * wt[6][e]; }
    DW_PIN_VALU();
#pragma unroll
    for (int kx = 1; kx < 3; ++kx) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            u[e] = fmaf(top[kx][e], wt[kx][e], u[e]);
            a[e] = fmaf(mid[kx][e], wt[3 + kx][e], a[e]);
            b[e] = fmaf(bot[kx][e], wt[6 + kx][e], b[e]);
        }
        DW_PIN_VALU();
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) u[e] += a[e] + b[e];
    DW_PIN_VALU();
}

// GELU pieces for NE values at once, stage by stage (same formula as gelu_parts in common.h)
template <int NE>
__device__ __forceinline__ void gelu_parts_v(const float (&x)[NE], float (&cdf)[NE], float (&ez)[NE]) {
    float z[NE], t[NE], p[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { z[e] = fabsf(x[e]) * 0.70710678118654752f; t[e] = fmaf(0.3275911f, z[e], 1.f); }
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) { t[e] = __builtin_amdgcn_rcpf(t[e]); z[e] = -z[e] * z[e]; }
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) { ez[e] = __expf(z[e]); p[e] = fmaf(t[e], 1.061405429f, -1.453152027f); }
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = fmaf(t[e], p[e], 1.421413741f);
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = fmaf(t[e], p[e], -0.284496736f);
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = fmaf(t[e], p[e], 0.254829592f);
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = 0.5f * (t[e] * p[e]);
    DW_PIN_VALU();
#pragma unroll
    for (int e = 0; e < NE; ++e) { const float hc = p[e] * ez[e]; cdf[e] = x[e] >= 0.f ? 1.f - hc : hc; }
    DW_PIN_VALU();
}

// GELU'(u) by linear interpolation in a table kept in LDS (bf16 storage only): 1024 intervals of 1/64 over [-8, 8), entry = (value, step to
// the next value).  Interpolation error <= h^2 / 8 * max|third derivative of GELU| = 2.5e-5, two orders below the rounding of the bf16
// gradient the result is multiplied into; outside the range the derivative is 0 / 1 to 1e-14.  Six vector instructions and one
// ds_read_b64 per value where the closed form (gelu_parts_v: reciprocal, exponential, five-term polynomial) takes twenty-one on a kernel
// that is vector-ALU bound; the LDS pipe is idle next to it.  A NaN in u reads entry 0 (the index conversion maps it there) -- the
// gradient it multiplies carries the NaN.
#ifndef DWF_LUT
#define DWF_LUT 1
#endif
constexpr int GELU_LUT_N = 1024;
__device__ __forceinline__ void gelu_grad_lut_fill(float2* lut, int tid, int nthreads) {
    for (int i = tid; i < GELU_LUT_N; i += nthreads) {
        const float f0 = gelu_erf_grad((float)(i - GELU_LUT_N / 2) * (1.f / 64.f)), f1 = gelu_erf_grad((float)(i + 1 - GELU_LUT_N / 2) * (1.f / 64.f));
        lut[i] = make_float2(f0, f1 - f0);
    }
}
template <int NE>
__device__ __forceinline__ void gelu_grad_lut(const float2* lut, const float (&u)[NE], float (&d)[NE]) {
    float t[NE]; unsigned idx[NE]; float2 en[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) t[e] = __builtin_amdgcn_fmed3f(fmaf(u[e], 64.f, (float)(GELU_LUT_N / 2)), 0.f, (float)GELU_LUT_N - 0.004f);
#pragma unroll
    for (int e = 0; e < NE; ++e) idx[e] = (unsigned)t[e];
#pragma unroll
    for (int e = 0; e < NE; ++e) en[e] = lut[idx[e]];
#pragma unroll
    for (int e = 0; e < NE; ++e) d[e] = fmaf(__builtin_amdgcn_fractf(t[e]), en[e].y, en[e].x);
}

// Row r of the strip reaches the window through a ring of 3 raw rows: its loads are issued three steps before the step that
// converts it, so a wave's own arithmetic covers its own HBM latency (two waves per SIMD suffice).
template <typename T, int NE, int ACT>
__global__ __launch_bounds__(256, 2) void dwconv3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
        T* __restrict__ y, int H, int W, int C, DwGeom g) {
    typedef vecn<T, NE> V;
    const int tid = threadIdx.x, chunk = blockIdx.y * g.cpb + tid % g.cpb, xs = tid / g.cpb;
    const int xt = blockIdx.x % g.xtiles, ys = blockIdx.x / g.xtiles;
    const int xx = xt * g.xl + xs, y0 = ys * g.ty, c = chunk * NE;
    if (chunk >= g.chunks || xx >= W) return;
    float wt[9][NE], bv[NE];
#pragma unroll
    for (int k = 0; k < 9; ++k) loadf<NE>(w + k * C + c, wt[k]);
    if (bias) loadf<NE>(bias + c, bv);
    else {
#pragma unroll
        for (int e = 0; e < NE; ++e) bv[e] = 0.f; }
    const bool lok = xx > 0, rok = xx + 1 < W;
    const unsigned mL = lok ? ~0u : 0u, mR = rok ? ~0u : 0u;
    const int offL = lok ? -C : 0, offR = rok ? C : 0;
    const int64_t rs = (int64_t)W * C;
    const int64_t off = (((int64_t)blockIdx.z * H + y0) * W + xx) * C + c;
    const T* px = x + off;                                 // row y0 of this thread's column
    T* py = y + off;
    const int yend = min(H, y0 + g.ty);
    const int ylast = min(H - 1, yend);                    // last row any output of the strip reads
    float win[3][3][NE];
    V ring[DW_FWD_RING][3];
    {
        V raw[3];
        unsigned k = y0 > 0 ? ~0u : 0u;
        dw_issue_row<T, NE>(px - (y0 > 0 ? rs : 0), offL, offR, raw); dw_cvt_row<T, NE>(raw, mL & k, k, mR & k, win[0]);
        dw_issue_row<T, NE>(px, offL, offR, raw); dw_cvt_row<T, NE>(raw, mL, ~0u, mR, win[1]);
        k = y0 + 1 < H ? ~0u : 0u;
        dw_issue_row<T, NE>(px + (y0 + 1 < H ? rs : 0), offL, offR, raw); dw_cvt_row<T, NE>(raw, mL & k, k, mR & k, win[2]);
#pragma unroll
        for (int j = 0; j < DW_FWD_RING; ++j) dw_issue_row<T, NE>(px + (min(y0 + 2 + j, ylast) - y0) * rs, offL, offR, ring[(2 + j) % DW_FWD_RING]);
    }
    // Branch-free steps (a row past the strip's end is computed from clamped loads and not stored), DW_UNROLL rows per loop
    // iteration: the compiler's wait counts are exact inside an iteration and conservative only across the back edge.
    for (int yy = y0; yy < yend; yy += DW_FWD_UNROLL) {
#pragma unroll
        for (int i = 0; i < DW_FWD_UNROLL; ++i) {           // window rows rotate through the three slots: no register moves
            const int yc = yy + i;
            float u[NE];
            dw_taps<NE>(win[i % 3], win[(i + 1) % 3], win[(i + 2) % 3], wt, bv, u);
            V o;
            if (ACT == 1) {
                float cdf[NE], ez[NE];
                gelu_parts_v<NE>(u, cdf, ez);
#pragma unroll
                for (int e = 0; e < NE; ++e) o[e] = (T)(u[e] * cdf[e]);
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) o[e] = (T)u[e];
            }
            if (yc < yend) *(V*)(py + i * rs) = o;
            const unsigned k = yc + 2 < H ? ~0u : 0u;
            dw_cvt_row<T, NE>(ring[(i + 2) % DW_FWD_RING], mL & k, k, mR & k, win[i % 3]);         // row yc+2 enters the window
            dw_issue_row<T, NE>(px + (min(yc + 2 + DW_FWD_RING, ylast) - yy) * rs, offL, offR, ring[(i + 2) % DW_FWD_RING]);  // the next row starts travelling
        }
        px += DW_FWD_UNROLL * rs; py += DW_FWD_UNROLL * rs;
    }
}

// g = dy * act'(u);  partial[blockIdx.x][10][C]: taps 0..8 = sum g * x(tap), 9 = sum g   (this workgroup's channel group only)
template <typename T, int NE, int ACT>
__global__ __launch_bounds__(256) void dwconv3_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ gout, float* __restrict__ partial, int B, int H, int W, int C, DwGeom g) {
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
        const int64_t rs = (int64_t)W * C;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int img = tile / tiles_per_img, rem = tile % tiles_per_img;
            const int xx = (rem % g.xtiles) * g.xl + xs, y0 = (rem / g.xtiles) * g.ty;
            if (xx >= W) continue;
            const bool lok = xx > 0, rok = xx + 1 < W;
            const unsigned mL = lok ? ~0u : 0u, mR = rok ? ~0u : 0u;
            const int offL = lok ? -C : 0, offR = rok ? C : 0;
            const int64_t off = (((int64_t)img * H + y0) * W + xx) * C + c;
            const T* px = x + off;
            const T* pd = dy + off;
            T* pg = gout + off;
            const int yend = min(H, y0 + g.ty);
            const int ylast = min(H - 1, yend);
            float win[3][3][NE];
            V ring[3][3], dring[3];
            {
                V raw[3];
                unsigned k = y0 > 0 ? ~0u : 0u;
                dw_issue_row<T, NE>(px - (y0 > 0 ? rs : 0), offL, offR, raw); dw_cvt_row<T, NE>(raw, mL & k, k, mR & k, win[0]);
                dw_issue_row<T, NE>(px, offL, offR, raw); dw_cvt_row<T, NE>(raw, mL, ~0u, mR, win[1]);
                k = y0 + 1 < H ? ~0u : 0u;
                dw_issue_row<T, NE>(px + (y0 + 1 < H ? rs : 0), offL, offR, raw); dw_cvt_row<T, NE>(raw, mL & k, k, mR & k, win[2]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    dw_issue_row<T, NE>(px + (min(y0 + 2 + j, ylast) - y0) * rs, offL, offR, ring[(2 + j) % 3]);
                    dring[j] = raw_load<T, NE>(pd + (min(y0 + j, yend - 1) - y0) * rs);
                }
            }
            for (int yy = y0; yy < yend; yy += DW_UNROLL) {
#pragma unroll
                for (int i = 0; i < DW_UNROLL; ++i) {
                    const int yc = yy + i;
                    float u[NE], go[NE];
                    dw_taps<NE>(win[i % 3], win[(i + 1) % 3], win[(i + 2) % 3], wt, bv, u);
                    union { V v; unsigned w[sizeof(V) / 4]; } d;
                    d.v = dring[i % 3];
                    const unsigned live = yc < yend ? ~0u : 0u;          // a row past the strip's end contributes nothing
#pragma unroll
                    for (int j = 0; j < (int)(sizeof(V) / 4); ++j) d.w[j] &= live;
                    dring[i % 3] = raw_load<T, NE>(pd + (min(yc + 3, yend - 1) - yy) * rs);   // row yc+3 of dy starts travelling
                    V gv;
                    float dact[NE];
                    if (ACT == 1) {
                        float cdf[NE], ez[NE];
                        gelu_parts_v<NE>(u, cdf, ez);
#pragma unroll
                        for (int e = 0; e < NE; ++e) dact[e] = fmaf(u[e] * 0.3989422804014327f, ez[e], cdf[e]);
                    }
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        go[e] = ACT == 1 ? (float)d.v[e] * dact[e] : (float)d.v[e];
                        gv[e] = (T)go[e];
                        go[e] = (float)gv[e];             // dW uses the stored (rounded) g, as a separate pass over g would
                        acc[9][e] += go[e];
                    }
                    if (yc < yend) *(V*)(pg + i * rs) = gv;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            acc[kx][e] = fmaf(go[e], win[i % 3][kx][e], acc[kx][e]);
                            acc[3 + kx][e] = fmaf(go[e], win[(i + 1) % 3][kx][e], acc[3 + kx][e]);
                            acc[6 + kx][e] = fmaf(go[e], win[(i + 2) % 3][kx][e], acc[6 + kx][e]);
                        }
                    const unsigned k = yc + 2 < H ? ~0u : 0u;
                    dw_cvt_row<T, NE>(ring[(i + 2) % 3], mL & k, k, mR & k, win[i % 3]);
                    dw_issue_row<T, NE>(px + (min(yc + 5, ylast) - yy) * rs, offL, offR, ring[(i + 2) % 3]);
                }
                px += DW_UNROLL * rs; pd += DW_UNROLL * rs; pg += DW_UNROLL * rs;
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

// ------------------------------------------------------------------ fused backward (data gradient + weight gradient in one pass)
// The two-pass backward above writes g = dy * act'(u) and a second forward-style pass reads it back for dx = dw_flipped(g):
// 5 tensor passes where 3 suffice (dy and x in, dx out).  Here a workgroup owns XL image columns (the outer two are halo) of
// one channel group and walks down a strip; a thread still keeps the 3x3 window of x in registers for ITS column, computes
// g for its column and row r, hands it to its two column neighbours through LDS (one barrier per row, two row buffers), and
// keeps the last three rows of g for the three columns in registers: dx[r-2] = sum_ky,kx w[ky][kx] g[r-1-ky][x+1-kx] follows
// two rows behind (the neighbours' g is read from LDS one step before it is used).  g is rounded to the storage dtype before it is used, as the stored g of the two-pass form is, and never
// reaches memory.  Halo rows (y0-1 and yend) and halo columns are computed twice (by both neighbours) and counted once.
#ifndef DWF_UNR
#define DWF_UNR 3
#endif
#ifndef DWF_NE
#define DWF_NE 4
#endif
struct DwFGeom { int cpb, xl, groups, xtiles, ystrips, chunks, ty; };
static inline DwFGeom dwf_geom(int B, int H, int W, int C, int ne) {
    DwFGeom g; g.chunks = C / ne;
    // XL columns per workgroup, XL - 2 of them produce output: wide enough to keep the halo share small, narrow enough that the
    // chunk lanes of one image column still cover whole 128-byte lines where C allows it
    int xl = 32;
    while (xl > 8 && (256 / xl) * 2 <= g.chunks && (256 / xl) * ne * 2 < 128) xl >>= 1;    // whole 128-byte lines (64 bf16 channels) per pixel if C has them
    while (xl > 4 && 256 / xl < 1) xl >>= 1;
    // narrow maps: 30 useful columns of 32 cover W = 64 with a third of the lanes idle (3 x 32 for 64); 14 of 16 waste less there
    if (xl == 32 && g.chunks >= 16 && cdiv(W, 14) * 16 < cdiv(W, 30) * 32) xl = 16;
    int cpb = 256 / xl;
    while (cpb > 1 && cpb / 2 >= g.chunks) { cpb >>= 1; xl <<= 1; }                          // few channels: spend the lanes on columns
    g.cpb = cpb; g.xl = xl; g.groups = (int)cdiv(g.chunks, cpb);
    g.xtiles = (int)cdiv(W, xl - 2);
    int best = 2 * DW_UNROLL; int64_t best_cost = -1;
    for (int ty = 2 * DW_UNROLL; ty <= 8 * DW_UNROLL; ty += DW_UNROLL) {
        const int64_t strips = cdiv(H, ty), cost = strips * (ty + 8);
        const int64_t blocks = strips * g.xtiles * g.groups * B;
        if (ty > 2 * DW_UNROLL && blocks < 1024) break;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = ty; }
    }
    g.ty = best; g.ystrips = (int)cdiv(H, g.ty);
    return g;
}

template <typename T, int NE, int ACT, int OCC, int UNR>
__global__ __launch_bounds__(256, OCC) void dwconv3_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ dxo, float* __restrict__ partial, int B, int H, int W, int C, DwFGeom g) {
    typedef vecn<T, NE> V;
    __shared__ float shf[256 * NE > 2 * 256 * NE * (int)sizeof(T) / 4 ? 256 * NE : 2 * 256 * NE * (int)sizeof(T) / 4];
    V* gx = (V*)shf;                                        // [2][256] rows of g in flight between column neighbours
    constexpr bool LUT = DWF_LUT && ACT == 1 && sizeof(T) == 2;
    __shared__ float2 lut[LUT ? GELU_LUT_N : 1];
    const int tid = threadIdx.x, cl = tid % g.cpb, xs = tid / g.cpb;
    if (LUT) { gelu_grad_lut_fill(lut, tid, 256); __syncthreads(); }
    const int chunk_raw = blockIdx.y * g.cpb + cl;
    const bool cok = chunk_raw < g.chunks;
    const int chunk = cok ? chunk_raw : g.chunks - 1, c = chunk * NE;
    const int tl = xs > 0 ? tid - g.cpb : tid, tr = xs + 1 < g.xl ? tid + g.cpb : tid;      // LDS slots of the column neighbours
    float wt[9][NE], bv[NE], acc[10][NE];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[k][e] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) loadf<NE>(w + k * C + c, wt[k]);
    loadf<NE>(bias + c, bv);
    const int tiles_per_img = g.xtiles * g.ystrips, ntiles = B * tiles_per_img;
    const int64_t rs = (int64_t)W * C;
    int par = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tiles_per_img, rem = tile % tiles_per_img;
        const int xx = (rem % g.xtiles) * (g.xl - 2) - 1 + xs, y0 = (rem / g.xtiles) * g.ty;
        const bool colok = xx >= 0 && xx < W;
        const bool inner = colok && cok && xs > 0 && xs + 1 < g.xl;          // this lane's dx / dW are its own
        const int xc = xx < 0 ? 0 : (xx < W ? xx : W - 1);
        const bool lok = colok && xx > 0, rok = colok && xx + 1 < W;
        const unsigned mC = colok ? ~0u : 0u, mL = lok ? ~0u : 0u, mR = rok ? ~0u : 0u;
        const int offL = lok ? -C : 0, offR = rok ? C : 0;
        const int yend = min(H, y0 + g.ty);
        const int ylast = min(H - 1, yend + 1);                  // last row of x any g of the strip reads
        const int dlast = min(H - 1, yend);                      // last row of dy
        const int rb = max(y0 - 1, 0);                           // row the base pointers stand on
        const int64_t off = (((int64_t)img * H + rb) * W + xc) * C + c;
        const T* px = x + off;
        const T* pd = dy + off;
        T* po = dxo + off;
        // g rows r = y0-1 .. yend, one per step; x rows r-1, r, r+1 in the window, r+2 .. r+4 travelling, dy rows r .. r+2 travelling
        const int r0 = y0 - 1;
        float win[3][3][NE], gw[3][3][NE];
        V ring[3][3], dring[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int e = 0; e < NE; ++e) gw[a][b][e] = 0.f;
        {
            V raw[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {                         // rows r0-1, r0, r0+1
                const int rr = r0 - 1 + j;
                const unsigned k = (rr >= 0 && rr < H) ? ~0u : 0u;
                dw_issue_row<T, NE>(px + (int64_t)(min(max(rr, 0), ylast) - rb) * rs, offL, offR, raw);
                dw_cvt_row<T, NE>(raw, mL & k, mC & k, mR & k, win[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dw_issue_row<T, NE>(px + (int64_t)(min(max(r0 + 2 + j, 0), ylast) - rb) * rs, offL, offR, ring[(2 + j) % 3]);
                dring[j] = raw_load<T, NE>(pd + (int64_t)(min(max(r0 + j, 0), dlast) - rb) * rs);
            }
        }
        const int nsteps = yend - y0 + 3;                        // rows y0-1 .. yend of g, and one more step: dx trails g by two rows
        V gl, gr;
#pragma unroll
        for (int e = 0; e < NE; ++e) { gl[e] = (T)0.f; gr[e] = (T)0.f; }
        DW_DRAIN_PROLOGUE();
        for (int s0 = 0; s0 < nsteps; s0 += UNR) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int r = r0 + s0 + i;                        // the row of g this step produces
                float u[NE], go[NE];
                dw_taps<NE>(win[i % 3], win[(i + 1) % 3], win[(i + 2) % 3], wt, bv, u);
                union { V v; unsigned w[sizeof(V) / 4]; } d;
                d.v = dring[i % 3];
                const unsigned live = (r >= 0 && r <= dlast && r <= yend && colok) ? ~0u : 0u;     // g is zero outside the image
#pragma unroll
                for (int j = 0; j < (int)(sizeof(V) / 4); ++j) d.w[j] &= live;
                dring[i % 3] = raw_load<T, NE>(pd + (int64_t)(min(max(r + 3, 0), dlast) - rb) * rs);
                V gv;
                float dact[NE];
                if (LUT) gelu_grad_lut<NE>(lut, u, dact);
                else if (ACT == 1) {
                    float cdf[NE], ez[NE];
                    gelu_parts_v<NE>(u, cdf, ez);
#pragma unroll
                    for (int e = 0; e < NE; ++e) dact[e] = fmaf(u[e] * 0.3989422804014327f, ez[e], cdf[e]);
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    go[e] = ACT == 1 ? (float)d.v[e] * dact[e] : (float)d.v[e];
                    gv[e] = (T)go[e];
                    go[e] = (float)gv[e];                         // the rounded g, as a stored g would be read back
                }
                gx[par * 256 + tid] = gv;
                // While the row travels: dx of row r-2 from g rows r-1 (ky = 0), r-2 (ky = 1), r-3 (ky = 2), column x+1-kx.  The
                // neighbours' g of row r-1 was read from LDS at the end of the previous step (gl / gr), a whole step of arithmetic
                // ago, so nothing here waits for the LDS round trip.
                {
#pragma unroll
                    for (int e = 0; e < NE; ++e) { gw[(i + 2) % 3][0][e] = (float)gl[e]; gw[(i + 2) % 3][2][e] = (float)gr[e]; }
                    float o[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) o[e] = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            o[e] = fmaf(wt[kx][e], gw[(i + 2) % 3][2 - kx][e], o[e]);
                            o[e] = fmaf(wt[3 + kx][e], gw[(i + 1) % 3][2 - kx][e], o[e]);
                            o[e] = fmaf(wt[6 + kx][e], gw[i % 3][2 - kx][e], o[e]);
                        }
                    V ov;
#pragma unroll
                    for (int e = 0; e < NE; ++e) ov[e] = (T)o[e];
                    if (inner && r - 2 >= y0 && r - 2 < yend) *(V*)(po + (int64_t)(r - 2 - rb) * rs) = ov;
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) gw[i % 3][1][e] = go[e];          // slot of row r-3 is free now: row r moves in
                // weight / bias gradient: own rows and own columns only (halo g belongs to the neighbouring workgroup)
                const float own = (inner && r >= y0 && r < yend) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < NE; ++e) { const float gm = go[e] * own; acc[9][e] += gm;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[kx][e] = fmaf(gm, win[i % 3][kx][e], acc[kx][e]);
                        acc[3 + kx][e] = fmaf(gm, win[(i + 1) % 3][kx][e], acc[3 + kx][e]);
                        acc[6 + kx][e] = fmaf(gm, win[(i + 2) % 3][kx][e], acc[6 + kx][e]);
                    } }
                __syncthreads();
                gl = gx[par * 256 + tl]; gr = gx[par * 256 + tr];                // consumed by the next step
                par ^= 1;
                const unsigned k = r + 2 < H ? ~0u : 0u;          // r + 2 >= y0 + 1 > 0
                dw_cvt_row<T, NE>(ring[(i + 2) % 3], mL & k, mC & k, mR & k, win[i % 3]);
                dw_issue_row<T, NE>(px + (int64_t)(min(max(r + 5, 0), ylast) - rb) * rs, offL, offR, ring[(i + 2) % 3]);
            }
        }
    }
    // combine the x lanes of each chunk lane through LDS in a fixed order
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) shf[tid * NE + e] = acc[k][e];
        __syncthreads();
        if (xs == 0 && cok) {
            float* dst = partial + ((int64_t)blockIdx.x * 10 + k) * C + c;
#pragma unroll
            for (int e = 0; e < NE; ++e) { float sum = 0.f;
                for (int j = 0; j < g.xl; ++j) sum += shf[(j * g.cpb + cl) * NE + e];
                dst[e] = sum; }
        }
    }
}

// ------------------------------------------------------------------ fused backward, a lane per image column (bf16, C % 64 == 0)
// The kernel above hands every row of g to the column neighbours through LDS behind a workgroup barrier and fetches the left / right
// window entries with loads of their own.  Here the 16 lanes of a DPP row ARE 16 adjacent image columns (tid = chunk lane * 16 +
// column lane; a wave = 16 columns x 4 chunks of 4 channels): the neighbours' x and g are row_shr:1 / row_shl:1 of a lane's own value,
// the row loop has no barrier and no LDS traffic, and a row costs two loads (the centre, and the column beyond the tile edge for
// lanes 0 and 15 -- the same address as the centre for the lanes between, a cache hit) instead of three.  A wave instruction touches
// 32 bytes of 16 pixels; the four waves of the workgroup read the four quarters of the same 128-byte lines.  Arithmetic and its order
// are those of the kernel above (same results bit for bit).
//
// KS > 0: `dy` is not the gradient of the depthwise output but dout [pixels, 32 KS], the gradient of the Linear that follows it
// (Mix-FFN fc2, modules/segformer.py:58-63), and the kernel forms its row of da2 = (dout . W2) * keep itself: this thread layout is
// the accumulator layout of v_mfma_f32_16x16x32_bf16 with A = 16 rows of W2^T (channels) and B = 16 pixels of dout -- lane (column
// j, k group) receives channels 4 kg .. 4 kg + 3 of column j.  The [pixels, 4C] tensor da2 (537 MB per block at MiT stage 1) is never
// written or read; dout is re-read by the C / 64 channel groups out of L2.
__device__ __forceinline__ float lane_left(float v, float edge) {    // value of lane - 1 within the row of 16; lane 0 receives `edge`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v, float edge) {   // value of lane + 1; lane 15 receives `edge`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
// the same with zero beyond the row of 16 (no register to preset)
__device__ __forceinline__ float lane_left0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
template <int NE>
__device__ __forceinline__ void dwl_cvt_row(const vecn<bf16, NE> (&raw)[2], unsigned keepC, unsigned keepO, float (&row)[3][NE]) {
    union { vecn<bf16, NE> v; unsigned u[NE / 2]; } c, o;
    c.v = raw[0]; o.v = raw[1];
#pragma unroll
    for (int i = 0; i < NE / 2; ++i) { c.u[i] &= keepC; o.u[i] &= keepO; }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const float cv = (float)c.v[e], ov = (float)o.v[e];
        row[1][e] = cv; row[0][e] = lane_left(cv, ov); row[2][e] = lane_right(cv, ov);
    }
}

// KS > 0: everything the row loop reads from memory arrives by LDS-DMA in wave-private queues of three slots, issued by hand
// (gemm_fast.h) -- the register file has no room for operands in flight (the kernel sits at 252 VGPRs without them), and with the
// builtin hipcc treats a pending LDS-DMA as a flat access and turns every wait of the loop into vmcnt(0).
//   x row:    one instruction per wave and row; lane l < 32 fetches 16 bytes (8 channels) of pixel l >> 1, lanes 32 .. 35 the two
//             columns beyond the tile edges; a lane reads back its own 8 bytes and the edge column with ds_read_b64.
//   dout row: KS instructions; instruction s leaves lane l's B fragment (pixel l & 15, k = 32 s + 8 (l >> 4) ..) at [s][l], where the
//             same lane reads it back.
// A step issues [x, dout x KS] in this order and nothing else but the dx store, so the waits are exact counts of what was issued since
// (a store in between makes a wait stricter, never laxer).  `dep` is a register the fill must not overtake: a value read from the slot
// being refilled, which makes the compiler wait for that LDS read before the fill is issued.
__device__ __forceinline__ void dwl_dma(const void* g, unsigned lds_uniform, unsigned dep) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_uniform), "v"(dep) : "memory");
}
template <int N> __device__ __forceinline__ void dwl_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int ACT, int KS, int DEPTH>
__global__ __launch_bounds__(256, 2) void dwconv3_bwd_lane_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, bf16* __restrict__ dxo, float* __restrict__ partial, int B, int H, int W, int C, DwFGeom g,
        const bf16* __restrict__ w2t, const float* __restrict__ keep) {
    constexpr int NE = 4, K = 32 * KS;
    typedef vecn<bf16, NE> V;
    constexpr bool LUT = DWF_LUT && ACT == 1;
    __shared__ float2 lut[LUT ? GELU_LUT_N : 1];
    const int tid = threadIdx.x, xs = tid & 15, cl = tid >> 4, kg = cl & 3;
    if (LUT) { gelu_grad_lut_fill(lut, tid, 256); __syncthreads(); }
    const int c = (blockIdx.y * 16 + cl) * NE;                  // C % 64 == 0: every chunk lane has its channels
    float wt[9][NE], bv[NE], acc[10][NE];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[k][e] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) loadf<NE>(w + k * C + c, wt[k]);
    loadf<NE>(bias + c, bv);
    // A fragments (row = channel blockIdx.y * 64 + 16 wave + xs, k = 32 s + 8 kg ..) wait in LDS in fragment order: a ds_read_b128 per
    // step and fragment instead of 4 KS registers held for the whole kernel
    __shared__ bf16x8 wfs[KS > 0 ? KS : 1][256];
    if (KS > 0) {
        const bf16* wr = w2t + (int64_t)(blockIdx.y * 64 + (cl & ~3) * NE + xs) * K + 8 * kg;
#pragma unroll
        for (int s = 0; s < KS; ++s) wfs[s][tid] = *(const bf16x8*)(wr + 32 * s);
    }
    constexpr int NB = KS > 0 ? KS : 1;                       // fills per row of the dy / dout queue
    __shared__ bf16x8 bst[4 * 3 * NB][64];                    // [wave][ring slot][s][lane]; KS = 0: the dy row, laid out like the x row
    constexpr int XSLOT = 640;                                // bytes per x slot: 36 lanes x 16, rounded up
    __shared__ __attribute__((aligned(16))) char xq[4 * 3 * XSLOT];                     // [wave][ring slot][lane]
    float* shf = (float*)&bst[0][0];                          // the final combine reuses the dy / dout queue (drained first)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned bbase = lds_addr(&bst[wave * 3 * NB][0]), xbase = lds_addr(xq + wave * 3 * XSLOT);
    const V* xqw = (const V*)(xq + wave * 3 * XSLOT);         // XSLOT / 8 V per slot: centre of lane (xs, kg) at xs * 4 + kg, edge columns from 64
    const V* dqw = (const V*)&bst[wave * 3 * NB][0];          // KS = 0: 128 V per slot, same order
    const int xci = xs * 4 + kg, xoi = 64 + (xs == 15 ? 4 : 0) + kg;
    const int tiles_per_img = g.xtiles * g.ystrips, ntiles = B * tiles_per_img;
    const int64_t rs = (int64_t)W * C, rsd = (int64_t)W * (KS > 0 ? K : C);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tiles_per_img, rem = tile % tiles_per_img;
        const int xx = (rem % g.xtiles) * 14 - 1 + xs, y0 = (rem / g.xtiles) * g.ty;
        const bool colok = xx >= 0 && xx < W;
        const bool inner = colok && xs > 0 && xs < 15;                       // this lane's dx / dW are its own
        const int xc = xx < 0 ? 0 : (xx < W ? xx : W - 1);
        const bool ook = colok && (xs == 0 ? xx > 0 : xs == 15 ? xx + 1 < W : false);   // the column beyond the tile edge exists
        const unsigned mC = colok ? ~0u : 0u, mO = ook ? ~0u : 0u;
        const int offO = ook ? (xs == 0 ? -C : C) : 0;
        const int yend = min(H, y0 + g.ty);
        const int ylast = min(H - 1, yend + 1);                  // last row of x any g of the strip reads
        const int dlast = min(H - 1, yend);                      // last row of dy
        const int rb = max(y0 - 1, 0);                           // row the base pointers stand on
        const int64_t pix = ((int64_t)img * H + rb) * W + xc, off = pix * C + c;
        const bf16* px = x + off;
        const bf16* pd = dy + pix * K + 8 * kg;
        bf16* po = dxo + off;
        const float kscale = (KS > 0 && keep) ? keep[img] : 1.f;
        const int r0 = y0 - 1;
        // the fill lane's own address: 16 bytes of a centre pixel (lanes 0 .. 31) or of an edge column (32 .. 35)
        const int t0 = (rem % g.xtiles) * 14 - 1;
        const int qc = lane < 32 ? t0 + (lane >> 1) : (lane < 34 ? t0 - 1 : t0 + 16);
        const int64_t qoff = (((int64_t)img * H + rb) * W + min(max(qc, 0), W - 1)) * C + blockIdx.y * 64 + wave * 16 + (lane & 1) * 8;
        const bf16* pq = x + qoff;
        if (KS == 0) pd = dy + qoff;                              // dy rows are fetched like x rows (lanes 0 .. 31)
        float win[3][3][NE], gw[3][3][NE];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int e = 0; e < NE; ++e) gw[a][b][e] = 0.f;
        {
            V raw[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) {                         // rows r0-1, r0, r0+1
                const int rr = r0 - 1 + j;
                const unsigned k = (rr >= 0 && rr < H) ? ~0u : 0u;
                const bf16* p = px + (int64_t)(min(max(rr, 0), ylast) - rb) * rs;
                raw[0] = raw_load<bf16, NE>(p); raw[1] = raw_load<bf16, NE>(p + offO);
                dwl_cvt_row<NE>(raw, mC & k, mO & k, win[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int64_t xo = (int64_t)(min(max(r0 + 2 + j, 0), ylast) - rb) * rs;
                const bf16* q = pd + (int64_t)(min(max(r0 + j, 0), dlast) - rb) * rsd;
                if (lane < 36) dwl_dma(pq + xo, xbase + ((2 + j) % 3) * XSLOT, 0u);
                if (KS == 0) { if (lane < 32) dwl_dma(q, bbase + j * 1024, 0u); }
                else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) dwl_dma(q + 32 * s, bbase + (j * KS + s) * 1024, 0u);
                }
            }
        }
        const int nsteps = yend - y0 + 3;                        // rows y0-1 .. yend of g, and one more step: dx trails g by two rows
        constexpr int UNR = 3;
        for (int s0 = 0; s0 < nsteps; s0 += UNR) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int r = r0 + s0 + i;                        // the row of g this step produces
                float u[NE], go[NE];
                union { V v; unsigned w[NE / 2]; } d;
                dwl_wait<2 + 2 * NB>();                             // the dy / dout row issued three steps ago: two steps of [x, NB fills] since
                if (KS > 0) {
                    floatx4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfs[s][tid], bst[(wave * 3 + i % 3) * KS + s][lane], a4, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < NE; ++e) d.v[e] = (bf16)(a4[e] * kscale);       // da2 as the GEMM would have stored it
                } else d.v = dqw[(i % 3) * 128 + xci];
                dw_taps<NE>(win[i % 3], win[(i + 1) % 3], win[(i + 2) % 3], wt, bv, u);
                const unsigned live = (r >= 0 && r <= dlast && r <= yend && colok) ? ~0u : 0u;     // g is zero outside the image
#pragma unroll
                for (int j = 0; j < NE / 2; ++j) d.w[j] &= live;
                float dact[NE];
                if (LUT) gelu_grad_lut<NE>(lut, u, dact);
                else if (ACT == 1) {
                    float cdf[NE], ez[NE];
                    gelu_parts_v<NE>(u, cdf, ez);
#pragma unroll
                    for (int e = 0; e < NE; ++e) dact[e] = fmaf(u[e] * 0.3989422804014327f, ez[e], cdf[e]);
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    go[e] = ACT == 1 ? (float)d.v[e] * dact[e] : (float)d.v[e];
                    go[e] = (float)(bf16)go[e];                   // the rounded g, as a stored g would be read back
                }
                // dx of row r-2 from g rows r-1 (ky = 0), r-2 (ky = 1), r-3 (ky = 2), column x+1-kx
                {
                    float o[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) o[e] = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            o[e] = fmaf(wt[kx][e], gw[(i + 2) % 3][2 - kx][e], o[e]);
                            o[e] = fmaf(wt[3 + kx][e], gw[(i + 1) % 3][2 - kx][e], o[e]);
                            o[e] = fmaf(wt[6 + kx][e], gw[i % 3][2 - kx][e], o[e]);
                        }
                    V ov;
#pragma unroll
                    for (int e = 0; e < NE; ++e) ov[e] = (bf16)o[e];
                    if (inner && r - 2 >= y0 && r - 2 < yend) *(V*)(po + (int64_t)(r - 2 - rb) * rs) = ov;
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {                    // slot of row r-3 is free now: row r moves in, with its neighbours
                    gw[i % 3][1][e] = go[e]; gw[i % 3][0][e] = lane_left0(go[e]); gw[i % 3][2][e] = lane_right0(go[e]);
                }
                // weight / bias gradient: own rows and own columns only (halo g belongs to the neighbouring workgroup)
                const float own = (inner && r >= y0 && r < yend) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < NE; ++e) { const float gm = go[e] * own; acc[9][e] += gm;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[kx][e] = fmaf(gm, win[i % 3][kx][e], acc[kx][e]);
                        acc[3 + kx][e] = fmaf(gm, win[(i + 1) % 3][kx][e], acc[3 + kx][e]);
                        acc[6 + kx][e] = fmaf(gm, win[(i + 2) % 3][kx][e], acc[6 + kx][e]);
                    } }
                const unsigned k = r + 2 < H ? ~0u : 0u;          // r + 2 >= y0 + 1 > 0
                const int64_t xo = (int64_t)(min(max(r + 5, 0), ylast) - rb) * rs;
                dwl_wait<2 + 3 * NB>();                             // the x row issued three steps ago, the NB fills after it and two whole steps since
                union { V v[2]; unsigned w[NE]; } xr;
                xr.v[0] = xqw[((i + 2) % 3) * (XSLOT / 8) + xci]; xr.v[1] = xqw[((i + 2) % 3) * (XSLOT / 8) + xoi];
                dwl_cvt_row<NE>(xr.v, mC & k, mO & k, win[i % 3]);
                if (lane < 36) dwl_dma(pq + xo, xbase + ((i + 2) % 3) * XSLOT, xr.w[0] | xr.w[2]);
                const bf16* q = pd + (int64_t)(min(max(r + 3, 0), dlast) - rb) * rsd;
                if (KS == 0) { if (lane < 32) dwl_dma(q, bbase + (i % 3) * 1024, d.w[0]); }
                else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) dwl_dma(q + 32 * s, bbase + ((i % 3) * KS + s) * 1024, d.w[0]);
                }
            }
        }
    }
    // combine the column lanes of each chunk lane through LDS in a fixed order
    dwl_wait<0>();                                            // fills issued past the last strip still land in the queue the combine reuses
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) shf[tid * NE + e] = acc[k][e];
        __syncthreads();
        if (xs == 0) {
            float* dst = partial + ((int64_t)blockIdx.x * 10 + k) * C + c;
#pragma unroll
            for (int e = 0; e < NE; ++e) { float sum = 0.f;
                for (int j = 0; j < 16; ++j) sum += shf[(cl * 16 + j) * NE + e];
                dst[e] = sum; }
        }
    }
}

extern "C" int ss_dwconv3_fwd(int dtype, const void* x, const float* w9c, const float* bias, void* y, int B, int H, int W, int C,
                              int act, hipStream_t stream) {
    SS_CHECK(x && w9c && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_fwd: bad arguments (C=%d)", C);
    SS_CHECK(aligned16(x) && aligned16(y) && aligned16(w9c) && (!bias || aligned16(bias)), "ss_dwconv3_fwd: pointers must be 16-byte aligned");
    DwGeom g = dw_geom(B, H, W, C, dtype == SS_BF16 ? 8 : 4, DW_FWD_UNROLL);
    SS_CHECK((int64_t)g.xtiles * g.ystrips < 2147483647LL && B < 65536 && g.groups < 65536, "ss_dwconv3_fwd: grid too large");
    dim3 grid((unsigned)(g.xtiles * g.ystrips), g.groups, B);
    SS_CHECK(act == 0 || act == 1, "ss_dwconv3_fwd: act must be 0 (none) or 1 (GELU)");
#define SS_DW_FWD(T, NE, ACT) hipLaunchKernelGGL((dwconv3_fwd_kernel<T, NE, ACT>), grid, dim3(256), 0, stream, (const T*)x, w9c, bias, (T*)y, H, W, C, g)
    if (dtype == SS_BF16) { if (act) SS_DW_FWD(bf16, 8, 1); else SS_DW_FWD(bf16, 8, 0); }
    else if (dtype == SS_F32) { if (act) SS_DW_FWD(float, 4, 1); else SS_DW_FWD(float, 4, 0); }
    else return ss_fail(-1, "ss_dwconv3_fwd: unsupported dtype %d", dtype);
#undef SS_DW_FWD
    SS_LAUNCH_CHECK();
    return 0;
}

// dw_db: [10][C] fp32 (9 taps tap-major, then bias).  workspace: 1024 * 10 * C floats.
extern "C" int ss_dwconv3_bwd(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* g, float* dw_db,
                              int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && w9c && bias && g && dw_db && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_bwd: bad arguments");
    SS_CHECK(aligned16(x) && aligned16(dy) && aligned16(g) && aligned16(w9c) && aligned16(bias), "ss_dwconv3_bwd: pointers must be 16-byte aligned");
    DwGeom geo = dw_geom(B, H, W, C, 4);
    const int64_t ntiles = (int64_t)B * geo.xtiles * geo.ystrips;
    SS_CHECK(ntiles < 2147483647LL && geo.groups < 65536, "ss_dwconv3_bwd: grid too large");
    int64_t nbg = 1024 / geo.groups; if (nbg < 1) nbg = 1; if (nbg > ntiles) nbg = ntiles;
    SS_CHECK(ws_bytes >= nbg * 10 * C * 4, "ss_dwconv3_bwd: workspace too small");
    dim3 grid((unsigned)nbg, geo.groups);
    SS_CHECK(act == 0 || act == 1, "ss_dwconv3_bwd: act must be 0 (none) or 1 (GELU)");
#define SS_DW_BWD(T, NE, ACT) hipLaunchKernelGGL((dwconv3_bwd_kernel<T, NE, ACT>), grid, dim3(256), 0, stream, (const T*)dy, (const T*)x, w9c, bias, (T*)g, workspace, B, H, W, C, geo)
    if (dtype == SS_BF16) { if (act) SS_DW_BWD(bf16, 4, 1); else SS_DW_BWD(bf16, 4, 0); }
    else if (dtype == SS_F32) { if (act) SS_DW_BWD(float, 4, 1); else SS_DW_BWD(float, 4, 0); }
    else return ss_fail(-1, "ss_dwconv3_bwd: unsupported dtype %d", dtype);
#undef SS_DW_BWD
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, (int)nbg, 10 * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// the lane-per-column kernel serves the geometry 16 columns x 16 chunk lanes with whole channel groups (SS_DW_LANE=0: the LDS-exchange kernel)
static inline bool dwl_ok(const DwFGeom& g, int C) {
    const char* e = getenv("SS_DW_LANE");
    return g.xl == 16 && g.cpb == 16 && C % 64 == 0 && !(e && e[0] == '0');
}

// dx and dw_db in one pass: dx = d/dx of sum(dy * act(dw(x) + b)), dw_db = [10][C] fp32.  workspace: 1024 * 10 * C floats.
extern "C" int ss_dwconv3_bwd_fused(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* dx, float* dw_db,
                                    int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dy && x && w9c && bias && dx && dw_db && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ss_dwconv3_bwd_fused: bad arguments");
    SS_CHECK(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(w9c) && aligned16(bias), "ss_dwconv3_bwd_fused: pointers must be 16-byte aligned");
    SS_CHECK(dx != dy && dx != x, "ss_dwconv3_bwd_fused: dx must not alias an input (neighbouring workgroups re-read the halo)");
    DwFGeom geo = dwf_geom(B, H, W, C, 4);
    const int64_t ntiles = (int64_t)B * geo.xtiles * geo.ystrips;
    SS_CHECK(ntiles < 2147483647LL && geo.groups < 65536, "ss_dwconv3_bwd_fused: grid too large");
    int64_t nbg = 1024 / geo.groups; if (nbg < 1) nbg = 1; if (nbg > ntiles) nbg = ntiles;
    SS_CHECK(ws_bytes >= nbg * 10 * C * 4, "ss_dwconv3_bwd_fused: workspace too small");
    dim3 grid((unsigned)nbg, geo.groups);
    SS_CHECK(act == 0 || act == 1, "ss_dwconv3_bwd_fused: act must be 0 (none) or 1 (GELU)");
    if (dtype == SS_BF16 && dwl_ok(geo, C)) {
        if (act) hipLaunchKernelGGL((dwconv3_bwd_lane_kernel<1, 0, 3>), grid, dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, w9c, bias, (bf16*)dx, workspace, B, H, W, C, geo, (const bf16*)nullptr, (const float*)nullptr);
        else hipLaunchKernelGGL((dwconv3_bwd_lane_kernel<0, 0, 3>), grid, dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, w9c, bias, (bf16*)dx, workspace, B, H, W, C, geo, (const bf16*)nullptr, (const float*)nullptr);
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, (int)nbg, 10 * C, 1.f, dw_db, 0);
        SS_LAUNCH_CHECK();
        return 0;
    }
    // fp32 (the parity path) keeps twice the bytes per window entry: one workgroup per SIMD pair there, two for bf16
#define SS_DW_BWDF(T, NE, ACT) hipLaunchKernelGGL((dwconv3_bwd_fused_kernel<T, NE, ACT, sizeof(T) == 2 ? 2 : 1, DWF_UNR>), grid, dim3(256), 0, stream, (const T*)dy, (const T*)x, w9c, bias, (T*)dx, workspace, B, H, W, C, geo)
    if (dtype == SS_BF16) { if (act) SS_DW_BWDF(bf16, DWF_NE, 1); else SS_DW_BWDF(bf16, DWF_NE, 0); }
    else if (dtype == SS_F32) { if (act) SS_DW_BWDF(float, 4, 1); else SS_DW_BWDF(float, 4, 0); }
    else return ss_fail(-1, "ss_dwconv3_bwd_fused: unsupported dtype %d", dtype);
#undef SS_DW_BWDF
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, (int)nbg, 10 * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

// The backward of Mix-FFN's depthwise stage fed by the gradient of fc2's OUTPUT (modules/segformer.py:58-63): da2 = (dout . W2) * keep is
// formed row by row on the matrix cores inside the lane kernel and never stored.  dout [B*H*W, K] bf16, w2t [C, K] bf16 (fc2's weight
// transposed: row = hidden channel), keep [B] fp32 or NULL, x = fc1's output [B*H*W, C].  ss_dwconv3_bwd_fc2_ok: 1 where it runs.
extern "C" int ss_dwconv3_bwd_fc2_ok(int dtype, int B, int H, int W, int C, int K) {
    if (dtype != SS_BF16 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 64 || !(K == 64 || K == 128)) return 0;
    const char* e = getenv("SS_DW_FC2");
    if (e && e[0] == '0') return 0;
    // K = 128 (MiT stage 2) is built and tested but loses to GEMM + lane kernel (493 against 444 us per block: four dout fills and eight
    // fragment reads per wave and row, and 80 KB of LDS per workgroup); SS_DW_FC2=2 runs it anyway
    if (K == 128 && !(e && e[0] == '2')) return 0;
    return dwl_ok(dwf_geom(B, H, W, C, 4), C) ? 1 : 0;
}
extern "C" int ss_dwconv3_bwd_fc2(int dtype, const void* dout, const void* w2t, const float* keep, const void* x, const float* w9c, const float* bias,
                                  void* dx, float* dw_db, int B, int H, int W, int C, int K, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(dout && w2t && x && w9c && bias && dx && dw_db && workspace, "ss_dwconv3_bwd_fc2: null argument");
    SS_CHECK(ss_dwconv3_bwd_fc2_ok(dtype, B, H, W, C, K), "ss_dwconv3_bwd_fc2: unsupported shape (bf16, C %% 64 == 0, K in {64, 128}; C=%d K=%d)", C, K);
    SS_CHECK(aligned16(x) && aligned16(dout) && aligned16(w2t) && aligned16(dx) && aligned16(w9c) && aligned16(bias), "ss_dwconv3_bwd_fc2: pointers must be 16-byte aligned");
    SS_CHECK(dx != x && dx != dout, "ss_dwconv3_bwd_fc2: dx must not alias an input");
    DwFGeom geo = dwf_geom(B, H, W, C, 4);
    const int64_t ntiles = (int64_t)B * geo.xtiles * geo.ystrips;
    SS_CHECK(ntiles < 2147483647LL && geo.groups < 65536, "ss_dwconv3_bwd_fc2: grid too large");
    int64_t nbg = 1024 / geo.groups; if (nbg < 1) nbg = 1; if (nbg > ntiles) nbg = ntiles;
    SS_CHECK(ws_bytes >= nbg * 10 * C * 4, "ss_dwconv3_bwd_fc2: workspace too small");
    dim3 grid((unsigned)nbg, geo.groups);
#define SS_DW_FC2(KS, DEPTH) hipLaunchKernelGGL((dwconv3_bwd_lane_kernel<1, KS, DEPTH>), grid, dim3(256), 0, stream, (const bf16*)dout, (const bf16*)x, w9c, bias, (bf16*)dx, workspace, B, H, W, C, geo, (const bf16*)w2t, keep)
    if (K == 64) SS_DW_FC2(2, 3); else SS_DW_FC2(4, 3);
#undef SS_DW_FC2
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(10 * C, 8), 1), dim3(256), 0, stream, workspace, (int)nbg, 10 * C, 1.f, dw_db, 0);
    SS_LAUNCH_CHECK();
    return 0;
}
