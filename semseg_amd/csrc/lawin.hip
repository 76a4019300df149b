// Lawin head: multi-scale large-window attention (reference modules/lawin.py:18-94,140-179).
//
// The reference materialises, per ratio r in {8,4,2}, an unfolded (8r x 8r) context patch per 8x8 window (64x/16x/4x
// the feature map) and pools it.  Here the context of ALL windows is 8x8 gathers from ONE pooled grid per r:
//   G_r[g] = 0.5*(max + avg) over feat rows [g*r - pad, g*r - pad + r),  pad = 4(r-1)  (zero padding participates),
//   window (i,j) reads cells (i*8/r + py, j*8/r + px)  -- bit-identical values, 1/64..1/4 of the traffic.
// Kernels (all deterministic, no atomics):
//   lawin_pool_fwd      feat -> G_r (+ arg-max position per cell for the backward)
//   lawin_mix           per-head position mixing  ctx' = ctx + Linear_{head}(ctx over the 64 positions)  (and its transpose)
//   lawin_gather_bwd    window-ordered d(ctx) -> dG_r (gather form over the <= 64 windows sharing a cell)
//   lawin_pool_bwd      dG_8, dG_4, dG_2 -> d(feat)  (avg share + arg-max share), one pass
//   lawin_mix_wgrad     position-mixing weight / bias gradients
//   lawin_attn_fwd/bwd  per (window, head) 64x64 softmax attention on the vector ALU (head dims 2..64 make MFMA tiles
//                       mostly padding; the op is bandwidth / VALU bound), one wave per (window, head), lane = query
//                       (and lane = key for the dphi/dg pass), phi/g/theta/dY rows broadcast from LDS.
#include "common.h"
#include <initializer_list>
#include "colreduce.h"

// bf16 MFMA window attention for head dims 16 / 32 / 64 (lawin_attn_mfma.hip)
bool lawin_attn_mfma_ok(int D, const void* theta, int64_t ldt, const void* pg, int inter, const void* p2, int64_t ld2, const void* p3,
                        int64_t ld3, const void* p4, int64_t ld4);
int lawin_attn_fwd_mfma(const void* theta, int64_t ldt, const void* pg, void* y, int64_t ldy, float* lse, int B, int H, int W, int inter,
                        int heads, hipStream_t stream);
int lawin_attn_bwd_mfma(const void* theta, int64_t ldt, const void* pg, const void* y, int64_t ldy, const void* dy, int64_t lddy,
                        const float* lse, void* dtheta, int64_t lddt, void* dpg, int B, int H, int W, int inter, int heads, hipStream_t stream);

#include "lawin_geom.h"

// ------------------------------------------------------------------ pooled context grid
template <typename T>
__global__ __launch_bounds__(256) void lawin_pool_fwd_kernel(const T* __restrict__ feat, int64_t ld, T* __restrict__ G,
                                                             unsigned char* __restrict__ amax, WinGeom g, int C) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)g.B * g.gh * g.gw * chunks;
    const float inv = 1.f / (g.r * g.r);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int gx = (int)(p % g.gw); int64_t q = p / g.gw; const int gy = (int)(q % g.gh); const int64_t b = q / g.gh;
        const int y0 = gy * g.r - g.pad, x0 = gx * g.r - g.pad;
        float mx[8], sm[8]; int am[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; sm[e] = 0.f; am[e] = 0; }
        for (int dy = 0; dy < g.r; ++dy)
            for (int dx = 0; dx < g.r; ++dx) {
                const int y = y0 + dy, x = x0 + dx;
                F8 v = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? load8(feat + ((b * g.H + y) * g.W + x) * ld + c) : zero8();
#pragma unroll
                for (int e = 0; e < 8; ++e) { sm[e] += v.v[e]; if (v.v[e] > mx[e]) { mx[e] = v.v[e]; am[e] = dy * g.r + dx; } }
            }
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = 0.5f * (mx[e] + sm[e] * inv);
        store8(G + p * C + c, o);
        unsigned long long packed = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(am[e] & 0xff) << (8 * e);
        *(unsigned long long*)(amax + p * C + c) = packed;
    }
}

// d(feat)[b,y,x,:] = base + sum_r ( 0.5/r^2 * dG_r[cell] + 0.5 * dG_r[cell] * [amax_r[cell] == position] )
struct PoolBwdArgs { const float* dG[3]; const unsigned char* amax[3]; WinGeom g[3]; };
template <typename T>
__global__ __launch_bounds__(256) void lawin_pool_bwd_kernel(PoolBwdArgs a, const T* __restrict__ base, int64_t ldb, const float* __restrict__ bvec,
                                                             T* __restrict__ dfeat, int64_t ldo, int C) {
    const int chunks = C / 8;
    const WinGeom g0 = a.g[0];
    const int64_t total = (int64_t)g0.B * g0.H * g0.W * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int x = (int)(p % g0.W); int64_t q = p / g0.W; const int y = (int)(q % g0.H); const int64_t b = q / g0.H;
        F8 acc = base ? load8(base + p * ldb + c) : zero8();
        if (bvec) { const F8 bv = load8(bvec + b * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += bv.v[e]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const WinGeom g = a.g[k];
            const int gy = (y + g.pad) / g.r, gx = (x + g.pad) / g.r;
            const int idx = ((y + g.pad) % g.r) * g.r + (x + g.pad) % g.r;
            const int64_t cell = ((b * g.gh + gy) * g.gw + gx) * C + c;
            F8 d = load8(a.dG[k] + cell);
            const unsigned long long am = *(const unsigned long long*)(a.amax[k] + cell);
            const float w = 0.5f / (g.r * g.r);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += d.v[e] * (w + ((int)((am >> (8 * e)) & 0xff) == idx ? 0.5f : 0.f));
        }
        store8(dfeat + p * ldo + c, acc);
    }
}

// ------------------------------------------------------------------ stand-alone PatchEmbed (modules/lawin.py:18-35)
// y[b,oy,ox,:] = 0.5*(max + mean) over the ps x ps cell at (oy*ps, ox*ps); cells that hang over the right / bottom edge see zeros there.
template <typename T>
__global__ __launch_bounds__(256) void patch_pool_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                             unsigned char* __restrict__ amax, int B, int H, int W, int Ho, int Wo, int C, int ps) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * Ho * Wo * chunks;
    const float inv = 1.f / (ps * ps);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int ox = (int)(p % Wo); int64_t q = p / Wo; const int oy = (int)(q % Ho); const int64_t b = q / Ho;
        float mx[8], sm[8]; int am[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; sm[e] = 0.f; am[e] = 0; }
        for (int dy = 0; dy < ps; ++dy)
            for (int dx = 0; dx < ps; ++dx) {
                const int yy = oy * ps + dy, xx = ox * ps + dx;
                F8 v = (yy < H && xx < W) ? load8(x + ((b * H + yy) * W + xx) * ldx + c) : zero8();
#pragma unroll
                for (int e = 0; e < 8; ++e) { sm[e] += v.v[e]; if (v.v[e] > mx[e]) { mx[e] = v.v[e]; am[e] = dy * ps + dx; } }
            }
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = 0.5f * (mx[e] + sm[e] * inv);
        store8(y + p * C + c, o);
        unsigned long long packed = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(am[e] & 0xff) << (8 * e);
        *(unsigned long long*)(amax + p * C + c) = packed;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void patch_pool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ amax,
                                                             T* __restrict__ dx, int B, int H, int W, int Ho, int Wo, int C, int ps) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * H * W * chunks;
    const float w = 0.5f / (ps * ps);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks) * 8; int64_t p = i / chunks;
        const int xx = (int)(p % W); int64_t q = p / W; const int yy = (int)(q % H); const int64_t b = q / H;
        const int idx = (yy % ps) * ps + xx % ps;
        const int64_t cell = ((b * Ho + yy / ps) * Wo + xx / ps) * C + c;
        F8 d = load8(dy + cell), o;
        const unsigned long long am = *(const unsigned long long*)(amax + cell);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = d.v[e] * (w + ((int)((am >> (8 * e)) & 0xff) == idx ? 0.5f : 0.f));
        store8(dx + p * C + c, o);
    }
}

// dG[b,gy,gx,:] = sum over the windows (i,j) whose 8x8 cell block contains (gy,gx) of dctx[(b,i,j), (gy-i*st, gx-j*st), :]
template <typename T>
__global__ __launch_bounds__(256) void lawin_gather_bwd_kernel(const T* __restrict__ dctx, float* __restrict__ dG, WinGeom g, int C) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)g.B * g.gh * g.gw * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % chunks) * 8; int64_t p = t / chunks;
        const int gx = (int)(p % g.gw); int64_t q = p / g.gw; const int gy = (int)(q % g.gh); const int64_t b = q / g.gh;
        F8 acc = zero8();
        const int ilo = max(0, (gy - 7 + g.st - 1) / g.st), ihi = min(g.nh - 1, gy / g.st);
        const int jlo = max(0, (gx - 7 + g.st - 1) / g.st), jhi = min(g.nw - 1, gx / g.st);
        for (int i = ilo; i <= ihi; ++i)
            for (int j = jlo; j <= jhi; ++j) {
                const int py = gy - i * g.st, px = gx - j * g.st;
                F8 v = load8(dctx + ((((b * g.nh + i) * g.nw + j) * 64) + py * 8 + px) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e];
            }
        store8(dG + p * C + c, acc);
    }
}

// ------------------------------------------------------------------ position mixing over the 64 window positions
// out[w, p', c] = sum_p M[head(c)][p'][p] * x[w, p, c] + bias,  M = W + I  (transpose: M = W^T + I, no bias)
// x[w, p, c] comes either from the pooled grid (gather) or from a window-ordered tensor.
template <typename T>
__global__ __launch_bounds__(256) void lawin_mix_kernel(const T* __restrict__ G, const T* __restrict__ xwin, const float* __restrict__ Wm,
        const float* __restrict__ bm, T* __restrict__ out, WinGeom g, int C, int heads, int transpose) {
    __shared__ __attribute__((aligned(16))) float Ws[64 * 64];
    __shared__ float bs[64];
    const int hd = blockIdx.y, cg = C / heads, wpb = 256 / cg, t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) {
        int pr = i >> 6, pc = i & 63;
        float v = transpose ? Wm[(int64_t)hd * 4096 + pc * 64 + pr] : Wm[(int64_t)hd * 4096 + i];
        Ws[i] = v + (pr == pc ? 1.f : 0.f);
    }
    if (t < 64) bs[t] = (bm && !transpose) ? bm[hd * 64 + t] : 0.f;
    __syncthreads();
    const int wl = t / cg, cl = t % cg;
    const int64_t wi = (int64_t)blockIdx.x * wpb + wl;
    const int64_t nwin = (int64_t)g.B * g.nh * g.nw;
    if (wl >= wpb || wi >= nwin) return;
    const int c = hd * cg + cl;
    float x[64];
    if (xwin) {
#pragma unroll
        for (int p = 0; p < 64; ++p) x[p] = to_f(xwin[(wi * 64 + p) * C + c]);
    } else {
        const int j = (int)(wi % g.nw); int64_t q = wi / g.nw; const int i = (int)(q % g.nh); const int64_t b = q / g.nh;
        const T* base = G + ((b * g.gh + i * g.st) * g.gw + j * g.st) * C + c;
#pragma unroll
        for (int p = 0; p < 64; ++p) x[p] = to_f(base[((int64_t)(p >> 3) * g.gw + (p & 7)) * C]);
    }
    for (int pr = 0; pr < 64; ++pr) {
        float acc = bs[pr];
        const floatx4* wrow = (const floatx4*)&Ws[pr * 64];
#pragma unroll
        for (int p4 = 0; p4 < 16; ++p4) {
            floatx4 wv = wrow[p4];
            acc += wv[0] * x[4 * p4] + wv[1] * x[4 * p4 + 1] + wv[2] * x[4 * p4 + 2] + wv[3] * x[4 * p4 + 3];
        }
        out[(wi * 64 + pr) * C + c] = from_f<T>(acc);
    }
}

// partial[chunk][hd][p'][0..63] = sum_{w in chunk, c in head} d[w,p',c] * x[w,p,c] ;  [..][64] = sum d[w,p',c]
template <typename T>
__global__ __launch_bounds__(256) void lawin_mix_wgrad_kernel(const T* __restrict__ G, const T* __restrict__ dwin, float* __restrict__ partial,
                                                              WinGeom g, int C, int heads, int win_per_chunk) {
    extern __shared__ float shm[];            // At[cg][64] | Xt[cg][64]
    const int hd = blockIdx.y, cg = C / heads, t = threadIdx.x;
    float* At = shm; float* Xt = shm + cg * 64;
    const int pr = t >> 2, pq = t & 3;
    const int64_t nwin = (int64_t)g.B * g.nh * g.nw;
    const int64_t w0 = (int64_t)blockIdx.x * win_per_chunk, w1 = min(nwin, w0 + win_per_chunk);
    float acc[16], accb = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int64_t wi = w0; wi < w1; ++wi) {
        const int j = (int)(wi % g.nw); int64_t q = wi / g.nw; const int i = (int)(q % g.nh); const int64_t b = q / g.nh;
        const T* gbase = G + ((b * g.gh + i * g.st) * g.gw + j * g.st) * C + hd * cg;
        __syncthreads();
        for (int e = t; e < 64 * cg; e += 256) {
            int p = e / cg, cl = e % cg;
            At[cl * 64 + p] = to_f(dwin[(wi * 64 + p) * C + hd * cg + cl]);
            Xt[cl * 64 + p] = to_f(gbase[((int64_t)(p >> 3) * g.gw + (p & 7)) * C + cl]);
        }
        __syncthreads();
        for (int cl = 0; cl < cg; ++cl) {
            const float a = At[cl * 64 + pr];
            const floatx4* xr = (const floatx4*)&Xt[cl * 64 + pq * 16];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) { floatx4 xv = xr[e4];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e4 * 4 + e] += a * xv[e]; }
            accb += a;
        }
    }
    float* dst = partial + (((int64_t)blockIdx.x * heads + hd) * 64 + pr) * 65;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[pq * 16 + e] = acc[e];
    if (pq == 0) dst[64] = accb;
}

// ------------------------------------------------------------------ window attention, one wave per (window, head)
// theta / y / dy / dtheta: image layout rows (b*H + 8i+qy)*W + 8j+qx, `inter` channels (ld given);
// pg / dpg: window layout rows w*64 + k, 2*inter channels (phi | g).
// D consecutive elements <-> fp32 registers: one 4 / 8 / 16-byte access when VEC (host-checked alignment), else element-wise
template <typename T, int D, bool VEC>
__device__ __forceinline__ void load_d(const T* p, float (&o)[D]) {
    if constexpr (VEC && D > 1 && D * sizeof(T) <= 16) {
        typedef T V __attribute__((ext_vector_type(D)));
        const V v = *(const V*)p;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = to_f(v[d]);
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = to_f(p[d]);
    }
}
template <typename T, int D, bool VEC>
__device__ __forceinline__ void store_d(T* p, const float (&o)[D]) {
    if constexpr (VEC && D > 1 && D * sizeof(T) <= 16) {
        typedef T V __attribute__((ext_vector_type(D)));
        V v;
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = from_f<T>(o[d]);
        *(V*)p = v;
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) p[d] = from_f<T>(o[d]);
    }
}

template <typename T, int D, bool VEC>
__global__ __launch_bounds__(256) void lawin_attn_fwd_kernel(const T* __restrict__ theta, int64_t ldt, const T* __restrict__ pg,
        T* __restrict__ y, int64_t ldy, float* __restrict__ lse, WinGeom g, int inter, int heads, float c2) {
    extern __shared__ __attribute__((aligned(16))) float shm[];     // per wave: phi[64][D] | g[64][D]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hd = blockIdx.y * (blockDim.x >> 6) + wave;
    if (hd >= heads) return;
    const int64_t wi = blockIdx.x;
    const int j = (int)(wi % g.nw); int64_t q = wi / g.nw; const int i = (int)(q % g.nh); const int64_t b = q / g.nh;
    float* ph = shm + wave * (2 * 64 * D); float* gg = ph + 64 * D;
    const T* prow = pg + (wi * 64 + lane) * (2 * (int64_t)inter) + hd * D;
    const int64_t row = (b * g.H + 8 * i + (lane >> 3)) * g.W + 8 * j + (lane & 7);
    float th[D], acc[D];
    {
        float a[D], c[D];
        load_d<T, D, VEC>(prow, a); load_d<T, D, VEC>(prow + inter, c); load_d<T, D, VEC>(theta + row * ldt + hd * D, th);
#pragma unroll
        for (int d = 0; d < D; ++d) { ph[lane * D + d] = a[d]; gg[lane * D + d] = c[d]; th[d] *= c2; acc[d] = 0.f; }
    }
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY, l = 0.f;
    if constexpr (D <= 16) {
        // two passes over the 64 keys with the scores in registers: 2 D fma + max + sub + v_exp + add per pair
        float sc[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s = fmaf(th[d], ph[k * D + d], s);
            sc[k] = s; m = fmaxf(m, s);
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float p = __builtin_amdgcn_exp2f(sc[k] - m);
            l += p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(p, gg[k * D + d], acc[d]);
        }
    } else {
#pragma unroll 2
        for (int k = 0; k < 64; ++k) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += th[d] * ph[k * D + d];
            const float mn = fmaxf(m, s), alpha = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
            m = mn; l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = acc[d] * alpha + p * gg[k * D + d];
        }
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] *= inv;
    store_d<T, D, VEC>(y + row * ldy + hd * D, acc);
    lse[(wi * heads + hd) * 64 + lane] = m + log2f(l);
}

template <typename T, int D, bool VEC>
__global__ __launch_bounds__(256) void lawin_attn_bwd_kernel(const T* __restrict__ theta, int64_t ldt, const T* __restrict__ pg,
        const T* __restrict__ y, int64_t ldy, const T* __restrict__ dy, int64_t lddy, const float* __restrict__ lse,
        T* __restrict__ dtheta, int64_t lddt, T* __restrict__ dpg, WinGeom g, int inter, int heads, float c2, float scale) {
    extern __shared__ __attribute__((aligned(16))) float shm[];     // per wave: phi | g | theta*c2 | dY  [64][D] each, lse[64], delta[64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hd = blockIdx.y * (blockDim.x >> 6) + wave;
    if (hd >= heads) return;
    const int64_t wi = blockIdx.x;
    const int j = (int)(wi % g.nw); int64_t q = wi / g.nw; const int i = (int)(q % g.nh); const int64_t b = q / g.nh;
    float* ph = shm + wave * (4 * 64 * D + 128); float* gg = ph + 64 * D; float* ts = gg + 64 * D; float* ds_ = ts + 64 * D;
    float* ls = ds_ + 64 * D; float* dl = ls + 64;
    const T* prow = pg + (wi * 64 + lane) * (2 * (int64_t)inter) + hd * D;
    const int64_t row = (b * g.H + 8 * i + (lane >> 3)) * g.W + 8 * j + (lane & 7);
    float delta = 0.f;
    const float mylse = lse[(wi * heads + hd) * 64 + lane];
    float th[D], dq[D], pk[D], gk[D];
    {
        float yv[D];
        load_d<T, D, VEC>(prow, pk); load_d<T, D, VEC>(prow + inter, gk);
        load_d<T, D, VEC>(theta + row * ldt + hd * D, th); load_d<T, D, VEC>(dy + row * lddy + hd * D, dq);
        load_d<T, D, VEC>(y + row * ldy + hd * D, yv);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            th[d] *= c2;
            ph[lane * D + d] = pk[d]; gg[lane * D + d] = gk[d]; ts[lane * D + d] = th[d]; ds_[lane * D + d] = dq[d];
            delta = fmaf(dq[d], yv[d], delta);
        }
    }
    ls[lane] = mylse; dl[lane] = delta;
    __builtin_amdgcn_wave_barrier();
    {   // pass A, lane = query: dtheta_q = scale * sum_k p (dp - delta) phi_k
        float acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = 0.f;
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {
            float s = -mylse, dp = -delta;
#pragma unroll
            for (int d = 0; d < D; ++d) { s = fmaf(th[d], ph[k * D + d], s); dp = fmaf(dq[d], gg[k * D + d], dp); }
            const float w = __builtin_amdgcn_exp2f(s) * dp;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(w, ph[k * D + d], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] *= scale;
        store_d<T, D, VEC>(dtheta + row * lddt + hd * D, acc);
    }
    T* drow = dpg + (wi * 64 + lane) * (2 * (int64_t)inter) + hd * D;
    {   // pass B, lane = key: dg_k = sum_q p dY_q ; dphi_k = scale * sum_q p (dp - delta_q) theta_q   (ts holds theta*c2: undone at the end)
        float ag[D], ap[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { ag[d] = 0.f; ap[d] = 0.f; }
#pragma unroll 4
        for (int qq = 0; qq < 64; ++qq) {
            float s = -ls[qq], dp = -dl[qq];
#pragma unroll
            for (int d = 0; d < D; ++d) { s = fmaf(pk[d], ts[qq * D + d], s); dp = fmaf(gk[d], ds_[qq * D + d], dp); }
            const float p = __builtin_amdgcn_exp2f(s), w = p * dp;
#pragma unroll
            for (int d = 0; d < D; ++d) { ag[d] = fmaf(p, ds_[qq * D + d], ag[d]); ap[d] = fmaf(w, ts[qq * D + d], ap[d]); }
        }
        const float f = scale / c2;
#pragma unroll
        for (int d = 0; d < D; ++d) ap[d] *= f;
        store_d<T, D, VEC>(drow, ap);
        store_d<T, D, VEC>(drow + inter, ag);
    }
}

// ------------------------------------------------------------------ host entry points
static int check_geom(int B, int H, int W, int r, int C, const char* who) {
    if (!(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (r == 8 || r == 4 || r == 2) && C > 0 && C % 8 == 0))
        return ss_fail(-1, "%s: need H,W multiples of 8, r in {8,4,2}, C multiple of 8 (B=%d H=%d W=%d r=%d C=%d)", who, B, H, W, r, C);
    return 0;
}
static inline int nblk(int64_t n, int cap = 16384) { int64_t b = cdiv(n, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

extern "C" int ss_lawin_pool_fwd(int dtype, const void* feat, int64_t ld, void* G, void* amax, int B, int H, int W, int C, int r,
                                 hipStream_t stream) {
    SS_CHECK(feat && G && amax && ld >= C && ld % 8 == 0, "ss_lawin_pool_fwd: bad arguments");
    if (check_geom(B, H, W, r, C, "ss_lawin_pool_fwd")) return -1;
    WinGeom g = win_geom(B, H, W, r);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_pool_fwd_kernel<T>), dim3(nblk((int64_t)B * g.gh * g.gw * (C / 8))), dim3(256), 0, stream,
                                          (const T*)feat, ld, (T*)G, (unsigned char*)amax, g, C));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_lawin_pool_bwd(int dtype, const float* dG8, const float* dG4, const float* dG2, const void* amax8, const void* amax4,
                                 const void* amax2, const void* base, int64_t ldb, const float* bvec, void* dfeat, int64_t ldo, int B, int H, int W, int C,
                                 hipStream_t stream) {
    SS_CHECK(dG8 && dG4 && dG2 && amax8 && amax4 && amax2 && dfeat && ldo >= C && ldo % 8 == 0 && (!base || (ldb >= C && ldb % 8 == 0)),
             "ss_lawin_pool_bwd: bad arguments");
    if (check_geom(B, H, W, 8, C, "ss_lawin_pool_bwd")) return -1;
    PoolBwdArgs a;
    a.dG[0] = dG8; a.dG[1] = dG4; a.dG[2] = dG2;
    a.amax[0] = (const unsigned char*)amax8; a.amax[1] = (const unsigned char*)amax4; a.amax[2] = (const unsigned char*)amax2;
    a.g[0] = win_geom(B, H, W, 8); a.g[1] = win_geom(B, H, W, 4); a.g[2] = win_geom(B, H, W, 2);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_pool_bwd_kernel<T>), dim3(nblk((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream, a,
                                          (const T*)base, ldb, bvec, (T*)dfeat, ldo, C));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_patch_pool_fwd(int dtype, const void* x, int64_t ldx, void* y, void* amax, int B, int H, int W, int C, int ps,
                                 hipStream_t stream) {
    SS_CHECK(x && y && amax && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 8 == 0 && ps >= 1 && ps <= 16,
             "ss_patch_pool_fwd: need C %% 8 == 0, 16-byte aligned rows, 1 <= patch_size <= 16 (B=%d H=%d W=%d C=%d ps=%d)", B, H, W, C, ps);
    const int Ho = (H + ps - 1) / ps, Wo = (W + ps - 1) / ps;
    SS_DISPATCH(dtype, hipLaunchKernelGGL((patch_pool_fwd_kernel<T>), dim3(nblk((int64_t)B * Ho * Wo * (C / 8))), dim3(256), 0, stream,
                                          (const T*)x, ldx, (T*)y, (unsigned char*)amax, B, H, W, Ho, Wo, C, ps));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_patch_pool_bwd(int dtype, const void* dy, const void* amax, void* dx, int B, int H, int W, int C, int ps, hipStream_t stream) {
    SS_CHECK(dy && dx && amax && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ps >= 1 && ps <= 16,
             "ss_patch_pool_bwd: need C %% 8 == 0, 1 <= patch_size <= 16 (B=%d H=%d W=%d C=%d ps=%d)", B, H, W, C, ps);
    const int Ho = (H + ps - 1) / ps, Wo = (W + ps - 1) / ps;
    SS_DISPATCH(dtype, hipLaunchKernelGGL((patch_pool_bwd_kernel<T>), dim3(nblk((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream,
                                          (const T*)dy, (const unsigned char*)amax, (T*)dx, B, H, W, Ho, Wo, C, ps));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_lawin_gather_bwd(int dtype, const void* dctx, float* dG, int B, int H, int W, int C, int r, hipStream_t stream) {
    SS_CHECK(dctx && dG, "ss_lawin_gather_bwd: null pointer");
    if (check_geom(B, H, W, r, C, "ss_lawin_gather_bwd")) return -1;
    WinGeom g = win_geom(B, H, W, r);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_gather_bwd_kernel<T>), dim3(nblk((int64_t)B * g.gh * g.gw * (C / 8))), dim3(256), 0, stream,
                                          (const T*)dctx, dG, g, C));
    SS_LAUNCH_CHECK();
    return 0;
}

// Wm: [heads][64][64] fp32, bm: [heads][64] fp32.  Exactly one of G (pooled grid, gather) / xwin (window-ordered) is given.
// workspace (optional): heads*64*64 bf16 for the packed weights of the matrix-core path (ss_lawin_mix_workspace bytes); without it,
// in fp32, or for head widths it does not cover, the vector-ALU kernel runs.
extern "C" int64_t ss_lawin_mix_workspace(int heads) { return lawin_mix_mfma_workspace(heads); }
extern "C" int ss_lawin_mix(int dtype, const void* G, const void* xwin, const float* Wm, const float* bm, void* out, int B, int H, int W,
                            int C, int r, int heads, int transpose, void* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK((G != nullptr) != (xwin != nullptr) && Wm && out, "ss_lawin_mix: need exactly one of G / xwin, and Wm, out");
    if (check_geom(B, H, W, r, C, "ss_lawin_mix")) return -1;
    SS_CHECK(heads > 0 && C % heads == 0 && C / heads <= 256, "ss_lawin_mix: C/heads must be an integer <= 256");
    if (dtype == SS_BF16 && workspace && ws_bytes >= lawin_mix_mfma_workspace(heads) && aligned16(workspace) &&
        lawin_mix_mfma_ok(C, heads, G ? G : xwin, out))
        return lawin_mix_mfma(G, xwin, Wm, bm, out, B, H, W, C, r, heads, transpose, workspace, stream);
    WinGeom g = win_geom(B, H, W, r);
    const int cg = C / heads, wpb = 256 / cg;
    const int64_t nwin = (int64_t)B * g.nh * g.nw;
    dim3 grid((unsigned)cdiv(nwin, wpb), heads);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_mix_kernel<T>), grid, dim3(256), 0, stream, (const T*)G, (const T*)xwin, Wm, bm, (T*)out, g, C,
                                          heads, transpose));
    SS_LAUNCH_CHECK();
    return 0;
}

// dWb: [heads][64][65] fp32 (64 weight columns + bias).  workspace >= nchunks*heads*64*65 floats.
extern "C" int64_t ss_lawin_mix_wgrad_workspace(int B, int H, int W, int C, int heads) {
    int64_t nwin = (int64_t)B * (H / 8) * (W / 8);
    int64_t chunks = cdiv(2048, heads); if (chunks > nwin) chunks = nwin; if (chunks < 1) chunks = 1;
    int64_t wpc = cdiv(nwin, chunks); chunks = cdiv(nwin, wpc);
    if (heads > 0 && C % 128 == 0 && C % heads == 0) {                   // slabs of the matrix-core path
        const int64_t m = lawin_mix_wgrad_mfma_slabs(B, H, W, C, heads);
        if (m > chunks) chunks = m;
    }
    return chunks * heads * 64 * 65 * 4;
}
extern "C" int ss_lawin_mix_wgrad(int dtype, const void* G, const void* dwin, float* dWb, int B, int H, int W, int C, int r, int heads,
                                  float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(G && dwin && dWb && workspace, "ss_lawin_mix_wgrad: null pointer");
    if (check_geom(B, H, W, r, C, "ss_lawin_mix_wgrad")) return -1;
    SS_CHECK(heads > 0 && C % heads == 0 && C / heads <= 256, "ss_lawin_mix_wgrad: C/heads must be an integer <= 256");
    WinGeom g = win_geom(B, H, W, r);
    const int64_t nwin = (int64_t)B * g.nh * g.nw;
    if (dtype == SS_BF16 && lawin_mix_mfma_ok(C, heads, G, dwin) &&
        ws_bytes >= lawin_mix_wgrad_mfma_slabs(B, H, W, C, heads) * heads * 64 * 65 * 4)
        return lawin_mix_wgrad_mfma(G, dwin, dWb, B, H, W, C, r, heads, workspace, stream);
    int64_t chunks = cdiv(2048, heads); if (chunks > nwin) chunks = nwin; if (chunks < 1) chunks = 1;
    const int64_t wpc = cdiv(nwin, chunks); chunks = cdiv(nwin, wpc);
    SS_CHECK(ws_bytes >= chunks * heads * 64 * 65 * 4, "ss_lawin_mix_wgrad: workspace too small");
    const int cg = C / heads;
    size_t shm = (size_t)2 * cg * 64 * sizeof(float);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_mix_wgrad_kernel<T>), dim3((unsigned)chunks, heads), dim3(256), shm, stream, (const T*)G,
                                          (const T*)dwin, workspace, g, C, heads, (int)wpc));
    SS_LAUNCH_CHECK();
    const int ncols = heads * 64 * 65;
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(ncols, 8), 1), dim3(256), 0, stream, workspace, (int)chunks, ncols, 1.f, dWb, 0);
    SS_LAUNCH_CHECK();
    return 0;
}

template <typename T, int D> static bool attn_vec_ok(std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
    for (const void* p : ptrs) if ((uintptr_t)p % (D * sizeof(T))) return false;
    for (int64_t l : lds) if (l % D) return false;
    return true;
}
template <typename T, int D, bool VEC>
static int attn_fwd_launch(const T* theta, int64_t ldt, const T* pg, T* y, int64_t ldy, float* lse, WinGeom g, int inter, int heads,
                           hipStream_t stream) {
    const size_t per_wave = (size_t)2 * 64 * D * sizeof(float);
    int waves = (int)(131072 / per_wave); if (waves > 4) waves = 4; if (waves > heads) waves = heads; if (waves < 1) waves = 1;
    const float scale = 1.f / sqrtf((float)D);
    dim3 grid((unsigned)((int64_t)g.B * g.nh * g.nw), (unsigned)cdiv(heads, waves));
    if (waves * per_wave > 65536)
        SS_HIP(hipFuncSetAttribute((const void*)lawin_attn_fwd_kernel<T, D, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(waves * per_wave)));
    hipLaunchKernelGGL((lawin_attn_fwd_kernel<T, D, VEC>), grid, dim3(64 * waves), waves * per_wave, stream, theta, ldt, pg, y, ldy, lse, g, inter, heads,
                       scale * 1.4426950408889634f);
    SS_LAUNCH_CHECK();
    return 0;
}
template <typename T, int D, bool VEC>
static int attn_bwd_launch(const T* theta, int64_t ldt, const T* pg, const T* y, int64_t ldy, const T* dy, int64_t lddy, const float* lse,
                           T* dtheta, int64_t lddt, T* dpg, WinGeom g, int inter, int heads, hipStream_t stream) {
    const size_t per_wave = (size_t)(4 * 64 * D + 128) * sizeof(float);
    int waves = (int)(131072 / per_wave); if (waves > 4) waves = 4; if (waves > heads) waves = heads; if (waves < 1) waves = 1;
    const float scale = 1.f / sqrtf((float)D);
    dim3 grid((unsigned)((int64_t)g.B * g.nh * g.nw), (unsigned)cdiv(heads, waves));
    if (waves * per_wave > 65536)
        SS_HIP(hipFuncSetAttribute((const void*)lawin_attn_bwd_kernel<T, D, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(waves * per_wave)));
    hipLaunchKernelGGL((lawin_attn_bwd_kernel<T, D, VEC>), grid, dim3(64 * waves), waves * per_wave, stream, theta, ldt, pg, y, ldy, dy, lddy, lse, dtheta,
                       lddt, dpg, g, inter, heads, scale * 1.4426950408889634f, scale);
    SS_LAUNCH_CHECK();
    return 0;
}
#define SS_D_CASE(N, ...) case N: { constexpr int DD = N; __VA_ARGS__; } break;
#define SS_D_SWITCH(D, ...) \
    switch (D) { SS_D_CASE(1, __VA_ARGS__) SS_D_CASE(2, __VA_ARGS__) SS_D_CASE(4, __VA_ARGS__) SS_D_CASE(8, __VA_ARGS__) \
                 SS_D_CASE(16, __VA_ARGS__) SS_D_CASE(32, __VA_ARGS__) SS_D_CASE(64, __VA_ARGS__) \
                 default: return ss_fail(-1, "lawin attention: unsupported head dim %d", D); }

// lse: [windows][heads][64] fp32
extern "C" int ss_lawin_attn_fwd(int dtype, const void* theta, int64_t ldt, const void* pg, void* y, int64_t ldy, float* lse, int B, int H,
                                 int W, int inter, int heads, hipStream_t stream) {
    SS_CHECK(theta && pg && y && lse && heads > 0 && inter % heads == 0 && ldt >= inter && ldy >= inter, "ss_lawin_attn_fwd: bad arguments");
    if (check_geom(B, H, W, 8, 8, "ss_lawin_attn_fwd")) return -1;
    WinGeom g = win_geom(B, H, W, 8);
    const int D = inter / heads;
    int rc = 0;
    if (dtype == SS_BF16 && lawin_attn_mfma_ok(D, theta, ldt, pg, inter, y, ldy, nullptr, 0, nullptr, 0))
        return lawin_attn_fwd_mfma(theta, ldt, pg, y, ldy, lse, B, H, W, inter, heads, stream);
    SS_DISPATCH(dtype, SS_D_SWITCH(D, {
        if (attn_vec_ok<T, DD>({theta, pg, y}, {ldt, ldy, (int64_t)inter}))
            rc = attn_fwd_launch<T, DD, true>((const T*)theta, ldt, (const T*)pg, (T*)y, ldy, lse, g, inter, heads, stream);
        else rc = attn_fwd_launch<T, DD, false>((const T*)theta, ldt, (const T*)pg, (T*)y, ldy, lse, g, inter, heads, stream); }));
    return rc;
}
extern "C" int ss_lawin_attn_bwd(int dtype, const void* theta, int64_t ldt, const void* pg, const void* y, int64_t ldy, const void* dy,
                                 int64_t lddy, const float* lse, void* dtheta, int64_t lddt, void* dpg, int B, int H, int W, int inter,
                                 int heads, hipStream_t stream) {
    SS_CHECK(theta && pg && y && dy && lse && dtheta && dpg && heads > 0 && inter % heads == 0, "ss_lawin_attn_bwd: bad arguments");
    SS_CHECK(ldt >= inter && ldy >= inter && lddy >= inter && lddt >= inter, "ss_lawin_attn_bwd: bad leading dimensions");
    if (check_geom(B, H, W, 8, 8, "ss_lawin_attn_bwd")) return -1;
    WinGeom g = win_geom(B, H, W, 8);
    const int D = inter / heads;
    int rc = 0;
    if (dtype == SS_BF16 && lawin_attn_mfma_ok(D, theta, ldt, pg, inter, y, ldy, dy, lddy, dtheta, lddt) && aligned16(dpg))
        return lawin_attn_bwd_mfma(theta, ldt, pg, y, ldy, dy, lddy, lse, dtheta, lddt, dpg, B, H, W, inter, heads, stream);
    SS_DISPATCH(dtype, SS_D_SWITCH(D, {
        if (attn_vec_ok<T, DD>({theta, pg, y, dy, dtheta, dpg}, {ldt, ldy, lddy, lddt, (int64_t)inter}))
            rc = attn_bwd_launch<T, DD, true>((const T*)theta, ldt, (const T*)pg, (const T*)y, ldy, (const T*)dy, lddy, lse, (T*)dtheta, lddt,
                                              (T*)dpg, g, inter, heads, stream);
        else rc = attn_bwd_launch<T, DD, false>((const T*)theta, ldt, (const T*)pg, (const T*)y, ldy, (const T*)dy, lddy, lse, (T*)dtheta, lddt,
                                                (T*)dpg, g, inter, heads, stream); }));
    return rc;
}

// ------------------------------------------------------------------ the branches' BatchNorm folded into the `cat` conv
// LawinHead concatenates [feat, feat + BN_8(z_8), feat + BN_4(z_4), feat + BN_2(z_2), pool] (modules/lawin.py:88-93,202-222) and runs
// a 1x1 conv W = [W_0 | W_1 | W_2 | W_3 | W_p] over it.  The three BatchNorms have no activation, so BN_r(z) = s_r * z + t_r per
// channel (s = gamma * rstd, t = beta - mean * s) and the conv over the concat equals
//     (W_0 + W_1 + W_2 + W_3) . feat  +  sum_r (W_r diag(s_r)) . z_r  +  sum_r W_r t_r  +  W_p . pool
// i.e. the same GEMM over [feat | z_8 | z_4 | z_2] with folded weights and a constant vector: the three normalise-and-add passes
// over the S/8 map (and their outputs) disappear.  ss_lawin_cat_fold builds the folded weight [E, 4E] in the compute dtype, the
// constant [E] and Wt = [(W_0 + W_1 + W_2 + W_3)^T ; W_1^T ; W_2^T ; W_3^T] ([4E, E]: the data-gradient GEMM over it leaves the SUM of the four
// slice gradients -- the gradient fan-in on feat -- in the first E columns, beside the three slice gradients the BatchNorm backward needs); ss_lawin_cat_wgrad_fix turns the weight gradient of the folded GEMM, G' = dz^T [feat | z_8 | z_4 | z_2], into the
// gradient of W:  dW_0 = G'_0,  dW_r = G'_0 + G'_r diag(s_r) + colsum(dz) (x) t_r.
struct CatFoldArgs { const float* stats[3]; const float* gamma[3]; const float* beta[3]; };
template <typename T>
__global__ __launch_bounds__(256) void lawin_cat_fold_kernel(const float* __restrict__ W, int64_t ldw, CatFoldArgs a, int E, T* __restrict__ Wo,
                                                             T* __restrict__ Wt, float* __restrict__ cconst) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    const float* w = W + (int64_t)n * ldw;
    float c = 0.f;
    for (int k = threadIdx.x; k < E; k += 256) {
        float w0 = w[k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float wr = w[(r + 1) * E + k];
            const float sr = a.gamma[r][k] * a.stats[r][E + k], tr = a.beta[r][k] - a.stats[r][k] * sr;
            Wo[(int64_t)n * 4 * E + (r + 1) * E + k] = from_f<T>(wr * sr);
            Wt[((int64_t)(r + 1) * E + k) * E + n] = from_f<T>(wr);
            c = fmaf(wr, tr, c);
            w0 += wr;
        }
        Wo[(int64_t)n * 4 * E + k] = from_f<T>(w0);
        Wt[(int64_t)k * E + n] = from_f<T>(w0);
    }
    red[threadIdx.x] = c; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) cconst[n] = red[0];
}
__global__ __launch_bounds__(256) void lawin_cat_wgrad_fix_kernel(float* __restrict__ G, int64_t ldg, CatFoldArgs a, const float* __restrict__ dvb, int B,
                                                                  int E) {
    const int n = blockIdx.x;
    float cs = 0.f;
    for (int b = 0; b < B; ++b) cs += dvb[(int64_t)b * E + n];
    float* g = G + (int64_t)n * ldg;
    for (int k = threadIdx.x; k < E; k += 256) {
        const float g0 = g[k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float sr = a.gamma[r][k] * a.stats[r][E + k], tr = a.beta[r][k] - a.stats[r][k] * sr;
            g[(r + 1) * E + k] = fmaf(g[(r + 1) * E + k], sr, fmaf(cs, tr, g0));
        }
    }
}
extern "C" int ss_lawin_cat_fold(int dtype, const float* W, int64_t ldw, const float* stats8, const float* stats4, const float* stats2,
                                 const float* gamma8, const float* gamma4, const float* gamma2, const float* beta8, const float* beta4,
                                 const float* beta2, int E, void* Wout, void* Wt, float* cconst, hipStream_t stream) {
    SS_CHECK(W && stats8 && stats4 && stats2 && gamma8 && gamma4 && gamma2 && beta8 && beta4 && beta2 && Wout && Wt && cconst && E > 0 && ldw >= 4 * (int64_t)E,
             "ss_lawin_cat_fold: bad arguments");
    CatFoldArgs a{{stats8, stats4, stats2}, {gamma8, gamma4, gamma2}, {beta8, beta4, beta2}};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((lawin_cat_fold_kernel<T>), dim3(E), dim3(256), 0, stream, W, ldw, a, E, (T*)Wout, (T*)Wt, cconst));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_lawin_cat_wgrad_fix(float* G, int64_t ldg, const float* stats8, const float* stats4, const float* stats2,
                                      const float* gamma8, const float* gamma4, const float* gamma2, const float* beta8, const float* beta4,
                                      const float* beta2, const float* dvb, int B, int E, hipStream_t stream) {
    SS_CHECK(G && stats8 && stats4 && stats2 && gamma8 && gamma4 && gamma2 && beta8 && beta4 && beta2 && dvb && B > 0 && E > 0 && ldg >= 4 * (int64_t)E,
             "ss_lawin_cat_wgrad_fix: bad arguments");
    CatFoldArgs a{{stats8, stats4, stats2}, {gamma8, gamma4, gamma2}, {beta8, beta4, beta2}};
    hipLaunchKernelGGL(lawin_cat_wgrad_fix_kernel, dim3(E), dim3(256), 0, stream, G, ldg, a, dvb, B, E);
    SS_LAUNCH_CHECK();
    return 0;
}

// First pass of the three branch BatchNorm backwards without touching the S/8 maps.  The gradient reaching BN_r's output is
// g_r = dz . W_r (dz: gradient of the cat conv's output), so its column sums are bilinear forms of quantities the step already has:
//     sum_m g_r[m][k]            = sum_n W_r[n][k] * colsum(dz)[n]
//     sum_m g_r[m][k] z_r[m][k]  = sum_n W_r[n][k] * G'_r[n][k],        G' = dz^T [feat | z_8 | z_4 | z_2]  (the folded weight gradient)
// -> dgb_r = (dgamma, dbeta) and coef_r = (mean g, mean g * xhat) for the second pass (ss_bn_bwd_coef).  Call BEFORE
// ss_lawin_cat_wgrad_fix rewrites G'.  A workgroup takes 16 channels of one branch, 16 row parts each (fixed-order combine).
__global__ __launch_bounds__(256) void lawin_branch_bn_sums_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ W, int64_t ldw,
                                                                   CatFoldArgs a, const float* __restrict__ dvb, int B, int E, double inv_rows, int training,
                                                                   float* __restrict__ dgb, float* __restrict__ coef) {
    extern __shared__ float cs[];                             // colsum(dz) [E], then 2 x 16 x 16 doubles of partial sums
    __shared__ double red[2][16][17];
    for (int n = threadIdx.x; n < E; n += 256) { float t = 0.f; for (int b = 0; b < B; ++b) t += dvb[(int64_t)b * E + n]; cs[n] = t; }
    __syncthreads();
    const int r = blockIdx.y, kl = threadIdx.x & 15, part = threadIdx.x >> 4, k = blockIdx.x * 16 + kl;
    double sg = 0.0, sgz = 0.0;
    if (k < E) {
        const float* w = W + (int64_t)(r + 1) * E + k;
        const float* g = G + (int64_t)(r + 1) * E + k;
#pragma unroll 4
        for (int n = part; n < E; n += 16) { const float wv = w[(int64_t)n * ldw]; sg += (double)wv * cs[n]; sgz += (double)wv * g[(int64_t)n * ldg]; }
    }
    red[0][part][kl] = sg; red[1][part][kl] = sgz;
    __syncthreads();
    if (part != 0 || k >= E) return;
    sg = 0.0; sgz = 0.0;
    for (int q = 0; q < 16; ++q) { sg += red[0][q][kl]; sgz += red[1][q][kl]; }
    const double mu = a.stats[r][k], rs = a.stats[r][E + k];
    const double sgx = rs * (sgz - mu * sg);
    dgb[((int64_t)r * 2 + 0) * E + k] = (float)sgx;
    dgb[((int64_t)r * 2 + 1) * E + k] = (float)sg;
    coef[((int64_t)r * 2 + 0) * E + k] = training ? (float)(sg * inv_rows) : 0.f;
    coef[((int64_t)r * 2 + 1) * E + k] = training ? (float)(sgx * inv_rows) : 0.f;
}
extern "C" int ss_lawin_branch_bn_sums(const float* G, int64_t ldg, const float* W, int64_t ldw, const float* stats8, const float* stats4,
                                       const float* stats2, const float* dvb, int B, int E, int64_t rows, int training, float* dgb, float* coef,
                                       hipStream_t stream) {
    SS_CHECK(G && W && stats8 && stats4 && stats2 && dvb && dgb && coef && B > 0 && E > 0 && rows > 0 && ldg >= 4 * (int64_t)E && ldw >= 4 * (int64_t)E &&
             E <= 8192, "ss_lawin_branch_bn_sums: bad arguments");
    CatFoldArgs a{{stats8, stats4, stats2}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipLaunchKernelGGL(lawin_branch_bn_sums_kernel, dim3((unsigned)cdiv(E, 16), 3), dim3(256), E * sizeof(float), stream, G, ldg, W, ldw, a, dvb, B, E,
                       1.0 / (double)rows, training, dgb, coef);
    SS_LAUNCH_CHECK();
    return 0;
}
