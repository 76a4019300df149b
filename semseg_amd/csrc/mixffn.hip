// The Mix-FFN of a MiT block in ONE kernel (modules/segformer.py:44-64: fc2(GELU(dw3x3(fc1(x)) + b)) and the block's residual,
// :96-99), for the stages whose hidden tensors dominate the step (C = 64 / 128, hidden 4C, bf16).
//
// As three launches the 4C-wide tensors h1 = fc1(x) and a2 = GELU(dw(h1) + b) are written once and read once each (537 MB a piece
// at stage 1 of MiT-B2, 16 x 1024^2) around a depthwise kernel that is bound by its vector arithmetic, not by memory.  Here a
// workgroup owns TH x TW pixels of one image and walks the hidden channels in slices of SW:
//   A  expand:    h1[halo pixel][slice] = x . W1[slice]^T + b1 on the matrix cores (v_mfma_f32_32x32x16_bf16, weights as the A operand
//                 so a lane is a pixel), zero outside the image (the depthwise conv pads h1), bf16 into LDS -- the rounding point of
//                 the three-launch form;
//   B  depthwise: 8 channels x 1 pixel per item from 16-byte LDS gathers, taps in registers, GELU in packed fp32, bf16 into a second
//                 LDS image (the second rounding point);
//   C  project:   out[pixel][C] += a2[pixel][slice] . W2[:, slice]^T, accumulators stay in registers across the slices;
// then bias, DropPath scale and residual in fp32 and one rounding, as the GEMM epilogue does.  The x fragments of the tile's halo
// pixels are loaded once and serve every slice; slice weights travel global -> registers -> LDS one phase ahead of their use.
// With h1_out / a2_out set the kernel also stores both hidden tensors (the three-launch backward reads them); the recomputing
// backward (mixffn_bwd) needs neither.
#include "common.h"

namespace {

struct MixArgs {
    const bf16* x; const bf16* w1; const float* b1; const float* w9; const float* bd; const bf16* w2; const float* b2;
    const bf16* res; const float* keep; bf16* out; bf16* h1; bf16* a2;
    int B, H, W, Ch, tiles_x, tiles_y, dbg;
};

// GELU(u) = u Phi(u) for a pair, Abramowitz-Stegun 7.1.26 as gelu_parts (common.h): Phi(u) = 1/2 + sign(u) (1/2 - hc),
// hc = 1/2 erfc(|u| / sqrt 2) = poly(t) exp(-u^2 / 2), t = 1 / (1 + p |u| / sqrt 2)  =>  GELU(u) = u / 2 + |u| (1/2 - hc).
// Packed fp32 where the ISA has it (v_pk_mul / v_pk_fma); |u| rides as the abs modifier of two plain fmas.
__device__ __forceinline__ floatx2 gelu2(floatx2 u, floatx2& hc_out, floatx2& ez_out) {
    floatx2 t, ez;
    t[0] = __builtin_amdgcn_rcpf(fmaf(fabsf(u[0]), 0.3275911f * 0.70710678118654752f, 1.f));
    t[1] = __builtin_amdgcn_rcpf(fmaf(fabsf(u[1]), 0.3275911f * 0.70710678118654752f, 1.f));
    const floatx2 e2 = (u * floatx2{-0.72134752044448170f, -0.72134752044448170f}) * u;      // -u^2 / 2 * log2(e)
    ez[0] = __builtin_amdgcn_exp2f(e2[0]); ez[1] = __builtin_amdgcn_exp2f(e2[1]);
    floatx2 p = __builtin_elementwise_fma(t, floatx2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, floatx2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
    p = __builtin_elementwise_fma(t, p, floatx2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
    p = __builtin_elementwise_fma(t, p, floatx2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
    p = __builtin_elementwise_fma(t, p, floatx2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
    const floatx2 hc = (t * p) * ez;
    hc_out = hc; ez_out = ez;
    const floatx2 q = floatx2{0.5f, 0.5f} - hc, hu = u * floatx2{0.5f, 0.5f};
    floatx2 a;
    a[0] = fmaf(fabsf(u[0]), q[0], hu[0]);
    a[1] = fmaf(fabsf(u[1]), q[1], hu[1]);
    return a;
}

// L2 prefetch without a register: one dword per lane by LDS-DMA into bytes nobody reads.  Issued as inline asm so that hipcc's wait
// counting does not see it (a builtin LDS-DMA in flight makes it wait for vmcnt(0) before every later LDS access); an operation it
// does not count only makes its own waits conservative (loads return in order).
typedef __attribute__((address_space(3))) void mix_lds_void;
__device__ __forceinline__ void prefetch_dword(const void* g, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_uniform) : "memory");
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    union { bf16x2 v; unsigned u; } r;
    r.v = bf16x2{(bf16)lo, (bf16)hi};
    return r.u;
}

template <int C, int SW, int TH, int TW>
struct MixGeom {
    static constexpr int IH = TH + 2, IW = TW + 2, IP = IH * IW, OP = TH * TW;
    static constexpr int NRT = (IP + 31) / 32, RTW = (NRT + 3) / 4;         // halo row tiles of 32 pixels, per wave at most
    static constexpr int KS1 = C / 16, CT1 = SW / 32, KS2 = SW / 16, CT2 = C / 32;
    static constexpr int NCH = SW / 8, NI = OP * NCH / 256;
    static constexpr int PH = SW * 2 + 16, PI = C * 2 + 16;                 // LDS row pitches (bytes): slice-wide images, W1 rows
    static constexpr int NWD = 11 * SW;                                     // 9 taps, depthwise bias, fc1 bias of a slice
    static constexpr size_t OFF_DS = (size_t)IP * PH, OFF_W1 = OFF_DS + (size_t)OP * PH, OFF_W2 = OFF_W1 + (size_t)SW * PI,
                            OFF_WD = OFF_W2 + (size_t)C * PH, OFF_EP = OFF_WD + (size_t)NWD * 4, OFF_PF = OFF_EP + 4 * 2048, LDS = OFF_PF + 4 * 256;
    static_assert(OP % 32 == 0 && (OP / 32) * CT2 == 8, "phase C: two accumulator tiles per wave");
    static_assert(NI * 256 == OP * NCH, "depthwise items must fill the workgroup");
    static constexpr int W1V = SW * C / 8 / 256, W2V = C * SW / 8 / 256;    // 16-byte weight chunks per thread and slice
    static_assert(W1V * 256 * 8 == SW * C && W2V >= 1, "weight staging");
};

template <int C, int SW, int TH, int TW>
__global__ __launch_bounds__(256, 2) void mixffn_fwd_kernel(MixArgs a, int ntiles, FastDiv d_tpi, FastDiv d_tx) {
    typedef MixGeom<C, SW, TH, TW> G;
    constexpr int IW = G::IW, IP = G::IP, OP = G::OP, RTW = G::RTW, KS1 = G::KS1, CT1 = G::CT1, KS2 = G::KS2, CT2 = G::CT2, NCH = G::NCH,
                  NI = G::NI, PH = G::PH, PI = G::PI, W1V = G::W1V, W2V = G::W2V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Hs = smem;                                   // [IP][SW] h1 of the halo tile
    char* Ds = smem + G::OFF_DS;                       // [OP][SW] a2
    char* W1s = smem + G::OFF_W1;                      // [SW][C]
    char* W2s = smem + G::OFF_W2;                      // [C][SW]
    float* Wd = (float*)(smem + G::OFF_WD);            // [9][SW] taps, [SW] depthwise bias, [SW] fc1 bias
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wl = smem + G::OFF_EP + wave * 2048;         // this wave's epilogue strip
    const int nsl = a.Ch / SW;
    const unsigned pf_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(mix_lds_void*)(smem + G::OFF_PF) + wave * 256);

    // ---- slice weights: global -> registers (prefetch) -> LDS (commit), each in the window its readers leave free
    bf16x8 pw1[W1V], pw2[W2V];
    floatx4 pwd, pwt;                                  // fc1 bias of the next slice; taps + depthwise bias of the next slice
    auto fetch_w1 = [&](int s) {                       // W1 rows s*SW .. +SW, and the fc1 bias of the slice
#pragma unroll
        for (int i = 0; i < W1V; ++i) { const int idx = tid + i * 256, r = idx / (C / 8), c = idx % (C / 8);
            pw1[i] = *(const bf16x8*)(a.w1 + (int64_t)(s * SW + r) * C + c * 8); }
    };
    auto commit_w1 = [&]() {
#pragma unroll
        for (int i = 0; i < W1V; ++i) { const int idx = tid + i * 256, r = idx / (C / 8), c = idx % (C / 8);
            *(bf16x8*)(W1s + (size_t)r * PI + c * 16) = pw1[i]; }
    };
    auto fetch_w2 = [&](int s) {                       // W2[:, s*SW .. +SW]
#pragma unroll
        for (int i = 0; i < W2V; ++i) { const int idx = tid + i * 256, r = idx / NCH, c = idx % NCH;
            pw2[i] = *(const bf16x8*)(a.w2 + (int64_t)r * a.Ch + s * SW + c * 8); }
    };
    auto commit_w2 = [&]() {
#pragma unroll
        for (int i = 0; i < W2V; ++i) { const int idx = tid + i * 256, r = idx / NCH, c = idx % NCH;
            *(bf16x8*)(W2s + (size_t)r * PH + c * 16) = pw2[i]; }
    };
    // taps + depthwise bias (rows 0..9) and fc1 bias (row 10): row k of the slice = SW floats = SW/4 vectors
    auto fetch_wd = [&](int s, int k0, int k1, floatx4& dst) {       // rows k0 .. k1-1; (k1 - k0) * SW / 4 <= 256
        const int v = tid, k = k0 + v / (SW / 4), j = v % (SW / 4);
        if (k < k1) {
            const float* src = k < 9 ? a.w9 + (int64_t)k * a.Ch : (k == 9 ? a.bd : a.b1);
            dst = *(const floatx4*)(src + s * SW + j * 4);
        }
    };
    auto commit_wd = [&](int k0, int k1, const floatx4& src) {
        const int v = tid, k = k0 + v / (SW / 4), j = v % (SW / 4);
        if (k < k1) *(floatx4*)(Wd + k * SW + j * 4) = src;
    };
    static_assert(10 * (SW / 4) <= 256, "tap staging: one vector per thread");

    // this thread's depthwise items: NI horizontally adjacent output pixels of one 8-channel chunk -- the same in every tile and slice.
    // A row of the 3 x 3 windows is then NI + 2 gathers for NI outputs, each widened to fp32 once.
    static_assert(TW % NI == 0, "item runs must not wrap around a tile row");
    const int it_c = tid % NCH;
    const int it_op = (tid / NCH) * NI, it_py = it_op / TW, it_px = it_op - it_py * TW;
    const int it_h = (it_py * IW + it_px) * PH + it_c * 16;            // halo pixel of the run's first window corner
    const int t0 = wave * 2, rt2 = t0 / CT2, ct2 = t0 % CT2;          // phase C: accumulator tiles (rt2, ct2), (rt2, ct2 + 1)

    // prologue: slice 0's fc1 operands
    fetch_w1(0); fetch_wd(0, 10, 11, pwd); fetch_wd(0, 0, 10, pwt);
    commit_w1(); commit_wd(10, 11, pwd); commit_wd(0, 10, pwt);
    __syncthreads();

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = (int)fdiv((unsigned)tile, d_tpi), trem = tile - img * (int)d_tpi.d;
        const int ty = (int)fdiv((unsigned)trem, d_tx), tx = trem - ty * (int)d_tx.d;
        const int oy0 = ty * TH, ox0 = tx * TW;

        // ---- the tile's halo pixels as B-operand fragments, once for all slices
        bf16x8 xf[RTW][KS1];
        bool pin[RTW], valid[RTW];
#pragma unroll
        for (int q = 0; q < RTW; ++q) {
            const int p = (wave + 4 * q) * 32 + li, iy = p / IW, ix = p - iy * IW;
            const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
            pin[q] = p < IP; valid[q] = pin[q] && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const bf16* xp = a.x + (((int64_t)img * a.H + (valid[q] ? gy : 0)) * a.W + (valid[q] ? gx : 0)) * C + lh * 8;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[q][ks] = valid[q] ? *(const bf16x8*)(xp + ks * 16) : zero8;
        }
        floatx16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

        for (int s = 0; s < nsl; ++s) {
            const int sn = s + 1 < nsl ? s + 1 : 0;    // the slice after this one (of the next tile, at the end)
            // ---- A: expand
#pragma unroll
            for (int q = 0; q < RTW; ++q) {
                if ((wave + 4 * q) * 32 >= IP || (a.dbg & 4)) break;   // wave-uniform
                const int p = (wave + 4 * q) * 32 + li;
#pragma unroll
                for (int ct = 0; ct < CT1; ++ct) {
                    floatx16 h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const floatx4 bv = *(const floatx4*)(Wd + 10 * SW + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[4 * g + e] = bv[e];
                    }
                    const char* wp = W1s + (size_t)(ct * 32 + li) * PI + lh * 16;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks)
                        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wp + ks * 32), xf[q][ks], h, 0, 0, 0);
                    if (pin[q]) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            unsigned lo = pack_bf16x2(h[4 * g], h[4 * g + 1]), hi = pack_bf16x2(h[4 * g + 2], h[4 * g + 3]);
                            if (!valid[q]) { lo = 0; hi = 0; }         // outside the image h1 is the conv's zero padding, not the bias
                            *(uint2*)(Hs + (size_t)p * PH + (ct * 32 + 8 * g + 4 * lh) * 2) = uint2{lo, hi};
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fetch_w1(sn); fetch_wd(sn, 10, 11, pwd); fetch_wd(sn, 0, 10, pwt); fetch_w2(s);
            __syncthreads();                           // B1: the h1 slice is in LDS; nobody still reads W1s / the fc1 bias

            // ---- B: depthwise + GELU, one window row at a time: its NI + 2 gathers and three taps are requested together
            if (!(a.dbg & 2)) {
                const float* wb = Wd + it_c * 8;
                floatx2 u[NI][4];
                uint4 centre[NI];
                { const floatx4 w0 = *(const floatx4*)(wb + 9 * SW), w1 = *(const floatx4*)(wb + 9 * SW + 4);
#pragma unroll
                  for (int i = 0; i < NI; ++i) { u[i][0] = floatx2{w0[0], w0[1]}; u[i][1] = floatx2{w0[2], w0[3]}; u[i][2] = floatx2{w1[0], w1[1]}; u[i][3] = floatx2{w1[2], w1[3]}; } }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    uint4 hv[NI + 2];
                    floatx2 wv[3][4];
#pragma unroll
                    for (int c = 0; c < NI + 2; ++c) hv[c] = *(const uint4*)(Hs + it_h + (size_t)(ky * IW + c) * PH);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const floatx4 w0 = *(const floatx4*)(wb + (ky * 3 + kx) * SW), w1 = *(const floatx4*)(wb + (ky * 3 + kx) * SW + 4);
                        wv[kx][0] = floatx2{w0[0], w0[1]}; wv[kx][1] = floatx2{w0[2], w0[3]}; wv[kx][2] = floatx2{w1[0], w1[1]}; wv[kx][3] = floatx2{w1[2], w1[3]};
                    }
#pragma unroll
                    for (int c = 0; c < NI + 2; ++c) {
                        if (ky == 1 && c >= 1 && c <= NI) centre[c - 1] = hv[c];
                        const unsigned w[4] = {hv[c].x, hv[c].y, hv[c].z, hv[c].w};
                        floatx2 hf[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) hf[e] = floatx2{__uint_as_float(w[e] << 16), __uint_as_float(w[e] & 0xffff0000u)};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int i = c - kx;                      // column c is tap kx of output i
                            if (i < 0 || i >= NI) continue;
#pragma unroll
                            for (int e = 0; e < 4; ++e) u[i][e] = __builtin_elementwise_fma(hf[e], wv[kx][e], u[i][e]);
                        }
                    }
                    // pin the row: without it hipcc keeps the gathers of all three rows alive and spills
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(u[i][e]));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    unsigned ow[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { floatx2 hc, ez; const floatx2 g2 = (a.dbg & 1) ? u[i][e] : gelu2(u[i][e], hc, ez); ow[e] = pack_bf16x2(g2[0], g2[1]); }
                    const uint4 o = uint4{ow[0], ow[1], ow[2], ow[3]};
                    *(uint4*)(Ds + (size_t)(it_op + i) * PH + it_c * 16) = o;
                    if (a.a2) {                        // the three-launch backward reads both hidden tensors
                        const int oy = oy0 + it_py, ox = ox0 + it_px + i;
                        if (oy < a.H && ox < a.W) {
                            const int64_t pix = ((int64_t)img * a.H + oy) * a.W + ox;
                            *(uint4*)(a.a2 + pix * a.Ch + s * SW + it_c * 8) = o;
                            *(uint4*)(a.h1 + pix * a.Ch + s * SW + it_c * 8) = centre[i];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            commit_w1(); commit_wd(10, 11, pwd); commit_w2();
            __syncthreads();                           // B2: a2 slice, W2 slice and the next slice's fc1 operands are in LDS
            commit_wd(0, 10, pwt);                     // the next slice's taps: phase B of this slice is over for everyone

            // ---- C: project, accumulators across slices
            if (!(a.dbg & 8)) {
                const char* dp = Ds + (size_t)(rt2 * 32 + li) * PH + lh * 16;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    const bf16x8 df = *(const bf16x8*)(dp + ks * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(W2s + (size_t)((ct2 + j) * 32 + li) * PH + lh * 16 + ks * 32), df, acc[j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: (acc + b2) * keep + res, one rounding; 32 x 32 tiles through the wave's strip, out as 64-byte row segments
        {
            const int op = rt2 * 32 + li, py = op / TW, px = op - py * TW, oy = oy0 + py, ox = ox0 + px;
            const bool ok = oy < a.H && ox < a.W;
            const int64_t pix = ((int64_t)img * a.H + (ok ? oy : 0)) * a.W + (ok ? ox : 0);
            const float rs = a.keep ? a.keep[img] : 1.f;
            const int wsw = ((li >> 2) & 3) << 1, rr = lane >> 2, c16 = lane & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = (ct2 + j) * 32 + 8 * g + 4 * lh;
                    const floatx4 bv = *(const floatx4*)(a.b2 + n);
                    bf16x4 rv = {0, 0, 0, 0};
                    if (a.res && ok) rv = *(const bf16x4*)(a.res + pix * C + n);
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)((acc[j][4 * g + e] + bv[e]) * rs + (float)rv[e]);
                    *(bf16x4*)(wl + li * 64 + (((2 * g + lh) ^ wsw) << 3)) = o;
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = h * 16 + rr;
                    const bf16x8 v = *(const bf16x8*)(wl + r * 64 + ((c16 ^ ((r >> 2) & 3)) << 4));
                    const int op2 = rt2 * 32 + r, py2 = op2 / TW, px2 = op2 - py2 * TW, oy2 = oy0 + py2, ox2 = ox0 + px2;
                    if (oy2 < a.H && ox2 < a.W)
                        *(bf16x8*)(a.out + (((int64_t)img * a.H + oy2) * a.W + ox2) * C + (ct2 + j) * 32 + c16 * 8) = v;
                }
                asm volatile("" ::: "memory");
            }
        }
    }
}

int mix_cus() {
    static int n = 0;
    if (!n) { int dev = 0, v = 0;
              if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
              n = v; }
    return n;
}

template <int C, int SW, int TH, int TW> int mix_fwd_launch(MixArgs a, hipStream_t st) {
    typedef MixGeom<C, SW, TH, TW> G;
    const size_t lds = G::LDS;
    static bool attr = false;                          // per instantiation; the attribute is per function, races only repeat the call
    if (!attr && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mixffn_fwd_kernel<C, SW, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ss_fail(-2, "ss_mixffn_fwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
        attr = true;
    }
    if (a.dbg & 16) { int nb = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mixffn_fwd_kernel<C, SW, TH, TW>, 256, lds);
                      fprintf(stderr, "mixffn_fwd<%d>: lds %zu, blocks per CU %d (%s)\n", C, lds, nb, hipGetErrorString(e)); }
    a.tiles_x = (int)cdiv(a.W, TW); a.tiles_y = (int)cdiv(a.H, TH);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.B;
    if (ntiles >= (1LL << 31)) return ss_fail(-1, "ss_mixffn_fwd: too many tiles");
    int64_t grid = (int64_t)mix_cus() * 2; if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((mixffn_fwd_kernel<C, SW, TH, TW>), dim3((unsigned)grid), dim3(256), lds, st, a, (int)ntiles,
                       fast_div((unsigned)(a.tiles_x * a.tiles_y)), fast_div((unsigned)a.tiles_x));
    SS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// 1 when ss_mixffn_fwd takes the shape: bf16, C = 64 or 128 input channels, hidden width 4C
extern "C" int ss_mixffn_ok(int dtype, int C, int Ch) {
    return dtype == SS_BF16 && (C == 64 || C == 128) && Ch == 4 * C;
}

// out[T, C] = (fc2(GELU(dw3x3(fc1(x) + b1) + bd)) + b2) * keep[image] + res;  x, res, out: [B*H*W, C] dense bf16 tokens of a
// channels-last map; w1 [Ch, C], w2 [C, Ch] bf16 (the nn.Linear layouts); b1, bd [Ch], b2 [C], w9 [9][Ch] tap-major, keep [B] (or
// null) fp32; res may be null.  h1_out / a2_out (both or neither): [T, Ch] bf16, the hidden tensors as the three-launch form stores them.
extern "C" int ss_mixffn_fwd(int dtype, const void* x, const void* w1, const float* b1, const float* w9, const float* bd, const void* w2,
                             const float* b2, const void* res, const float* keep, void* out, void* h1_out, void* a2_out, int B, int H,
                             int W, int C, int Ch, hipStream_t stream) {
    SS_CHECK(x && w1 && b1 && w9 && bd && w2 && b2 && out, "ss_mixffn_fwd: null operand");
    SS_CHECK(ss_mixffn_ok(dtype, C, Ch), "ss_mixffn_fwd: needs bf16, C = 64 or 128 and a hidden width of 4C (C=%d Ch=%d)", C, Ch);
    SS_CHECK(B > 0 && H > 0 && W > 0 && (int64_t)B * H * W < (1LL << 31), "ss_mixffn_fwd: bad geometry");
    SS_CHECK((h1_out == nullptr) == (a2_out == nullptr), "ss_mixffn_fwd: h1_out and a2_out come together");
    SS_CHECK(aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(out) && aligned16(b1) && aligned16(w9) && aligned16(bd) && aligned16(b2) &&
             (!res || aligned16(res)) && (!h1_out || (aligned16(h1_out) && aligned16(a2_out))), "ss_mixffn_fwd: pointers must be 16-byte aligned");
    SS_CHECK(out != x, "ss_mixffn_fwd: out must not alias x (neighbouring tiles re-read the halo)");
    MixArgs a{};
    a.x = (const bf16*)x; a.w1 = (const bf16*)w1; a.b1 = b1; a.w9 = w9; a.bd = bd; a.w2 = (const bf16*)w2; a.b2 = b2;
    a.res = (const bf16*)res; a.keep = keep; a.out = (bf16*)out; a.h1 = (bf16*)h1_out; a.a2 = (bf16*)a2_out;
    a.B = B; a.H = H; a.W = W; a.Ch = Ch;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SS_MIX_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }     // ablation switches (tools/bench_mixffn.py)
    if (C == 64) return mix_fwd_launch<64, 64, 8, 16>(a, stream);
    return mix_fwd_launch<128, 64, 8, 8>(a, stream);
}
