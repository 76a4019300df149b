// One inverted-residual block of the TopFormer token pyramid in ONE kernel (modules/topformer.py InvertedResidual: 1x1 expand + BN +
// ReLU -> k x k depthwise (stride 1 / 2, AutoPadding) + BN + ReLU -> 1x1 project + BN (+ skip), BatchNorm folded, inference).
//
// As three launches the block writes and re-reads the expanded tensor twice (537 MB each way for the second block of TopFormer-B at
// 4 x 2048^2: 1.5 GB of traffic for 134 MB in and 67 MB out).  Here a workgroup owns TH x 8 output pixels:
//   A  expand:   the input pixels under the tile's taps ((TH-1)S+K rows x 7S+K columns) times W_in on the matrix cores
//                (v_mfma_f32_32x32x16_bf16, weights as the A operand so a lane is a pixel), bias + activation, pixels outside the
//                image set to zero (the depthwise conv pads the EXPANDED tensor), bf16 into LDS [pixel][hidden];
//   B  depthwise: 8 channels x 1 output pixel per item from 16-byte LDS reads, taps and bias from an LDS copy, bf16 into a second
//                LDS image [output pixel][hidden] -- the same rounding points as the three-launch form;
//   C  project:  that image times W_out on the matrix cores, bias (+ the block input as the skip), 8-byte stores.
// Channel counts are padded by the host (hidden and output channels to 32, input channels to 16, zero weights) so the tile loops
// need no channel tails; x and y keep their real widths (loads / stores masked).  A block without expansion (expand_ratio 1) passes
// an identity W_in with no activation: one exact product per element.
#include "common.h"

namespace {

struct IrArgs {
    const bf16* x; bf16* y; const bf16* w_in; const float* b_in; const float* w_dw; const float* b_dw; const bf16* w_out; const float* b_out;
    int B, H, W, Cin, cinp, hidp, Cout, coutp, pt, pl, Ho, Wo, act_in, act_dw, residual, tiles_x, tiles_y;
};

template <int K, int S, int TH, bool HI>
__global__ __launch_bounds__(256, 4) void ir_block_kernel(IrArgs a, int ntiles, FastDiv d_tpi, FastDiv d_tx) {
    constexpr int TW = 8, IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, IP = IH * IW, OP = TH * TW;
    constexpr int RTW = (IP + 127) / 128;                        // row tiles of 32 input pixels per wave
    constexpr int UKY = K == 3 ? 3 : 1;                          // tap rows unrolled (all 25 taps of a 5 x 5 spill at 128 registers)
    constexpr int NI = 4;                                        // depthwise items per thread, at most (OP * hidp / 8 <= 1024)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PH = a.hidp * 2 + 16, PI = a.cinp * 2 + 16;       // row pitches (bytes): hidden-wide images, W_in
    char* Hs = smem;                                             // [IP][hidp] expanded input tile
    char* Ds = Hs + (size_t)IP * PH;                             // [OP][hidp] depthwise output
    char* Wo = Ds + (size_t)OP * PH;                             // [coutp][hidp] project weights
    char* Wi = Wo + (size_t)a.coutp * PH;                        // [hidp][cinp] expand weights
    float* Wd = (float*)(Wi + (size_t)a.hidp * PI);              // [K*K + 1][hidp] taps, then the depthwise bias
    float* Bi = Wd + (K * K + 1) * a.hidp;                       // [hidp] expand bias
    float* Bo = Bi + a.hidp;                                     // [coutp] project bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;

    // the block's weights, once per (persistent) workgroup
    for (int i = tid; i < (K * K + 1) * a.hidp; i += 256) Wd[i] = i < K * K * a.hidp ? a.w_dw[i] : a.b_dw[i - K * K * a.hidp];
    for (int i = tid; i < a.hidp; i += 256) Bi[i] = a.b_in[i];
    for (int i = tid; i < a.coutp; i += 256) Bo[i] = a.b_out[i];
    { const int cpr = a.hidp >> 3;
      for (int i = tid; i < a.coutp * cpr; i += 256) { const int r = i / cpr, c = i - r * cpr;
          *(bf16x8*)(Wo + (size_t)r * PH + c * 16) = *(const bf16x8*)(a.w_out + (int64_t)r * a.hidp + c * 8); } }
    { const int cpr = a.cinp >> 3;
      for (int i = tid; i < a.hidp * cpr; i += 256) { const int r = i / cpr, c = i - r * cpr;
          *(bf16x8*)(Wi + (size_t)r * PI + c * 16) = *(const bf16x8*)(a.w_in + (int64_t)r * a.cinp + c * 8); } }

    const int nks_in = a.cinp >> 4, nct_h = a.hidp >> 5, nch = a.hidp >> 3, nks_h = a.hidp >> 4, nct_o = a.coutp >> 5;
    // activations as clamps (0 none, 2 ReLU, 3 ReLU6; the upper clamp only in the HI instantiation): a per-element switch compiles to
    // a scalar branch tree
    const float lo_in = a.act_in ? 0.f : -INFINITY, hi_in = a.act_in == 3 ? 6.f : INFINITY;
    const float lo_dw = a.act_dw ? 0.f : -INFINITY, hi_dw = a.act_dw == 3 ? 6.f : INFINITY;
    // this thread's depthwise items (output pixel, 8-channel chunk): the same in every tile
    int it_h[NI], it_d[NI], it_w[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int item = tid + i * 256, live = item < OP * nch;
        const int op = live ? item / nch : 0, ch = live ? item - op * nch : 0;
        const int py = op / TW, px = op - py * TW;
        it_h[i] = ((py * S) * IW + px * S) * PH + ch * 16;
        it_d[i] = live ? op * PH + ch * 16 : -1;
        it_w[i] = ch * 8;
    }
    __syncthreads();

    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = (int)fdiv((unsigned)tile, d_tpi), trem = tile - img * (int)d_tpi.d;
        const int ty = (int)fdiv((unsigned)trem, d_tx), tx = trem - ty * (int)d_tx.d;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int gy0 = oy0 * S - a.pt, gx0 = ox0 * S - a.pl;

        // ---- A: expand.  All of this wave's input pixels are requested before the first product waits for any of them.
        bf16x8 xf[RTW][4];
        bool pin[RTW], valid[RTW];
#pragma unroll
        for (int q = 0; q < RTW; ++q) {
            const int p = (wave + 4 * q) * 32 + li, iy = p / IW, ix = p - iy * IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            pin[q] = p < IP; valid[q] = pin[q] && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const bf16* xp = a.x + (((int64_t)img * a.H + (valid[q] ? gy : 0)) * a.W + (valid[q] ? gx : 0)) * a.Cin;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k0 = ks * 16 + lh * 8;
                xf[q][ks] = (ks < nks_in && valid[q] && k0 < a.Cin) ? *(const bf16x8*)(xp + k0) : zero;
            }
        }
#pragma unroll
        for (int q = 0; q < RTW; ++q) {
            const int p = (wave + 4 * q) * 32 + li;
            if ((wave + 4 * q) * 32 >= IP) break;                 // wave-uniform
            // pixels outside the image are zeros of the EXPANDED tensor (the depthwise conv's padding), not act(bias): interior tiles
            // (wave-uniform test) skip the per-element select
            const bool border = __builtin_amdgcn_ballot_w64(pin[q] && !valid[q]) != 0;
            for (int ct = 0; ct < nct_h; ++ct) {
                floatx16 acc;                                     // starts at the bias of its channels (register quad g = 4 consecutive channels)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 bv = *(const floatx4*)(Bi + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * g + e] = bv[e];
                }
                const char* wp = Wi + (size_t)(ct * 32 + li) * PI + lh * 16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < nks_in) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wp + ks * 32), xf[q][ks], acc, 0, 0, 0);
                if (pin[q]) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaxf(acc[4 * g + e], lo_in);
                            if (HI) v = fminf(v, hi_in);
                            o[e] = (bf16)v;
                        }
                        if (border && !valid[q]) o = bf16x4{0, 0, 0, 0};
                        *(bf16x4*)(Hs + (size_t)p * PH + (ct * 32 + 8 * g + 4 * lh) * 2) = o;
                    }
                }
            }
        }
        __syncthreads();

        // ---- B: depthwise
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (it_d[i] < 0) continue;
            const float* wb = Wd + it_w[i];
            floatx2 u[4];
            { const floatx4 b0 = *(const floatx4*)(wb + K * K * a.hidp), b1 = *(const floatx4*)(wb + K * K * a.hidp + 4);
              u[0] = floatx2{b0[0], b0[1]}; u[1] = floatx2{b0[2], b0[3]}; u[2] = floatx2{b1[0], b1[1]}; u[3] = floatx2{b1[2], b1[3]}; }
            const char* hp = Hs + it_h[i];
#pragma unroll UKY
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    union { bf16x8 v; unsigned w[4]; } h;
                    h.v = *(const bf16x8*)(hp + (size_t)(ky * IW + kx) * PH);
                    const floatx4 w0 = *(const floatx4*)(wb + (ky * K + kx) * a.hidp), w1 = *(const floatx4*)(wb + (ky * K + kx) * a.hidp + 4);
                    const floatx2 wv[4] = {floatx2{w0[0], w0[1]}, floatx2{w0[2], w0[3]}, floatx2{w1[0], w1[1]}, floatx2{w1[2], w1[3]}};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                 // a bf16 pair widens with one shift and one mask; the products are packed fp32 fmas
                        const floatx2 hv = {__uint_as_float(h.w[e] << 16), __uint_as_float(h.w[e] & 0xffff0000u)};
                        u[e] = __builtin_elementwise_fma(hv, wv[e], u[e]);
                    }
                }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v0 = fmaxf(u[e][0], lo_dw), v1 = fmaxf(u[e][1], lo_dw);
                if (HI) { v0 = fminf(v0, hi_dw); v1 = fminf(v1, hi_dw); }
                o[2 * e] = (bf16)v0; o[2 * e + 1] = (bf16)v1;
            }
            *(bf16x8*)(Ds + it_d[i]) = o;
        }
        __syncthreads();

        // ---- C: project (+ skip).  The next tile's phase A writes Hs only, and its barrier comes after every wave has left this phase.
        for (int t = wave; t < (OP / 32) * nct_o; t += 4) {
            const int rt = t / nct_o, ct = t - rt * nct_o;
            const int op = rt * 32 + li, py = op / TW, px = op - py * TW;
            const int oy = oy0 + py, ox = ox0 + px;
            const bool ok = oy < a.Ho && ox < a.Wo;
            const int64_t pix = ((int64_t)img * a.Ho + (ok ? oy : 0)) * a.Wo + (ok ? ox : 0);
            bf16x4 rv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = ct * 32 + 8 * g + 4 * lh;
                const bf16x4 z4 = {0, 0, 0, 0};
                rv[g] = (a.residual && ok && c0 < a.Cout) ? *(const bf16x4*)(a.x + pix * a.Cin + c0) : z4;   // stride 1, Cin == Cout: the same pixel
            }
            floatx16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 bv = *(const floatx4*)(Bo + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * g + e] = bv[e];
            }
            const char* wp = Wo + (size_t)(ct * 32 + li) * PH + lh * 16;
            const char* dp = Ds + (size_t)(rt * 32 + li) * PH + lh * 16;
            for (int ks = 0; ks < nks_h; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wp + ks * 32), *(const bf16x8*)(dp + ks * 32), acc, 0, 0, 0);
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = ct * 32 + 8 * g + 4 * lh;
                    if (c0 < a.Cout) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[4 * g + e] + (float)rv[g][e]);
                        *(bf16x4*)(a.y + pix * a.Cout + c0) = o;
                    }
                }
            }
        }
    }
}

size_t ir_lds_bytes(int k, int s, int th, int cinp, int hidp, int coutp) {
    const int ih = (th - 1) * s + k, iw = 7 * s + k;
    return (size_t)(ih * iw + th * 8 + coutp) * (hidp * 2 + 16) + (size_t)hidp * (cinp * 2 + 16) + (size_t)((k * k + 2) * hidp + coutp) * 4;
}
int ir_tile_rows(int k, int s, int cinp, int hidp, int coutp) {  // 8 output rows unless that leaves one workgroup per CU
    return ir_lds_bytes(k, s, 8, cinp, hidp, coutp) <= 80 * 1024 ? 8 : 4;
}

int ir_cus() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
              n = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return n;
}

template <int K, int S, int TH, bool HI> int ir_launch(IrArgs a, hipStream_t st) {
    const size_t lds = ir_lds_bytes(K, S, TH, a.cinp, a.hidp, a.coutp);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ir_block_kernel<K, S, TH, HI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ss_fail(-2, "ss_ir_block_fwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
    }
    a.tiles_x = (int)cdiv(a.Wo, 8); a.tiles_y = (int)cdiv(a.Ho, TH);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.B;
    if (ntiles >= (1LL << 31)) return ss_fail(-1, "ss_ir_block_fwd: too many tiles");
    int per_cu = (int)((160 * 1024) / lds); if (per_cu > 4) per_cu = 4; if (per_cu < 1) per_cu = 1;      // 128 registers: 4 workgroups per CU
    int64_t grid = (int64_t)ir_cus() * per_cu; if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((ir_block_kernel<K, S, TH, HI>), dim3((unsigned)grid), dim3(256), lds, st, a, (int)ntiles,
                       fast_div((unsigned)(a.tiles_x * a.tiles_y)), fast_div((unsigned)a.tiles_x));
    SS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int ss_ir_block_ok(int dtype, int Cin, int hidp, int Cout, int k, int s) {
    if (dtype != SS_BF16 || (k != 3 && k != 5) || (s != 1 && s != 2)) return 0;
    if (Cin <= 0 || Cin % 8 || Cin > 64 || Cout <= 0 || Cout % 8 || hidp <= 0 || hidp % 32) return 0;
    const int cinp = (Cin + 15) / 16 * 16, coutp = (Cout + 31) / 32 * 32;
    const int th = ir_tile_rows(k, s, cinp, hidp, coutp);
    if ((int64_t)th * hidp > 1024) return 0;                                                             // the depthwise item table of ss_ir_block_fwd
    return ir_lds_bytes(k, s, th, cinp, hidp, coutp) <= 80 * 1024;                                       // two workgroups per CU
}

// x [B, H, W, Cin] -> y [B, Ho, Wo, Cout] (channels-last bf16).  w_in [hidp][cinp] (cinp = Cin rounded up to 16), b_in [hidp],
// w_dw [k*k][hidp] tap-major fp32, b_dw [hidp], w_out [coutp][hidp] (coutp = Cout rounded up to 32), b_out [coutp]; the padding is zero.
// residual: y += x (needs stride 1, Cin == Cout, Ho == H, Wo == W).
extern "C" int ss_ir_block_fwd(int dtype, const void* x, void* y, int B, int H, int W, int Cin, int hidp, int Cout, int k, int s, int pad_top,
                               int pad_left, int Ho, int Wo, const void* w_in, const float* b_in, int act_in, const float* w_dw,
                               const float* b_dw, int act_dw, const void* w_out, const float* b_out, int residual, hipStream_t stream) {
    SS_CHECK(x && y && w_in && b_in && w_dw && b_dw && w_out && b_out, "ss_ir_block_fwd: null operand");
    SS_CHECK(ss_ir_block_ok(dtype, Cin, hidp, Cout, k, s), "ss_ir_block_fwd: needs bf16, k 3 or 5, stride 1 or 2, Cin <= 64 and Cin, Cout multiples of 8, hidden padded to 32");
    SS_CHECK((act_in == 0 || act_in == 2 || act_in == 3) && (act_dw == 0 || act_dw == 2 || act_dw == 3), "ss_ir_block_fwd: activations must be 0 (none), 2 (ReLU) or 3 (ReLU6)");
    SS_CHECK(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && pad_top >= 0 && pad_left >= 0 && pad_top < k && pad_left < k,
             "ss_ir_block_fwd: bad geometry");
    SS_CHECK((Ho - 1) * s - pad_top < H && (Wo - 1) * s - pad_left < W, "ss_ir_block_fwd: output larger than the input allows");
    SS_CHECK(!residual || (s == 1 && Cin == Cout && Ho == H && Wo == W), "ss_ir_block_fwd: the skip needs stride 1 and equal shapes");
    SS_CHECK(aligned16(x) && aligned16(y) && aligned16(w_in) && aligned16(w_out) && aligned16(b_in) && aligned16(b_out) && aligned16(w_dw) && aligned16(b_dw),
             "ss_ir_block_fwd: pointers must be 16-byte aligned");
    IrArgs a{};
    a.x = (const bf16*)x; a.y = (bf16*)y; a.w_in = (const bf16*)w_in; a.b_in = b_in; a.w_dw = w_dw; a.b_dw = b_dw; a.w_out = (const bf16*)w_out;
    a.b_out = b_out; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.cinp = (Cin + 15) / 16 * 16; a.hidp = hidp; a.Cout = Cout;
    a.coutp = (Cout + 31) / 32 * 32; a.pt = pad_top; a.pl = pad_left; a.Ho = Ho; a.Wo = Wo; a.act_in = act_in; a.act_dw = act_dw; a.residual = residual;
    const int th = ir_tile_rows(k, s, a.cinp, hidp, a.coutp);
    const bool hi = act_in == 3 || act_dw == 3;
    SS_CHECK((int64_t)th * 8 * hidp / 8 <= 1024, "ss_ir_block_fwd: hidden width too large for the depthwise item table");
#define SS_IR(K, S) (th == 8 ? (hi ? ir_launch<K, S, 8, true>(a, stream) : ir_launch<K, S, 8, false>(a, stream)) \
                             : (hi ? ir_launch<K, S, 4, true>(a, stream) : ir_launch<K, S, 4, false>(a, stream)))
    if (k == 3) return s == 1 ? SS_IR(3, 1) : SS_IR(3, 2);
    return s == 1 ? SS_IR(5, 1) : SS_IR(5, 2);
#undef SS_IR
}
