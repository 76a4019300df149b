// Lawin position mixing on the matrix cores (reference modules/lawin.py:60-71,86-88): per head h a 64 x 64
// matrix mixes the 64 positions of every 8x8 context window,
//     out[w, p', c] = x[w, p', c] + b_h[p'] + sum_p W_h[p'][p] x[w, p, c],      c in head h  (C / heads channels each)
// and its weight gradient  dW_h[p'][p] = sum_{w, c in h} d[w, p', c] x[w, p, c],  db_h[p'] = sum d.
// That is 2 * 64 * 64 flop per element of a [windows * 64, C] tensor -- 17 GFLOP per call at the headline size, which the
// vector ALUs pay for 4x over (one broadcast LDS read per 4 fma) while v_mfma_f32_16x16x32_bf16 does it in the shadow of the
// tensor's one trip through HBM.
//
//   mix   : a workgroup owns 128 channels of a run of windows.  Each wave keeps the weight fragments of ITS heads in registers
//           for the whole run (B operand: rows of W_h, k-contiguous); a window's [64 positions x 128 channels] slice is staged
//           in LDS, its transpose is the A operand (ds_read_b64_tr_b16), so a lane ends up with 4 consecutive channels of one
//           position: the result (+ identity + bias) goes back into the LDS tile IN PLACE and leaves as whole 256-byte rows.
//   wgrad : a workgroup owns 64 channels of a chunk of windows; k runs over (window, channel), both operands are plain
//           16-byte LDS reads (channels are contiguous), 4 windows per step.  Per-chunk fp32 slabs, fixed-order final sum.
#include "common.h"
#include "attn_common.h"
#include "lawin_geom.h"
#include "colreduce.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct MixArgs { const bf16* G; const bf16* xwin; const bf16* Wp; const float* bm; bf16* out; WinGeom g; int C, heads, wpb; int64_t nwin; };

// source row of position p of window wi (channel offset c0): window-ordered tensor, or a gather from the pooled grid
__device__ __forceinline__ const bf16* mix_src_row(const bf16* G, const bf16* xwin, const WinGeom& g, int C, int64_t wi, int p, int c0) {
    if (xwin) return xwin + (wi * 64 + p) * C + c0;
    const int j = (int)(wi % g.nw); const int64_t q = wi / g.nw; const int i = (int)(q % g.nh); const int64_t b = q / g.nh;
    return G + ((b * g.gh + i * g.st + (p >> 3)) * g.gw + j * g.st + (p & 7)) * C + c0;
}

constexpr int MIX_PT = 128 * 2 + 16;                    // LDS row pitch of the 128-channel slice (bytes)

template <int CG>
__global__ __launch_bounds__(256) void lawin_mix_mfma_kernel(MixArgs a) {
    constexpr int HPT = CG < 16 ? 16 / CG : 1;           // heads inside one 16-channel tile
    constexpr int WH = CG == 8 ? 4 : CG == 16 ? 2 : 1;   // heads a wave works for (it owns 2 tiles = 32 channels)
    __shared__ __attribute__((aligned(16))) char tile[64 * MIX_PT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 128;
    bf16x8 wf[WH][4][2];
    float bias[WH][4];
#pragma unroll
    for (int hw = 0; hw < WH; ++hw) {
        const int head = (c0 + wave * 32 + (CG == 8 ? hw * 8 : CG == 16 ? hw * 16 : 0)) / CG;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[hw][pt][ks] = *(const bf16x8*)(a.Wp + ((int64_t)head * 64 + pt * 16 + j16) * 64 + ks * 32 + 8 * g);
            bias[hw][pt] = a.bm ? a.bm[head * 64 + pt * 16 + j16] : 0.f;
        }
    }
    const int64_t w0 = (int64_t)blockIdx.x * a.wpb, w1 = w0 + a.wpb < a.nwin ? w0 + a.wpb : a.nwin;
    bf16x8 st[4];
    auto gload = [&](int64_t wi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = tid + k * 256; st[k] = *(const bf16x8*)(mix_src_row(a.G, a.xwin, a.g, a.C, wi, idx >> 4, c0) + (idx & 15) * 8); }
    };
    if (w0 < w1) gload(w0);
    for (int64_t wi = w0; wi < w1; ++wi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = tid + k * 256; *(bf16x8*)(tile + (idx >> 4) * MIX_PT + (idx & 15) * 16) = st[k]; }
        __syncthreads();
        if (wi + 1 < w1) gload(wi + 1);                   // next window travels under this one's arithmetic
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n0 = (wave * 2 + tt) * 16;
            bf16x8 af[2];                                 // A = X^T: lane (channel n0 + j16, k group g) <- positions 32 ks + 8 g .. + 7
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* p0 = tile + (ks * 32 + 8 * g + (j16 >> 2)) * MIX_PT + (n0 + 4 * (j16 & 3)) * 2;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * MIX_PT));
#pragma unroll
                for (int e = 0; e < 4; ++e) { af[ks][e] = lo[e]; af[ks][4 + e] = hi[e]; }
            }
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) {
                const int hw = CG == 8 ? tt * 2 + hh : CG == 16 ? tt : 0;
                const bool mine = HPT == 1 || (g >> 1) == hh;     // accumulator rows 4g .. 4g+3 = channels n0 + 4g ..: which head's
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], wf[hw][pt][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], wf[hw][pt][1], acc, 0, 0, 0);
                    if (mine) {                               // position p' = 16 pt + j16, channels n0 + 4g .. + 3: identity + bias, in place
                        bf16x4* cell = (bf16x4*)(tile + (pt * 16 + j16) * MIX_PT + (n0 + 4 * g) * 2);
                        const bf16x4 xo = *cell;
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[r] + (float)xo[r] + bias[hw][pt]);
                        *cell = o;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 256;
            *(bf16x8*)(a.out + (wi * 64 + (idx >> 4)) * a.C + c0 + (idx & 15) * 8) = *(const bf16x8*)(tile + (idx >> 4) * MIX_PT + (idx & 15) * 16);
        }
        __syncthreads();
    }
}

// Wp[h][r][c] = bf16(transpose ? Wm[h][c][r] : Wm[h][r][c])
__global__ void lawin_mix_pack_kernel(const float* __restrict__ Wm, bf16* __restrict__ Wp, int heads, int transpose) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= heads * 4096) return;
    const int h = i >> 12, r = (i >> 6) & 63, c = i & 63;
    Wp[i] = (bf16)(transpose ? Wm[(h << 12) + c * 64 + r] : Wm[i]);
}

// ------------------------------------------------------------------ weight gradient
struct MixWgArgs { const bf16* G; const bf16* dwin; float* partial; WinGeom g; int C, heads, wpc, split; int64_t nwin; };
constexpr int WG_PT = 64 * 2 + 16;                       // row pitch of a 64-channel slice
constexpr int WG_WIN = 64 * WG_PT;                       // one window's slice

template <int CG>
__global__ __launch_bounds__(256) void lawin_mix_wgrad_mfma_kernel(MixWgArgs a) {
    constexpr int CGG = CG < 64 ? CG : 64;               // channels of one head inside this workgroup's 64
    constexpr int HG = 64 / CGG;                         // heads in the group: 8, 4, 2, 1
    constexpr int HPW = HG >= 4 ? HG / 4 : 1;            // heads per wave
    constexpr int WPH = HG >= 4 ? 1 : 4 / HG;            // waves per head (they split the 4 p' tile rows)
    constexpr int NPT = 4 / WPH;                         // p' tiles per wave and head
    constexpr int KS = 4 * CGG / 32;                     // k steps over (4 windows x CGG channels)
    __shared__ __attribute__((aligned(16))) char lds[2 * 4 * WG_WIN];     // d slices of 4 windows | x slices of 4 windows
    char* Dt = lds; char* Xt = lds + 4 * WG_WIN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64;
    const int64_t w0 = (int64_t)blockIdx.x * a.wpc, w1 = w0 + a.wpc < a.nwin ? w0 + a.wpc : a.nwin;
    f32x4 acc[HPW][NPT][4], accb[HPW][NPT];
#pragma unroll
    for (int h = 0; h < HPW; ++h)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            accb[h][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[h][pt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    const int head_lo = HG >= 4 ? wave * HPW : wave / WPH;            // first head (within the group) of this wave
    const int pt_lo = HG >= 4 ? 0 : (wave % WPH) * NPT;
    for (int64_t wb = w0; wb < w1; wb += 4) {
        __syncthreads();
        // 4 windows x 64 rows x 8 chunks for each of d and x: 4096 chunks, 16 per thread
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid + k * 256, wl = idx >> 9, row = (idx >> 3) & 63, ch = idx & 7;
            const int64_t wi = wb + wl;
            bf16x8 dv = zero, xv = zero;
            if (wi < w1) {
                dv = *(const bf16x8*)(a.dwin + (wi * 64 + row) * a.C + c0 + ch * 8);
                xv = *(const bf16x8*)(mix_src_row(a.G, nullptr, a.g, a.C, wi, row, c0) + ch * 8);
            }
            *(bf16x8*)(Dt + wl * WG_WIN + row * WG_PT + ch * 16) = dv;
            *(bf16x8*)(Xt + wl * WG_WIN + row * WG_PT + ch * 16) = xv;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            const int hoff = (head_lo + h) * CGG;                     // channel offset of the head inside the slice
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kk = ks * 32 + 8 * g, wl = kk / CGG, cc = kk % CGG;       // this lane group's (window, channel) run of 8
                bf16x8 af[NPT], bfr[4];
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) af[pt] = *(const bf16x8*)(Dt + wl * WG_WIN + ((pt_lo + pt) * 16 + j16) * WG_PT + (hoff + cc) * 2);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) bfr[jt] = *(const bf16x8*)(Xt + wl * WG_WIN + (jt * 16 + j16) * WG_PT + (hoff + cc) * 2);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) acc[h][pt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pt], bfr[jt], acc[h][pt][jt], 0, 0, 0);
                    accb[h][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pt], ones, accb[h][pt], 0, 0, 0);
                }
            }
        }
    }
    // slab [slab][head][p'][65]: D rows = p' (4g + r), columns = p (j16)
    const int slab = blockIdx.x * a.split + (blockIdx.y % a.split);
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int head = (c0 / CG) + (CG < 64 ? head_lo + h : 0);
        float* base = a.partial + ((int64_t)slab * a.heads + head) * 64 * 65;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* rowp = base + ((pt_lo + pt) * 16 + 4 * g + r) * 65;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) rowp[jt * 16 + j16] = acc[h][pt][jt][r];
                if (j16 == 0) rowp[64] = accb[h][pt][r];
            }
    }
}

// ------------------------------------------------------------------ host side (called from lawin.hip)
bool lawin_mix_mfma_ok(int C, int heads, const void* src, const void* out) {
    if (heads <= 0 || C % heads || C % 128) return false;
    const int cg = C / heads;
    return (cg == 8 || cg == 16 || cg == 32 || cg == 64 || cg == 128) && aligned16(src) && aligned16(out);
}
int64_t lawin_mix_mfma_workspace(int heads) { return (int64_t)heads * 4096 * 2; }

int lawin_mix_mfma(const void* G, const void* xwin, const float* Wm, const float* bm, void* out, int B, int H, int W, int C, int r, int heads,
                   int transpose, void* workspace, hipStream_t stream) {
    MixArgs a{};
    a.G = (const bf16*)G; a.xwin = (const bf16*)xwin; a.Wp = (const bf16*)workspace; a.bm = transpose ? nullptr : bm; a.out = (bf16*)out;
    a.g = win_geom(B, H, W, r); a.C = C; a.heads = heads;
    a.nwin = (int64_t)B * a.g.nh * a.g.nw;
    hipLaunchKernelGGL(lawin_mix_pack_kernel, dim3((unsigned)cdiv(heads * 4096, 256)), dim3(256), 0, stream, Wm, (bf16*)workspace, heads, transpose);
    SS_LAUNCH_CHECK();
    const int groups = C / 128;
    int64_t blocks_x = cdiv(2048, groups); if (blocks_x > a.nwin) blocks_x = a.nwin; if (blocks_x < 1) blocks_x = 1;
    a.wpb = (int)cdiv(a.nwin, blocks_x);
    dim3 grid((unsigned)cdiv(a.nwin, a.wpb), (unsigned)groups);
    switch (C / heads) {
        case 8: hipLaunchKernelGGL(lawin_mix_mfma_kernel<8>, grid, dim3(256), 0, stream, a); break;
        case 16: hipLaunchKernelGGL(lawin_mix_mfma_kernel<16>, grid, dim3(256), 0, stream, a); break;
        case 32: hipLaunchKernelGGL(lawin_mix_mfma_kernel<32>, grid, dim3(256), 0, stream, a); break;
        case 64: hipLaunchKernelGGL(lawin_mix_mfma_kernel<64>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(lawin_mix_mfma_kernel<128>, grid, dim3(256), 0, stream, a); break;
    }
    SS_LAUNCH_CHECK();
    return 0;
}

static void wgrad_plan(int B, int H, int W, int C, int heads, int64_t* chunks, int64_t* wpc, int* split) {
    const int64_t nwin = (int64_t)B * (H / 8) * (W / 8);
    const int groups = C / 64, cg = C / heads;
    int64_t ch = cdiv(512, groups); if (ch > cdiv(nwin, 4)) ch = cdiv(nwin, 4); if (ch < 1) ch = 1;
    int64_t per = cdiv(cdiv(nwin, ch), 4) * 4;
    *wpc = per; *chunks = cdiv(nwin, per); *split = cg > 64 ? cg / 64 : 1;
}
int64_t lawin_mix_wgrad_mfma_slabs(int B, int H, int W, int C, int heads) {
    int64_t chunks, wpc; int split;
    wgrad_plan(B, H, W, C, heads, &chunks, &wpc, &split);
    return chunks * split;
}
int lawin_mix_wgrad_mfma(const void* G, const void* dwin, float* dWb, int B, int H, int W, int C, int r, int heads, float* workspace,
                         hipStream_t stream) {
    MixWgArgs a{};
    a.G = (const bf16*)G; a.dwin = (const bf16*)dwin; a.partial = workspace; a.g = win_geom(B, H, W, r); a.C = C; a.heads = heads;
    a.nwin = (int64_t)B * a.g.nh * a.g.nw;
    int64_t chunks, wpc;
    wgrad_plan(B, H, W, C, heads, &chunks, &wpc, &a.split);
    a.wpc = (int)wpc;
    dim3 grid((unsigned)chunks, (unsigned)(C / 64));
    switch (C / heads) {
        case 8: hipLaunchKernelGGL(lawin_mix_wgrad_mfma_kernel<8>, grid, dim3(256), 0, stream, a); break;
        case 16: hipLaunchKernelGGL(lawin_mix_wgrad_mfma_kernel<16>, grid, dim3(256), 0, stream, a); break;
        case 32: hipLaunchKernelGGL(lawin_mix_wgrad_mfma_kernel<32>, grid, dim3(256), 0, stream, a); break;
        case 64: hipLaunchKernelGGL(lawin_mix_wgrad_mfma_kernel<64>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(lawin_mix_wgrad_mfma_kernel<128>, grid, dim3(256), 0, stream, a); break;
    }
    SS_LAUNCH_CHECK();
    const int ncols = heads * 64 * 65;
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((unsigned)cdiv(ncols, 8), 1), dim3(256), 0, stream, workspace, (int)(chunks * a.split), ncols, 1.f,
                       dWb, 0);
    SS_LAUNCH_CHECK();
    return 0;
}
