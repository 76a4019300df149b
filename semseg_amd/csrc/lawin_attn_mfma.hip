// Lawin window attention on the matrix cores (bf16, head dim D in {16, 32, 64}; reference modules/lawin.py:72-91).
//
// One wave per (window, head): 64 queries (the window's 8x8 pixels, image-ordered rows of theta / y / dY) against the 64
// context positions of that window (window-ordered rows of pg = phi | g).  Same transposed formulation as the MiT
// attention (attn_common.h): S^T[key][q] = phi . theta^T with lane = query, softmax lane-local (all 64 keys are resident:
// no online rescale), O^T[d][q] = g^T . P^T with the g^T operand produced by ds_read_b64_tr_b16 from a wave-private
// row-major LDS image.  Row operands (phi, g, theta, dY fragments) are read once per wave straight from global memory
// (no other wave shares them).  Backward = pass 1 (lane = query: dtheta) + pass 2 (lane = key: dphi, dg), S recomputed
// from the forward's base-2 log-sum-exp.  Outputs leave as 8-byte pieces of 4 consecutive channels per lane.
#include "attn_common.h"

// Workgroups b, b + 8, ... run on one XCD (one L2).  The waves of a window's heads read 32-128-byte slices of the same rows: handing
// each XCD a CONTIGUOUS run of (window, head) pairs keeps every 128-byte line in one L2 instead of fetching it into several
// (FETCH_SIZE of the d = 4 / d = 16 backward kernels was 3x the operand bytes with the round-robin order).
__device__ __forceinline__ unsigned xcd_block() {
    const unsigned n = gridDim.x, q = n >> 3, r = n & 7, x = blockIdx.x & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
}

struct LawinAttnArgs {
    const bf16 *theta, *pg, *y, *dy; const float* lse; bf16 *out, *dtheta, *dpg; float* lse_out;
    int64_t ldt, ldy, lddy, lddt;
    int H, W, nh, nw, inter, heads; int64_t nwin; float c2, scale;
};

template <int D> struct LwCfg {
    static constexpr int PT = D == 64 ? 192 : 64;        // transpose-read image pitch (bytes)
    static constexpr int IMG = 64 * PT;
    static constexpr int CH = D / 8, KS = D / 16, TD = D >= 32 ? D / 32 : 1;
};

__device__ __forceinline__ int64_t win_row(const LawinAttnArgs& a, int64_t wi, int t) {
    const int j = (int)(wi % a.nw); const int64_t q = wi / a.nw; const int i = (int)(q % a.nh); const int64_t b = q / a.nh;
    return (b * a.H + 8 * i + (t >> 3)) * a.W + 8 * j + (t & 7);
}

// 64 rows x D channels -> row-major LDS image (pitch PT); rowptr(t) = global pointer of row t's first channel
template <int D, typename F>
__device__ __forceinline__ void fill_image(char* img, F rowptr) {
    constexpr int CH = LwCfg<D>::CH, PT = LwCfg<D>::PT;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int c = lane + it * 64, row = c / CH, ch = c % CH;
        *(bf16x8*)(img + row * PT + ch * 16) = *(const bf16x8*)(rowptr(row) + ch * 8);
    }
}

// registers of a lane-major accumulator tile -> 8-byte pieces of one global row (channels d0 .. d0+3)
template <int D>
__device__ __forceinline__ void store_lane_row(const floatx16 (&acc)[LwCfg<D>::TD], float mul, bf16* rowp) {
    const int h = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int t = 0; t < LwCfg<D>::TD; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = t * 32 + 8 * g + 4 * h;
            if (d0 < D) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[t][4 * g + e] * mul);
                *(bf16x4*)(rowp + d0) = v;
            }
        }
}

template <int D>
__global__ __launch_bounds__(128) void lawin_attn_fwd_mfma_kernel(LawinAttnArgs a) {
    typedef LwCfg<D> G;
    constexpr int KS = G::KS, TD = G::TD, PT = G::PT;
    __shared__ __attribute__((aligned(16))) char smem[2 * G::IMG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int64_t idx = (int64_t)xcd_block() * 2 + wave;
    if (idx >= a.nwin * a.heads) return;
    const int64_t wi = idx / a.heads; const int hd = (int)(idx % a.heads);
    char* Vt = smem + wave * G::IMG;
    const bf16* pgw = a.pg + wi * 64 * 2 * (int64_t)a.inter + hd * D;
    fill_image<D>(Vt, [&](int r) { return pgw + (int64_t)r * 2 * a.inter + a.inter; });
    bf16x8 qf[2][KS], kf[2][KS];
    int64_t qrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        qrow[t] = win_row(a, wi, t * 32 + j);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[t][ks] = *(const bf16x8*)(a.theta + qrow[t] * a.ldt + hd * D + ks * 16 + 8 * h);
            kf[t][ks] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + ks * 16 + 8 * h);
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        floatx16 s[2];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[qt][ks], s[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = xor32_max(mx) * a.c2;      // c2 > 0: scale and log2 e go into the exponent's fma
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], a.c2, -mx)); s[kt][r] = p; l += p; }
        l = xor32_sum(l);
        floatx16 oacc[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf = cvt_frag<bf16>(s[kt], st);
#pragma unroll
                for (int t = 0; t < TD; ++t)
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, PT, kt * 32 + 16 * st, t * 32), pf, oacc[t], 0, 0, 0);
            }
        store_lane_row<D>(oacc, 1.f / l, a.out + qrow[qt] * a.ldy + hd * D);
        if (h == 0) a.lse_out[(wi * a.heads + hd) * 64 + qt * 32 + j] = mx + log2f(l);
    }
}

template <int D>
__global__ __launch_bounds__(128) void lawin_attn_bwd_mfma_kernel(LawinAttnArgs a) {
    typedef LwCfg<D> G;
    constexpr int KS = G::KS, TD = G::TD, PT = G::PT;
    constexpr int PER_WAVE = 2 * G::IMG + 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * PER_WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int64_t idx = (int64_t)xcd_block() * 2 + wave;
    if (idx >= a.nwin * a.heads) return;
    const int64_t wi = idx / a.heads; const int hd = (int)(idx % a.heads);
    char* R0 = smem + wave * PER_WAVE; char* R1 = R0 + G::IMG; float* lsd = (float*)(R1 + G::IMG);     // lse[64] | delta[64]
    const bf16* pgw = a.pg + wi * 64 * 2 * (int64_t)a.inter + hd * D;
    fill_image<D>(R0, [&](int r) { return pgw + (int64_t)r * 2 * a.inter; });                               // phi (K)
    fill_image<D>(R1, [&](int r) { return a.dy + win_row(a, wi, r) * a.lddy + hd * D; });                    // dY
    bf16x8 qf[2][KS], dof[2][KS], kf[2][KS], vf[2][KS];
    int64_t qrow[2]; float ls[2], dl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        qrow[t] = win_row(a, wi, t * 32 + j);
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int co = hd * D + ks * 16 + 8 * h;
            qf[t][ks] = *(const bf16x8*)(a.theta + qrow[t] * a.ldt + co);
            dof[t][ks] = *(const bf16x8*)(a.dy + qrow[t] * a.lddy + co);
            const bf16x8 of = *(const bf16x8*)(a.y + qrow[t] * a.ldy + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += (float)dof[t][ks][e] * (float)of[e];
            kf[t][ks] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + ks * 16 + 8 * h);
            vf[t][ks] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + a.inter + ks * 16 + 8 * h);
        }
        dl[t] = xor32_sum(dot);
        ls[t] = a.lse[(wi * a.heads + hd) * 64 + t * 32 + j];
        if (h == 0) { lsd[t * 32 + j] = ls[t]; lsd[64 + t * 32 + j] = dl[t]; }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass 1, lane = query: dtheta_q = scale * sum_k P (dP - delta) phi_k
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        floatx16 acc[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[qt][ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][ks], dof[qt][ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], a.c2, -ls[qt])) * fmaf(dp[r], a.scale, -dl[qt] * a.scale);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 df = cvt_frag<bf16>(s, st);
#pragma unroll
                for (int t = 0; t < TD; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(R0, PT, kt * 32 + 16 * st, t * 32), df, acc[t], 0, 0, 0);
            }
        }
        store_lane_row<D>(acc, 1.f, a.dtheta + qrow[qt] * a.lddt + hd * D);
    }
    // ---- pass 2, lane = key: dg_k = sum_q P dY_q ; dphi_k = scale * sum_q P (dP - delta_q) theta_q
    __builtin_amdgcn_wave_barrier();
    fill_image<D>(R0, [&](int r) { return a.theta + win_row(a, wi, r) * a.ldt + hd * D; });                  // theta (Q)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        floatx16 dk[TD], dv[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[qt][ks], kf[kt][ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[qt][ks], vf[kt][ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = qt * 32 + 8 * g + 4 * h;
                const floatx4 l4 = *(const floatx4*)(lsd + ql), d4 = *(const floatx4*)(lsd + 64 + ql);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], a.c2, -l4[e]));
                    s[r] = p;
                    dp[r] = p * (dp[r] - d4[e]) * a.scale;
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf = cvt_frag<bf16>(s, st), df = cvt_frag<bf16>(dp, st);
#pragma unroll
                for (int t = 0; t < TD; ++t) {
                    dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(R1, PT, qt * 32 + 16 * st, t * 32), pf, dv[t], 0, 0, 0);
                    dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(R0, PT, qt * 32 + 16 * st, t * 32), df, dk[t], 0, 0, 0);
                }
            }
        }
        bf16* drow = a.dpg + (wi * 64 + kt * 32 + j) * 2 * (int64_t)a.inter + hd * D;
        store_lane_row<D>(dk, 1.f, drow);
        store_lane_row<D>(dv, 1.f, drow + a.inter);
    }
}

// ------------------------------------------------------------------ head dims 2 / 4 / 8: a wave takes a GROUP of 16 / DH heads
// The 16 channels of a head group are one MFMA k-step.  Per head the query-side fragment (theta, dY) keeps only that head's
// channels (zeroed dwords: DH >= 2 channels = whole dwords), so S and dP come out per head from fragments loaded once per group;
// the products that end in channels (O = P g, dtheta, dphi, dg) mask the channel-side operand by LANE (the transposed image
// fragment has lane = channel; lanes 16..31 read image columns nobody filled and are always masked) and accumulate all heads of the group into ONE tile, normalised per channel at the store.
// (On the vector ALU this op needed ~13 instructions per (query, key) pair and head; here the pair costs the softmax only.)
template <int DH> struct GrpCfg { static constexpr int G = 16 / DH, PT = 64, IMG = 64 * PT; };

template <int DH>
__device__ __forceinline__ bf16x8 head_channels(const bf16x8& f, int hh, int h) {     // lane half h holds channels 8h .. 8h+7
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 v = __builtin_bit_cast(u4, f);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((8 * h + 2 * i) / DH == hh) ? v[i] : 0u;
    return __builtin_bit_cast(bf16x8, v);
}
template <int DH>
__device__ __forceinline__ bf16x8 head_lanes(const bf16x8& f, int hh, int li) {        // lane li = channel li of the group (>= 16: padding)
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 v = __builtin_bit_cast(u4, f);
    const bool keep = li / DH == hh;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = keep ? v[i] : 0u;
    return __builtin_bit_cast(bf16x8, v);
}
// rows 0..15 of a lane-major accumulator tile -> 8-byte pieces (channels 8g + 4h .. +3, g = 0, 1), each channel times mul[its head]
template <int DH>
__device__ __forceinline__ void store_group_row(const floatx16& acc, const float (&mul)[16 / DH], int h, bf16* rowp) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        bf16x4 v;                                       // both candidates, then a select: a select of the INDEX sends mul[] to scratch
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = acc[4 * g + e] * mul[(8 * g + e) / DH], hi = acc[4 * g + e] * mul[(8 * g + 4 + e) / DH];
            v[e] = (bf16)(h ? hi : lo);
        }
        *(bf16x4*)(rowp + 8 * g + 4 * h) = v;
    }
}

template <int DH>
__global__ __launch_bounds__(128) void lawin_attn_fwd_grp_kernel(LawinAttnArgs a) {
    typedef GrpCfg<DH> C;
    constexpr int G = C::G, PT = C::PT;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::IMG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int groups = a.heads / G;
    const int64_t idx = (int64_t)xcd_block() * 2 + wave;
    if (idx >= a.nwin * groups) return;
    const int64_t wi = idx / groups; const int gp = (int)(idx % groups);
    char* Vt = smem + wave * C::IMG;
    const bf16* pgw = a.pg + wi * 64 * 2 * (int64_t)a.inter + gp * 16;
    fill_image<16>(Vt, [&](int r) { return pgw + (int64_t)r * 2 * a.inter + a.inter; });
    bf16x8 qf[2], kf[2];
    int64_t qrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        qrow[t] = win_row(a, wi, t * 32 + j);
        qf[t] = *(const bf16x8*)(a.theta + qrow[t] * a.ldt + gp * 16 + 8 * h);
        kf[t] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + 8 * h);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        floatx16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
        float linv[G];
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            const bf16x8 qm = head_channels<DH>(qf[qt], hh, h);
            floatx16 s[2];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt], qm, s[kt], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
            }
            mx = xor32_max(mx) * a.c2;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], a.c2, -mx)); s[kt][r] = p; l += p; }
            l = xor32_sum(l);
            linv[hh] = 1.f / l;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_lanes<DH>(tr_frag(Vt, PT, kt * 32 + 16 * st, 0), hh, j), cvt_frag<bf16>(s[kt], st),
                                                                   oacc, 0, 0, 0);
            if (h == 0) a.lse_out[(wi * a.heads + gp * G + hh) * 64 + qt * 32 + j] = mx + log2f(l);
        }
        store_group_row<DH>(oacc, linv, h, a.out + qrow[qt] * a.ldy + gp * 16);
    }
}

template <int DH>
__global__ __launch_bounds__(128) void lawin_attn_bwd_grp_kernel(LawinAttnArgs a) {
    typedef GrpCfg<DH> C;
    constexpr int G = C::G, PT = C::PT;
    constexpr int PER_WAVE = 2 * C::IMG + G * 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * PER_WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int groups = a.heads / G;
    const int64_t idx = (int64_t)xcd_block() * 2 + wave;
    if (idx >= a.nwin * groups) return;
    const int64_t wi = idx / groups; const int gp = (int)(idx % groups);
    char* R0 = smem + wave * PER_WAVE; char* R1 = R0 + C::IMG; float* lsd = (float*)(R1 + C::IMG);     // [G][ lse[64] | delta[64] ]
    const bf16* pgw = a.pg + wi * 64 * 2 * (int64_t)a.inter + gp * 16;
    fill_image<16>(R0, [&](int r) { return pgw + (int64_t)r * 2 * a.inter; });                               // phi
    fill_image<16>(R1, [&](int r) { return a.dy + win_row(a, wi, r) * a.lddy + gp * 16; });                  // dY
    bf16x8 qf[2], dof[2], kf[2], vf[2];
    int64_t qrow[2]; float ls[2][G], dl[2][G];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        qrow[t] = win_row(a, wi, t * 32 + j);
        const int co = gp * 16 + 8 * h;
        qf[t] = *(const bf16x8*)(a.theta + qrow[t] * a.ldt + co);
        dof[t] = *(const bf16x8*)(a.dy + qrow[t] * a.lddy + co);
        const bf16x8 of = *(const bf16x8*)(a.y + qrow[t] * a.ldy + co);
        kf[t] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + 8 * h);
        vf[t] = *(const bf16x8*)(pgw + (int64_t)(t * 32 + j) * 2 * a.inter + a.inter + 8 * h);
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            float dot = 0.f;                             // this lane half's share of head hh (zero when the head lives in the other half)
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += ((8 * h + e) / DH == hh) ? (float)dof[t][e] * (float)of[e] : 0.f;
            dl[t][hh] = xor32_sum(dot) * a.scale;
            ls[t][hh] = a.lse[(wi * a.heads + gp * G + hh) * 64 + t * 32 + j];
            if (h == 0) { lsd[hh * 128 + t * 32 + j] = ls[t][hh]; lsd[hh * 128 + 64 + t * 32 + j] = dl[t][hh]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    float one[G];
#pragma unroll
    for (int hh = 0; hh < G; ++hh) one[hh] = 1.f;
    // ---- pass 1, lane = query: dtheta_q = scale * sum_k P (dP - delta) phi_k
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            const bf16x8 qm = head_channels<DH>(qf[qt], hh, h), dom = head_channels<DH>(dof[qt], hh, h);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                floatx16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt], qm, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt], dom, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], a.c2, -ls[qt][hh])) * fmaf(dp[r], a.scale, -dl[qt][hh]);
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_lanes<DH>(tr_frag(R0, PT, kt * 32 + 16 * st, 0), hh, j), cvt_frag<bf16>(s, st),
                                                                  acc, 0, 0, 0);
            }
        }
        store_group_row<DH>(acc, one, h, a.dtheta + qrow[qt] * a.lddt + gp * 16);
    }
    // ---- pass 2, lane = key: dg_k = sum_q P dY_q ; dphi_k = scale * sum_q P (dP - delta_q) theta_q
    __builtin_amdgcn_wave_barrier();
    fill_image<16>(R0, [&](int r) { return a.theta + win_row(a, wi, r) * a.ldt + gp * 16; });               // theta (columns 16..31 stay zero)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        floatx16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int hh = 0; hh < G; ++hh) {
                floatx16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_channels<DH>(qf[qt], hh, h), kf[kt], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_channels<DH>(dof[qt], hh, h), vf[kt], dp, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = qt * 32 + 8 * g + 4 * h;
                    const floatx4 l4 = *(const floatx4*)(lsd + hh * 128 + ql), d4 = *(const floatx4*)(lsd + hh * 128 + 64 + ql);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], a.c2, -l4[e]));
                        s[r] = p;
                        dp[r] = p * fmaf(dp[r], a.scale, -d4[e]);
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const bf16x8 pf = cvt_frag<bf16>(s, st), df = cvt_frag<bf16>(dp, st);
                    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_lanes<DH>(tr_frag(R1, PT, qt * 32 + 16 * st, 0), hh, j), pf, dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(head_lanes<DH>(tr_frag(R0, PT, qt * 32 + 16 * st, 0), hh, j), df, dk, 0, 0, 0);
                }
            }
        bf16* drow = a.dpg + (wi * 64 + kt * 32 + j) * 2 * (int64_t)a.inter + gp * 16;
        store_group_row<DH>(dk, one, h, drow);
        store_group_row<DH>(dv, one, h, drow + a.inter);
    }
}

// ------------------------------------------------------------------ host side (called from lawin.hip)
bool lawin_attn_mfma_ok(int D, const void* theta, int64_t ldt, const void* pg, int inter, const void* p2, int64_t ld2, const void* p3,
                        int64_t ld3, const void* p4, int64_t ld4) {
    const bool grouped = (D == 2 || D == 4 || D == 8) && inter % 16 == 0;        // 16 / D heads per wave
    if (!(D == 16 || D == 32 || D == 64 || grouped) || inter % 8 || ldt % 8 || !aligned16(theta) || !aligned16(pg)) return false;
    if (p2 && (ld2 % 8 || !aligned16(p2))) return false;
    if (p3 && (ld3 % 8 || !aligned16(p3))) return false;
    if (p4 && (ld4 % 8 || !aligned16(p4))) return false;
    return true;
}
int lawin_attn_fwd_mfma(const void* theta, int64_t ldt, const void* pg, void* y, int64_t ldy, float* lse, int B, int H, int W, int inter,
                        int heads, hipStream_t stream) {
    LawinAttnArgs a{};
    const int D = inter / heads;
    a.theta = (const bf16*)theta; a.pg = (const bf16*)pg; a.out = (bf16*)y; a.lse_out = lse; a.ldt = ldt; a.ldy = ldy;
    a.H = H; a.W = W; a.nh = H / 8; a.nw = W / 8; a.inter = inter; a.heads = heads; a.nwin = (int64_t)B * a.nh * a.nw;
    a.scale = 1.f / sqrtf((float)D); a.c2 = a.scale * 1.4426950408889634f;
    const unsigned blocks = (unsigned)cdiv(a.nwin * heads, 2);
    if (D <= 8) {
        const unsigned gb = (unsigned)cdiv(a.nwin * (heads / (16 / D)), 2);
        if (D == 8) hipLaunchKernelGGL(lawin_attn_fwd_grp_kernel<8>, dim3(gb), dim3(128), 0, stream, a);
        else if (D == 4) hipLaunchKernelGGL(lawin_attn_fwd_grp_kernel<4>, dim3(gb), dim3(128), 0, stream, a);
        else hipLaunchKernelGGL(lawin_attn_fwd_grp_kernel<2>, dim3(gb), dim3(128), 0, stream, a);
        SS_LAUNCH_CHECK();
        return 0;
    }
    if (D == 64) hipLaunchKernelGGL(lawin_attn_fwd_mfma_kernel<64>, dim3(blocks), dim3(128), 0, stream, a);
    else if (D == 32) hipLaunchKernelGGL(lawin_attn_fwd_mfma_kernel<32>, dim3(blocks), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL(lawin_attn_fwd_mfma_kernel<16>, dim3(blocks), dim3(128), 0, stream, a);
    SS_LAUNCH_CHECK();
    return 0;
}
int lawin_attn_bwd_mfma(const void* theta, int64_t ldt, const void* pg, const void* y, int64_t ldy, const void* dy, int64_t lddy,
                        const float* lse, void* dtheta, int64_t lddt, void* dpg, int B, int H, int W, int inter, int heads, hipStream_t stream) {
    LawinAttnArgs a{};
    const int D = inter / heads;
    a.theta = (const bf16*)theta; a.pg = (const bf16*)pg; a.y = (const bf16*)y; a.dy = (const bf16*)dy; a.lse = lse;
    a.dtheta = (bf16*)dtheta; a.dpg = (bf16*)dpg; a.ldt = ldt; a.ldy = ldy; a.lddy = lddy; a.lddt = lddt;
    a.H = H; a.W = W; a.nh = H / 8; a.nw = W / 8; a.inter = inter; a.heads = heads; a.nwin = (int64_t)B * a.nh * a.nw;
    a.scale = 1.f / sqrtf((float)D); a.c2 = a.scale * 1.4426950408889634f;
    const unsigned blocks = (unsigned)cdiv(a.nwin * heads, 2);
    if (D <= 8) {
        const unsigned gb = (unsigned)cdiv(a.nwin * (heads / (16 / D)), 2);
        if (D == 8) hipLaunchKernelGGL(lawin_attn_bwd_grp_kernel<8>, dim3(gb), dim3(128), 0, stream, a);
        else if (D == 4) hipLaunchKernelGGL(lawin_attn_bwd_grp_kernel<4>, dim3(gb), dim3(128), 0, stream, a);
        else hipLaunchKernelGGL(lawin_attn_bwd_grp_kernel<2>, dim3(gb), dim3(128), 0, stream, a);
        SS_LAUNCH_CHECK();
        return 0;
    }
    if (D == 64) hipLaunchKernelGGL(lawin_attn_bwd_mfma_kernel<64>, dim3(blocks), dim3(128), 0, stream, a);
    else if (D == 32) hipLaunchKernelGGL(lawin_attn_bwd_mfma_kernel<32>, dim3(blocks), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL(lawin_attn_bwd_mfma_kernel<16>, dim3(blocks), dim3(128), 0, stream, a);
    SS_LAUNCH_CHECK();
    return 0;
}
