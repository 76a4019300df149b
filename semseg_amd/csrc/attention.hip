// MiT efficient self-attention, flash-style (reference modules/segformer.py:32-39).
//
// Shapes: Q [B,N,C] (head hd at channels hd*D..), KV [B,Nk,2C] (K = channels [0,C), V = [C,2C)), O like Q.
// Nk = (S/32)^2 is small (<= 4096) while N is large, so K/V of one (image, head) stay L2-resident and every operand
// fragment is fetched with 16-byte global loads straight into MFMA registers: no LDS in the main loops.
//
// Everything is computed "transposed" so that the softmax axis is lane-local (wave = 64 lanes, MFMA 32x32x16):
//   S^T[key][q] = K . Q^T           A = K rows,            B = Q rows   -> lane = query, 16 regs = 16 of the tile's 32 keys
//   O^T[d][q]  += Vt[d][key] . P^T  A = Vt (pre-transposed, keys pre-permuted to the accumulator's k order), B = P^T regs
// The accumulator of the first product is used directly as the B operand of the second one (k order inside a 16-key
// step: element e of lane-half h is key 16s + 8(e>>2) + 4h + (e&3)); attn_transpose_perm_kernel stores the transposed
// operands in exactly that order so their fragments are contiguous 16-byte loads.
// fp32 tensors run the same code with eight v_mfma_f32_32x32x2_f32 per k-step (exact fp32 products).
//
// Backward = two kernels (deterministic, no atomics): dQ (lane = query, loops keys) and dK/dV (lane = key, loops a
// slice of the queries, fp32 partial slabs summed by attn_bwd_finish_kernel).  P is recomputed from the forward LSE.
#include "common.h"
#include "attn_common.h"

// ------------------------------------------------------------------ operand preparation
// Xt[b][hd][dd][pos] = X[b][16*(pos>>4) + perm16(pos&15)][off + hd*D + dd]   (zero for rows >= n);  np = n rounded up to 32
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_transpose_perm_kernel(const T* __restrict__ x, int64_t ld, int off, T* __restrict__ xt,
                                                                  int n, int np, int heads) {
    __shared__ float tile[64 * (D + 1)];
    const int b = blockIdx.z, hd = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int c = tid; c < 64 * D / 8; c += 256) {
        int row = c / (D / 8), cc = c % (D / 8);
        F8 v = (t0 + row < n) ? load8(x + ((int64_t)b * n + t0 + row) * ld + off + hd * D + cc * 8) : zero8();
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[row * (D + 1) + cc * 8 + e] = v.v[e];
    }
    __syncthreads();
    for (int i = tid; i < 64 * D; i += 256) {
        int dd = i / 64, pos = i % 64;
        if (t0 + pos >= np) continue;
        int src = 16 * (pos >> 4) + perm16(pos & 15);
        xt[((int64_t)(b * heads + hd) * D + dd) * np + t0 + pos] = from_f<T>(tile[src * (D + 1) + dd]);
    }
}

// delta[b][hd][q] = sum_dd dO[b,q,hd*D+dd] * O[b,q,hd*D+dd]
template <typename T>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const T* __restrict__ a, const T* __restrict__ bb, float* __restrict__ delta,
                                                         int64_t rows_total, int n, int np, int C, int D, int heads) {
    const int chunks = C / 8, gl = D / 8;
    const int64_t total = rows_total * chunks;                 // multiple of gl
    const int64_t padded = cdiv(total, 64) * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        const bool ok = i < total;
        const int64_t row = ok ? i / chunks : 0; const int ch = ok ? (int)(i % chunks) : 0;
        if (ok) { F8 u = load8(a + row * C + ch * 8), v = load8(bb + row * C + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += u.v[e] * v.v[e]; }
        s = group_sum(s, gl);
        if (ok && (ch % gl) == 0) { int64_t b = row / n; int q = (int)(row % n), hd = ch / gl; delta[(b * heads + hd) * np + q] = s; }
    }
}

// ------------------------------------------------------------------ forward
template <typename T, int D>
__global__ __launch_bounds__(256) void mit_attn_fwd_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ kv, int64_t ldkv,
        const T* __restrict__ vtp, T* __restrict__ o, int64_t ldo, float* __restrict__ lse, int N, int Np, int Nk, int Nkp, int heads, float c) {
    typedef typename AT<T>::frag frag;
    __shared__ float shm[4 * 32 * (D + 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= N) return;
    const int qrow = min(q0 + j, N - 1);
    const T* qp = q + ((int64_t)b * N + qrow) * ldq + hd * D;
    frag qf[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) qf[s] = ldg_frag(qp + 16 * s + 8 * h);
    const T* kbase = kv + (int64_t)b * Nk * ldkv + hd * D;
    const T* vt = vtp + (int64_t)(b * heads + hd) * D * Nkp;
    floatx16 oacc[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int key0 = 0; key0 < Nkp; key0 += 32) {
        const int krow = min(key0 + j, Nk - 1);
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) s = mma32(ldg_frag(kbase + (int64_t)krow * ldkv + 16 * ks + 8 * h), qf[ks], s);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { float v = (key0 + acc_row(r, h) < Nk) ? s[r] * c : -INFINITY; s[r] = v; mx = fmaxf(mx, v); }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx), alpha = exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { float p = exp2f(s[r] - mn); s[r] = p; ps += p; }
        l = l * alpha + ps;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            frag pf = cvt_frag<T>(s, st);
#pragma unroll
            for (int t = 0; t < D / 32; ++t)
                oacc[t] = mma32(ldg_frag(vt + (int64_t)(t * 32 + j) * Nkp + key0 + 16 * st + 8 * h), pf, oacc[t]);
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (h == 0 && q0 + j < N) lse[(int64_t)(b * heads + hd) * Np + q0 + j] = m + log2f(l);
    store_transposed<T, D>(oacc, 1.f / l, shm + wave * 32 * (D + 1), o + ((int64_t)b * N + q0) * ldo + hd * D, ldo, N - q0);
}

// ------------------------------------------------------------------ backward: dQ
template <typename T, int D>
__global__ __launch_bounds__(256) void mit_attn_bwd_dq_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ kv, int64_t ldkv,
        const T* __restrict__ ktp, const T* __restrict__ dout, int64_t ldo, const float* __restrict__ lse, const float* __restrict__ delta,
        T* __restrict__ dq, int64_t lddq, int N, int Np, int Nk, int Nkp, int heads, int C, float c, float scale) {
    typedef typename AT<T>::frag frag;
    __shared__ float shm[4 * 32 * (D + 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= N) return;
    const int qrow = min(q0 + j, N - 1);
    frag qf[D / 16], dof[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
        qf[s] = ldg_frag(q + ((int64_t)b * N + qrow) * ldq + hd * D + 16 * s + 8 * h);
        dof[s] = ldg_frag(dout + ((int64_t)b * N + qrow) * ldo + hd * D + 16 * s + 8 * h);
    }
    const float ls = lse[(int64_t)(b * heads + hd) * Np + qrow], dl = delta[(int64_t)(b * heads + hd) * Np + qrow];
    const T* kbase = kv + (int64_t)b * Nk * ldkv + hd * D;
    const T* vbase = kbase + C;
    const T* kt = ktp + (int64_t)(b * heads + hd) * D * Nkp;
    floatx16 acc[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int key0 = 0; key0 < Nkp; key0 += 32) {
        const int krow = min(key0 + j, Nk - 1);
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            s = mma32(ldg_frag(kbase + (int64_t)krow * ldkv + 16 * ks + 8 * h), qf[ks], s);
            dp = mma32(ldg_frag(vbase + (int64_t)krow * ldkv + 16 * ks + 8 * h), dof[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = (key0 + acc_row(r, h) < Nk) ? exp2f(s[r] * c - ls) : 0.f;
            s[r] = p * (dp[r] - dl) * scale;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            frag df = cvt_frag<T>(s, st);
#pragma unroll
            for (int t = 0; t < D / 32; ++t)
                acc[t] = mma32(ldg_frag(kt + (int64_t)(t * 32 + j) * Nkp + key0 + 16 * st + 8 * h), df, acc[t]);
        }
    }
    store_transposed<T, D>(acc, 1.f, shm + wave * 32 * (D + 1), dq + ((int64_t)b * N + q0) * lddq + hd * D, lddq, N - q0);
}

// ------------------------------------------------------------------ backward: dK, dV  (lane = key)
// slabs: dk_slab/dv_slab [nsplit][B][heads][Nkp][D] fp32
template <typename T, int D>
__global__ __launch_bounds__(256) void mit_attn_bwd_dkv_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ qtp,
        const T* __restrict__ dout, int64_t ldo, const T* __restrict__ dotp, const T* __restrict__ kv, int64_t ldkv,
        const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dk_slab, float* __restrict__ dv_slab,
        int B, int N, int Np, int Nk, int Nkp, int heads, int C, float c, float scale, int nsplit, int q_per_split) {
    typedef typename AT<T>::frag frag;
    __shared__ float shm[4 * 32 * (D + 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / nsplit, split = blockIdx.z % nsplit, hd = blockIdx.y;
    const int key0 = (blockIdx.x * 4 + wave) * 32;
    if (key0 >= Nkp) return;
    const int krow = min(key0 + j, Nk - 1);
    const bool kvalid = key0 + j < Nk;
    const T* kp = kv + ((int64_t)b * Nk + krow) * ldkv + hd * D;
    frag kf[D / 16], vf[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) { kf[s] = ldg_frag(kp + 16 * s + 8 * h); vf[s] = ldg_frag(kp + C + 16 * s + 8 * h); }
    const float* lsep = lse + (int64_t)(b * heads + hd) * Np;
    const float* dlp = delta + (int64_t)(b * heads + hd) * Np;
    const T* qt = qtp + (int64_t)(b * heads + hd) * D * Np;
    const T* dot = dotp + (int64_t)(b * heads + hd) * D * Np;
    floatx16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }
    const int qbeg = split * q_per_split, qend = min(Np, qbeg + q_per_split);
    for (int q0 = qbeg; q0 < qend; q0 += 32) {
        const int qrow = min(q0 + j, N - 1);
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            s = mma32(ldg_frag(q + ((int64_t)b * N + qrow) * ldq + hd * D + 16 * ks + 8 * h), kf[ks], s);
            dp = mma32(ldg_frag(dout + ((int64_t)b * N + qrow) * ldo + hd * D + 16 * ks + 8 * h), vf[ks], dp);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 l4 = *(const floatx4*)(lsep + q0 + 8 * g + 4 * h), d4 = *(const floatx4*)(dlp + q0 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const bool ok = kvalid && (q0 + 8 * g + 4 * h + e < N);
                float p = ok ? exp2f(s[r] * c - l4[e]) : 0.f;
                s[r] = p;
                dp[r] = p * (dp[r] - d4[e]) * scale;
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            frag pf = cvt_frag<T>(s, st), df = cvt_frag<T>(dp, st);
#pragma unroll
            for (int t = 0; t < D / 32; ++t) {
                dv[t] = mma32(ldg_frag(dot + (int64_t)(t * 32 + j) * Np + q0 + 16 * st + 8 * h), pf, dv[t]);
                dk[t] = mma32(ldg_frag(qt + (int64_t)(t * 32 + j) * Np + q0 + 16 * st + 8 * h), df, dk[t]);
            }
        }
    }
    const int64_t slab = (((int64_t)split * B + b) * heads + hd) * Nkp + key0;
    store_transposed<float, D>(dk, 1.f, shm + wave * 32 * (D + 1), dk_slab + slab * D, D, 32);
    store_transposed<float, D>(dv, 1.f, shm + wave * 32 * (D + 1), dv_slab + slab * D, D, 32);
}

// dkv[(b*Nk + key)*ldkv + {0,C} + hd*D + dd] = sum_split slab
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_finish_kernel(const float* __restrict__ dk_slab, const float* __restrict__ dv_slab, T* __restrict__ dkv,
                                       int64_t ldkv, int B, int Nk, int Nkp, int heads, int D, int C, int nsplit) {
    // a thread owns 8 consecutive channels of one (image, key) row: D % 8 == 0, so they lie inside one head and are contiguous
    // in the slab; the splits are summed in a fixed order (deterministic)
    const int cpr = 2 * C / 8;                                        // 8-channel chunks per output row (dK then dV)
    const int64_t total = (int64_t)B * Nk * cpr;
    const int64_t slab_sz = (int64_t)B * heads * Nkp * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int chunk = (int)(i % cpr); const int64_t row = i / cpr;
        const int key = (int)(row % Nk); const int64_t b = row / Nk;
        const int ch = chunk * 8;
        const float* src = ch < C ? dk_slab : dv_slab;
        const int cc = ch < C ? ch : ch - C, hd = cc / D, dd = cc % D;
        const float* p = src + ((b * heads + hd) * Nkp + key) * D + dd;
        F8 acc = load8(p);
        for (int z = 1; z < nsplit; ++z) { const F8 v = load8(p + z * slab_sz);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e]; }
        store8(dkv + row * ldkv + ch, acc);
    }
}

// ------------------------------------------------------------------ host entry points
static bool attn_shape_ok(int B, int N, int Nk, int heads, int D, int C) {
    return B > 0 && N > 0 && Nk > 0 && heads > 0 && (D == 32 || D == 64) && C == heads * D;
}
static inline int round32(int n) { return (n + 31) / 32 * 32; }

// workspace (fwd): Vtp [B,heads,D,Nkp] elements of T.   lse: [B,heads,Np] fp32 with Np = round32(N).
extern "C" int64_t ss_mit_attn_fwd_workspace(int dtype, int B, int Nk, int heads, int D) {
    return (int64_t)B * heads * D * round32(Nk) * (dtype == SS_BF16 ? 2 : 4);
}
extern "C" int ss_mit_attn_fwd(int dtype, const void* q, const void* kv, void* o, float* lse, int B, int N, int Nk, int heads, int D,
                               float scale, void* workspace, int64_t ws_bytes, hipStream_t stream) {
    const int C = heads * D;
    SS_CHECK(q && kv && o && lse && workspace, "ss_mit_attn_fwd: null pointer");
    SS_CHECK(attn_shape_ok(B, N, Nk, heads, D, C), "ss_mit_attn_fwd: unsupported shape B=%d N=%d Nk=%d heads=%d D=%d", B, N, Nk, heads, D);
    SS_CHECK(ws_bytes >= ss_mit_attn_fwd_workspace(dtype, B, Nk, heads, D), "ss_mit_attn_fwd: workspace too small");
    SS_CHECK(aligned16(q) && aligned16(kv) && aligned16(o) && aligned16(workspace), "ss_mit_attn_fwd: pointers must be 16-byte aligned");
    const int Np = round32(N), Nkp = round32(Nk);
    const float c = scale * 1.4426950408889634f;
    if (dtype == SS_BF16) {
        AttnBf16Args a{};
        a.q = (const bf16*)q; a.kv = (const bf16*)kv; a.out = (bf16*)o; a.lse_out = lse;
        a.B = B; a.N = N; a.Np = Np; a.Nk = Nk; a.Nkp = Nkp; a.heads = heads; a.D = D; a.C = C; a.c = c; a.scale = scale;
        return attn_fwd_bf16_launch(a, stream);
    }
    SS_DISPATCH(dtype, {
        dim3 tg((unsigned)cdiv(Nkp, 64), heads, B), g((unsigned)cdiv(cdiv(N, 32), 4), heads, B);
        if (D == 64) {
            hipLaunchKernelGGL((attn_transpose_perm_kernel<T, 64>), tg, dim3(256), 0, stream, (const T*)kv, (int64_t)2 * C, C, (T*)workspace, Nk, Nkp, heads);
            hipLaunchKernelGGL((mit_attn_fwd_kernel<T, 64>), g, dim3(256), 0, stream, (const T*)q, (int64_t)C, (const T*)kv, (int64_t)2 * C,
                               (const T*)workspace, (T*)o, (int64_t)C, lse, N, Np, Nk, Nkp, heads, c);
        } else {
            hipLaunchKernelGGL((attn_transpose_perm_kernel<T, 32>), tg, dim3(256), 0, stream, (const T*)kv, (int64_t)2 * C, C, (T*)workspace, Nk, Nkp, heads);
            hipLaunchKernelGGL((mit_attn_fwd_kernel<T, 32>), g, dim3(256), 0, stream, (const T*)q, (int64_t)C, (const T*)kv, (int64_t)2 * C,
                               (const T*)workspace, (T*)o, (int64_t)C, lse, N, Np, Nk, Nkp, heads, c);
        }
    });
    SS_LAUNCH_CHECK();
    return 0;
}

struct BwdArgs {
    const void *q, *kv, *o, *dout; const float* lse; void *dq, *dkv; float* delta; void *ktp, *qtp, *dotp; float *dks, *dvs;
    int B, N, Np, Nk, Nkp, heads, C; float c, scale; int ns, qps;
};
template <typename T, int DD>
static void attn_bwd_launch(const BwdArgs& a, hipStream_t stream) {
    const int B = a.B, N = a.N, Np = a.Np, Nk = a.Nk, Nkp = a.Nkp, heads = a.heads, C = a.C;
    const int64_t rows = (int64_t)B * N;
    int rb = (int)(cdiv(rows * (C / 8), 256) < 8192 ? cdiv(rows * (C / 8), 256) : 8192);
    hipLaunchKernelGGL((attn_rowdot_kernel<T>), dim3(rb), dim3(256), 0, stream, (const T*)a.dout, (const T*)a.o, a.delta, rows, N, Np, C, DD, heads);
    dim3 tk((unsigned)cdiv(Nkp, 64), heads, B), tq((unsigned)cdiv(Np, 64), heads, B);
    dim3 gq((unsigned)cdiv(cdiv(N, 32), 4), heads, B), gk((unsigned)cdiv(Nkp / 32, 4), heads, B * a.ns);
    int64_t tot = (int64_t)B * Nk * 2 * C / 8;
    int fb = (int)(cdiv(tot, 256) < 4096 ? cdiv(tot, 256) : 4096);
    hipLaunchKernelGGL((attn_transpose_perm_kernel<T, DD>), tk, dim3(256), 0, stream, (const T*)a.kv, (int64_t)2 * C, 0, (T*)a.ktp, Nk, Nkp, heads);
    hipLaunchKernelGGL((attn_transpose_perm_kernel<T, DD>), tq, dim3(256), 0, stream, (const T*)a.q, (int64_t)C, 0, (T*)a.qtp, N, Np, heads);
    hipLaunchKernelGGL((attn_transpose_perm_kernel<T, DD>), tq, dim3(256), 0, stream, (const T*)a.dout, (int64_t)C, 0, (T*)a.dotp, N, Np, heads);
    hipLaunchKernelGGL((mit_attn_bwd_dq_kernel<T, DD>), gq, dim3(256), 0, stream, (const T*)a.q, (int64_t)C, (const T*)a.kv, (int64_t)2 * C, (const T*)a.ktp,
                       (const T*)a.dout, (int64_t)C, a.lse, a.delta, (T*)a.dq, (int64_t)C, N, Np, Nk, Nkp, heads, C, a.c, a.scale);
    hipLaunchKernelGGL((mit_attn_bwd_dkv_kernel<T, DD>), gk, dim3(256), 0, stream, (const T*)a.q, (int64_t)C, (const T*)a.qtp, (const T*)a.dout, (int64_t)C,
                       (const T*)a.dotp, (const T*)a.kv, (int64_t)2 * C, a.lse, a.delta, a.dks, a.dvs, B, N, Np, Nk, Nkp, heads, C, a.c, a.scale, a.ns, a.qps);
    hipLaunchKernelGGL((attn_bwd_finish_kernel<T>), dim3(fb), dim3(256), 0, stream, a.dks, a.dvs, (T*)a.dkv, (int64_t)2 * C, B, Nk, Nkp, heads, DD, C, a.ns);
}
static void bwd_plan(int B, int N, int Nk, int heads, int* nsplit, int* qps) {
    const int Np = round32(N), Nkp = round32(Nk);
    int64_t blocks = (int64_t)cdiv(Nkp / 32, 4) * heads * B;
    int64_t want = cdiv(1024, blocks);
    int64_t maxs = cdiv(Np, 256);
    int64_t s = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (s < 1) s = 1;
    int per = (int)(cdiv(cdiv(Np, s), 32) * 32);
    *qps = per; *nsplit = (int)cdiv(Np, per);
}
// workspace layout (bytes): delta fp32 [B,heads,Np] | Ktp | Qtp | dOtp (T, [B,heads,D,*]) | dk slabs | dv slabs (fp32)
extern "C" int64_t ss_mit_attn_bwd_workspace(int dtype, int B, int N, int Nk, int heads, int D) {
    const int64_t Np = round32(N), Nkp = round32(Nk), es = dtype == SS_BF16 ? 2 : 4;
    int ns, qps; bwd_plan(B, N, Nk, heads, &ns, &qps);
    int64_t bytes = (int64_t)B * heads * Np * 4;
    bytes += (int64_t)B * heads * D * Nkp * es + 2 * (int64_t)B * heads * D * Np * es;
    bytes = (bytes + 255) / 256 * 256;
    bytes += 2 * (int64_t)ns * B * heads * Nkp * D * 4;
    return bytes + 1024;
}
extern "C" int ss_mit_attn_bwd(int dtype, const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq,
                               void* dkv, int B, int N, int Nk, int heads, int D, float scale, void* workspace, int64_t ws_bytes,
                               hipStream_t stream) {
    const int C = heads * D;
    SS_CHECK(q && kv && o && dout && lse && dq && dkv && workspace, "ss_mit_attn_bwd: null pointer");
    SS_CHECK(attn_shape_ok(B, N, Nk, heads, D, C), "ss_mit_attn_bwd: unsupported shape B=%d N=%d Nk=%d heads=%d D=%d", B, N, Nk, heads, D);
    SS_CHECK(ws_bytes >= ss_mit_attn_bwd_workspace(dtype, B, N, Nk, heads, D), "ss_mit_attn_bwd: workspace too small");
    SS_CHECK(aligned16(q) && aligned16(kv) && aligned16(o) && aligned16(dout) && aligned16(dq) && aligned16(workspace), "ss_mit_attn_bwd: pointers must be 16-byte aligned");
    const int Np = round32(N), Nkp = round32(Nk);
    const int64_t es = dtype == SS_BF16 ? 2 : 4;
    int ns, qps; bwd_plan(B, N, Nk, heads, &ns, &qps);
    char* w = (char*)workspace;
    float* delta = (float*)w; w += (int64_t)B * heads * Np * 4;
    void* ktp = w; w += (int64_t)B * heads * D * Nkp * es;
    void* qtp = w; w += (int64_t)B * heads * D * Np * es;
    void* dotp = w; w += (int64_t)B * heads * D * Np * es;
    w = (char*)workspace + (((w - (char*)workspace) + 255) / 256 * 256);
    float* dks = (float*)w; w += (int64_t)ns * B * heads * Nkp * D * 4;
    float* dvs = (float*)w;
    const float c = scale * 1.4426950408889634f;
    // (delta needs no clearing: every entry a kernel reads, q < N, is written by the dQ / row-dot kernel before)
    if (dtype == SS_BF16) {
        const int64_t rows = (int64_t)B * N;
        int rb = (int)(cdiv(rows * (C / 8), 256) < 8192 ? cdiv(rows * (C / 8), 256) : 8192);
        (void)rb; (void)rows;                            // delta is produced inside the bf16 dQ kernel
        AttnBf16Args f{};
        f.q = (const bf16*)q; f.kv = (const bf16*)kv; f.o = (const bf16*)o; f.dout = (const bf16*)dout; f.lse = lse; f.dq = (bf16*)dq;
        f.dkv = (bf16*)dkv; f.delta = delta; f.dks = dks; f.dvs = dvs;
        f.B = B; f.N = N; f.Np = Np; f.Nk = Nk; f.Nkp = Nkp; f.heads = heads; f.D = D; f.C = C; f.c = c; f.scale = scale; f.ns = ns; f.qps = qps;
        int rc = attn_bwd_bf16_launch(f, stream);
        if (rc) return rc;
        int64_t tot = (int64_t)B * Nk * 2 * C / 8;
        int fb = (int)(cdiv(tot, 256) < 4096 ? cdiv(tot, 256) : 4096);
        hipLaunchKernelGGL((attn_bwd_finish_kernel<bf16>), dim3(fb), dim3(256), 0, stream, dks, dvs, (bf16*)dkv, (int64_t)2 * C, B, Nk, Nkp, heads, D, C, ns);
        SS_LAUNCH_CHECK();
        return 0;
    }
    BwdArgs a{q, kv, o, dout, lse, dq, dkv, delta, ktp, qtp, dotp, dks, dvs, B, N, Np, Nk, Nkp, heads, C, c, scale, ns, qps};
    SS_DISPATCH(dtype, { if (D == 64) attn_bwd_launch<T, 64>(a, stream); else attn_bwd_launch<T, 32>(a, stream); });
    SS_LAUNCH_CHECK();
    return 0;
}
