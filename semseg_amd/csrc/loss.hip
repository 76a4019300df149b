// Per-pixel losses and the confusion-matrix accumulator (reference loss/seg_loss.py, helper/weights.py,
// helper/metrics.py:90-95).  Logits/probabilities are addressed with explicit (n, c, h, w) element strides so both
// NCHW-contiguous tensors and the channels-last views the networks return are read in place.  Labels: int64 [B,H,W].
//
// OHEM cross-entropy runs entirely on the device (no host sync): true-class probability + NLL per pixel, an exact radix
// select (3 histogram passes over the float bit patterns) of the k-th smallest valid probability with the reference's
// stable-order tie rule AND its quirk (the index found among the VALID pixels is applied to the ALL-pixel vector,
// loss/seg_loss.py:141-143), the stateful never-decreasing threshold (:143-144), then a fixed-order masked mean.
#include "common.h"
#include <mutex>

struct Strides { int64_t n, c, h, w; };
struct OhemCtl {
    long long num_valid, k, krem, kept, j;
    unsigned prefix; int do_select; int chunk; long long valid_before, eq_before;
    float thr; float loss_sum;
};
#define OHEM_CHUNK 4096
#define OHEM_BINS 2048

template <typename T>
__device__ __forceinline__ void softmax_stats(const T* __restrict__ p, int64_t sc, int C, float& mx, float& sum) {
    mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, to_f(p[c * sc]));
    sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(to_f(p[c * sc]) - mx);
}
__device__ __forceinline__ int64_t pix_off(int64_t i, int H, int W, const Strides& s) {
    const int x = (int)(i % W); const int64_t q = i / W; const int y = (int)(q % H); const int64_t b = q / H;
    return b * s.n + y * s.h + x * s.w;
}

// prob[i] = softmax(logits)[label] (0 if the label is outside [0,C)), nll[i] = -log prob, vkey[i] = float bits of prob for valid
// pixels (label != ignore) else 0xFFFFFFFF; ctl->num_valid += #valid
template <typename T>
__global__ __launch_bounds__(256) void ce_prob_kernel(const T* __restrict__ logits, Strides s, const long long* __restrict__ labels,
        float* __restrict__ prob, float* __restrict__ nll, unsigned* __restrict__ vkey, OhemCtl* ctl, int64_t npix, int H, int W, int C,
        long long ignore) {
    long long nv = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const T* p = logits + pix_off(i, H, W, s);
        const long long lab = labels[i];
        float mx, sum; softmax_stats(p, s.c, C, mx, sum);
        float pr = 0.f, nl = 0.f;
        if (lab >= 0 && lab < C) { const float z = to_f(p[lab * s.c]) - mx; nl = logf(sum) - z; pr = expf(z) / sum; }
        const bool valid = lab != ignore;
        prob[i] = pr; nll[i] = nl; vkey[i] = valid ? __float_as_uint(pr) : 0xFFFFFFFFu;
        nv += valid;
    }
    __shared__ long long sh[256];
    sh[threadIdx.x] = nv; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && sh[0]) atomicAdd((unsigned long long*)&ctl->num_valid, (unsigned long long)sh[0]);
}

// ---- channels-last fast path (what the networks of this package return: pixel-major, C <= 8 contiguous channels per pixel, `pitch`
// elements apart): the pixel's logits are loaded once into registers (4-byte pairs for bf16) and no index is divided.
template <typename T>
__device__ __forceinline__ void load_pixel(const T* __restrict__ p, int C, bool pairs, float (&v)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = -INFINITY;
    if (pairs) {
#pragma unroll
        for (int c = 0; c < 8; c += 2)
            if (c < C) { const bf16x2 t = *(const bf16x2*)((const bf16*)p + c); v[c] = (float)t[0]; v[c + 1] = (float)t[1]; }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) v[c] = to_f(p[c]);
    }
}
// A thread keeps CE_UNR pixels in flight (all their loads are issued before the first is reduced): with one pixel per iteration the
// 20 bytes a lane has outstanding leave the kernel latency-bound at 2 TB/s.  Per-pixel arithmetic and the order of every sum are unchanged.
#define CE_UNR 4
template <typename T>
__global__ __launch_bounds__(256) void ce_prob_cl_kernel(const T* __restrict__ logits, int64_t pitch, const long long* __restrict__ labels,
        float* __restrict__ prob, float* __restrict__ nll, unsigned* __restrict__ vkey, OhemCtl* ctl, int64_t npix, int C, long long ignore, int pairs) {
    long long nv = 0;
    const int64_t S = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < npix; i0 += CE_UNR * S) {
      float vv[CE_UNR][8]; long long labv[CE_UNR];
#pragma unroll
      for (int u = 0; u < CE_UNR; ++u) {
          const int64_t ii = min(i0 + u * S, npix - 1);          // clamped: the loads stay unconditional
          load_pixel(logits + ii * pitch, C, pairs != 0, vv[u]); labv[u] = labels[ii];
      }
#pragma unroll
      for (int u = 0; u < CE_UNR; ++u) {
        const int64_t i = i0 + u * S;
        if (i >= npix) break;
        float (&v)[8] = vv[u];
        const long long lab = labv[u];
        float mx = v[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) mx = fmaxf(mx, v[c]);
        float sum = 0.f, zl = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) { sum += expf(v[c] - mx); if (c == lab) zl = v[c] - mx; }
        float pr = 0.f, nl = 0.f;
        if (lab >= 0 && lab < C) { nl = logf(sum) - zl; pr = expf(zl) / sum; }
        const bool valid = lab != ignore;
        prob[i] = pr; nll[i] = nl; vkey[i] = valid ? __float_as_uint(pr) : 0xFFFFFFFFu;
        nv += valid;
      }
    }
    __shared__ long long sh[256];
    sh[threadIdx.x] = nv; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && sh[0]) atomicAdd((unsigned long long*)&ctl->num_valid, (unsigned long long)sh[0]);
}
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_cl_kernel(const T* __restrict__ logits, int64_t pitch, const long long* __restrict__ labels,
        const float* __restrict__ prob, const OhemCtl* ctl, const float* __restrict__ gout, T* __restrict__ dlogits, int64_t opitch,
        int64_t npix, int C, long long ignore, int pairs) {
    const int sel = ctl->do_select; const float thr = ctl->thr;
    const float g = gout[0] / (float)ctl->kept;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const long long lab = labels[i];
        const bool keep = lab != ignore && (!sel || prob[i] < thr);
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = 0.f;
        if (keep) {
            float v[8]; load_pixel(logits + i * pitch, C, pairs != 0, v);
            float mx = v[0];
#pragma unroll
            for (int c = 1; c < 8; ++c) mx = fmaxf(mx, v[c]);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
            const float inv = 1.f / sum;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < C) o[c] = g * (v[c] * inv - (c == lab ? 1.f : 0.f));
        }
        T* d = dlogits + i * opitch;
        if (pairs) {
#pragma unroll
            for (int c = 0; c < 8; c += 2) if (c < C) { bf16x2 t = {(bf16)o[c], (bf16)o[c + 1]}; *(bf16x2*)((bf16*)d + c) = t; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < C) d[c] = from_f<T>(o[c]);
        }
    }
}
static inline bool ce_channels_last(int64_t sn, int64_t sc, int64_t sh, int64_t sw, int C, int H, int W) {
    return sc == 1 && C <= 8 && sw >= C && sh == (int64_t)W * sw && sn == (int64_t)H * sh;
}

__global__ void ohem_plan_kernel(OhemCtl* ctl, double min_kept) {
    const long long nv = ctl->num_valid;
    const long long k = (long long)(min_kept * (double)nv);          // int(self.min_kept * valid_prob.numel())
    ctl->k = k; ctl->krem = k; ctl->prefix = 0;
    ctl->do_select = (k < nv && nv > 0 && k > 0) ? 1 : 0;
}

// pass 0: bits 31..21, pass 1: bits 20..10, pass 2: bits 9..0
__device__ __forceinline__ void pass_bits(int pass, int& shift, unsigned& mask, unsigned& himask) {
    if (pass == 0) { shift = 21; mask = 0x7FF; himask = 0; }
    else if (pass == 1) { shift = 10; mask = 0x7FF; himask = 0xFFE00000u; }
    else { shift = 0; mask = 0x3FF; himask = 0xFFFFFC00u; }
}
__global__ __launch_bounds__(256) void ohem_hist_kernel(const unsigned* __restrict__ vkey, int64_t npix, const OhemCtl* ctl,
                                                        unsigned* __restrict__ hist, int pass) {
    if (!ctl->do_select) return;
    __shared__ unsigned sh[OHEM_BINS];
    for (int i = threadIdx.x; i < OHEM_BINS; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    int shift; unsigned mask, himask; pass_bits(pass, shift, mask, himask);
    const unsigned prefix = ctl->prefix;
    const int64_t S = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < npix; i0 += CE_UNR * S) {
        unsigned key[CE_UNR];
#pragma unroll
        for (int u = 0; u < CE_UNR; ++u) key[u] = i0 + u * S < npix ? vkey[i0 + u * S] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < CE_UNR; ++u)
            if (key[u] != 0xFFFFFFFFu && (key[u] & himask) == prefix) atomicAdd(&sh[(key[u] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < OHEM_BINS; i += blockDim.x) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}
// one block of 256 threads: thread t owns bins 8t .. 8t+7; block-wide exclusive scan of the per-thread totals, then the owner of the
// bin where the running count first reaches krem records it (the serial walk over 2048 bins took 60 us per pass)
__global__ __launch_bounds__(256) void ohem_scan_kernel(OhemCtl* ctl, const unsigned* __restrict__ hist, int pass) {
    if (!ctl->do_select) return;
    int shift; unsigned mask, himask; pass_bits(pass, shift, mask, himask);
    __shared__ long long tot[256];
    const int t = threadIdx.x;
    const long long rem = ctl->krem;
    const unsigned prefix = ctl->prefix;
    unsigned h[8]; long long mine = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const unsigned b = t * 8 + e; h[e] = b <= mask ? hist[b] : 0u; mine += h[e]; }
    tot[t] = mine;
    __syncthreads();
    long long before = 0;
    for (int u = 0; u < t; ++u) before += tot[u];
    if (before < rem && before + mine >= rem) {              // exactly one thread: the running count crosses krem inside its bins
        long long cum = before;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (cum + h[e] >= rem) { ctl->prefix = prefix | ((unsigned)(t * 8 + e) << shift); ctl->krem = rem - cum; break; }
            cum += h[e];
        }
    }
}
// per chunk of OHEM_CHUNK pixels: number of valid pixels and of valid pixels whose key equals the selected value
__global__ __launch_bounds__(256) void ohem_chunk_count_kernel(const unsigned* __restrict__ vkey, int64_t npix, const OhemCtl* ctl,
                                                               unsigned* __restrict__ cnt_valid, unsigned* __restrict__ cnt_eq) {
    if (!ctl->do_select) return;
    const unsigned target = ctl->prefix;
    const int64_t base = (int64_t)blockIdx.x * OHEM_CHUNK;
    unsigned nv = 0, ne = 0;
    for (int e = threadIdx.x; e < OHEM_CHUNK; e += 256) {
        const int64_t i = base + e;
        if (i < npix) { const unsigned key = vkey[i]; nv += key != 0xFFFFFFFFu; ne += key == target; }
    }
    __shared__ unsigned s1[256], s2[256];
    s1[threadIdx.x] = nv; s2[threadIdx.x] = ne; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { cnt_valid[blockIdx.x] = s1[0]; cnt_eq[blockIdx.x] = s2[0]; }
}
// one block: find the chunk holding the krem-th occurrence (in pixel order) of the selected value
__global__ __launch_bounds__(1024) void ohem_locate_chunk_kernel(OhemCtl* ctl, const unsigned* __restrict__ cnt_valid,
                                                                 const unsigned* __restrict__ cnt_eq, int nchunks) {
    if (!ctl->do_select) return;
    __shared__ long long sv[1024], se[1024];
    const int t = threadIdx.x, per = (nchunks + 1023) / 1024;
    long long v = 0, e = 0;
    for (int c = t * per; c < min(nchunks, (t + 1) * per); ++c) { v += cnt_valid[c]; e += cnt_eq[c]; }
    sv[t] = v; se[t] = e; __syncthreads();
    if (t == 0) {
        long long cv = 0, ce = 0; const long long rem = ctl->krem;
        int seg = 0;
        for (; seg < 1024; ++seg) { if (ce + se[seg] >= rem) break; cv += sv[seg]; ce += se[seg]; }
        for (int c = seg * per; c < min(nchunks, (seg + 1) * per); ++c) {
            if (ce + cnt_eq[c] >= rem) { ctl->chunk = c; ctl->valid_before = cv; ctl->eq_before = ce; return; }
            cv += cnt_valid[c]; ce += cnt_eq[c];
        }
        ctl->chunk = -1;
    }
}
// one block walks the located chunk (4096 keys, 16 consecutive keys per thread): finds the position of the needed occurrence of
// the selected value, applies the reference's index quirk and updates the stateful threshold
__global__ __launch_bounds__(256) void ohem_threshold_kernel(OhemCtl* ctl, const unsigned* __restrict__ vkey, const float* __restrict__ prob, int64_t npix,
                                                             float* __restrict__ thr_state) {
    __shared__ int seq[256], sval[256];
    const int t = threadIdx.x;
    const bool active = ctl->do_select && ctl->chunk >= 0;
    if (active) {
        const unsigned target = ctl->prefix;
        const long long need = ctl->krem - ctl->eq_before;          // 1-based occurrence inside the chunk
        const int64_t base = (int64_t)ctl->chunk * OHEM_CHUNK + t * 16;
        unsigned key[16]; int ne = 0, nv = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) { key[e] = base + e < npix ? vkey[base + e] : 0xFFFFFFFFu; ne += key[e] == target; nv += key[e] != 0xFFFFFFFFu; }
        seq[t] = ne; sval[t] = nv;
        __syncthreads();
        long long eq_before = 0, valid_before = 0;
        for (int u = 0; u < t; ++u) { eq_before += seq[u]; valid_before += sval[u]; }
        if (eq_before < need && eq_before + ne >= need) {          // the occurrence lies in this thread's 16 keys
            long long left = need - eq_before, j = ctl->valid_before + valid_before;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (key[e] == target && --left == 0) break;
                j += key[e] != 0xFFFFFFFFu;
            }
            ctl->j = j;
            const float cand = prob[j];           // sic: index among the valid pixels applied to the all-pixel vector
            if (cand > *thr_state) *thr_state = cand;
        }
    }
    __syncthreads();
    if (t == 0) { __threadfence(); ctl->thr = *thr_state; }
}

// masked mean of the NLL: kept = valid && (!do_select || prob < thr)
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ prob, const float* __restrict__ nll,
        const unsigned* __restrict__ vkey, const OhemCtl* ctl, int64_t npix, float* __restrict__ psum, unsigned* __restrict__ pcnt) {
    const int sel = ctl->do_select; const float thr = ctl->thr;
    float s = 0.f; unsigned n = 0;
    const int64_t per = cdiv(npix, gridDim.x), beg = blockIdx.x * per, end = min(npix, beg + per);
    // a valid pixel's key IS its probability's bit pattern (ce_prob_*_kernel): `prob` is not read again
    for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += CE_UNR * 256) {
        unsigned key[CE_UNR]; float nl[CE_UNR];
#pragma unroll
        for (int u = 0; u < CE_UNR; ++u) {
            const int64_t i = i0 + u * 256;
            key[u] = i < end ? vkey[i] : 0xFFFFFFFFu; nl[u] = i < end ? nll[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CE_UNR; ++u) {
            const bool keep = key[u] != 0xFFFFFFFFu && (!sel || __uint_as_float(key[u]) < thr);
            if (keep) { s += nl[u]; ++n; }
        }
    }
    __shared__ float ss[256]; __shared__ unsigned sn[256];
    ss[threadIdx.x] = s; sn[threadIdx.x] = n; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sn[threadIdx.x] += sn[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { psum[blockIdx.x] = ss[0]; pcnt[blockIdx.x] = sn[0]; }
}
__global__ void ce_finalize_kernel(OhemCtl* ctl, const float* __restrict__ psum, const unsigned* __restrict__ pcnt, int nb, float* __restrict__ loss) {
    // 256 threads, thread t adds partials t, t + 256, ...; fixed-order tree in LDS (was one thread walking all nb partials: 64 us)
    __shared__ double ss[256]; __shared__ long long sn[256];
    double s = 0.0; long long n = 0;
    for (int b = threadIdx.x; b < nb; b += 256) { s += psum[b]; n += pcnt[b]; }
    ss[threadIdx.x] = s; sn[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sn[threadIdx.x] += sn[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x != 0) return;
    s = ss[0]; n = sn[0];
    ctl->kept = n; ctl->loss_sum = (float)s;
    loss[0] = (float)(s / (double)n);            // mean over an empty selection is NaN, as torch
}

// dlogits = gout * (softmax - onehot) * keep / kept
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, Strides s, const long long* __restrict__ labels,
        const float* __restrict__ prob, const OhemCtl* ctl, const float* __restrict__ gout, T* __restrict__ dlogits, Strides so,
        int64_t npix, int H, int W, int C, long long ignore) {
    const int sel = ctl->do_select; const float thr = ctl->thr;
    const float g = gout[0] / (float)ctl->kept;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const T* p = logits + pix_off(i, H, W, s);
        T* d = dlogits + pix_off(i, H, W, so);
        const long long lab = labels[i];
        const bool keep = lab != ignore && (!sel || prob[i] < thr);
        if (!keep) { for (int c = 0; c < C; ++c) d[c * so.c] = from_f<T>(0.f); continue; }
        float mx, sum; softmax_stats(p, s.c, C, mx, sum);
        const float inv = 1.f / sum;
        for (int c = 0; c < C; ++c) d[c * so.c] = from_f<T>(g * (expf(to_f(p[c * s.c]) - mx) * inv - (c == lab ? 1.f : 0.f)));
    }
}

// ------------------------------------------------------------------ Dice / Tversky sums:  sums[b][c] = {sum p*[y=c], sum p, #[y=c]}
template <typename T>
__global__ __launch_bounds__(256) void dice_sums_kernel(const T* __restrict__ probs, Strides s, const long long* __restrict__ labels,
                                                        float* __restrict__ partial, int H, int W, int C) {
    extern __shared__ float sh[];                 // [C][3] per block
    const int b = blockIdx.y; const int64_t hw = (int64_t)H * W;
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sh[i] = 0.f;
    __syncthreads();
    const int64_t per = cdiv(hw, gridDim.x), beg = blockIdx.x * per, end = min(hw, beg + per);
    for (int c = 0; c < C; ++c) {
        float tp = 0.f, sp = 0.f, cnt = 0.f;
        for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
            const float p = to_f(probs[b * s.n + (i / W) * s.h + (i % W) * s.w + c * s.c]);
            const bool hit = labels[b * hw + i] == c;
            sp += p; if (hit) { tp += p; cnt += 1.f; }
        }
        tp = wave_sum(tp); sp = wave_sum(sp); cnt = wave_sum(cnt);
        __shared__ float w3[4][3];
        if ((threadIdx.x & 63) == 0) { w3[threadIdx.x >> 6][0] = tp; w3[threadIdx.x >> 6][1] = sp; w3[threadIdx.x >> 6][2] = cnt; }
        __syncthreads();
        if (threadIdx.x < 3) sh[c * 3 + threadIdx.x] = w3[0][threadIdx.x] + w3[1][threadIdx.x] + w3[2][threadIdx.x] + w3[3][threadIdx.x];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) partial[((int64_t)b * gridDim.x + blockIdx.x) * 3 * C + i] = sh[i];
}
__global__ void dice_sums_final_kernel(const float* __restrict__ partial, int nb, int n3c, float* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n3c) return;
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += partial[((int64_t)b * nb + k) * n3c + i];
    sums[(int64_t)b * n3c + i] = s;
}
// dprobs[b,c,pix] = gout * (coef_hit[b][c] * [y=c] + coef_all[b][c])
template <typename T>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const long long* __restrict__ labels, const float* __restrict__ coef_hit,
        const float* __restrict__ coef_all, const float* __restrict__ gout, T* __restrict__ dprobs, Strides so, int64_t npix, int H, int W, int C) {
    const float g = gout[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ((int64_t)H * W);
        T* d = dprobs + pix_off(i, H, W, so);
        const long long lab = labels[i];
        for (int c = 0; c < C; ++c) d[c * so.c] = from_f<T>(g * (coef_all[b * C + c] + (c == lab ? coef_hit[b * C + c] : 0.f)));
    }
}

// ------------------------------------------------------------------ confusion matrix: cm[target*C + argmax] += 1  (int64, exact)
template <typename T>
__global__ __launch_bounds__(256) void confusion_kernel(const T* __restrict__ logits, Strides s, const long long* __restrict__ labels,
                                                        unsigned long long* __restrict__ cm, int64_t npix, int H, int W, int C, int copies) {
    extern __shared__ unsigned bins[];           // `copies` private C*C matrices per block: lane l counts in copy l % copies, so the
    const int cc = C * C;                        // lanes of a wave that meet the same (target, class) pair do not queue on one LDS word
    for (int i = threadIdx.x; i < cc * copies; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    unsigned* mine = bins + (threadIdx.x % copies) * cc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const T* p = logits + pix_off(i, H, W, s);
        float best = to_f(p[0]); int arg = 0;
        for (int c = 1; c < C; ++c) { const float v = to_f(p[c * s.c]); if (v > best) { best = v; arg = c; } }   // first max wins
        const long long lab = labels[i];
        if (lab >= 0 && lab < C) atomicAdd(&mine[(int)lab * C + arg], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cc; i += blockDim.x) {
        unsigned long long t = 0;
        for (int k = 0; k < copies; ++k) t += bins[k * cc + i];
        if (t) atomicAdd(&cm[i], t);
    }
}

// the same accumulator for predictions that already are class indices (torchmetrics takes both forms, helper/metrics.py:90-95):
// pixels whose target or prediction lies outside [0, C) are not counted
__global__ __launch_bounds__(256) void confusion_indices_kernel(const long long* __restrict__ pred, const long long* __restrict__ labels,
                                                                unsigned long long* __restrict__ cm, int64_t npix, int C, int copies) {
    extern __shared__ unsigned bins[];
    const int cc = C * C;
    for (int i = threadIdx.x; i < cc * copies; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    unsigned* mine = bins + (threadIdx.x % copies) * cc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const long long lab = labels[i], arg = pred[i];
        if (lab >= 0 && lab < C && arg >= 0 && arg < C) atomicAdd(&mine[(int)lab * C + (int)arg], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cc; i += blockDim.x) {
        unsigned long long t = 0;
        for (int k = 0; k < copies; ++k) t += bins[k * cc + i];
        if (t) atomicAdd(&cm[i], t);
    }
}

// ------------------------------------------------------------------ label histogram: hist[v] += #(labels == v), v in [0, nbins)
// helper/weights.py:18-27 builds a [pixels x classes] boolean matrix to count the labels; one pass over the int64 labels with
// per-workgroup LDS bins (lane l counts in copy l % copies) and one 64-bit global add per non-empty bin.  Exact.
__global__ __launch_bounds__(256) void label_hist_kernel(const long long* __restrict__ labels, unsigned long long* __restrict__ hist,
                                                         int64_t n, int nbins, int copies) {
    extern __shared__ unsigned bins[];
    for (int i = threadIdx.x; i < nbins * copies; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    unsigned* mine = bins + (threadIdx.x % copies) * nbins;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = labels[i];
        if (v >= 0 && v < nbins) atomicAdd(&mine[(int)v], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
        unsigned long long t = 0;
        for (int k = 0; k < copies; ++k) t += bins[k * nbins + i];
        if (t) atomicAdd(&hist[i], t);
    }
}

// ------------------------------------------------------------------ Softmax2d (torch.nn.Softmax2d of modules/layers.py:41; the
// probabilities DiceLoss takes, loss/seg_loss.py:82-99) for logits that have no fused producer: BASELINE configs[4] trains
// MiT + Lawin with OHEM-CE on the logits AND Dice on their softmax.  A thread owns one pixel; exponentials in fp32.
//   forward : y[c] = exp(x[c] - max) / sum        backward: dx[c] = y[c] * (dy[c] - sum_k dy[k] y[k])
template <typename T>
__global__ __launch_bounds__(256) void softmax2d_fwd_kernel(const T* __restrict__ x, Strides sx, T* __restrict__ y, Strides sy, int64_t npix,
                                                            int H, int W, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const T* p = x + pix_off(i, H, W, sx);
        T* q = y + pix_off(i, H, W, sy);
        float m = to_f(p[0]);
        for (int c = 1; c < C; ++c) m = fmaxf(m, to_f(p[c * sx.c]));
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(to_f(p[c * sx.c]) - m);
        const float inv = 1.f / sum;
        for (int c = 0; c < C; ++c) q[c * sy.c] = from_f<T>(expf(to_f(p[c * sx.c]) - m) * inv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax2d_bwd_kernel(const T* __restrict__ y, Strides sy, const T* __restrict__ dy, Strides sd,
                                                            T* __restrict__ dx, Strides sx, int64_t npix, int H, int W, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const T* p = y + pix_off(i, H, W, sy);
        const T* g = dy + pix_off(i, H, W, sd);
        T* q = dx + pix_off(i, H, W, sx);
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(to_f(g[c * sd.c]), to_f(p[c * sy.c]), dot);
        for (int c = 0; c < C; ++c) q[c * sx.c] = from_f<T>(to_f(p[c * sy.c]) * (to_f(g[c * sd.c]) - dot));
    }
}

// ------------------------------------------------------------------ prediction writer / evaluation script
// helper/callbacks.py:52-54: prediction.softmax(dim=1).argmax(dim=1) -> class map.  softmax is increasing in the logit, so the
// arg-max is taken on the logits themselves; the first maximum wins (torch.argmax); a pixel that holds a NaN gets class 0
// (softmax turns the whole pixel into NaN and torch.argmax returns the first of them).  One byte per pixel instead of the
// reference's int64.  A thread owns 4 consecutive pixels (one 4-byte store).
// A thread owns ONE pixel (lanes walk the image in storage order, so channels-last and NCHW logits both load coalesced; VEC:
// the channels of a pixel are contiguous and 16-byte aligned, read as whole 16-byte chunks); the classes of 4 neighbouring
// lanes are packed into one 4-byte store.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void class_map_kernel(const T* __restrict__ logits, Strides s, unsigned char* __restrict__ out, int64_t npix,
                                                        int H, int W, int C) {
    constexpr int PER = 16 / (int)sizeof(T);
    typedef T VT __attribute__((ext_vector_type(PER)));
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < npix; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const T* p = logits + pix_off(i < npix ? i : npix - 1, H, W, s);
        float best = 0.f; int arg = 0; bool nan = false;
        if (VEC) {
            for (int c0 = 0; c0 < C; c0 += PER) {
                const VT v8 = *(const VT*)(p + c0);
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const float v = to_f(v8[e]);
                    if (c0 + e < C) { nan |= v != v; if (c0 + e == 0 || v > best) { best = v; arg = c0 + e; } }
                }
            }
        } else {
            for (int c = 0; c < C; ++c) { const float v = to_f(p[c * s.c]); nan |= v != v; if (c == 0 || v > best) { best = v; arg = c; } }
        }
        const unsigned mine = nan ? 0u : (unsigned)arg;
        const unsigned packed = mine | (__shfl_down(mine, 1) << 8) | (__shfl_down(mine, 2) << 16) | (__shfl_down(mine, 3) << 24);
        if (base + blockDim.x <= npix) { if ((threadIdx.x & 3) == 0) *(unsigned*)(out + i) = packed; }
        else if (i < npix) out[i] = (unsigned char)mine;
    }
}

// scripts/evaluate.py:130-135: sklearn confusion_matrix(y_true, y_pred, labels) of two class-map rasters, restricted to the
// pixels `valid` marks (null = all).  lut[256]: raster value -> row / column index, or -1 for a value that is not among
// `labels` (sklearn drops a pixel whose truth or prediction is not listed).  cm: int64 [C][C], accumulated in place.
__global__ __launch_bounds__(256) void confusion_labels_kernel(const unsigned char* __restrict__ truth, const unsigned char* __restrict__ pred,
        const unsigned char* __restrict__ valid, const int* __restrict__ lut, unsigned long long* __restrict__ cm, int64_t n, int C, int copies) {
    extern __shared__ unsigned bins[];           // `copies` private matrices, lane l counts in copy l % copies (see confusion_kernel)
    __shared__ int slut[256];
    const int cc = C * C;
    slut[threadIdx.x] = lut[threadIdx.x];
    for (int i = threadIdx.x; i < cc * copies; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    unsigned* mine = bins + (threadIdx.x % copies) * cc;
    const bool wide = ((((uintptr_t)truth | (uintptr_t)pred | (uintptr_t)valid) & 3) == 0);
    const int64_t nq = wide ? n / 4 : 0;                              // 4 pixels per thread through 4-byte loads, then the tail one by one
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const unsigned t4 = ((const unsigned*)truth)[q], p4 = ((const unsigned*)pred)[q], v4 = valid ? ((const unsigned*)valid)[q] : 0x01010101u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = slut[(t4 >> (8 * j)) & 255], p = slut[(p4 >> (8 * j)) & 255];
            if (((v4 >> (8 * j)) & 255) && t >= 0 && p >= 0) atomicAdd(&mine[t * C + p], 1u);
        }
    }
    for (int64_t i = nq * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (valid && !valid[i]) continue;
        const int t = slut[truth[i]], p = slut[pred[i]];
        if (t >= 0 && p >= 0) atomicAdd(&mine[t * C + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cc; i += blockDim.x) {
        unsigned long long t = 0;
        for (int k = 0; k < copies; ++k) t += bins[k * cc + i];
        if (t) atomicAdd(&cm[i], t);
    }
}

// ------------------------------------------------------------------ host entry points
static inline int nblk(int64_t n, int cap) { int64_t b = cdiv(n, 256); return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }
#define CE_REDUCE_BLOCKS 1024
// private copies of the C x C count matrix per workgroup: up to 16, within 32 KB of LDS
static inline int cm_copies(int C) { int c = 8192 / (C * C); return c < 1 ? 1 : (c > 16 ? 16 : c); }

// workspace layout helper: all regions 256-byte aligned
extern "C" int64_t ss_ohem_ce_workspace(int64_t npix) {
    int64_t nchunks = cdiv(npix, OHEM_CHUNK);
    int64_t b = 256 /*ctl*/ + 3 * OHEM_BINS * 4 + 2 * ((nchunks * 4 + 255) / 256 * 256) + 2 * CE_REDUCE_BLOCKS * 4 + 1024;
    b += 2 * ((npix * 4 + 255) / 256 * 256);     // nll, vkey
    return b;
}

// prob: [npix] fp32 (kept by the caller for the backward), ctl_out: 256-byte device record kept for the backward,
// thr_state: device float holding the module's stateful threshold, loss: device float.
extern "C" int ss_ohem_ce_fwd(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                              int B, int C, int H, int W, int64_t ignore_index, double min_kept, float* thr_state, float* prob,
                              void* ctl_out, float* loss, void* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(logits && labels && thr_state && prob && ctl_out && loss && workspace, "ss_ohem_ce_fwd: null pointer");
    SS_CHECK(B > 0 && C > 0 && C <= 1024 && H > 0 && W > 0 && min_kept >= 0.0 && min_kept <= 1.0, "ss_ohem_ce_fwd: bad arguments");
    const int64_t npix = (int64_t)B * H * W;
    SS_CHECK(ws_bytes >= ss_ohem_ce_workspace(npix), "ss_ohem_ce_fwd: workspace too small");
    SS_CHECK(((uintptr_t)workspace & 255) == 0, "ss_ohem_ce_fwd: workspace must be 256-byte aligned");
    const int64_t nchunks = cdiv(npix, OHEM_CHUNK);
    char* w = (char*)workspace;
    OhemCtl* ctl = (OhemCtl*)w; w += 256;
    unsigned* hist = (unsigned*)w; w += 3 * OHEM_BINS * 4;
    unsigned* cntv = (unsigned*)w; w += (nchunks * 4 + 255) / 256 * 256;
    unsigned* cnte = (unsigned*)w; w += (nchunks * 4 + 255) / 256 * 256;
    float* psum = (float*)w; w += CE_REDUCE_BLOCKS * 4;
    unsigned* pcnt = (unsigned*)w; w += CE_REDUCE_BLOCKS * 4;
    w = (char*)workspace + (((w - (char*)workspace) + 255) / 256 * 256);
    float* nll = (float*)w; w += (npix * 4 + 255) / 256 * 256;
    unsigned* vkey = (unsigned*)w;
    Strides s{sn, sc, sh, sw};
    SS_HIP(hipMemsetAsync(workspace, 0, 256 + 3 * OHEM_BINS * 4, stream));
    const int gb = nblk(npix, 4096);
    if (ce_channels_last(sn, sc, sh, sw, C, H, W)) {
        const int pairs = dtype == SS_BF16 && C % 2 == 0 && sw % 2 == 0 && ((uintptr_t)logits & 3) == 0;
        SS_DISPATCH(dtype, hipLaunchKernelGGL((ce_prob_cl_kernel<T>), dim3(nblk(npix, 16384)), dim3(256), 0, stream, (const T*)logits, sw,
                                              (const long long*)labels, prob, nll, vkey, ctl, npix, C, (long long)ignore_index, pairs));
    } else {
        SS_DISPATCH(dtype, hipLaunchKernelGGL((ce_prob_kernel<T>), dim3(gb), dim3(256), 0, stream, (const T*)logits, s, (const long long*)labels, prob, nll,
                                              vkey, ctl, npix, H, W, C, (long long)ignore_index));
    }
    hipLaunchKernelGGL(ohem_plan_kernel, dim3(1), dim3(1), 0, stream, ctl, min_kept);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(ohem_hist_kernel, dim3(gb < 1024 ? gb : 1024), dim3(256), 0, stream, vkey, npix, ctl, hist + pass * OHEM_BINS, pass);
        hipLaunchKernelGGL(ohem_scan_kernel, dim3(1), dim3(256), 0, stream, ctl, hist + pass * OHEM_BINS, pass);
    }
    hipLaunchKernelGGL(ohem_chunk_count_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, vkey, npix, ctl, cntv, cnte);
    hipLaunchKernelGGL(ohem_locate_chunk_kernel, dim3(1), dim3(1024), 0, stream, ctl, cntv, cnte, (int)nchunks);
    hipLaunchKernelGGL(ohem_threshold_kernel, dim3(1), dim3(256), 0, stream, ctl, vkey, prob, npix, thr_state);
    const int rb = (int)(cdiv(npix, 1024) < CE_REDUCE_BLOCKS ? cdiv(npix, 1024) : CE_REDUCE_BLOCKS);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(rb), dim3(256), 0, stream, prob, nll, vkey, ctl, npix, psum, pcnt);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, stream, ctl, psum, pcnt, rb, loss);
    SS_LAUNCH_CHECK();
    SS_HIP(hipMemcpyAsync(ctl_out, ctl, 256, hipMemcpyDeviceToDevice, stream));
    return 0;
}

extern "C" int ss_ohem_ce_bwd(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                              const float* prob, const void* ctl, const float* gout, void* dlogits, int64_t on, int64_t oc, int64_t oh,
                              int64_t ow, int B, int C, int H, int W, int64_t ignore_index, hipStream_t stream) {
    SS_CHECK(logits && labels && prob && ctl && gout && dlogits && B > 0 && C > 0 && H > 0 && W > 0, "ss_ohem_ce_bwd: bad arguments");
    const int64_t npix = (int64_t)B * H * W;
    Strides s{sn, sc, sh, sw}, so{on, oc, oh, ow};
    if (ce_channels_last(sn, sc, sh, sw, C, H, W) && ce_channels_last(on, oc, oh, ow, C, H, W)) {
        const int pairs = dtype == SS_BF16 && C % 2 == 0 && sw % 2 == 0 && ow % 2 == 0 && ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 3) == 0;
        SS_DISPATCH(dtype, hipLaunchKernelGGL((ce_bwd_cl_kernel<T>), dim3(nblk(npix, 16384)), dim3(256), 0, stream, (const T*)logits, sw,
                                              (const long long*)labels, prob, (const OhemCtl*)ctl, gout, (T*)dlogits, ow, npix, C,
                                              (long long)ignore_index, pairs));
    } else {
        SS_DISPATCH(dtype, hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(nblk(npix, 8192)), dim3(256), 0, stream, (const T*)logits, s, (const long long*)labels,
                                              prob, (const OhemCtl*)ctl, gout, (T*)dlogits, so, npix, H, W, C, (long long)ignore_index));
    }
    SS_LAUNCH_CHECK();
    return 0;
}

// sums: [B][C][3] fp32.  workspace >= B*64*3*C floats.
extern "C" int ss_dice_sums(int dtype, const void* probs, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels, int B,
                            int C, int H, int W, float* sums, float* workspace, int64_t ws_bytes, hipStream_t stream) {
    SS_CHECK(probs && labels && sums && workspace && B > 0 && C > 0 && C <= 1024 && H > 0 && W > 0, "ss_dice_sums: bad arguments");
    const int nb = (int)(cdiv((int64_t)H * W, 4096) < 64 ? cdiv((int64_t)H * W, 4096) : 64);
    SS_CHECK(ws_bytes >= (int64_t)B * nb * 3 * C * 4, "ss_dice_sums: workspace too small");
    Strides s{sn, sc, sh, sw};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dice_sums_kernel<T>), dim3(nb, B), dim3(256), 3 * C * sizeof(float), stream, (const T*)probs, s,
                                          (const long long*)labels, workspace, H, W, C));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_sums_final_kernel, dim3((unsigned)cdiv(3 * C, 256), B), dim3(256), 0, stream, workspace, nb, 3 * C, sums);
    SS_LAUNCH_CHECK();
    return 0;
}
// ---- the same sums for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc): the
// per-workgroup partial sums go to a file-scope device array instead of a caller workspace.  Each GPU the library runs on holds
// its own copy of the array at its own address until the process ends; the address is looked up once per device
// (hipGetSymbolAddress with that device current) and only that pointer is ever handed to kernels.  Every partial the finalize
// kernel reads is written by the launch before it, so nothing needs clearing between calls; calls that overlap on one GPU
// (two streams) race on the array.
#define SS_DICE_SCRATCH_FLOATS (256 * 1024)          // 1 MiB per GPU
static __device__ __align__(256) float scratch[SS_DICE_SCRATCH_FLOATS];
#define SS_MAX_DEVICES 64
static float* g_scratch_ptr[SS_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int dice_scratch_for_current_device(float** out) {
    int dev = -1;
    SS_HIP(hipGetDevice(&dev));
    SS_CHECK(dev >= 0 && dev < SS_MAX_DEVICES, "ss_dice_sums_static: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        SS_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        SS_CHECK(p != nullptr, "ss_dice_sums_static: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int64_t ss_dice_sums_static_max_floats(void) { return SS_DICE_SCRATCH_FLOATS; }

extern "C" int ss_dice_sums_static(int dtype, const void* probs, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                                   int B, int C, int H, int W, float* sums, hipStream_t stream) {
    SS_CHECK(probs && labels && sums && B > 0 && C > 0 && C <= 1024 && H > 0 && W > 0, "ss_dice_sums_static: bad arguments");
    const int nb = (int)(cdiv((int64_t)H * W, 4096) < 64 ? cdiv((int64_t)H * W, 4096) : 64);
    const int64_t need = (int64_t)B * nb * 3 * C;
    SS_CHECK(need <= SS_DICE_SCRATCH_FLOATS, "ss_dice_sums_static: B * blocks * 3 * C = %lld partial sums exceed the static scratch of %d "
             "floats (B %d, C %d, %d blocks per image); use ss_dice_sums with a workspace", (long long)need, SS_DICE_SCRATCH_FLOATS, B, C, nb);
    float* part = nullptr;
    if (int rc = dice_scratch_for_current_device(&part)) return rc;
    Strides s{sn, sc, sh, sw};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dice_sums_kernel<T>), dim3(nb, B), dim3(256), 3 * C * sizeof(float), stream, (const T*)probs, s,
                                          (const long long*)labels, part, H, W, C));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_sums_final_kernel, dim3((unsigned)cdiv(3 * C, 256), B), dim3(256), 0, stream, part, nb, 3 * C, sums);
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_dice_bwd(int dtype, const int64_t* labels, const float* coef_hit, const float* coef_all, const float* gout, void* dprobs,
                           int64_t on, int64_t oc, int64_t oh, int64_t ow, int B, int C, int H, int W, hipStream_t stream) {
    SS_CHECK(labels && coef_hit && coef_all && gout && dprobs && B > 0 && C > 0 && H > 0 && W > 0, "ss_dice_bwd: bad arguments");
    const int64_t npix = (int64_t)B * H * W;
    Strides so{on, oc, oh, ow};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((dice_bwd_kernel<T>), dim3(nblk(npix, 8192)), dim3(256), 0, stream, (const long long*)labels, coef_hit,
                                          coef_all, gout, (T*)dprobs, so, npix, H, W, C));
    SS_LAUNCH_CHECK();
    return 0;
}

// cm: int64 [C][C] (rows = target, cols = prediction), accumulated in place
extern "C" int ss_confusion_update(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                                   int64_t* cm, int B, int C, int H, int W, hipStream_t stream) {
    SS_CHECK(logits && labels && cm && B > 0 && C > 0 && C <= 100 && H > 0 && W > 0, "ss_confusion_update: bad arguments (C <= 100)");
    const int64_t npix = (int64_t)B * H * W;
    Strides s{sn, sc, sh, sw};
    const int copies = cm_copies(C);
    SS_DISPATCH(dtype, hipLaunchKernelGGL((confusion_kernel<T>), dim3(nblk(npix, 2048)), dim3(256), (size_t)C * C * copies * sizeof(unsigned), stream,
                                          (const T*)logits, s, (const long long*)labels, (unsigned long long*)cm, npix, H, W, C, copies));
    SS_LAUNCH_CHECK();
    return 0;
}

// pred, labels: int64 [n] class indices; cm as above
extern "C" int ss_confusion_indices(const int64_t* pred, const int64_t* labels, int64_t* cm, int64_t n, int C, hipStream_t stream) {
    SS_CHECK(pred && labels && cm && n > 0 && C > 0 && C <= 100, "ss_confusion_indices: bad arguments (1 <= C <= 100)");
    const int copies = cm_copies(C);
    hipLaunchKernelGGL(confusion_indices_kernel, dim3(nblk(n, 2048)), dim3(256), (size_t)C * C * copies * sizeof(unsigned), stream,
                       (const long long*)pred, (const long long*)labels, (unsigned long long*)cm, n, C, copies);
    SS_LAUNCH_CHECK();
    return 0;
}

// labels: int64 [n]; hist: int64 [nbins], accumulated in place (values outside [0, nbins) are not counted)
extern "C" int ss_label_hist(const int64_t* labels, int64_t n, int nbins, int64_t* hist, hipStream_t stream) {
    SS_CHECK(labels && hist && n > 0 && nbins > 0 && nbins <= 4096, "ss_label_hist: bad arguments (1 <= nbins <= 4096)");
    int copies = 8192 / nbins; copies = copies < 1 ? 1 : (copies > 16 ? 16 : copies);
    hipLaunchKernelGGL(label_hist_kernel, dim3(nblk(n, 2048)), dim3(256), (size_t)nbins * copies * sizeof(unsigned), stream,
                       (const long long*)labels, (unsigned long long*)hist, n, nbins, copies);
    SS_LAUNCH_CHECK();
    return 0;
}

// out: uint8 [B][H][W] (dense, 4-byte aligned)
extern "C" int ss_class_map_u8(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, uint8_t* out, int B, int C,
                               int H, int W, hipStream_t stream) {
    SS_CHECK(logits && out && B > 0 && C > 0 && C <= 256 && H > 0 && W > 0, "ss_class_map_u8: bad arguments (1 <= C <= 256)");
    SS_CHECK(((uintptr_t)out & 3) == 0, "ss_class_map_u8: out must be 4-byte aligned");
    const int64_t npix = (int64_t)B * H * W;
    Strides s{sn, sc, sh, sw};
    const int64_t esz = dtype == SS_BF16 ? 2 : 4, per = 16 / esz;
    const bool vec = sc == 1 && aligned16(logits) && ((sn | sh | sw) * esz) % 16 == 0 && sw >= (C + per - 1) / per * per;
    dim3 grid(nblk(npix, 16384));
    if (vec) SS_DISPATCH(dtype, hipLaunchKernelGGL((class_map_kernel<T, true>), grid, dim3(256), 0, stream, (const T*)logits, s, out, npix, H, W, C));
    else SS_DISPATCH(dtype, hipLaunchKernelGGL((class_map_kernel<T, false>), grid, dim3(256), 0, stream, (const T*)logits, s, out, npix, H, W, C));
    SS_LAUNCH_CHECK();
    return 0;
}

// truth, pred, valid (or null): uint8 [n]; lut: int32 [256] on the device; cm: int64 [C][C], accumulated in place
extern "C" int ss_confusion_labels_u8(const uint8_t* truth, const uint8_t* pred, const uint8_t* valid, const int* lut, int64_t* cm, int64_t n,
                                      int C, hipStream_t stream) {
    SS_CHECK(truth && pred && lut && cm && n > 0 && C > 0 && C <= 100, "ss_confusion_labels_u8: bad arguments (1 <= C <= 100)");
    const int copies = cm_copies(C);
    hipLaunchKernelGGL(confusion_labels_kernel, dim3(nblk(cdiv(n, 4), 512)), dim3(256), (size_t)C * C * copies * sizeof(unsigned), stream, truth, pred,
                       valid, lut, (unsigned long long*)cm, n, C, copies);
    SS_LAUNCH_CHECK();
    return 0;
}

// x, y (and dy, dx): [B][C][H][W] with element strides, any dense layout
extern "C" int ss_softmax2d_fwd(int dtype, const void* x, int64_t xn, int64_t xc, int64_t xh, int64_t xw, void* y, int64_t yn, int64_t yc,
                                int64_t yh, int64_t yw, int B, int C, int H, int W, hipStream_t stream) {
    SS_CHECK(x && y && B > 0 && C > 0 && H > 0 && W > 0, "ss_softmax2d_fwd: bad arguments");
    const int64_t npix = (int64_t)B * H * W;
    Strides sx{xn, xc, xh, xw}, sy{yn, yc, yh, yw};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((softmax2d_fwd_kernel<T>), dim3(nblk(npix, 16384)), dim3(256), 0, stream, (const T*)x, sx, (T*)y, sy, npix,
                                          H, W, C));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_softmax2d_bwd(int dtype, const void* y, int64_t yn, int64_t yc, int64_t yh, int64_t yw, const void* dy, int64_t gn, int64_t gc,
                                int64_t gh, int64_t gw, void* dx, int64_t xn, int64_t xc, int64_t xh, int64_t xw, int B, int C, int H, int W,
                                hipStream_t stream) {
    SS_CHECK(y && dy && dx && B > 0 && C > 0 && H > 0 && W > 0, "ss_softmax2d_bwd: bad arguments");
    const int64_t npix = (int64_t)B * H * W;
    Strides sy{yn, yc, yh, yw}, sd{gn, gc, gh, gw}, sx{xn, xc, xh, xw};
    SS_DISPATCH(dtype, hipLaunchKernelGGL((softmax2d_bwd_kernel<T>), dim3(nblk(npix, 16384)), dim3(256), 0, stream, (const T*)y, sy, (const T*)dy, sd,
                                          (T*)dx, sx, npix, H, W, C));
    SS_LAUNCH_CHECK();
    return 0;
}
