// Flat-buffer optimizer step: the whole model's fp32 parameters, gradients and moment buffers are
// four contiguous arrays, so one launch updates every tensor (no per-parameter launches, 16-byte coalesced streams).
#include "common.h"

// AdaBound (Luo et al., ICLR 2019; the reference trains with torch_optimizer.AdaBound(lr=1e-3, final_lr=0.1, amsbound=False),
// train.py:53-58).  Per element, as the published algorithm:
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   step = clamp(step_size / (sqrt(v) + eps), lower, upper) * m;  p -= step
// with step_size = lr * sqrt(1-b2^t) / (1-b1^t), lower = flr (1 - 1/(gamma t + 1)), upper = flr (1 + 1/(gamma t)) computed on the host.
__global__ __launch_bounds__(256) void adabound_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, int64_t n4, int64_t n, float b1, float b2, float omb1, float omb2, float eps, float step_size, float lower, float upper,
        float wd, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        floatx4 pv = ((floatx4*)p)[i], gv = ((const floatx4*)g)[i], mv = ((floatx4*)m)[i], vv = ((floatx4*)v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = gv[e] * gscale + wd * pv[e];
            mv[e] = b1 * mv[e] + omb1 * gr;
            vv[e] = b2 * vv[e] + omb2 * gr * gr;
            float s = step_size / (sqrtf(vv[e]) + eps);
            s = fminf(fmaxf(s, lower), upper);
            pv[e] -= s * mv[e];
        }
        ((floatx4*)p)[i] = pv; ((floatx4*)m)[i] = mv; ((floatx4*)v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gscale + wd * p[i];
        float mm = b1 * m[i] + omb1 * gr, vv = b2 * v[i] + omb2 * gr * gr;
        float s = fminf(fmaxf(step_size / (sqrtf(vv) + eps), lower), upper);
        p[i] -= s * mm; m[i] = mm; v[i] = vv;
    }
}

extern "C" int ss_adabound_step(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, float eps,
                                float step_size, float lower, float upper, float weight_decay, float grad_scale, hipStream_t stream) {
    SS_CHECK(n >= 0, "ss_adabound_step: n < 0");
    if (n == 0) return 0;
    SS_CHECK(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "ss_adabound_step: buffers must be 16-byte aligned");
    int64_t n4 = n / 4;
    int blocks = (int)(cdiv(n4 > 0 ? n4 : 1, 256) < 4096 ? cdiv(n4 > 0 ? n4 : 1, 256) : 4096);
    hipLaunchKernelGGL(adabound_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n4, n, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2) /* rounded once from the double, as torch rounds alpha = 1 - beta */, eps, step_size, lower,
                       upper, weight_decay, grad_scale);
    SS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ gradient delivery into the flat buffer
// dst_i[0..n_i) += src_i for up to SS_MULTI_MAX tensors in ONE launch: a backward node hands all of its parameter gradients to
// their slots of the flat gradient buffer at once (autograd's AccumulateGrad would launch one add per parameter: ~550
// four-microsecond launches per step on MiT-B2 + Lawin).  A source is `rows` rows of `cols` contiguous floats `ld` apart
// (rows = 1 for a contiguous tensor; the Lawin position-mixing gradients are [64][65] slabs whose 65th column is the bias
// gradient); the destination is contiguous.  The table travels in the kernel arguments; a block finds its tensor by scanning
// the (wave-uniform) block prefix.
#define SS_MULTI_MAX 40
struct MultiAddTable {
    const float* src[SS_MULTI_MAX];
    float* dst[SS_MULTI_MAX];
    int64_t n[SS_MULTI_MAX];
    int64_t ld[SS_MULTI_MAX];
    int cols[SS_MULTI_MAX];
    int first_block[SS_MULTI_MAX + 1];
    int count;
};
#define SS_MULTI_PER_BLOCK (256 * 4 * 4)           // 4 vectors of 4 floats per thread

__global__ __launch_bounds__(256) void multi_add_kernel(MultiAddTable t) {
    int e = 0;
    while (e + 1 < t.count && (int)blockIdx.x >= t.first_block[e + 1]) ++e;
    const float* __restrict__ s = t.src[e]; float* __restrict__ d = t.dst[e];
    const int64_t n = t.n[e], base = (int64_t)(blockIdx.x - t.first_block[e]) * SS_MULTI_PER_BLOCK;
    const int cols = t.cols[e];
    const bool flat = cols == n;
    const bool vec = flat && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    if (vec && base + SS_MULTI_PER_BLOCK <= n) {
        floatx4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int64_t o = base + (int64_t)(i * 256 + threadIdx.x) * 4; a[i] = *(const floatx4*)(s + o); b[i] = *(const floatx4*)(d + o); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int64_t o = base + (int64_t)(i * 256 + threadIdx.x) * 4; *(floatx4*)(d + o) = a[i] + b[i]; }
    } else {
        const int64_t end = base + SS_MULTI_PER_BLOCK < n ? base + SS_MULTI_PER_BLOCK : n;
        if (flat) { for (int64_t o = base + threadIdx.x; o < end; o += 256) d[o] += s[o]; }
        else {
            const int64_t ld = t.ld[e];
            for (int64_t o = base + threadIdx.x; o < end; o += 256) { const int64_t r = o / cols; d[o] += s[r * ld + (o - r * cols)]; }
        }
    }
}

extern "C" int ss_multi_add_f32(int count, const float* const* src, const int64_t* src_ld, const int64_t* cols, float* const* dst,
                                const int64_t* numel, hipStream_t stream) {
    SS_CHECK(count >= 0 && (count == 0 || (src && src_ld && cols && dst && numel)), "ss_multi_add_f32: bad arguments");
    for (int i0 = 0; i0 < count; i0 += SS_MULTI_MAX) {
        MultiAddTable t;
        t.count = count - i0 < SS_MULTI_MAX ? count - i0 : SS_MULTI_MAX;
        int64_t blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int k = i0 + i;
            SS_CHECK(src[k] && dst[k] && numel[k] > 0 && cols[k] > 0 && cols[k] < (1LL << 31) && numel[k] % cols[k] == 0 &&
                     (cols[k] == numel[k] || src_ld[k] >= cols[k]), "ss_multi_add_f32: bad tensor %d (numel %lld cols %lld ld %lld)", k,
                     (long long)numel[k], (long long)cols[k], (long long)src_ld[k]);
            t.src[i] = src[k]; t.dst[i] = dst[k]; t.n[i] = numel[k]; t.ld[i] = src_ld[k]; t.cols[i] = (int)cols[k];
            t.first_block[i] = (int)blocks;
            blocks += cdiv(numel[k], SS_MULTI_PER_BLOCK);
            SS_CHECK(blocks < (1LL << 30), "ss_multi_add_f32: too many elements");
        }
        for (int i = t.count; i <= SS_MULTI_MAX; ++i) t.first_block[i] = (int)blocks;
        for (int i = t.count; i < SS_MULTI_MAX; ++i) { t.src[i] = nullptr; t.dst[i] = nullptr; t.n[i] = 0; t.ld[i] = 0; t.cols[i] = 1; }
        hipLaunchKernelGGL(multi_add_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
        SS_LAUNCH_CHECK();
    }
    return 0;
}
