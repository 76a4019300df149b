// Flat-buffer optimizer step: the whole model's fp32 parameters, gradients and moment buffers are
// four contiguous arrays, so one launch updates every tensor (no per-parameter launches, 16-byte coalesced streams).
#include "common.h"

// AdaBound (Luo et al., ICLR 2019; the reference trains with torch_optimizer.AdaBound(lr=1e-3, final_lr=0.1, amsbound=False),
// train.py:53-58).  Per element, as the published algorithm:
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   step = clamp(step_size / (sqrt(v) + eps), lower, upper) * m;  p -= step
// with step_size = lr * sqrt(1-b2^t) / (1-b1^t), lower = flr (1 - 1/(gamma t + 1)), upper = flr (1 + 1/(gamma t)) computed on the host.
__global__ __launch_bounds__(256) void adabound_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, int64_t n4, int64_t n, float b1, float b2, float eps, float step_size, float lower, float upper,
        float wd, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        floatx4 pv = ((floatx4*)p)[i], gv = ((const floatx4*)g)[i], mv = ((floatx4*)m)[i], vv = ((floatx4*)v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = gv[e] * gscale + wd * pv[e];
            mv[e] = b1 * mv[e] + (1.f - b1) * gr;
            vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
            float s = step_size / (sqrtf(vv[e]) + eps);
            s = fminf(fmaxf(s, lower), upper);
            pv[e] -= s * mv[e];
        }
        ((floatx4*)p)[i] = pv; ((floatx4*)m)[i] = mv; ((floatx4*)v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gscale + wd * p[i];
        float mm = b1 * m[i] + (1.f - b1) * gr, vv = b2 * v[i] + (1.f - b2) * gr * gr;
        float s = fminf(fmaxf(step_size / (sqrtf(vv) + eps), lower), upper);
        p[i] -= s * mm; m[i] = mm; v[i] = vv;
    }
}

extern "C" int ss_adabound_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                                float step_size, float lower, float upper, float weight_decay, float grad_scale, hipStream_t stream) {
    SS_CHECK(n >= 0, "ss_adabound_step: n < 0");
    if (n == 0) return 0;
    SS_CHECK(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "ss_adabound_step: buffers must be 16-byte aligned");
    int64_t n4 = n / 4;
    int blocks = (int)(cdiv(n4 > 0 ? n4 : 1, 256) < 4096 ? cdiv(n4 > 0 ? n4 : 1, 256) : 4096);
    hipLaunchKernelGGL(adabound_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n4, n, beta1, beta2, eps, step_size, lower,
                       upper, weight_decay, grad_scale);
    SS_LAUNCH_CHECK();
    return 0;
}
