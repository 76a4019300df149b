// Stand-alone activation passes (ConvNeXt's nn.GELU between pw_conv1 and pw_conv2, reference modules/convnext.py:29-31,42-44):
// a = GELU(z) and dz = da * GELU'(z).  z (the pre-activation) is what the backward keeps.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ z, T* __restrict__ a, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        F8 v = load8(z + i * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = gelu_erf(v.v[e]);
        store8(a + i * 8, o);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ da, const T* __restrict__ z, T* __restrict__ dz, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        F8 v = load8(z + i * 8), d = load8(da + i * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = d.v[e] * gelu_erf_grad(v.v[e]);
        store8(dz + i * 8, o);
    }
}
static inline int act_blocks(int64_t n8) { int64_t b = cdiv(n8, 256); return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

extern "C" int ss_gelu_fwd(int dtype, const void* z, void* a, int64_t n, hipStream_t stream) {
    SS_CHECK(z && a && n > 0 && n % 8 == 0 && aligned16(z) && aligned16(a), "ss_gelu_fwd: need n %% 8 == 0 and 16-byte aligned pointers");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(act_blocks(n / 8)), dim3(256), 0, stream, (const T*)z, (T*)a, n / 8));
    SS_LAUNCH_CHECK();
    return 0;
}
extern "C" int ss_gelu_bwd(int dtype, const void* da, const void* z, void* dz, int64_t n, hipStream_t stream) {
    SS_CHECK(da && z && dz && n > 0 && n % 8 == 0 && aligned16(z) && aligned16(da) && aligned16(dz), "ss_gelu_bwd: need n %% 8 == 0 and 16-byte aligned pointers");
    SS_DISPATCH(dtype, hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(act_blocks(n / 8)), dim3(256), 0, stream, (const T*)da, (const T*)z, (T*)dz, n / 8));
    SS_LAUNCH_CHECK();
    return 0;
}
