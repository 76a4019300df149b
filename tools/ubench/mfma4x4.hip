// Lane layout and rate of v_mfma_f32_4x4x4_16B_bf16 on gfx950 (16 independent 4x4x4 products per wave instruction: block = lane / 4).
// Assumed: lane 4b + i holds A[i][0..3] of block b, lane 4b + j holds B[0..3][j]; D[i][j] sits in register i of lane 4b + j.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short short4_;
typedef __attribute__((ext_vector_type(4))) float floatx4;
__device__ short bf(float f) { return (short)(__float_as_uint(f) >> 16); }
__global__ void layout(float* out) {
    const int lane = threadIdx.x, b = lane >> 2, i = lane & 3;
    short4_ a, bm;
    for (int k = 0; k < 4; ++k) { a[k] = bf((float)(i * 4 + k + 1 + 16 * (b & 1))); bm[k] = bf(k == i ? (float)(b + 1) : 0.f); }   // B = (b+1) * identity
    floatx4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bm, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    short4_ a = {0x3f80, 0x3f80, 0x3f80, 0x3f80}, b = a;
    floatx4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* d; hipMalloc(&d, 1 << 22);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int b = lane >> 2, j = lane & 3;
        const float want = (float)((r * 4 + j + 1 + 16 * (b & 1)) * (b + 1));     // D[i=r][j] = A[r][j] * (b+1)
        if (h[lane * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", lane, r, h[lane * 4 + r], want); ++bad; }
    }
    printf("layout: %s (%d mismatches)\n", bad ? "DIFFERENT from the assumption" : "as assumed", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1, 2}) {
        hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(256), 0, 0, d, 1000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(256), 0, 0, d, 100000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = 4.0 * 100000 * 4 * 256 * waves;       // wave instructions
        printf("%d wave(s) per SIMD: %.2f ns per instruction per SIMD = %.1f clocks at 2.4 GHz, %.0f TFLOP/s\n", waves, ms * 1e6 / (n / (256 * 4)), ms * 1e6 / (n / 1024) * 2.4,
               n * 2048 / ms / 1e9);
    }
    return 0;
}
