// What does the whole chip sustain on v_mfma_f32_32x32x16_bf16 with nothing else going on, and at which shader clock?
// Every CU runs `waves` waves per SIMD of back-to-back MFMAs on 4 independent accumulators for `iters` iterations; the kernel
// reports s_memtime (shader clock) and s_memrealtime (100 MHz) deltas of one wave, the host the wall time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

__global__ __launch_bounds__(512) void mfma_loop(float* out, uint64_t* clk, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.f; }
    floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    float* out; uint64_t* clk;
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) for (int blocks : {256, 512}) for (int iters : {20000, 200000}) {
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, out, clk, 1000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 4.0 * iters * 32768.0 * (threads / 64) * blocks;
        printf("blocks %4d x %d waves, %6d iters: %8.3f ms  %7.1f TFLOP/s   shader clocks %llu over %llu ticks of 100 MHz -> %.0f MHz, %.1f clocks per MFMA per wave\n",
               blocks, threads / 64, iters, ms, flop / ms / 1e9, (unsigned long long)h[0], (unsigned long long)h[1], h[0] / (h[1] / 100.0),
               (double)h[0] / (4.0 * iters));
    }
    return 0;
}
