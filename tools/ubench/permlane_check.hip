#include <hip/hip_runtime.h>
__device__ __forceinline__ void swap32(float& a, float& b) {          // a.hi <-> b.lo
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor32_max(float v) { float a = v, b = v; swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_sum(float v) { float a = v, b = v; swap32(a, b); return a + b; }
__global__ void k(float* o, float* o2) {
    o[threadIdx.x] = xor32_max(o[threadIdx.x]);
    o2[threadIdx.x] = xor32_sum(o2[threadIdx.x]);
}
int main() {
    float h[64], *d, *d2; for (int i = 0; i < 64; ++i) h[i] = (i * 37 % 64) - 20.f;
    (void)hipMalloc(&d, 256); (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    (void)hipMalloc(&d2, 256); (void)hipMemcpy(d2, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d2);
    float o[64], o2[64]; (void)hipMemcpy(o, d, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(o2, d2, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (o[i] != fmaxf(h[i], h[i ^ 32]) || o2[i] != h[i] + h[i ^ 32]) ++bad;
    printf("permlane32 max / sum: %d mismatches\n", bad); return bad;
}
