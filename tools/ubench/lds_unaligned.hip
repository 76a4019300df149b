// Does a 64-bit LDS read at a 2-byte-aligned address work on gfx950, and what does it cost?  (A Toeplitz operand window starts at an
// arbitrary bf16 element: lane i of a 4x4x4 MFMA block wants elements [x0 + i, x0 + i + 4).)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short short4v;
struct __attribute__((packed, aligned(2))) U8 { short4v v; };
template <int SHIFT>
__global__ __launch_bounds__(256) void rd(float* out, int iters, int check) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int e = ((wave * 64 + lane) * 4 + (lane & 3) * SHIFT + it * 8) & 16383 & ~0 ;
        const int ee = e > 16380 ? 0 : e;
        const short4v v = ((const U8*)(lds + ee))->v;
        acc += v[0] + v[1] * 3 + v[2] * 5 + v[3] * 7;
        if (check && it == 0) { out[threadIdx.x * 4 + 0] = v[0] - ee; out[threadIdx.x * 4 + 1] = v[1] - ee - 1; out[threadIdx.x * 4 + 2] = v[2] - ee - 2; out[threadIdx.x * 4 + 3] = v[3] - ee - 3; }
    }
    if (!check) out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}
template <int SHIFT> void run(float* d) {
    hipLaunchKernelGGL((rd<SHIFT>), dim3(1), dim3(256), 0, 0, d, 1, 1);
    float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += h[i] != 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rd<SHIFT>), dim3(1024), dim3(256), 0, 0, d, 100, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rd<SHIFT>), dim3(1024), dim3(256), 0, 0, d, 20000, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("window start offset = lane %% 4 * %d elements: %s, %.2f ns per wave read per CU\n", SHIFT, bad ? "WRONG VALUES" : "values ok", ms * 1e6 / (20000.0 * 4 * 1024 / 256));
}
int main() {
    float* d; hipMalloc(&d, 1 << 22);
    run<0>(d); run<1>(d); run<2>(d); run<4>(d);
    return 0;
}
