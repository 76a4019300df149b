// LDS read throughput of the GEMM fragment pattern on gfx950: 8 waves of one workgroup per CU, each issuing ds_read_b128 with
// the NT kernels' addressing (32 rows x one 16-byte chunk per half-wave, XOR-swizzled slots) -- and, for comparison, a linear
// lane-contiguous pattern (lane * 16 bytes) and a 2-way / deliberately conflicting one.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_read.hip -o tools/ubench/lds_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int slot(int row, int chunk) { int R = row >> 1; return R * 16 + ((((row & 1) << 3) | chunk) ^ (R & 15)); }

template <int MODE>
__global__ __launch_bounds__(512) void rd(float* out, int iters, long long* cycles) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 131072 / 4; i += 512) ((float*)smem)[i] = (float)i;
    __syncthreads();
    f4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            int off;
            const int base = ((wave * 24 + j + it) & 7) * 16384 + (j & 3) * 4096;       // walks the 128 KiB image
            if (MODE == 0) off = slot((j & 1) * 32 + li + ((j >> 1) & 1) * 64, ((j >> 2) & 3) * 2 + lh) * 16;      // fragment pattern
            else if (MODE == 1) off = lane * 16;                                                                 // linear
            else off = li * 128 + lh * 16;                                                                       // unswizzled rows (conflicts)
            acc += *(const f4*)(smem + ((base + off) & 131071));
        }
    }
    const long long t1 = clock64();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = 1.f;
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc; CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char* names[3] = {"GEMM fragment pattern (swizzled)", "linear lane * 16 B", "unswizzled 128-byte rows"};
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&] { if (mode == 0) rd<0><<<256, 512>>>(out, iters, cyc); else if (mode == 1) rd<1><<<256, 512>>>(out, iters, cyc); else rd<2><<<256, 512>>>(out, iters, cyc); };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double bytes_per_cu = 8.0 * iters * 24 * 1024;
        printf("%-36s %8.1f us  %6.1f B/clk/CU by clock64 (%lld ticks), %6.2f TB/s chip-wide, %.1f ns per wave-instruction per CU\n", names[mode], ms * 1e3,
               bytes_per_cu / (double)c, c, bytes_per_cu * 256 / ms / 1e9, ms * 1e6 / (8.0 * iters * 24));
    }
    return 0;
}
