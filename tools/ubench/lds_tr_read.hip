// Rate of ds_read_b64_tr_b16 (the transposing fragment read of the weight-gradient kernels) on gfx950 against ds_read_b64 and
// ds_read_b128: 8 waves of one workgroup per CU, addressing of gemm_tn_bf16_256_kernel (512-byte rows, slot ^ ((row & 3) << 2)), of the
// wide kernel (slot ^ ((row & 1) << 2)) and without any swizzle.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_tr_read.hip -o tools/ubench/lds_tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: tr read, (row & 3) swizzle; 1: tr read, (row & 1) swizzle; 2: tr read, no swizzle; 3: plain ds_read_b64 at the same addresses as 0;
// 4: ds_read_b128 linear
template <int MODE>
__global__ __launch_bounds__(512) void rd(float* out, int iters, long long* cycles) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5;
    for (int i = tid; i < 131072 / 4; i += 512) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3, gsel = (lane >> 4) & 1;
    const int PITCH = 512;
    const int row_off = (8 * lh + tq) * PITCH;
    const int sub_off = (gsel * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
    const int sw = MODE == 1 ? ((tq & 1) << 2) : (MODE == 2 ? 0 : (tq << 2));
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            // fragment j of a k-step: column base (j % 6) * 32 + wave-dependent, k-step (j / 6), half (j & 1) * 4 rows
            const int colbase = ((j % 6) * 32 + (wave & 1) * 64) & 255, ks = (j / 6 + it) & 3;
            const int off = (ks * 16 * PITCH + row_off + ((((colbase >> 3) ^ sw)) << 4) + sub_off + ((j & 1) ? 4 * PITCH : 0) + (wave >> 1) * 32768) & 131071;
            if (MODE <= 2) {
                bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + off));
                acc += (float)v[0] + (float)v[3];
            } else if (MODE == 3) {
                f2 v = *(const f2*)(smem + off);
                acc += v[0] + v[1];
            } else {
                f4 v = *(const f4*)(smem + ((lane * 16 + j * 1024 + wave * 16384) & 131071));
                acc += v[0] + v[3];
            }
        }
    }
    const long long t1 = clock64();
    if (acc == 1.2345f) out[0] = 1.f;
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc; CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char* names[5] = {"ds_read_b64_tr_b16, slot ^ (row&3)<<2", "ds_read_b64_tr_b16, slot ^ (row&1)<<2", "ds_read_b64_tr_b16, no swizzle",
                            "ds_read_b64, addresses of the first", "ds_read_b128 linear"};
    for (int mode = 0; mode < 5; ++mode) {
        auto launch = [&] {
            if (mode == 0) rd<0><<<256, 512>>>(out, iters, cyc); else if (mode == 1) rd<1><<<256, 512>>>(out, iters, cyc);
            else if (mode == 2) rd<2><<<256, 512>>>(out, iters, cyc); else if (mode == 3) rd<3><<<256, 512>>>(out, iters, cyc);
            else rd<4><<<256, 512>>>(out, iters, cyc);
        };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes_per_cu = 8.0 * iters * 24 * (mode == 4 ? 1024 : 512);
        printf("%-40s %8.1f us  %6.2f TB/s chip-wide, %5.2f ns per wave-instruction per CU (%.1f B/ns/CU)\n", names[mode], ms * 1e3,
               bytes_per_cu * 256 / ms / 1e9, ms * 1e6 / (8.0 * iters * 24), bytes_per_cu / (ms * 1e6));
    }
    return 0;
}
