// Microbenchmark: do v_exp_f32 (transcendental) and v_mfma_f32_32x32x16_bf16 overlap on one SIMD of gfx950 -- inside one wave
// (independent instruction streams) and across two waves?  Prints cycles per instruction from s_memtime deltas.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
#define N_IT 2000

// mode 0: exp only (8 independent chains), 1: mfma only (2 independent accumulators), 2: both interleaved in one wave,
// 3: waves 0-3 mfma, waves 4-7 exp (a SIMD holds one of each at 8 waves), 4: fma only (8 chains), 5: waves 0-3 mfma, 4-7 fma
__global__ __launch_bounds__(512) void k(int mode, float* out, long long* clk) {
    const int wave = threadIdx.x >> 6;
    float e[8]; floatx16 acc[2];
    for (int i = 0; i < 8; ++i) e[i] = 0.001f * (threadIdx.x + i);
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * (threadIdx.x & 7)); }
    const int grp = (wave >> 2) & 1;          // waves w and w + 4 share a SIMD
    const bool do_exp = mode == 0 || mode == 2 || (mode == 3 && grp);
    const bool do_mfma = mode == 1 || mode == 2 || ((mode == 3 || mode == 5) && !grp);
    const bool do_fma = mode == 4 || (mode == 5 && grp);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (do_exp && do_mfma) {
        for (int it = 0; it < N_IT; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[1], 0, 0, 0);
#pragma unroll
            for (int i = 4; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
        }
    } else if (do_exp) {
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
        }
    } else if (do_mfma) {
        for (int it = 0; it < N_IT; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[1], 0, 0, 0);
        }
    } else if (do_fma) {
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = __builtin_fmaf(e[i], 1.0001f, 0.5f);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += e[i];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 1024 * 8 * 8);
    long long h[16];
    const char* names[] = {"exp only", "mfma only", "exp + mfma in one wave", "exp waves + mfma waves", "fma only", "fma waves + mfma waves"};
    for (int waves = 4; waves <= 8; waves += 4)                 // 4 waves = 1 per SIMD, 8 waves = 2 per SIMD (one workgroup on one CU)
        for (int mode = 0; mode < 6; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<1, waves * 64>>>(mode, out, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<256, waves * 64>>>(mode, out, clk);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, waves * 8, hipMemcpyDeviceToHost);
            printf("waves/CU %d  %-26s  %8.1f us for %d iterations (8 exp / 2 mfma / 8 fma each)  -> %6.2f ns per iteration; wave0 counter delta %lld, last wave %lld\n",
                   waves, names[mode], ms * 1e3, N_IT, ms * 1e6 / N_IT, h[0], h[waves - 1]);
        }
    return 0;
}
