// What does the k-loop STRUCTURE of the 256 x 256 NT tile cost, apart from memory?  512 threads (8 waves, 2 per SIMD), per k-tile each wave
// issues 32 MFMAs on 8 accumulators; variants add a workgroup barrier per k-tile, 24 conflict-free ds_read_b128 per k-tile feeding the
// MFMAs (the real fragment traffic), and both.   hipcc --offload-arch=gfx950 -O3 tools/ubench/kloop.hip -o tools/ubench/kloop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

template <bool BAR, bool LDS>
__global__ __launch_bounds__(512) void kloop(float* out, int nk, int random, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // operand values: all 1.0 (nothing toggles in the multipliers) or pseudo-random bf16 in (-2, 2) like real activations / weights
    for (int i = tid; i < 128 * 1024 / 4; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        ((unsigned*)smem)[i] = random ? ((h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u)) : 0x3f803f80u;
    }
    __syncthreads();
    floatx16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 fa[2], fb[4];
    for (int e = 0; e < 8; ++e) { fa[0][e] = fa[1][e] = (__bf16)1.f; fb[0][e] = fb[1][e] = fb[2][e] = fb[3][e] = (__bf16)1.f; }
    const char* base = smem + ((wave * 64 + lane) * 16) % (64 * 1024);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int kt = 0; kt < nk; ++kt) {
        if (BAR) asm volatile("s_barrier" ::: "memory");
        const char* st = base + (kt & 1) * 65536;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (LDS) {
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(st + ((ks * 6 + a) * 8192) % 65536);
#pragma unroll
                for (int b = 0; b < 4; ++b) fb[b] = *(const bf16x8*)(st + ((ks * 6 + 2 + b) * 8192) % 65536);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool BAR, bool LDS> void run(const char* name, float* d, int random, int nk = 4096) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipLaunchKernelGGL((kloop<BAR, LDS>), dim3(256), dim3(512), 0, 0, d, 64, random, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kloop<BAR, LDS>), dim3(256), dim3(512), 0, 0, d, nk, random, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-34s %-7s %6d k-tiles %7.3f us per k-tile (ideal 0.853 at 2.4 GHz)  %6.0f TFLOP/s  shader clock %4.0f MHz\n", name, random ? "random" : "ones", nk,
           ms * 1e3 / nk, 256.0 * 8 * 32 * 32768.0 * nk / ms / 1e9, h[0] / (h[1] / 100.0));
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int random = 0; random < 2; ++random) {
        run<false, false>("MFMAs only", d, random);
        run<true, false>("+ barrier per k-tile", d, random);
        run<false, true>("+ 24 fragment reads per k-tile", d, random);
        run<true, true>("+ both", d, random);
        run<true, true>("+ both, long run", d, random, 65536);
    }
    return 0;
}
