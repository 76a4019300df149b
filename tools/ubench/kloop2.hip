// k-loop of a 256 x 256 x 64 NT tile with REAL traffic shapes but no epilogue: two 64 KiB stages filled by LDS-DMA from an L2-resident
// buffer, fragment reads from the stage, MFMAs, one barrier per k-tile.  Compares 8 waves x (64 x 128) [24 ds_read_b128 per wave and
// k-tile] against 4 waves x (128 x 128) [32 per wave, but half the waves: 128 KiB instead of 192 KiB of fragment reads per k-tile].
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kloop2.hip -o tools/ubench/kloop2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_void*)p; }
__device__ __forceinline__ void dma_b128(const void* g, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_uniform) : "memory");
}

template <int WAVES, bool FILL, bool READ>
__global__ __launch_bounds__(WAVES * 64) void kloop(const char* __restrict__ src, float* out, int nk, unsigned long long* clk) {
    constexpr int TA = WAVES == 8 ? 2 : 4, TB = 4;                 // 32 x 32 accumulators per wave: TA x TB
    constexpr int NF = 128 / WAVES;                                 // dma_b128 per wave and k-tile (64 KiB / 1 KiB / WAVES)
    __shared__ __attribute__((aligned(1024))) char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * 1024 / 4; i += WAVES * 64) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        ((unsigned*)smem)[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
    }
    __syncthreads();
    floatx16 acc[TA][TB];
    for (int a = 0; a < TA; ++a) for (int b = 0; b < TB; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 fa[TA], fb[TB];
    for (int e = 0; e < 8; ++e) { for (int a = 0; a < TA; ++a) fa[a][e] = (__bf16)1.f; for (int b = 0; b < TB; ++b) fb[b][e] = (__bf16)1.f; }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * (NF * 1024));
    const char* g0 = src + ((size_t)(blockIdx.x % 64) * 65536) + wave * (NF * 1024) + lane * 16;     // 4 MiB window: L2-resident
    const char* frag = smem + lane * 16 + wave * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (FILL) {
            const unsigned st = lds0 + ((kt + 1) & 1) * 65536;
#pragma unroll
            for (int i = 0; i < NF; ++i) dma_b128(g0 + (size_t)(kt & 63) * 4096 + i * 1024, st + i * 1024);
        }
        const char* st = frag + (kt & 1) * 65536;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (READ) {
#pragma unroll
                for (int a = 0; a < TA; ++a) fa[a] = *(const bf16x8*)(st + ((ks * 8 + a) * 8192) % 57344);
#pragma unroll
                for (int b = 0; b < TB; ++b) fb[b] = *(const bf16x8*)(st + ((ks * 8 + 4 + b) * 8192) % 57344);
            }
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
    for (int a = 0; a < TA; ++a) for (int b = 0; b < TB; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int WAVES, bool FILL, bool READ> void run(const char* name, const char* src, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    const int nk = 8192;
    hipLaunchKernelGGL((kloop<WAVES, FILL, READ>), dim3(256), dim3(WAVES * 64), 0, 0, src, d, 64, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kloop<WAVES, FILL, READ>), dim3(256), dim3(WAVES * 64), 0, 0, src, d, nk, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%d waves  %-28s %7.3f us per k-tile  %6.0f TFLOP/s  shader clock %4.0f MHz\n", WAVES, name, ms * 1e3 / nk, 256.0 * 256 * 256 * 64 * 2 * nk / ms / 1e9,
           h[0] / (h[1] / 100.0));
}
int main() {
    char* src; hipMalloc(&src, 8 << 20); hipMemset(src, 0x3f, 8 << 20);
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<8, false, false>("MFMAs only", src, d);
    run<8, false, true>("+ fragment reads", src, d);
    run<8, true, false>("+ fills", src, d);
    run<8, true, true>("+ fragment reads + fills", src, d);
    run<4, false, false>("MFMAs only", src, d);
    run<4, false, true>("+ fragment reads", src, d);
    run<4, true, false>("+ fills", src, d);
    run<4, true, true>("+ fragment reads + fills", src, d);
    return 0;
}
