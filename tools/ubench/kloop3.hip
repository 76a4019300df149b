// Prefetch depth of the LDS-DMA fills in the k-loop of a 256 x 256 tile: stages of 32 KiB (a 256 x 256 x 32 half k-tile: 16 KiB per operand),
// NST of them, fills issued PD stages ahead, 16 MFMAs per wave and stage (8 waves x 64 x 128).  Reported per 64-deep k-tile (two stages).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kloop3.hip -o tools/ubench/kloop3
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
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NST, int PD, bool HBM>
__global__ __launch_bounds__(512) void kloop(const char* __restrict__ src, float* out, int nst, unsigned long long* clk, int wregions) {
    constexpr int SB = 32768, NF = 4;                               // 4 dma_b128 per wave and stage
    __shared__ __attribute__((aligned(1024))) char smem[NST * SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NST * SB / 4; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        ((unsigned*)smem)[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
    }
    __syncthreads();
    floatx16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * (NF * 1024));
    // HBM: every workgroup streams its own 32 MiB region once; else a 4 MiB window shared by all (L2 hits)
    const char* g0 = src + (HBM ? (size_t)blockIdx.x * (32u << 20) : (size_t)(blockIdx.x % wregions) * 65536) + wave * (NF * 1024) + lane * 16;
    auto issue = [&](int s) {
        const unsigned st = lds0 + (s % NST) * SB;
        const char* g = g0 + (HBM ? (size_t)(s & 1023) * SB : (size_t)(s & 7) * 4096);
#pragma unroll
        for (int i = 0; i < NF; ++i) dma_b128(g + i * 1024, st + i * 1024);
    };
    for (int s = 0; s < PD; ++s) issue(s);
    const char* frag = smem + lane * 16 + wave * 512;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int s = 0; s < nst; ++s) {
        wait_vm<(PD - 1) * NF>();                                   // stage s has landed; the PD - 1 newer ones may be in flight
        asm volatile("s_barrier" ::: "memory");
        issue(s + PD);                                              // into the stage everyone finished reading one barrier ago (NST >= PD + 1)
        const char* st = frag + (s % NST) * SB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const bf16x8*)(st + ((ks * 6 + a) * 4096) % 28672);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = *(const bf16x8*)(st + ((ks * 6 + 2 + b) * 4096) % 28672);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    wait_vm<0>();
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = sum;
}

template <int NST, int PD, bool HBM> void run(const char* src, float* d, int wregions = 64) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    const int nst = HBM ? 1000 : 8192;
    hipLaunchKernelGGL((kloop<NST, PD, HBM>), dim3(256), dim3(512), 0, 0, src, d, 64, clk, wregions);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kloop<NST, PD, HBM>), dim3(256), dim3(512), 0, 0, src, d, nst, clk, wregions);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    if (!HBM) printf("window %4d KiB  ", wregions * 64);
    printf("%s  %d stages of 32 KiB, %d in flight (%3d KiB per CU): %6.3f us per 64-deep k-tile  %5.0f TFLOP/s  fills %5.1f TB/s  clock %4.0f MHz\n",
           HBM ? "HBM stream" : "L2 window ", NST, PD, PD * 32, ms * 1e3 / nst * 2, 256.0 * 256 * 256 * 32 * 2 * nst / ms / 1e9, 256.0 * 32768 * nst / ms / 1e9,
           h[0] / (h[1] / 100.0));
}
int main() {
    char* src; hipMalloc(&src, (size_t)8 << 30); hipMemset(src, 0x3f, (size_t)8 << 30);
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int wr : {1, 8, 16, 32, 64, 256}) { run<2, 1, false>(src, d, wr); run<4, 3, false>(src, d, wr); }
    run<2, 1, true>(src, d); run<3, 2, true>(src, d); run<4, 3, true>(src, d); run<5, 4, true>(src, d);
    return 0;
}
