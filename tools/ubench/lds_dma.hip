// What the L2 -> LDS path delivers on gfx950 with global_load_lds_dwordx4, the way the 256-tile GEMM fills its operand stages:
// 256 persistent workgroups x 8 waves, every wave issues 8 x 1 KiB per step (64 KiB per workgroup and step), `STAGES` steps in flight,
// a barrier per step.  Sources: hot (a 256 KiB window per workgroup that stays in L2), stream (every byte new, from HBM), or half/half
// (the GEMM's mix: weights hot, activations streamed), optionally with a second workgroup of the same XCD re-reading the stream.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma.hip -o tools/ubench/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int STAGES, bool BARRIER>
__global__ __launch_bounds__(512) void dma_loop(const char* src, int64_t stream_stride, int steps, int mode, int share) {
    __shared__ __attribute__((aligned(1024))) char smem[STAGES * 65536 > 163840 ? 163840 : STAGES * 65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // share = 2: workgroups b and b + 8 (same XCD) read the same stream, like the two n-tiles of one A panel
    const int sid = share == 2 ? ((blockIdx.x >> 4) << 3 | (blockIdx.x & 7)) : blockIdx.x;
    const char* hot = src + (int64_t)((blockIdx.x >> 3) & 1) * 262144 + wave * 8192 + lane * 16;   // two weight tiles, in every L2
    const char* strm = src + (1LL << 30) + (int64_t)sid * stream_stride + wave * 8192 + lane * 16;
    for (int s = 0; s < steps; ++s) {
        char* st = smem + (s % STAGES) * (STAGES * 65536 > 163840 ? 163840 / STAGES : 65536) + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool h = mode == 0 || (mode == 2 && i >= 4);
            const char* g = h ? hot + ((s & 3) * 65536) + i * 1024 : strm + (int64_t)s * 65536 + i * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(st + i * 1024), 16, 0, 0);
        }
        if (STAGES == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (BARRIER) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the same traffic through registers: global_load_dwordx4, then ds_write_b128 (what the 128-tile kernel does)
typedef __attribute__((ext_vector_type(4))) float f4;
template <bool TO_LDS>
__global__ __launch_bounds__(512) void reg_loop(const char* src, int64_t stream_stride, int steps, int mode, int share, float* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sid = share == 2 ? ((blockIdx.x >> 4) << 3 | (blockIdx.x & 7)) : blockIdx.x;
    const char* hot = src + (int64_t)((blockIdx.x >> 3) & 1) * 262144 + wave * 8192 + lane * 16;   // two weight tiles, in every L2
    const char* strm = src + (1LL << 30) + (int64_t)sid * stream_stride + wave * 8192 + lane * 16;
    f4 acc = {0, 0, 0, 0};
    f4 v[8];
    auto load = [&](int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool h = mode == 0 || (mode == 2 && i >= 4);
            v[i] = *(const f4*)(h ? hot + ((s & 3) * 65536) + i * 1024 : strm + (int64_t)s * 65536 + i * 1024);
        }
    };
    load(0);
    for (int s = 0; s < steps; ++s) {
        char* st = smem + (s & 1) * 65536 + wave * 8192 + lane * 16;
        if (TO_LDS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(f4*)(st + i * 1024) = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        if (s + 1 < steps) load(s + 1);
        __syncthreads();
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f || smem[threadIdx.x] == 77) sink[0] = 1.f;
}

int main() {
    char* buf;
    const int64_t total = (1LL << 30) + 256LL * 64 * 65536 + (1 << 20);
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 64;
    auto run = [&](int stages, bool barrier, int mode, int share, const char* what) {
        auto launch = [&] {
            if (stages == 1) dma_loop<1, true><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share);
            else if (stages == 2 && barrier) dma_loop<2, true><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share);
            else if (stages == 2) dma_loop<2, false><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share);
            else dma_loop<3, true><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share);
        };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-40s stages %d barrier %d  %7.1f us  %6.2f TB/s into LDS\n", what, stages, (int)barrier, ms * 1e3, 256.0 * steps * 65536 / ms / 1e9);
    };
    for (int stages = 1; stages <= 3; ++stages) {
        run(stages, true, 0, 1, "hot (L2 hits)");
        run(stages, true, 1, 1, "stream (HBM)");
        run(stages, true, 2, 1, "half hot / half stream");
        run(stages, true, 2, 2, "half hot / half stream shared by 2 wg");
        run(stages, true, 1, 2, "stream shared by 2 wg");
    }
    float* sink; CK(hipMalloc(&sink, 64));
    auto runreg = [&](bool to_lds, int mode, int share, const char* what) {
        auto launch = [&] {
            if (to_lds) reg_loop<true><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share, sink);
            else reg_loop<false><<<256, 512>>>(buf, 64LL * 65536, steps, mode, share, sink);
        };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-40s via registers, ds_write %d  %7.1f us  %6.2f TB/s\n", what, (int)to_lds, ms * 1e3, 256.0 * steps * 65536 / ms / 1e9);
    };
    for (int to_lds = 0; to_lds < 2; ++to_lds) {
        runreg(to_lds, 0, 1, "hot (L2 hits)");
        runreg(to_lds, 1, 1, "stream (HBM)");
        runreg(to_lds, 2, 2, "half hot / half stream shared by 2 wg");
        runreg(to_lds, 1, 2, "stream shared by 2 wg");
    }
    run(2, false, 0, 1, "hot, no barrier");
    run(2, false, 2, 2, "half/half shared, no barrier");
    return 0;
}
