// Pure-memory ceilings of GEMM tile shapes on gfx950: how fast can workgroups WRITE a [M x N] bf16 matrix tile by tile (16-byte
// stores, lanes along a row), and READ an [M x K] operand panel in 64-wide k-tiles, as a function of the tile shape -- no MFMA, no LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_stream.hip -o tools/ubench/tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// each workgroup writes one BM x BN tile; thread t covers 16-byte chunk (t % (BN/8)) of rows t / (BN/8) + pass * (T / (BN/8))
template <int T>
__global__ __launch_bounds__(T) void store_tiles(uint16_t* C, int64_t M, int N, int BM, int BN, int remap) {
    const int ntn = N / BN;
    const int wg = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int64_t m0 = (int64_t)(wg / ntn) * BM; const int n0 = (wg % ntn) * BN;
    const int cpr = BN / 8, rpp = T / cpr;
    const int er = threadIdx.x / cpr, ec = threadIdx.x % cpr;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int r = er; r < BM; r += rpp) *(f4*)(C + (m0 + r) * N + n0 + ec * 8) = v;
}

// each workgroup reads a BM x K panel in k-tiles of 64 columns (8 lanes x 16 B per row and k-tile), all n-tiles re-read it (ntn times)
template <int T>
__global__ __launch_bounds__(T) void load_panels(const uint16_t* A, float* sink, int64_t M, int K, int BM, int ntn, int remap, int whole_rows) {
    const int wg = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int64_t m0 = (int64_t)(wg / ntn) * BM;
    f4 acc = {0, 0, 0, 0};
    if (whole_rows) {
        const int cpr = K / 8, rpp = T / cpr, er = threadIdx.x / cpr, ec = threadIdx.x % cpr;
        for (int r = er; r < BM; r += rpp) acc += *(const f4*)(A + (m0 + r) * K + ec * 8);
    } else {
        const int er = threadIdx.x >> 3, ec = threadIdx.x & 7;
        for (int k0 = 0; k0 < K; k0 += 64)
            for (int r = er; r < BM; r += T / 8) acc += *(const f4*)(A + (m0 + r) * K + k0 + ec * 8);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const int64_t M = 262144;
    uint16_t* buf; float* sink;
    CK(hipMalloc(&buf, M * 2560 * 2)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, M * 2560 * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                                   float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10; };
    struct S { int N, BM, BN, T; };
    for (S s : std::vector<S>{{512, 128, 32, 256}, {512, 256, 32, 512}, {512, 256, 256, 512}, {512, 128, 512, 512}, {512, 128, 512, 256}, {512, 64, 512, 256}, {512, 128, 128, 256}, {512, 128, 256, 256},
                              {512, 256, 128, 256}, {256, 128, 128, 256}, {256, 128, 256, 256}, {256, 256, 256, 512}, {128, 128, 128, 256}, {64, 128, 64, 256},
                              {1280, 256, 256, 512}, {1280, 128, 1280, 512}, {1280, 64, 1280, 256}, {2560, 256, 256, 512}, {2560, 64, 2560, 512}, {2560, 128, 128, 256}})
        for (int remap = 0; remap < 2; ++remap) {
            int grid = (int)((M / s.BM) * (s.N / s.BN));
            float ms = s.T == 512 ? time([&] { store_tiles<512><<<grid, 512>>>(buf, M, s.N, s.BM, s.BN, remap); })
                                  : time([&] { store_tiles<256><<<grid, 256>>>(buf, M, s.N, s.BM, s.BN, remap); });
            printf("store N=%4d tile %3dx%4d T=%d remap=%d  %7.1f us  %6.0f GB/s\n", s.N, s.BM, s.BN, s.T, remap, ms * 1e3, M * s.N * 2 / ms / 1e6);
        }
    struct L { int K, BM, ntn, T, whole; };
    for (L l : std::vector<L>{{512, 256, 2, 512, 0}, {512, 128, 4, 256, 0}, {512, 128, 1, 256, 0}, {512, 128, 1, 256, 1}, {512, 64, 1, 256, 1}, {128, 128, 4, 256, 0}, {128, 128, 1, 256, 0},
                              {128, 128, 1, 256, 1}, {320, 256, 5, 512, 0}, {320, 128, 1, 256, 0}, {2560, 256, 2, 512, 0}, {2560, 128, 1, 256, 0}, {64, 128, 1, 256, 0}, {64, 128, 2, 256, 0}})
        for (int remap = 0; remap < 2; ++remap) {
            int grid = (int)((M / l.BM) * l.ntn);
            float ms = l.T == 512 ? time([&] { load_panels<512><<<grid, 512>>>(buf, sink, M, l.K, l.BM, l.ntn, remap, l.whole); })
                                  : time([&] { load_panels<256><<<grid, 256>>>(buf, sink, M, l.K, l.BM, l.ntn, remap, l.whole); });
            printf("load  K=%4d panel %3d rows x%d n-tiles T=%d whole=%d remap=%d  %7.1f us  %6.0f GB/s (unique bytes)\n", l.K, l.BM, l.ntn, l.T, l.whole, remap, ms * 1e3,
                   M * l.K * 2 / ms / 1e6);
        }
    return 0;
}
