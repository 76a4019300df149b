// Deterministic column reductions over channels-last [rows, C] tensors.
//
// A 256-thread block is CPB chunk-columns (8 channels each) x RPB row lanes; every thread keeps K x 8 fp32 partial
// sums for its fixed 8 channels while it strides over rows, the block combines its row lanes through LDS in a fixed
// order and writes one partial row  partial[(g * nb + blk)][k][C];  colreduce_kernel then sums the nb partial rows
// in a fixed order.  No atomics: results are bitwise reproducible.
#pragma once
#include "common.h"

#define SS_COLRED_MAX_BLOCKS 1024

struct ColGeom { int cpb, rpb, nb; };
static inline ColGeom col_geom(int64_t rows, int C) {
    ColGeom g; int chunks = C / 8;
    g.cpb = chunks < 256 ? chunks : 256;
    g.rpb = 256 / g.cpb;
    int64_t nb = cdiv(rows, (int64_t)g.rpb * 4);
    g.nb = (int)(nb < 1 ? 1 : (nb > SS_COLRED_MAX_BLOCKS ? SS_COLRED_MAX_BLOCKS : nb));
    return g;
}

// F: struct with  static constexpr int K;  __device__ void operator()(int64_t row, int c0, float (&acc)[K][8]) const
template <typename F>
__global__ __launch_bounds__(256) void colreduce_partial_kernel(F f, int64_t rows, int C, int cpb, int rpb, float* __restrict__ partial) {
    constexpr int K = F::K;
    __shared__ float sh[256 * 8];
    const int t = threadIdx.x, ch = t % cpb, rl = t / cpb;
    const int nb = gridDim.x, g = blockIdx.y;
    const int chunks = C / 8;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {
        const int chunk = cbase + ch;
        float acc[K][8];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
        if (rl < rpb && chunk < chunks) {
#pragma unroll 4
            for (int64_t r = (int64_t)blockIdx.x * rpb + rl; r < rows; r += (int64_t)nb * rpb) f((int64_t)g * rows + r, chunk * 8, acc);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[t * 8 + e] = acc[k][e];
            __syncthreads();
            if (rl == 0 && chunk < chunks) {
                float* dst = partial + (((int64_t)g * nb + blockIdx.x) * K + k) * C + chunk * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = 0.f;
                    for (int j = 0; j < rpb; ++j) s += sh[(j * cpb + ch) * 8 + e];
                    dst[e] = s;
                }
            }
        }
    }
}

// out[g][col] = scale * sum_b partial[g][b][col]   (ncols = K*C).  8 columns x 32 row-parts per block: each thread sums
// nb/32 rows with independent loads, the parts are combined through LDS in a fixed order (bitwise reproducible).
static __global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ partial, int nb, int ncols, float scale,
                                                                   float* __restrict__ out, int accumulate) {
    __shared__ float sh[32][9];
    const int cx = threadIdx.x & 7, part = threadIdx.x >> 3;
    const int col = blockIdx.x * 8 + cx, g = blockIdx.y;
    float s = 0.f;
    if (col < ncols) {
        const float* p = partial + (int64_t)g * nb * ncols + col;
#pragma unroll 4
        for (int b = part; b < nb; b += 32) s += p[(int64_t)b * ncols];
    }
    sh[part][cx] = s;
    __syncthreads();
    if (part == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][cx];
        float* o = out + (int64_t)g * ncols + col;
        *o = accumulate ? *o + scale * t : scale * t;
    }
}
