// Geometric augmentation on the device (reference data_factory/transforms.py:168-475 and train.py:69-113):
// horizontal / vertical flips and the NEAREST affine / perspective resampling that the reference gets from
// torchvision.transforms.functional.affine / perspective on tensors (torchvision is not vendored: its published tensor algorithm is
// restated here -- inverse-map grid in float32, grid_sample(mode="nearest", padding_mode="zeros", align_corners=False), out-of-image
// pixels take `fill`).  Nearest sampling moves elements without arithmetic, so the kernels are typed by element SIZE only: uint8
// images before staging, float images, int64 labels all take the same path, bit for bit.
// Planes are [P][H][W] dense; plane p uses transform p / planes_per_xform (a batch with one transform per sample is one launch)
// and fill pattern p % nfill.
#include "common.h"

struct WarpFill { unsigned long long v[16]; int n; };

// source coordinates in grid_sample's un-normalised form, every float32 operation spelled out (no contraction): the CPU oracle
// repeats this sequence and agrees on every pixel
__device__ __forceinline__ void warp_source(const float* __restrict__ c, int kind, int x, int y, int H, int W, float& sx, float& sy) {
    float gx, gy;
    if (kind == 0) {                                       // affine: c = torchvision's inverse matrix [m0 m1 m2; m3 m4 m5]
        const float bx = (float)x + (0.5f - 0.5f * W), by = (float)y + (0.5f - 0.5f * H);
        const float hw = 0.5f * W, hh = 0.5f * H;
        gx = __fadd_rn(__fadd_rn(__fmul_rn(bx, __fdiv_rn(c[0], hw)), __fmul_rn(by, __fdiv_rn(c[1], hw))), __fdiv_rn(c[2], hw));
        gy = __fadd_rn(__fadd_rn(__fmul_rn(bx, __fdiv_rn(c[3], hh)), __fmul_rn(by, __fdiv_rn(c[4], hh))), __fdiv_rn(c[5], hh));
    } else {                                               // perspective: c = the 8 coefficients of _get_perspective_coeffs
        const float bx = (float)x + 0.5f, by = (float)y + 0.5f;
        const float hw = 0.5f * W, hh = 0.5f * H;
        const float nx = __fadd_rn(__fadd_rn(__fmul_rn(bx, __fdiv_rn(c[0], hw)), __fmul_rn(by, __fdiv_rn(c[1], hw))), __fdiv_rn(c[2], hw));
        const float ny = __fadd_rn(__fadd_rn(__fmul_rn(bx, __fdiv_rn(c[3], hh)), __fmul_rn(by, __fdiv_rn(c[4], hh))), __fdiv_rn(c[5], hh));
        const float den = __fadd_rn(__fadd_rn(__fmul_rn(bx, c[6]), __fmul_rn(by, c[7])), 1.0f);
        gx = __fadd_rn(__fdiv_rn(nx, den), -1.0f);
        gy = __fadd_rn(__fdiv_rn(ny, den), -1.0f);
    }
    sx = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)W), -1.0f), 2.0f);
    sy = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)H), -1.0f), 2.0f);
}

// grid: (blocks of 256 output columns, H rows, P planes)
template <typename E>
__global__ __launch_bounds__(256) void warp_nearest_kernel(const E* __restrict__ src, E* __restrict__ dst, const float* __restrict__ coef,
                                                           int planes_per_xform, int H, int W, int kind, WarpFill fill) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, p = blockIdx.z;
    if (x >= W) return;
    float sx, sy;
    warp_source(coef + (int64_t)(p / planes_per_xform) * 8, kind, x, y, H, W, sx, sy);
    const float rx = rintf(sx), ry = rintf(sy);            // std::nearbyint: ties to even
    E v = (E)fill.v[p % fill.n];
    if (rx >= 0.f && rx < (float)W && ry >= 0.f && ry < (float)H)            // NaN / Inf fall through to the fill
        v = src[((int64_t)p * H + (int)ry) * W + (int)rx];
    dst[((int64_t)p * H + y) * W + x] = v;
}

template <typename E>
__global__ __launch_bounds__(256) void flip_kernel(const E* __restrict__ src, E* __restrict__ dst, int H, int W, int horizontal) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, p = blockIdx.z;
    if (x >= W) return;
    const int64_t base = (int64_t)p * H;
    dst[(base + y) * W + x] = horizontal ? src[(base + y) * W + (W - 1 - x)] : src[(base + (H - 1 - y)) * W + x];
}

#define SS_BY_SIZE(ES, CALL)                                                                          \
    switch (ES) {                                                                                     \
        case 1: { typedef unsigned char E; CALL; } break;                                             \
        case 2: { typedef unsigned short E; CALL; } break;                                            \
        case 4: { typedef unsigned int E; CALL; } break;                                              \
        case 8: { typedef unsigned long long E; CALL; } break;                                        \
        default: return ss_fail(-1, "element size must be 1, 2, 4 or 8 bytes (got %d)", ES);          \
    }

extern "C" int ss_warp_nearest(int kind, int elem_bytes, const void* src, void* dst, int planes, int H, int W, const float* coef,
                               int nxform, const uint64_t* fill_host, int nfill, hipStream_t stream) {
    SS_CHECK(src && dst && coef && src != dst && (kind == 0 || kind == 1), "ss_warp_nearest: bad pointers (src must not alias dst) or kind");
    SS_CHECK(planes > 0 && planes <= 65535 && H > 0 && H <= 65535 && W > 0 && nxform > 0 && planes % nxform == 0,
             "ss_warp_nearest: planes (<= 65535) must be a multiple of the transform count, H <= 65535 (planes=%d nxform=%d H=%d W=%d)", planes,
             nxform, H, W);
    SS_CHECK(nfill >= 0 && nfill <= 16 && (nfill == 0 || fill_host), "ss_warp_nearest: at most 16 fill patterns");
    WarpFill f{};
    f.n = nfill > 0 ? nfill : 1;
    for (int i = 0; i < nfill; ++i) f.v[i] = fill_host[i];
    dim3 grid((unsigned)cdiv(W, 256), (unsigned)H, (unsigned)planes);
    SS_BY_SIZE(elem_bytes, hipLaunchKernelGGL((warp_nearest_kernel<E>), grid, dim3(256), 0, stream, (const E*)src, (E*)dst, coef, planes / nxform, H,
                                              W, kind, f));
    SS_LAUNCH_CHECK();
    return 0;
}

extern "C" int ss_flip(int elem_bytes, const void* src, void* dst, int planes, int H, int W, int horizontal, hipStream_t stream) {
    SS_CHECK(src && dst && src != dst && planes > 0 && planes <= 65535 && H > 0 && H <= 65535 && W > 0,
             "ss_flip: bad arguments (src must not alias dst; planes, H <= 65535)");
    dim3 grid((unsigned)cdiv(W, 256), (unsigned)H, (unsigned)planes);
    SS_BY_SIZE(elem_bytes, hipLaunchKernelGGL((flip_kernel<E>), grid, dim3(256), 0, stream, (const E*)src, (E*)dst, H, W, horizontal));
    SS_LAUNCH_CHECK();
    return 0;
}
