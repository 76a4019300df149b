// Shared device/host helpers for the semseg_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(8))) float floatx8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

#define SS_F32 0
#define SS_BF16 1

// ---------------------------------------------------------------- errors
extern thread_local char ss_err_buf[512];
int ss_fail(int code, const char* fmt, ...);
#define SS_CHECK(cond, ...) do { if (!(cond)) return ss_fail(-1, __VA_ARGS__); } while (0)
#define SS_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return ss_fail(-2, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define SS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) \
    return ss_fail(-3, "kernel launch failed at %s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

__host__ __device__ static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- element access
template <typename T> struct VecOf;            // 16-byte vector of T
template <> struct VecOf<float> { static constexpr int N = 4; typedef floatx4 type; };
template <> struct VecOf<bf16> { static constexpr int N = 8; typedef bf16x8 type; };

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// 8 consecutive elements as floats (the unit of work of the bandwidth kernels)
struct F8 { float v[8]; };
__device__ __forceinline__ F8 load8(const float* p) {
    F8 r; floatx4 a = *(const floatx4*)p, b = *(const floatx4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
    return r;
}
__device__ __forceinline__ F8 load8(const bf16* p) {
    F8 r; bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
    return r;
}
__device__ __forceinline__ void store8(float* p, const F8& r) {
    floatx4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = r.v[i]; b[i] = r.v[4 + i]; }
    *(floatx4*)p = a; *(floatx4*)(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16* p, const F8& r) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
    *(bf16x8*)p = a;
}
__device__ __forceinline__ F8 zero8() { F8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
    return r; }

// ---------------------------------------------------------------- division by a launch-invariant integer
// Flat element indices are decomposed into (pixel, tap, channel chunk, ...) in every layout kernel; a hardware integer
// division costs dozens of instructions (hundreds for 64-bit), so the divisors are turned into multiply + shift on the host.
// Valid for dividends < 2^31.
struct FastDiv { unsigned d, mul, sh; };
static inline FastDiv fast_div(unsigned d) {
    FastDiv f; f.d = d; f.mul = 0; f.sh = 0;
    if (d > 1) {
        unsigned lg = 31 - __builtin_clz(d); if (d & (d - 1)) ++lg;
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + d - 1) / d); f.sh = p - 32;
    }
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh); }

// ---------------------------------------------------------------- wave helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over aligned groups of `lanes` lanes (power of two <= 64), every lane receives it.  The first four steps are DPP operand
// modifiers of the add itself (quad permutes, then the half-row and row mirrors: inside a group whose quads / halves already agree a
// mirror is as good as an xor); __shfl_xor is a ds_bpermute_b32, an LDS-crossbar round trip per step.
template <int CTRL> __device__ __forceinline__ float dpp_add_t(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum(float v, int lanes) {      // lanes: power of two <= 64
    if (lanes >= 2) v = dpp_add_t<0xB1>(v);             // quad_perm [1, 0, 3, 2]
    if (lanes >= 4) v = dpp_add_t<0x4E>(v);             // quad_perm [2, 3, 0, 1]
    if (lanes >= 8) v = dpp_add_t<0x141>(v);            // row_half_mirror
    if (lanes >= 16) v = dpp_add_t<0x140>(v);           // row_mirror
    for (int o = 16; o < lanes; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v, int lanes) {
    for (int o = lanes >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Exchange between the two 32-lane halves of a wave: v_permlane32_swap_b32 swaps a's upper half with b's lower half (gfx950), one vector
// instruction where __shfl_xor(v, 32) is a ds_bpermute_b32 round trip through the LDS crossbar.  (The instruction by hand: the builtin's
// two results were folded into one by hipcc 7.0 when both inputs hold the same value.)  tools/ubench/permlane_check.hip checks both.
__device__ __forceinline__ void swap_halves(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xor32_max(float v) { float a = v, b = v; swap_halves(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_sum(float v) { float a = v, b = v; swap_halves(a, b); return a + b; }

// ---------------------------------------------------------------- math
// GELU(x) = x Phi(x), Phi(x) = 0.5 (1 + erf(x / sqrt 2)).  erfc(|z|) = poly(t) exp(-z^2), t = 1 / (1 + p |z|)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute): one exp and one reciprocal serve Phi AND the density term of the
// derivative, instead of libm erff + expf (the GELU kernels are otherwise VALU-bound on an HBM-bound op).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& ez) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);     // v_rcp_f32 (1 ulp), not the IEEE division sequence
    ez = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float hc = 0.5f * poly * ez;                       // 0.5 erfc(|z|)
    cdf = x >= 0.f ? 1.f - hc : hc;
}
__device__ __forceinline__ float gelu_erf(float x) { float cdf, ez; gelu_parts(x, cdf, ez); return x * cdf; }
__device__ __forceinline__ float gelu_erf_grad(float x) { float cdf, ez; gelu_parts(x, cdf, ez); return cdf + x * 0.3989422804014327f * ez; }
// epilogue activation codes shared by the GEMM / depthwise / elementwise kernels: 0 none, 1 GELU(erf), 2 ReLU, 3 ReLU6,
// 4 h-sigmoid = ReLU6(x + 3) / 6 (activation/activations.py), 5 sigmoid
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return gelu_erf(v);
        case 2: return fmaxf(v, 0.f);
        case 3: return fminf(fmaxf(v, 0.f), 6.f);
        case 4: return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case 5: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// Mish(x) = x * tanh(softplus(x)).  With e = exp(x) and n = e (e + 2):  tanh(log(1 + e)) = n / (n + 2)  (no cancellation),
// 1 - tanh^2 = 4 (n + 1) / (n + 2)^2, sigmoid = e / (1 + e): one expf and two divisions instead of expf + log1pf + tanhf.
// x > 20: Mish(x) = x to fp32 precision (torch's softplus threshold); x << 0: e underflows to 0 and both forms return 0.
// Branch-free: e is taken at min(x, 20) (beyond it both forms are 1 to fp32 precision); v_exp_f32 and v_rcp_f32 (~1 ulp each).
__device__ __forceinline__ float mish_f(float x) {
    const float e = __expf(fminf(x, 20.f)), n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
__device__ __forceinline__ float mish_grad(float x) {
    const float e = __expf(fminf(x, 20.f)), n = e * (e + 2.f), d = __builtin_amdgcn_rcpf(n + 2.f);
    const float t = n * d, omt2 = 4.f * (n + 1.f) * d * d;
    return t + x * omt2 * (e * __builtin_amdgcn_rcpf(1.f + e));
}

// dtype dispatch for extern "C" entry points
#define SS_DISPATCH(dtype, ...) \
    do { if ((dtype) == SS_F32) { typedef float T; __VA_ARGS__; } \
         else if ((dtype) == SS_BF16) { typedef bf16 T; __VA_ARGS__; } \
         else return ss_fail(-1, "unsupported dtype code %d", (int)(dtype)); } while (0)
