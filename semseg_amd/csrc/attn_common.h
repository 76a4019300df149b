// Shared device helpers of the MiT attention kernels (attention.hip: generic / fp32; attention_bf16.hip: LDS-staged bf16).
#pragma once
#include "common.h"

__device__ __forceinline__ floatx16 mma32(bf16x8 a, bf16x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx16 mma32(floatx8 a, floatx8 b, floatx16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
template <typename T> struct AT;
template <> struct AT<bf16> { typedef bf16x8 frag; };
template <> struct AT<float> { typedef floatx8 frag; };

__device__ __forceinline__ bf16x8 ldg_frag(const bf16* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ floatx8 ldg_frag(const float* p) {
    floatx4 a = *(const floatx4*)p, b = *(const floatx4*)(p + 4); floatx8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}
template <typename T> __device__ __forceinline__ typename AT<T>::frag cvt_frag(const floatx16& s, int st) {
    typename AT<T>::frag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (T)s[8 * st + e];
    return f;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int perm16(int pos) { return 8 * ((pos & 7) >> 2) + 4 * (pos >> 3) + (pos & 3); }


// LDS transpose of a lane-major accumulator (acc[t][r] = X^T[dd = acc_row(r,h)+32t][col = lane&31]) into row-major
// dst[(row0 + col)*ld + dd], rows < nrows.   `sh` = this wave's 32*(D+1) floats.
template <typename TO, int D>
__device__ __forceinline__ void store_transposed(const floatx16 (&acc)[D / 32], float mul, float* sh, TO* dst, int64_t ld, int nrows) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[j * (D + 1) + t * 32 + acc_row(r, h)] = acc[t][r] * mul;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 * (D / 8) / 64; ++it) {
        int c = lane + it * 64, row = c / (D / 8), cc = c % (D / 8);
        F8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = sh[row * (D + 1) + cc * 8 + e];
        if (row < nrows) store8(dst + (int64_t)row * ld + cc * 8, v);
    }
    __builtin_amdgcn_wave_barrier();
}


typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int CH> __device__ __forceinline__ int img_slot(int row, int chunk) {
    constexpr int RPB = 16 / CH;
    const int R = row / RPB;
    return R * 16 + ((((row % RPB) * CH) + chunk) ^ (R & 15));
}
__device__ __forceinline__ bf16x8 row_frag(const char* img, int slot) { return *(const bf16x8*)(img + slot * 16); }

// A operand X^T[col][row-perm] of a row-major [row][col] image: lane (li = col within the 32-wide tile, lh) gets rows
// base + 8(e>>2) + 4 lh + (e&3), e = 0..7.
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int pitch, int base, int col0) {
    const int lane = threadIdx.x & 63, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3, gsel = (lane >> 4) & 1, lh = lane >> 5;
    const char* a0 = img + (base + 4 * lh + tq) * pitch + (col0 + gsel * 16 + 4 * tp) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 8 * pitch));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}


// bf16 fast path (attention_bf16.hip)
struct AttnBf16Args {
    const bf16 *q, *kv, *o, *dout; const float* lse; bf16 *out, *dq, *dkv; float *lse_out, *delta, *dks, *dvs;
    int B, N, Np, Nk, Nkp, heads, D, C; float c, scale; int ns, qps;
};
int attn_fwd_bf16_launch(const AttnBf16Args& a, hipStream_t stream);
int attn_bwd_bf16_launch(const AttnBf16Args& a, hipStream_t stream);
