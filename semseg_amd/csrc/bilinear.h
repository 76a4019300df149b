// Bilinear interpolation taps (align_corners = False, torch semantics), shared by the resize kernels (resample.hip) and by the
// BatchNorm kernels that read their input through a resize (norm.hip: ss_bn_*_up), so that both produce the same bits.
#pragma once
#include "common.h"

struct LinTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ LinTap lin_tap(int o, float scale, int in) {
    float src = scale * (o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    LinTap t; t.i0 = (int)src; if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = src - t.i0; t.l0 = 1.f - t.l1;
    return t;
}

// One interpolated value, with the roundings spelled out (explicit fma) so that every kernel below that interpolates -- the
// resize kernels and the fused resize + arg-max of the prediction writer -- produces the same bits from the same taps.
__device__ __forceinline__ float bilerp(const LinTap& ty, const LinTap& tx, float a, float b, float c, float d) {
    const float top = fmaf(tx.l1, b, tx.l0 * a), bot = fmaf(tx.l1, d, tx.l0 * c);
    return fmaf(ty.l1, bot, ty.l0 * top);
}
