// Host interface of the bf16 fast-path GEMM kernels (gemm_bf16.hip), used by the C-ABI entry points in gemm.hip.
#pragma once
#include "common.h"

struct FastGemm {
    const bf16* A; const bf16* B; void* C;
    int64_t M; int N; int64_t K;
    int64_t lda, ldb, ldc;
    const float* bias; const bf16* res; int64_t ldr;
    const float* rowscale; int64_t rows_per_scale;
    int act, ntn, ntiles;
    int64_t k_per_split;
    int dbg;
    float* colsum_ws;                       // TN: per-split column sums of A, [nsplit][M] (or null)
    bf16* C2; int64_t ldc2;                 // NT: second output (or null): C2 gets the tile as the epilogue left it, C gets GELU of it
};

// Copy-out of one 16-byte chunk of the NT kernels' LDS tile image: plain, or -- when C2 is set -- the pre-activation to C2 and
// GELU of the (bf16-rounded) pre-activation to C, the very values a separate GELU pass over C2 would have produced.
__device__ __forceinline__ void nt_store_chunk(const FastGemm& p, bf16* C, int64_t row, int col, const bf16x8& v) {
    if (p.C2) {
        *(bf16x8*)(p.C2 + row * p.ldc2 + col) = v;
        bf16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (bf16)gelu_erf((float)v[e]);
        *(bf16x8*)(C + row * p.ldc + col) = a;
    } else {
        *(bf16x8*)(C + row * p.ldc + col) = v;
    }
}
bool ss_fast_nt_ok(const void* A, int64_t lda, const void* B, int64_t ldb, const void* C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, const void* res, int64_t ldr, const float* bias);
int ss_fast_nt_launch(FastGemm p, hipStream_t st);
bool ss_fast_tn_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_fast_tn_launch(FastGemm p, int nsplit, hipStream_t st);
bool ss_skinny_nt_ok(const FastGemm& p);
int ss_skinny_nt_launch(const FastGemm& p, hipStream_t st);
bool ss_skinny_tn_ok(const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_skinny_tn_launch(const FastGemm& p, int nsplit, hipStream_t st);
bool ss_big_nt_ok(const FastGemm& p);
int ss_big_nt_launch(FastGemm p, hipStream_t st);
bool ss_big_tn_plan(int64_t M, int64_t N, int64_t R, int* nsplit, int64_t* rps);
bool ss_big_tn_ok(const FastGemm& p, int nsplit);
int ss_big_tn_launch(FastGemm p, int nsplit, hipStream_t st);

// Epilogue of the NT kernels: every lane turns its accumulators (TA x TB tiles of 32 x 32, transposed orientation: the lane
// is output row `ml`, register quad g holds 4 consecutive columns) into bf16 cells of the LDS [m][n] image.  The activation
// and the residual are compile-time so that the 16..128 elements a lane owns run branch-free.
template <int ACT, bool RES, int TA, int TB>
__device__ __forceinline__ void nt_epilogue_cells(floatx16 (&acc)[TA][TB], const FastGemm& p, char* smem, int pitch, int64_t m0,
                                                  int n0, int mbase, int nbase, int li, int lh) {
#pragma unroll
    for (int a = 0; a < TA; ++a) {
        const int ml = mbase + a * 32 + li;
        const int64_t m = m0 + ml;
        float rs = 1.f;
        if (p.rowscale && m < p.M) rs = p.rowscale[m / p.rows_per_scale];
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = nbase + b * 32 + 8 * g + 4 * lh;
                floatx4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n0 + nl < p.N) bv = *(const floatx4*)(p.bias + n0 + nl);
                bf16x4* cell = (bf16x4*)(smem + ml * pitch + nl * 2);
                bf16x4 rv = {0, 0, 0, 0};
                if (RES) rv = *cell;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][b][4 * g + e] + bv[e];
                    if (ACT == 6) v = v * rs * gelu_erf_grad((float)rv[e]);       // RES is true: the image holds the pre-activation
                    else { v = apply_act(v, ACT); v = RES ? v * rs + (float)rv[e] : v * rs; }
                    o[e] = (bf16)v;
                }
                *cell = o;
            }
    }
}

template <int TA, int TB>
__device__ __forceinline__ void nt_epilogue_dispatch(floatx16 (&acc)[TA][TB], const FastGemm& p, char* smem, int pitch, int64_t m0,
                                                     int n0, int mbase, int nbase, int li, int lh) {
#define SS_EPI(ACT, RES) nt_epilogue_cells<ACT, RES, TA, TB>(acc, p, smem, pitch, m0, n0, mbase, nbase, li, lh)
    if (p.res) {
        switch (p.act) {
            case 1: SS_EPI(1, true); break;
            case 2: SS_EPI(2, true); break;
            case 3: SS_EPI(3, true); break;
            case 4: SS_EPI(4, true); break;
            case 5: SS_EPI(5, true); break;
            case 6: SS_EPI(6, true); break;
            default: SS_EPI(0, true); break;
        }
    } else {
        switch (p.act) {
            case 1: SS_EPI(1, false); break;
            case 2: SS_EPI(2, false); break;
            case 3: SS_EPI(3, false); break;
            case 4: SS_EPI(4, false); break;
            case 5: SS_EPI(5, false); break;
            default: SS_EPI(0, false); break;
        }
    }
#undef SS_EPI
}
