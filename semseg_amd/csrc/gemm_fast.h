// Host interface of the bf16 fast-path GEMM kernels (gemm_bf16.hip), used by the C-ABI entry points in gemm.hip.
#pragma once
#include "common.h"

struct FastGemm {
    const bf16* A; const bf16* B; void* C;
    int64_t M; int N; int64_t K;
    int64_t lda, ldb, ldc;
    const float* bias; const bf16* res; int64_t ldr;
    const float* rowscale; int64_t rows_per_scale;
    int act, ntn;
    int64_t k_per_split;
};
bool ss_fast_nt_ok(const void* A, int64_t lda, const void* B, int64_t ldb, const void* C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, const void* res, int64_t ldr, const float* bias);
int ss_fast_nt_launch(FastGemm p, hipStream_t st);
bool ss_fast_tn_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_fast_tn_launch(FastGemm p, int nsplit, hipStream_t st);
