// Host interface of the bf16 fast-path GEMM kernels (gemm_bf16.hip), used by the C-ABI entry points in gemm.hip.
#pragma once
#include "common.h"

// One operand of a GEMM addressed as the non-overlapping s x s patches of a channels-last image [B, H, W, C] (a convolution with
// kernel = stride = s needs no patch matrix: a patch row is s runs of s*C contiguous elements, W*C apart).  Logical row r (patch
// (b*Ho + ho)*Wo + wo) starts at (r / inner) * outer + (r % inner) * ld, logical column k = (ky*s + kx)*C + c sits (k / kc) * cs + k % kc
// further on: inner = Wo, outer = s*W*C, ld = kc = s*C, cs = W*C.  Tiles never straddle a run: kc % 64 == 0 (k-tiles of the NT A
// operand), kc % BN == 0 (column tiles of the TN B operand), kc % 8 == 0 (16-byte chunks of the NT output).
struct PatchMap {
    int side;                               // 0 none; 1 the NT A operand; 2 the NT output; 4 the TN B operand
    int kc; int64_t cs, outer;
    FastDiv inner;
};
__device__ __forceinline__ int64_t patch_row(const PatchMap& pm, int64_t r, int64_t ld) {
    const unsigned q = fdiv((unsigned)r, pm.inner);
    return q * pm.outer + (int64_t)((unsigned)r - q * pm.inner.d) * ld;
}
__device__ __forceinline__ int64_t patch_col(const PatchMap& pm, int k) { const int q = k / pm.kc; return q * pm.cs + (k - q * pm.kc); }

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
    PatchMap pm;                            // patch-addressed operand (side = 0: none); only the 128-row tile kernels take it
    // second product on the finished tile (gemm_nt_bf16_256_kernel, N == 256 only): out2[M, n2] = epilogue(A . B^T) . w2[n2, N]^T, n2 <= 32;
    // C is then not written (the classifier of a 1x1-conv head: the 256-channel activation never reaches memory)
    const bf16* w2; int64_t ldw2; int n2; bf16* out2; int64_t ldo2;
    // NT, one tile per workgroup only: per-row-tile column sums of the stored (rounded) output, stat_ws[(m0 / BM) * 2 + k][N] with
    // k = 0: sum, k = 1: sum of squares -- BatchNorm batch statistics of a 1x1 conv's output from the GEMM that produces it
    float* stat_ws;
};
// Column statistics of the tile image during the copy-out: a thread keeps one 16-byte column chunk over its rows; the row lanes are
// combined through LDS in a fixed order (bitwise reproducible).  Call from all threads after the copy-out loop.
template <int BN, int RPP, int NT>
__device__ __forceinline__ void nt_tile_stats(const FastGemm& p, char* smem, const float (&sx)[8], const float (&sq)[8], int erow, int ech,
                                              int64_t tile_row, int n0, int tid) {
    float* red = (float*)smem;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[((erow * BN) + ech * 8 + e) * 2] = sx[e]; red[((erow * BN) + ech * 8 + e) * 2 + 1] = sq[e]; }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
        if (n0 + c < p.N) {
            float s = 0.f, q = 0.f;
            for (int r = 0; r < RPP; ++r) { s += red[((r * BN) + c) * 2]; q += red[((r * BN) + c) * 2 + 1]; }
            float* dst = p.stat_ws + tile_row * 2 * p.N + n0 + c;
            dst[0] = s; dst[p.N] = q;
        }
    }
}

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
int64_t ss_fast_nt_stat_tiles(const FastGemm& p);      // row tiles (= partial rows of stat_ws) of the kernel ss_fast_nt_launch picks
bool ss_fast_tn_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_fast_tn_launch(FastGemm p, int nsplit, hipStream_t st);
int ss_patch_nt_launch(FastGemm p, hipStream_t st);
int ss_patch_tn_launch(FastGemm p, int nsplit, hipStream_t st);
bool ss_rows_nt_ok(int dtype, const void* A, int64_t lda, const void* C, int64_t ldc, int64_t M, int64_t N, int64_t K);
int ss_rows_nt_launch(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      const float* bias, hipStream_t st);
bool ss_smallmn_tn_ok(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_smallmn_tn_launch(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* slabs, int64_t N, int64_t R, int64_t rps, int nsplit,
                         hipStream_t st);
bool ss_skinny_nt_ok(const FastGemm& p);
int ss_skinny_nt_launch(const FastGemm& p, hipStream_t st);
bool ss_tinym_nt_ok(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb);      // M <= 16 rows: a wave per output column
int ss_tinym_nt_launch(const bf16* A, int64_t lda, const bf16* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int64_t M, int64_t N,
                       int64_t K, const float* bias, hipStream_t st);
bool ss_skinny_tn_ok(const void* B, int64_t ldb, int64_t M, int64_t N);
int ss_skinny_tn_launch(const FastGemm& p, int nsplit, hipStream_t st);
bool ss_stream_nt_ok(const FastGemm& p);                 // gemm_bf16_stream.hip: K = 64 / 128, N <= 256, every wave a pipeline of its own
int ss_stream_nt_launch(FastGemm p, hipStream_t st);
bool ss_wide_nt_ok(const FastGemm& p);                   // gemm_bf16_big.hip: 256 x 320 tiles for N % 320 == 0 (MiT stage 3)
int ss_wide_nt_launch(FastGemm p, hipStream_t st);
bool ss_big_nt_ok(const FastGemm& p);
int ss_big_nt_launch(FastGemm p, hipStream_t st);
int ss_big_nt2_launch(FastGemm p, hipStream_t st);      // the two-product form (p.w2 set): always the one-tile-per-workgroup kernel
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

// Per-WAVE epilogue of the persistent NT kernels: no workgroup barrier and no tile-sized LDS image, so the operand stages stay free
// for the next tile's fills while this tile drains.  A wave turns its TA x TB accumulators into C through 2 KiB of LDS of its own,
// one 32 x 32 accumulator tile at a time: the lane (output row li, 4 consecutive columns per register quad) writes 8-byte cells, the
// tile is read back as 16-byte chunks with 4 lanes along a row, and leaves as 64-byte row segments (the neighbouring tile completes
// the 128-byte line a few instructions later).  8-byte cell q of row r sits at q ^ (((r >> 2) & 3) << 1): cell pairs stay adjacent
// for the 16-byte reads, rows that share banks spread over the bank pairs.  The residual (and the pre-activation of act 6) is read
// in the accumulator layout (8 bytes per lane; the reads that share a line follow each other and hit in L1); bias / activation /
// row scale / residual are applied in fp32 before the one rounding, as in nt_epilogue_cells.
template <int ACT, bool RES, int TA, int TB>
__device__ __forceinline__ void nt_epilogue_wave_cells(floatx16 (&acc)[TA][TB], const FastGemm& p, char* wl, int64_t m0, int n0,
                                                       int mbase, int nbase, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int wsw = ((li >> 2) & 3) << 1;
    const int rr = lane >> 2, c16 = lane & 3;
    bf16* C = (bf16*)p.C;
    // The residual (or the pre-activation of act 6) of tile t+1 is requested before tile t is worked on: one round trip to memory per
    // 32 x 32 tile was the whole cost of a residual epilogue (10 tiles per wave on the 256 x 320 kernel: 45 vs 31 us at K = 320).  The
    // loads are unconditional (rows clamped; what a clamped row fetched is never stored): a load under a branch is waited for at the join.
    bf16x4 rnext[4] = {};
    auto res_load = [&](int t) {
        const int a = t / TB, b = t % TB;
        int64_t m = m0 + mbase + a * 32 + li; if (m >= p.M) m = p.M - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) rnext[g] = *(const bf16x4*)(p.res + m * p.ldr + n0 + nbase + b * 32 + 8 * g + 4 * lh);
    };
    if (RES) res_load(0);
#pragma unroll
    for (int a = 0; a < TA; ++a) {
        const int64_t m = m0 + mbase + a * 32 + li;
        const bool mok = m < p.M;
        float rs = 1.f;
        if (p.rowscale && mok) rs = p.rowscale[m / p.rows_per_scale];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            bf16x4 rcur[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rcur[g] = rnext[g];
            if (RES && a * TB + b + 1 < TA * TB) res_load(a * TB + b + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + nbase + b * 32 + 8 * g + 4 * lh;
                floatx4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *(const floatx4*)(p.bias + n);
                const bf16x4 rv = rcur[g];
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][b][4 * g + e] + bv[e];
                    if (ACT == 6) v = v * rs * gelu_erf_grad((float)rv[e]);
                    else { v = apply_act(v, ACT); v = RES ? v * rs + (float)rv[e] : v * rs; }
                    o[e] = (bf16)v;
                }
                *(bf16x4*)(wl + li * 64 + (((2 * g + lh) ^ wsw) << 3)) = o;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = j * 16 + rr;
                const bf16x8 v = *(const bf16x8*)(wl + r * 64 + ((c16 ^ ((r >> 2) & 3)) << 4));
                const int64_t row = m0 + mbase + a * 32 + r;
                if (row < p.M) nt_store_chunk(p, C, row, n0 + nbase + b * 32 + c16 * 8, v);
            }
            asm volatile("" ::: "memory");
        }
    }
}

template <int TA, int TB>
__device__ __forceinline__ void nt_epilogue_wave(floatx16 (&acc)[TA][TB], const FastGemm& p, char* wl, int64_t m0, int n0, int mbase,
                                                 int nbase, int lane) {
#define SS_EPW(ACT, RES) nt_epilogue_wave_cells<ACT, RES, TA, TB>(acc, p, wl, m0, n0, mbase, nbase, lane)
    if (p.res) {
        switch (p.act) {
            case 1: SS_EPW(1, true); break;
            case 2: SS_EPW(2, true); break;
            case 3: SS_EPW(3, true); break;
            case 4: SS_EPW(4, true); break;
            case 5: SS_EPW(5, true); break;
            case 6: SS_EPW(6, true); break;
            default: SS_EPW(0, true); break;
        }
    } else {
        switch (p.act) {
            case 1: SS_EPW(1, false); break;
            case 2: SS_EPW(2, false); break;
            case 3: SS_EPW(3, false); break;
            case 4: SS_EPW(4, false); break;
            case 5: SS_EPW(5, false); break;
            default: SS_EPW(0, false); break;
        }
    }
#undef SS_EPW
}

// LDS-DMA by hand.  hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of an LDS access it cannot prove disjoint from an LDS-DMA
// builtin in flight (always so for the transpose-read intrinsic, and for plain reads of the same array), which makes every k-iteration
// wait for the prefetch it has just issued; instructions it does not see are waited for by the kernel itself (s_waitcnt vmcnt(N) before
// the barrier that hands a stage over).  M0 (the LDS destination) is written right before each use and nothing else in these kernels reads it.
typedef __attribute__((address_space(3))) void ss_lds_void;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(ss_lds_void*)p; }
__device__ __forceinline__ void dma_b128(const void* g, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ void dma_b32(const void* g, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_uniform) : "memory");
}
