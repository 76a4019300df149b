/* semseg_hip.h — C ABI of libsemseg_hip.so, the MI355X (gfx950) forward/backward kernels behind the
 * reference's nn.Module / loss / metric API (digital-idiot/SemSeg).
 *
 * The reference is pure Python and has no FFI of its own; each entry point below names the reference
 * Python code (file:line under /root/reference) whose ATen work it replaces.  The Python host side
 * (semseg_amd/_lib.py) builds its ctypes signatures by parsing THIS file, so header and binding cannot drift.
 *
 * Conventions
 *   - dtype codes: SS_F32 = 0 (float), SS_BF16 = 1 (bfloat16 storage, fp32 accumulation).
 *   - activations are channels-last: [B, H, W, C] == token layout [B, N, C]; "ld*" = elements between rows.
 *   - every pointer is a DEVICE pointer borrowed for the call; nothing is kept or freed; workspaces are caller-owned.
 *   - launches are asynchronous on `stream`; return 0 on success, negative on error (ss_last_error() = message,
 *     thread-local); no C++ exception crosses the ABI.
 */
#ifndef SEMSEG_HIP_H
#define SEMSEG_HIP_H
#include <stdint.h>
#ifndef __HIP__
typedef void* hipStream_t;
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define SS_F32 0
#define SS_BF16 1

const char* ss_last_error(void);
int ss_version(void);

/* ---- linear / 1x1-conv projections.  Replaces nn.Linear / 1x1 nn.Conv2d calls:
 * modules/segformer.py:13-15,23,32,40,59,61 (q, kv, proj, fc1, fc2); :18,29 (sr conv as patch GEMM); :70-80 (PatchEmbed
 * conv as im2col GEMM); modules/lawin.py:10,14 (MLP.proj), :49-53,72-80 (g/theta/phi/conv_out), :100 (ConvModule.conv),
 * :137 (linear_pred).
 * C[M,N] = (A[M,K] . B[N,K]^T + bias[N]) * rowscale[row / rows_per_scale] + res[M,N];  act: 0 none, 1 GELU(erf) (before scale).
 * out_f32 != 0 stores C (and reads res) as fp32 even for bf16 operands. */
int ss_gemm_nt(int dtype, int out_f32, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldr,
               const float* rowscale, int64_t rows_per_scale, int act, hipStream_t stream);
/* weight gradient of the same layers: C[M,N] (fp32) (+)= sum_r A[r,M]^T . B[r,N]; deterministic split-R slab reduce. */
int64_t ss_gemm_tn_workspace(int dtype, int64_t M, int64_t N, int64_t R);
int ss_gemm_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
               int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, hipStream_t stream);

/* ---- nn.LayerNorm over C on token layout (modules/segformer.py:19,30,77,83,90,93,145-198). mean/rstd: [rows] fp32. */
int ss_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int64_t rows, int C, float eps, hipStream_t stream);
/* dgamma/dbeta: one [2,C] fp32 buffer (dbeta == dgamma + C); workspace >= 1024*2*C floats. */
int ss_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                     void* dx, float* dgamma, float* dbeta, int64_t rows, int C, float* workspace, int64_t ws_bytes,
                     hipStream_t stream);

/* ---- column sums: bias gradients of every projection above; nn.AdaptiveAvgPool2d(1) of modules/lawin.py:130 (groups = B).
 * out[g][c] (+)= scale * sum_{r<rows} x[g*rows + r][c]; workspace >= groups*2048*C floats. */
int ss_colsum(int dtype, const void* x, int64_t ld, int64_t rows, int C, int groups, float scale, float* out,
              int accumulate, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- nn.BatchNorm2d (+ nn.Mish) of modules/lawin.py:54,101-105 on channels-last rows.
 * stats = [2,C] fp32 (mean, rstd).  ss_bn_stats also updates running_mean/var (unbiased) and num_batches_tracked like
 * torch (pass NULLs to skip).  ss_bn_apply: y = res + mask[row/rows_per_image, c] * act(gamma*(z-mean)*rstd + beta),
 * act: 0 none, 1 Mish; mask = the Dropout2d(0.1) multiplier of modules/lawin.py:138,228 (NULL = none). */
int ss_bn_stats(int dtype, const void* x, int64_t ld, int64_t rows, int C, float eps, float momentum, float* stats,
                float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                int64_t ws_bytes, hipStream_t stream);
int ss_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* stats, hipStream_t stream);
int ss_bn_apply(int dtype, const void* z, int64_t ldz, const float* stats, const float* gamma, const float* beta,
                const void* res, int64_t ldr, const float* mask, int64_t rows_per_image, void* y, int64_t ldy,
                int64_t rows, int C, int act, hipStream_t stream);
/* dz and [2,C] (dgamma, dbeta); workspace >= (2048*2*C + 2*C) floats. */
int ss_bn_bwd(int dtype, const void* dy, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
              const float* beta, const float* mask, int64_t rows_per_image, int training, void* dz, int64_t ldo,
              float* dgamma_dbeta, int64_t rows, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- Mix-FFN depthwise 3x3 (+bias, +GELU) on token layout (modules/segformer.py:44-53,64).
 * w9c: weights repacked tap-major [9][C] fp32.  act: 0 none, 1 GELU(erf).
 * bwd: g = dy*act'(dw(x)+b) (written), dw_db = [10][C] fp32 (9 taps + bias); dx = ss_dwconv3_fwd(g, flipped taps). */
int ss_dwconv3_fwd(int dtype, const void* x, const float* w9c, const float* bias, void* y, int B, int H, int W, int C,
                   int act, hipStream_t stream);
int ss_dwconv3_bwd(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* g, float* dw_db,
                   int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- MiT efficient self-attention (modules/segformer.py:32-39): softmax(q k^T * scale) v without the [B,h,N,Nk] matrix.
 * q,o: [B,N,heads*D]; kv: [B,Nk,2*heads*D] (k | v); lse: [B,heads,round32(N)] fp32 (base-2 log-sum-exp); D in {32,64}. */
int64_t ss_mit_attn_fwd_workspace(int dtype, int B, int Nk, int heads, int D);
int ss_mit_attn_fwd(int dtype, const void* q, const void* kv, void* o, float* lse, int B, int N, int Nk, int heads, int D,
                    float scale, void* workspace, int64_t ws_bytes, hipStream_t stream);
int64_t ss_mit_attn_bwd_workspace(int dtype, int B, int N, int Nk, int heads, int D);
int ss_mit_attn_bwd(int dtype, const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq,
                    void* dkv, int B, int N, int Nk, int heads, int D, float scale, void* workspace, int64_t ws_bytes,
                    hipStream_t stream);

/* ---- im2col / col2im (column order ky,kx,c) feeding ss_gemm_*: PatchEmbed conv k7s4p3 / k3s2p1
 * (modules/segformer.py:70-80) and the spatial-reduction conv k=s=sr (:18,28-29).  Kp = k*k*C rounded up to 8. */
int ss_im2col(int dtype, const void* x, void* col, int B, int H, int W, int C, int k, int s, int pad, int Kp, hipStream_t stream);
int ss_col2im(int dtype, const void* col, void* dx, int B, int H, int W, int C, int k, int s, int pad, hipStream_t stream);

/* ---- F.interpolate(mode='bilinear', align_corners=False) (modules/lawin.py:194-196,205-208,213-218; the S/4 -> S logits
 * resize of SURVEY.md §0.2), reading/writing channel slices (ldx/ldy) of concat buffers. */
int ss_bilinear_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                    hipStream_t stream);
int ss_bilinear_bwd(int dtype, const void* dy, int64_t ldy, void* dx, int64_t ldx, int B, int hi, int wi, int ho, int wo, int C,
                    int accumulate, hipStream_t stream);

/* ---- casts / packing / small elementwise (autocast weight casts, layout changes implied by the reference's
 * permute/reshape/flatten calls, DropPath scaling modules/helpers.py:17-24, torch.cat of a broadcast modules/lawin.py:205-210). */
int ss_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t stream);
int ss_transpose_cast(int dst_dtype, const float* src, void* dst, int R, int Cc, int reverse_cols, hipStream_t stream);
int ss_pack_conv_weight(int dst_dtype, const float* w, void* wp, void* wpt, int O, int I, int k, int Kp, hipStream_t stream);
int ss_unpack_conv_wgrad(const float* gp, float* g, int O, int I, int k, int Kp, hipStream_t stream);
int ss_nchw_to_nhwc(int dst_dtype, const float* x, void* y, int B, int C, int64_t HW, hipStream_t stream);
int ss_add(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t stream);
int ss_rowscale(int dtype, const void* x, const float* scale, int64_t rows_per_scale, void* out, int64_t rows, int C, hipStream_t stream);
int ss_bcast_rows(int dtype, const void* v, void* y, int64_t ldy, int64_t rows_per_image, int64_t rows, int C, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMSEG_HIP_H */
