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
 * C[M,N] = act(A[M,K] . B[N,K]^T + bias[N]) * rowscale[row / rows_per_scale] + res[M,N];  act: 0 none, 1 GELU(erf), 2 ReLU, 3 ReLU6,
 * 4 h-sigmoid, 5 sigmoid (the conv-BN-act blocks of modules/topformer.py with BN folded into B and bias in eval mode);
 * 6: C = (A . B^T + bias) * rowscale * GELU'(res) -- the gradient through the GELU between ConvNeXt's pointwise convolutions
 * (modules/convnext.py:43-45) taken in the epilogue of the data-gradient GEMM, res = the saved pre-activation.
 * out_f32 != 0 stores C (and reads res) as fp32 even for bf16 operands. */
int ss_gemm_nt(int dtype, int out_f32, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldr,
               const float* rowscale, int64_t rows_per_scale, int act, hipStream_t stream);
/* Linear + GELU with both tensors kept (modules/convnext.py:43-45, pw_conv1 -> act: the backward needs the pre-activation for GELU'
 * and the activation for pw_conv2's weight gradient): Z [M, ldz] = A . B^T + bias, G [M, ldg] = GELU_erf(Z as stored).  bf16 tile
 * kernels write both from one epilogue; other shapes / fp32 (dense Z, G) run ss_gemm_nt and ss_gelu_fwd.  Bit-identical either way. */
int ss_gemm_nt_gelu2(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* Z, int64_t ldz, void* G, int64_t ldg,
                     int64_t M, int64_t N, int64_t K, const float* bias, hipStream_t stream);
/* weight gradient of the same layers: C[M,N] (fp32) (+)= sum_r A[r,M]^T . B[r,N]; deterministic split-R slab reduce. */
int64_t ss_gemm_tn_workspace(int dtype, int64_t M, int64_t N, int64_t R);
int ss_gemm_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
               int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, hipStream_t stream);
/* nn.Linear / 1x1-conv backward in one pass over dY: C as above AND colsum[m] (+)= sum_r A[r, m], the bias gradient
 * (autograd of modules/segformer.py:14-18,58-62 and the ConvModules of modules/lawin.py).  M and lda multiples of 8. */
int ss_gemm_tn_colsum(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum,
                      int colsum_accumulate, hipStream_t stream);

/* Weight + bias gradient of a Linear inside a DropPath branch (x + drop_path(f(x)), modules/segformer.py:96-97, modules/helpers.py:9-24)
 * from the UNSCALED output gradient: rows r of A carry the multiplier rowkeep[r / rows_per_keep], which is 0 (sample dropped) or one common
 * positive value 1/(1-p).  The bf16 tile kernels skip the k-tiles of dropped samples and scale their fp32 slabs, so the scaled copy of the
 * gradient (one pass over it per branch) is never made.  ss_gemm_tn_keep_ok: 1 when they take the call (bf16, rows_per_keep a multiple
 * of 64 dividing R, tile-kernel layout); otherwise scale the gradient (ss_rowscale) and call ss_gemm_tn_colsum. */
int ss_gemm_tn_keep_ok(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t R,
                       int64_t rows_per_keep);
int ss_gemm_tn_colsum_keep(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t R, void* workspace, int64_t ws_bytes, int accumulate, float* colsum,
                           int colsum_accumulate, const float* rowkeep, int64_t rows_per_keep, hipStream_t stream);

/* ---- nn.LayerNorm over C on token layout (modules/segformer.py:19,30,77,83,90,93,145-198). mean/rstd: [rows] fp32. */
int ss_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int64_t rows, int C, float eps, hipStream_t stream);
/* dgamma/dbeta: one [2,C] fp32 buffer (dbeta == dgamma + C); workspace >= 1024*2*C floats.  dy2: optional second gradient of the same
 * output, added to dy on the fly (a stage's feature map feeds the head and the next stage, modules/segformer.py:180-198). */
int ss_layernorm_bwd(int dtype, const void* dy, const void* dy2 /* optional */, const void* x, const float* mean, const float* rstd, const float* gamma,
                     const void* res /* optional [rows,C]: added to dx (gradient of the residual path around the norm) */,
                     void* dx, float* dgamma, float* dbeta, int64_t rows, int C, float* workspace, int64_t ws_bytes,
                     hipStream_t stream);

/* ---- column sums: bias gradients of every projection above; nn.AdaptiveAvgPool2d(1) of modules/lawin.py:130 (groups = B).
 * out[g][c] (+)= scale * sum_{r<rows} x[g*rows + r][c]; workspace >= groups*2048*C floats. */
int ss_colsum(int dtype, const void* x, int64_t ld, int64_t rows, int C, int groups, float scale, float* out,
              int accumulate, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* narrow variant (any C <= 32): bias gradient of the nc-class classifier modules/lawin.py:137; workspace >= 512*C floats */
int ss_colsum_small(int dtype, const void* x, int64_t ld, int64_t rows, int C, float* out, float* workspace, int64_t ws_bytes,
                    hipStream_t stream);
/* gradient fan-in of torch.cat slices (modules/lawin.py:210,221): out[r][c] = sum_k x[r][k*slice_stride + c] + bvec[r/rows_per_image][c] */
int ss_sum_slices(int dtype, const void* x, int64_t ld, int nslices, int slice_stride, const float* bvec, int64_t rows_per_image,
                  void* out, int64_t rows, int C, hipStream_t stream);

/* ---- nn.BatchNorm2d (+ nn.Mish) of modules/lawin.py:54,101-105 on channels-last rows.
 * stats = [2,C] fp32 (mean, rstd).  ss_bn_stats also updates running_mean/var (unbiased) and num_batches_tracked like
 * torch (pass NULLs to skip).  ss_bn_apply: y = res + mask[row/rows_per_image, c] * act(gamma*(z-mean)*rstd + beta),
 * act: 0 none, 1 Mish, 2 ReLU, 3 ReLU6 (the conv-BN-act units of modules/topformer.py in train mode); mask = the Dropout2d(0.1) multiplier of modules/lawin.py:138,228 (NULL = none). */
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

/* Bias-free 1x1 conv + the batch statistics of its output in one pass (conv -> BatchNorm2d of modules/lawin.py:49-54,97-105):
 * C = A . B^T + res as ss_gemm_nt, stats / running buffers as ss_bn_stats(C).  On the bf16 tile kernels the column sums of the
 * stored tile ride in the GEMM's copy-out; elsewhere the two calls run.  workspace >= max(cdiv(M,128), 1024) * 2 * N floats. */
int ss_gemm_nt_bn_stats(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t K, const void* res, int64_t ldr, float eps, float momentum, float* stats, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* second pass of ss_bn_bwd alone (act 0, no mask) with the first-pass means given: coef = [2, C] (mean dy, mean dy * xhat) */
int ss_bn_bwd_coef(int dtype, const void* dy, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
                   const float* beta, const float* coef, void* dz, int64_t ldo, int64_t rows, int C, hipStream_t stream);

/* The same three steps for a BatchNorm whose input is a bilinear resize (tnf.interpolate(..., mode='bilinear', align_corners=False),
 * modules/lawin.py:213-222, followed by ConvModule `cat`'s BatchNorm2d + Mish, lawin.py:101-105) of the coarse map x [B*hi*wi, C]:
 * the kernels read z = resize(x) through its four taps (the bits ss_bilinear_fwd would store), so the resized [B*ho*wo, C]
 * tensor is never written.  y, dy and dz are [B*ho*wo, C]; dz is the gradient of the resized map (ss_bilinear_bwd takes it to x).
 * bvec (NULL = none; x2 resize only): [B, C] fp32 added to the resized value of image b -- the image-pool branch of the concat in
 * front of `cat` (modules/lawin.py:205-210) is constant over an image, so its share of the conv is one vector per image that
 * commutes with the resize instead of E of the conv's 5E input columns; its gradient is the per-image column sum of dx.
 * workspace (ss_bn_stats_up) >= 1024*2*C floats, (ss_bn_bwd_up, ss_bn_bwd_up2_dx) >= (1024*2*C + 2*C) floats: enough for every
 * geometry (the partial rows of the first pass never exceed 1024); the calls check the exact need. */
int ss_bn_stats_up(int dtype, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo, int C, float eps, float momentum,
                   float* stats, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                   int64_t ws_bytes, hipStream_t stream);
int ss_bn_apply_up(int dtype, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo, const float* stats,
                   const float* gamma, const float* beta, void* y, int64_t ldy, int C, int act, hipStream_t stream);
int ss_bn_bwd_up(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi, int ho, int wo,
                 const float* stats, const float* gamma, const float* beta, int training, void* dz, int64_t ldo,
                 float* dgamma_dbeta, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* The x2 case (ho = 2 hi, wo = 2 wi) with the transpose of the resize folded into the second pass: returns
 * dx [B*hi*wi, C] = ss_bilinear_bwd(dz) and never writes dz (same weights; the sums run in another order).
 * g_scratch (optional, [B*ho*wo, C] with row pitch ldg in the compute dtype; MAY BE dy itself when the caller no longer needs dy):
 * with act = 1 the first pass leaves g = dy * Mish'(.) there and the second pass reads it instead of evaluating the derivative a
 * second time (g is rounded to the compute dtype, as a separately stored activation gradient would be). */
int ss_bn_bwd_up2_dx(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* bvec, int B, int hi, int wi,
                     const float* stats, const float* gamma, const float* beta, int training, void* dx, int64_t ldo,
                     float* dgamma_dbeta, int C, int act, float* workspace, int64_t ws_bytes, void* g_scratch, int64_t ldg,
                     hipStream_t stream);

/* LawinHead's tail seg = linear_pred(dropout(low_level_fuse's Mish(BatchNorm(z)))) (modules/lawin.py:101-105,227-228) without the
 * E-channel activation in memory.  wp [nc, C] / bp [nc] are the classifier's fp32 parameters (nc <= 8; C = 8 * a power of two,
 * <= 512); seg and dseg are [rows, nc] in the compute dtype.  ss_bn_bwd_classify returns dz, [2,C] (dgamma, dbeta) and the
 * classifier's weight gradient dwp [nc, C] fp32 (its bias gradient is the column sum of dseg: ss_colsum_small);
 * workspace >= (1024*(2+nc)*C + 2*C) floats. */
int ss_bn_apply_classify(int dtype, const void* z, int64_t ldz, const float* stats, const float* gamma, const float* beta,
                         const float* mask, int64_t rows_per_image, const float* wp, const float* bp, int nc, void* seg,
                         int64_t lds_, int64_t rows, int C, hipStream_t stream);
int ss_bn_bwd_classify(int dtype, const void* dseg, int64_t ldd, const void* z, int64_t ldz, const float* stats, const float* gamma,
                       const float* beta, const float* mask, int64_t rows_per_image, int training, const float* wp, int nc, void* dz,
                       int64_t ldo, float* dgamma_dbeta, float* dwp, int64_t rows, int C, float* workspace, int64_t ws_bytes,
                       hipStream_t stream);

/* ---- The whole Mix-FFN of a MiT block in one kernel (modules/segformer.py:56-64 MLP.forward: fc2(GELU(dw_conv(fc1(x)))), with the
 * block's residual and DropPath of :96-99): out = (fc2(GELU(dw3x3(fc1(x) + b1) + bd)) + b2) * keep[image] + res.  The 4C-wide
 * hidden tensors live in LDS, slice by slice.  x, res (or NULL), out: dense [B*H*W, C] bf16 tokens; w1 [Ch, C], w2 [C, Ch] bf16;
 * b1, bd [Ch], b2 [C], w9 [9][Ch] (tap-major), keep [B] (or NULL) fp32.  h1_out / a2_out (both or neither): [B*H*W, Ch] bf16 copies of
 * fc1's output and of the activation, for a backward that reads them.  ss_mixffn_ok: 1 for bf16, C = 64 or 128, Ch = 4C. */
int ss_mixffn_ok(int dtype, int C, int Ch);
int ss_mixffn_fwd(int dtype, const void* x, const void* w1, const float* b1, const float* w9, const float* bd, const void* w2,
                  const float* b2, const void* res, const float* keep, void* out, void* h1_out, void* a2_out, int B, int H,
                  int W, int C, int Ch, hipStream_t stream);

/* ---- Mix-FFN depthwise 3x3 (+bias, +GELU) on token layout (modules/segformer.py:44-53,64).
 * w9c: weights repacked tap-major [9][C] fp32.  act: 0 none, 1 GELU(erf).
 * bwd: g = dy*act'(dw(x)+b) (written), dw_db = [10][C] fp32 (9 taps + bias); dx = ss_dwconv3_fwd(g, flipped taps). */
int ss_dwconv3_fwd(int dtype, const void* x, const float* w9c, const float* bias, void* y, int B, int H, int W, int C,
                   int act, hipStream_t stream);
int ss_dwconv3_bwd(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* g, float* dw_db,
                   int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* The backward in ONE pass (3 tensor passes instead of 5): dx = the gradient of the dwconv input, dw_db as above; g never reaches
 * memory (column neighbours exchange it through DPP lane shifts -- bf16 with C % 64 == 0 -- or through LDS, rounded to the storage
 * dtype as the stored g is; both kernels give the same bits).  dx must not alias dy / x. */
int ss_dwconv3_bwd_fused(int dtype, const void* dy, const void* x, const float* w9c, const float* bias, void* dx, float* dw_db,
                         int B, int H, int W, int C, int act, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* The same pass fed by the gradient of fc2's OUTPUT (modules/segformer.py:58-63 backward: Mix-FFN's fc2 follows the depthwise stage):
 * da2 = (dout . W2) * keep[image] is formed row by row on the matrix cores inside the kernel, the [B*H*W, C] tensor da2 is neither
 * written nor read.  dout [B*H*W, K] bf16 (gradient of the block output on this branch), w2t [C, K] bf16 (fc2.weight transposed: one row
 * per hidden channel), keep [B] fp32 DropPath multipliers or NULL, x = fc1's output [B*H*W, C]; GELU between the two.  bf16, C % 64 == 0,
 * K in {64, 128} (ss_dwconv3_bwd_fc2_ok says where it runs; SS_DW_FC2=0 switches it off).  workspace as above. */
int ss_dwconv3_bwd_fc2_ok(int dtype, int B, int H, int W, int C, int K);
int ss_dwconv3_bwd_fc2(int dtype, const void* dout, const void* w2t, const float* keep, const void* x, const float* w9c, const float* bias,
                       void* dx, float* dw_db, int B, int H, int W, int C, int K, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- general depthwise k x k convolution on channels-last maps: ConvNeXt 7x7 s1 p3 (modules/convnext.py:27,39) and the TopFormer
 * token-pyramid 3x3 / 5x5, stride 1 / 2 with the asymmetric AutoPadding of modules/layers.py:146-164 (modules/topformer.py:166-184).
 * wkkc: tap-major [k*k][C] fp32.  act: 0 none, 1 GELU(erf), 2 ReLU, 3 ReLU6.  The data gradient (stride 1) is this entry point with
 * flipped taps and padding k-1-pad.  ss_dwconvk_wgrad: dw_db = [(k*k + 1)][C] fp32 (taps, then the bias gradient); stride 1,
 * symmetric padding, k in {3,5,7}; workspace >= 1024*(k*k+1)*C floats. */
int ss_dwconvk_fwd(int dtype, const void* x, const float* wkkc, const float* bias, void* y, int B, int H, int W, int C, int k,
                   int stride, int pad_top, int pad_left, int Ho, int Wo, int act, hipStream_t stream);
/* The same with a residual: y = act(dw(x) + bias) + res, res shaped like y.  modules/convnext.py:36-50 backward: the gradient of the
 * block's skip connection joins the depthwise data gradient in the kernel that produces it (was a separate add pass). */
int ss_dwconvk_fwd_res(int dtype, const void* x, const float* wkkc, const float* bias, const void* res, void* y, int B, int H, int W,
                       int C, int k, int stride, int pad_top, int pad_left, int Ho, int Wo, int act, hipStream_t stream);
int ss_dwconvk_wgrad(int dtype, const void* g, const void* x, float* dw_db, int B, int H, int W, int C, int k, int pad,
                     float* workspace, int64_t ws_bytes, hipStream_t stream);
/* out[r] = sum_c a[r][c] * b[r][c] (fp32): layer-scale gradient of modules/convnext.py:33-35,46-47 from the folded weight gradient */
int ss_rowdot_f32(const float* a, const float* b, float* out, int rows, int C, hipStream_t stream);

/* nn.GELU between pw_conv1 and pw_conv2 (modules/convnext.py:30,43): a = GELU(z);  dz = da * GELU'(z).  n elements, n % 8 == 0. */
int ss_gelu_fwd(int dtype, const void* z, void* a, int64_t n, hipStream_t stream);
int ss_gelu_bwd(int dtype, const void* da, const void* z, void* dz, int64_t n, hipStream_t stream);

/* ---- MiT efficient self-attention (modules/segformer.py:32-39): softmax(q k^T * scale) v without the [B,h,N,Nk] matrix.
 * q,o: [B,N,heads*D]; kv: [B,Nk,2*heads*D] (k | v); lse: [B,heads,round32(N)] fp32 (base-2 log-sum-exp); D in {32,64}. */
int64_t ss_mit_attn_fwd_workspace(int dtype, int B, int Nk, int heads, int D);
int ss_mit_attn_fwd(int dtype, const void* q, const void* kv, void* o, float* lse, int B, int N, int Nk, int heads, int D,
                    float scale, void* workspace, int64_t ws_bytes, hipStream_t stream);
int64_t ss_mit_attn_bwd_workspace(int dtype, int B, int N, int Nk, int heads, int D);
int ss_mit_attn_bwd(int dtype, const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq,
                    void* dkv, int B, int N, int Nk, int heads, int D, float scale, void* workspace, int64_t ws_bytes,
                    hipStream_t stream);

/* ---- Two chained 1x1 convolutions, the second one narrow: a head's conv + BN + act followed by its classifier (modules/heads.py:120-170
 * SimpleHead.conv, core/models.py:103-115 classifier).  out2[M, N2] = act(A[M,K] . B[256,K]^T + bias) . W2[N2,256]^T; the 256-wide
 * activation stays in the GEMM's LDS tile and is never written.  ss_gemm_nt_nt_ok: 1 when the tile kernel takes the shape (bf16, N == 256,
 * K % 64 == 0, K >= 128, N2 <= 32 and a multiple of 4), else 0 -- callers then chain two ss_gemm_nt calls. */
int ss_gemm_nt_nt_ok(int dtype, int64_t M, int64_t N, int64_t K, int64_t N2);
int ss_gemm_nt_nt(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, int act, const void* W2,
                  int64_t ldw2, void* out2, int64_t ldo2, int64_t M, int64_t N, int64_t K, int64_t N2, hipStream_t stream);

/* ---- im2col / col2im (column order ky,kx,c) feeding ss_gemm_*: PatchEmbed conv k7s4p3 / k3s2p1
 * (modules/segformer.py:70-80) and the spatial-reduction conv k=s=sr (:18,28-29).  Kp = k*k*C rounded up to 8. */
int ss_im2col(int dtype, const void* x, void* col, int B, int H, int W, int C, int k, int s, int pad, int Kp, hipStream_t stream);
int ss_col2im(int dtype, const void* col, void* dx, int B, int H, int W, int C, int k, int s, int pad, hipStream_t stream);

/* ---- nn.Conv2d(C, N, kernel_size=s, stride=s) on a channels-last image [B, H, W, C] WITHOUT the patch matrix: the spatial-reduction
 * conv of the MiT attention (modules/segformer.py:27-29 `sr`, sr = 8/4/2) and ConvNeXt's 2x2 down-sampling (modules/convnext.py:96-104).
 * The tile GEMM reads / writes the image itself (a patch row is s runs of s*C contiguous elements, W*C apart), so ss_im2col / ss_col2im
 * and their [M, s*s*C] matrix never exist.  M = B*(H/s)*(W/s) patches, weight columns in (ky, kx, c) order as ss_im2col's.
 * ss_conv_patch_ok: 1 when the three entry points take the shape (bf16, H % s == 0, W % s == 0, (s*C) % 64 == 0, C % 8 == 0,
 * N % 8 == 0), else 0 -- callers then use ss_im2col + ss_gemm_*.
 * fwd: y [M, N] = patches(x) . w [N, s*s*C]^T + bias.   dgrad: dx [B, H, W, C] = dy [M, N] . wt [s*s*C, N]^T, every pixel written once.
 * wgrad: dw [N, s*s*C] fp32 = dy^T . patches(x), dbias [N] = column sums of dy (or null); workspace as ss_gemm_tn's. */
int ss_conv_patch_ok(int dtype, int B, int H, int W, int C, int s, int N);
int ss_conv_patch_fwd(int dtype, const void* x, int B, int H, int W, int C, int s, const void* w, int64_t ldw, const float* bias,
                      void* y, int64_t ldy, int N, hipStream_t stream);
int ss_conv_patch_dgrad(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldw, void* dx, int B, int H, int W, int C,
                        int s, int N, hipStream_t stream);
int64_t ss_conv_patch_wgrad_workspace(int dtype, int B, int H, int W, int C, int s, int N);
int ss_conv_patch_wgrad(int dtype, const void* dy, int64_t lddy, const void* x, int B, int H, int W, int C, int s, int N, float* dw,
                        int64_t lddw, float* dbias, void* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- F.interpolate(mode='bilinear', align_corners=False) (modules/lawin.py:194-196,205-208,213-218; the S/4 -> S logits
 * resize of SURVEY.md §0.2), reading/writing channel slices (ldx/ldy) of concat buffers. */
int ss_bilinear_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                    hipStream_t stream);
/* y += resize(x) (fp32 add, one rounding): the sum of up-sampled per-level maps of LawinHead.forward's concat + 1x1 conv
 * (modules/lawin.py:183-199), with the conv commuted below the resize */
int ss_bilinear_fwd_add(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                        hipStream_t stream);
/* y = resize(x) + resize(x2) in one pass (LawinHead.linear_fuse run per level: up(c4-part) + up(c3-part), modules/lawin.py:183-199);
 * the sum is formed in fp32 and rounded once.  C % 8 == 0. */
int ss_bilinear_fwd2(int dtype, const void* x, int64_t ldx, int hi, int wi, const void* x2, int64_t ldx2, int hi2, int wi2,
                     void* y, int64_t ldy, int B, int ho, int wo, int C, hipStream_t stream);
int ss_bilinear_bwd(int dtype, const void* dy, int64_t ldy, void* dx, int64_t ldx, int B, int hi, int wi, int ho, int wo, int C,
                    int accumulate, hipStream_t stream);

/* ---- casts / packing / small elementwise (autocast weight casts, layout changes implied by the reference's
 * permute/reshape/flatten calls, DropPath scaling modules/helpers.py:17-24, torch.cat of a broadcast modules/lawin.py:205-210). */
int ss_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t stream);
int ss_transpose_cast(int dst_dtype, const float* src, void* dst, int R, int Cc, int reverse_cols, hipStream_t stream);
int ss_pack_conv_weight(int dst_dtype, const float* w, void* wp, void* wpt, int O, int I, int k, int Kp, hipStream_t stream);
int ss_unpack_conv_wgrad(const float* gp, float* g, int O, int I, int k, int Kp, hipStream_t stream);
/* ss_weight_prep_batch: the three conversions above for MANY weights in one launch.  table_dev: DEVICE array of nent entries of
 * 8 int64 { src (fp32), dst, n_out, kind | dst_dtype << 8 | reverse_cols << 16, a, b, c, first_block }: kind 0 cast (n_out elements),
 * 1 transpose_cast (a = R, b = Cc), 2 / 3 pack_conv_weight's wp / wpt (a = O, b = I, c = k | Kp << 32); an entry owns blocks
 * first_block .. first_block + ceil(n_out / 1024) - 1 of the launch, total_blocks in all. */
int ss_weight_prep_batch(const int64_t* table_dev, int nent, int64_t total_blocks, hipStream_t stream);
int ss_nchw_to_nhwc(int dst_dtype, const float* x, void* y, int B, int C, int64_t HW, hipStream_t stream);
int ss_add(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t stream);
int ss_rowscale(int dtype, const void* x, const float* scale, int64_t rows_per_scale, void* out, int64_t rows, int C, hipStream_t stream);
/* ss_droppath_keep: keep[i] = floor(keep_prob + u[i]) / keep_prob for the torch.rand draw u (activation dtype; the sum is rounded
 * to it first), modules/helpers.py:20-24 -- the fp32 per-sample multiplier handed to ss_gemm_nt's rowscale. */
int ss_droppath_keep(int dtype, const void* u, float keep_prob, float* keep, int n, hipStream_t stream);
int ss_bcast_rows(int dtype, const void* v, void* y, int64_t ldy, int64_t rows_per_image, int64_t rows, int C, hipStream_t stream);

/* ---- Lawin large-window attention (modules/lawin.py:18-94,140-179).  feat: [B,H,W,C] channels-last (H,W multiples of 8).
 * For ratio r in {8,4,2} the reference's F.unfold(8r x 8r, pad 4(r-1)) + PatchEmbed pooling (:27-35,153-168) is replaced by ONE
 * pooled grid G_r [B,gh,gw,C], gh = (H/8-1)*(8/r)+8: window (i,j) = the 8x8 cells at (i*8/r, j*8/r) (bit-identical values).
 * amax: uint8 [B,gh,gw,C] arg-max position inside each r x r cell for the max-pool backward. */
int ss_lawin_pool_fwd(int dtype, const void* feat, int64_t ld, void* G, void* amax, int B, int H, int W, int C, int r,
                      hipStream_t stream);
/* d(feat) = base + sum over r of the avg and arg-max shares of dG_r (fp32 [B,gh,gw,C]) */
int ss_lawin_pool_bwd(int dtype, const float* dG8, const float* dG4, const float* dG2, const void* amax8, const void* amax4,
                      const void* amax2, const void* base, int64_t ldb, const float* bvec /* optional [B, C]: added per image */,
                      void* dfeat, int64_t ldo, int B, int H, int W, int C,
                      hipStream_t stream);
/* Stand-alone PatchEmbed.forward (modules/lawin.py:27-35): y [B,ceil(H/ps),ceil(W/ps),C] = 0.5*(max_pool + avg_pool) over ps x ps cells of
 * x [B,H,W,C] (row stride ldx), zeros beyond the right / bottom edge taking part as F.pad's do; amax: uint8 arg-max position per cell.
 * ss_patch_pool_bwd: dx [B,H,W,C] from dy and amax (an arg-max that fell on the padding keeps no gradient, as in the reference). */
int ss_patch_pool_fwd(int dtype, const void* x, int64_t ldx, void* y, void* amax, int B, int H, int W, int C, int ps, hipStream_t stream);
int ss_patch_pool_bwd(int dtype, const void* dy, const void* amax, void* dx, int B, int H, int W, int C, int ps, hipStream_t stream);
/* window-ordered d(ctx) [B*nh*nw*64, C] -> dG_r (gather form, deterministic) */
int ss_lawin_gather_bwd(int dtype, const void* dctx, float* dG, int B, int H, int W, int C, int r, hipStream_t stream);
/* The branches' BatchNorm (no activation: BN_r(z) = s_r z + t_r) folded into the `cat` conv over [feat, feat + BN_8(z_8), feat + BN_4(z_4),
 * feat + BN_2(z_2), pool] (modules/lawin.py:88-93,202-222): the conv runs over [feat | z_8 | z_4 | z_2] with Wout = [W_0 + W_1 + W_2 + W_3 |
 * W_r diag(s_r)] (compute dtype, [E, 4E]) and the constant cconst = sum_r W_r t_r, so the normalised maps are never written.  W: the conv
 * weight [E, >= 4E] fp32, stats_r = [2, E] (mean, rstd).  ss_lawin_cat_wgrad_fix turns G = dz^T [feat | z_8 | z_4 | z_2] (in place, the
 * first 4E columns of [E, ldg]) into the gradient of W: dW_r = G_0 + G_r diag(s_r) + colsum(dz) (x) t_r, dvb = per-image column sums of dz [B, E]. */
int ss_lawin_cat_fold(int dtype, const float* W, int64_t ldw, const float* stats8, const float* stats4, const float* stats2,
                      const float* gamma8, const float* gamma4, const float* gamma2, const float* beta8, const float* beta4,
                      const float* beta2, int E, void* Wout, void* Wt /* [4E, E]: (W_0+W_1+W_2+W_3)^T, W_1^T, W_2^T, W_3^T */, float* cconst,
                      hipStream_t stream);
/* First pass of the three branch BatchNorm backwards from G = dz^T [feat | z_8 | z_4 | z_2] (BEFORE ss_lawin_cat_wgrad_fix), the cat conv
 * weight W and dvb: the gradient at BN_r's output is dz . W_r, so its column sums are sum_n W_r[n][k] colsum(dz)[n] and
 * sum_n W_r[n][k] G_r[n][k].  dgb = [3][2][E] (dgamma, dbeta per branch), coef = [3][2][E] for ss_bn_bwd_coef; rows = B*H*W. */
int ss_lawin_branch_bn_sums(const float* G, int64_t ldg, const float* W, int64_t ldw, const float* stats8, const float* stats4,
                            const float* stats2, const float* dvb, int B, int E, int64_t rows, int training, float* dgb, float* coef,
                            hipStream_t stream);
int ss_lawin_cat_wgrad_fix(float* G, int64_t ldg, const float* stats8, const float* stats4, const float* stats2,
                           const float* gamma8, const float* gamma4, const float* gamma2, const float* beta8, const float* beta4,
                           const float* beta2, const float* dvb, int B, int E, hipStream_t stream);
/* LawinAttn.position_mixing (:44-46,62-69): out[w,p',c] = x[w,p',c] + sum_p Wm[head(c)][p'][p] x[w,p,c] + bm[head(c)][p'];
 * x gathered from G (xwin NULL) or read window-ordered (G NULL); transpose != 0 applies Wm^T without bias (backward).
 * Wm: [heads][64][64] fp32, bm: [heads][64] fp32; out: [windows*64, C].
 * workspace (optional, ss_lawin_mix_workspace(heads) bytes): enables the bf16 matrix-core path (C % 128 == 0, C / heads in 8..128). */
int64_t ss_lawin_mix_workspace(int heads);
int ss_lawin_mix(int dtype, const void* G, const void* xwin, const float* Wm, const float* bm, void* out, int B, int H, int W,
                 int C, int r, int heads, int transpose, void* workspace, int64_t ws_bytes, hipStream_t stream);
int64_t ss_lawin_mix_wgrad_workspace(int B, int H, int W, int C, int heads);
/* dWb: [heads][64][65] fp32 (64 weight columns, then the bias column) */
int ss_lawin_mix_wgrad(int dtype, const void* G, const void* dwin, float* dWb, int B, int H, int W, int C, int r, int heads,
                       float* workspace, int64_t ws_bytes, hipStream_t stream);
/* per (window, head) softmax(theta phi^T / sqrt(d)) g (:72-91).  theta / y / dy / dtheta: image-ordered rows, `inter` channels;
 * pg / dpg: window-ordered rows [windows*64, 2*inter] = (phi | g).  lse: [windows][heads][64] fp32.  d = inter/heads in {2..64}. */
int ss_lawin_attn_fwd(int dtype, const void* theta, int64_t ldt, const void* pg, void* y, int64_t ldy, float* lse, int B, int H,
                      int W, int inter, int heads, hipStream_t stream);
int ss_lawin_attn_bwd(int dtype, const void* theta, int64_t ldt, const void* pg, const void* y, int64_t ldy, const void* dy,
                      int64_t lddy, const float* lse, void* dtheta, int64_t lddt, void* dpg, int B, int H, int W, int inter,
                      int heads, hipStream_t stream);

/* ---- OhemCrossEntropyLoss.forward (loss/seg_loss.py:121-167), fully on device.  logits: [B,C,H,W] addressed by element strides
 * (sn,sc,sh,sw); labels int64 [B,H,W].  thr_state: the module's stateful threshold (device float, updated in place, :143-144).
 * prob [B*H*W] fp32 and the 256-byte ctl record are kept by the caller for the backward. */
int64_t ss_ohem_ce_workspace(int64_t npix);
int ss_ohem_ce_fwd(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                   int B, int C, int H, int W, int64_t ignore_index, double min_kept, float* thr_state, float* prob,
                   void* ctl_out, float* loss, void* workspace, int64_t ws_bytes, hipStream_t stream);
int ss_ohem_ce_bwd(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                   const float* prob, const void* ctl, const float* gout, void* dlogits, int64_t on, int64_t oc, int64_t oh,
                   int64_t ow, int B, int C, int H, int W, int64_t ignore_index, hipStream_t stream);
/* ---- DiceLoss / dice_loss (loss/seg_loss.py:15-55,82-99) and class_weights (helper/weights.py:18-27):
 * sums[b][c] = {sum p*[y=c], sum p, #[y=c]}; backward dprobs = gout*(coef_all[b][c] + coef_hit[b][c]*[y=c]). */
int ss_dice_sums(int dtype, const void* probs, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels, int B,
                 int C, int H, int W, float* sums, float* workspace, int64_t ws_bytes, hipStream_t stream);
int ss_dice_bwd(int dtype, const int64_t* labels, const float* coef_hit, const float* coef_all, const float* gout, void* dprobs,
                int64_t on, int64_t oc, int64_t oh, int64_t ow, int B, int C, int H, int W, hipStream_t stream);
/* ss_dice_sums_static: ss_dice_sums for callers that CANNOT ALLOCATE -- code under HIP graph capture, tight serving / evaluation
 * loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same kernels, same results bit for bit; the
 * per-workgroup partial sums live in a 1 MiB file-scope device array (one copy per GPU, held until the process ends) instead of a
 * caller workspace, and the call performs no allocation of any kind.  Limit: B * min(64, ceil(H*W / 4096)) * 3 * C <=
 * ss_dice_sums_static_max_floats() (= 262144; e.g. B = 16, C = 85 at 1024 x 1024), beyond which it fails with a message.  Calls
 * that overlap on one GPU (different streams) race on the array: serialise them. */
int64_t ss_dice_sums_static_max_floats(void);
int ss_dice_sums_static(int dtype, const void* probs, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels, int B,
                        int C, int H, int W, float* sums, hipStream_t stream);
/* helper/weights.py:18-22 (class_weights: the label frequencies; the reference compares every pixel with every class id in a
 * [pixels x classes] boolean matrix): hist[v] += #(labels == v) for v in [0, nbins), int64, exact; other values are not counted. */
int ss_label_hist(const int64_t* labels, int64_t n, int nbins, int64_t* hist, hipStream_t stream);
/* ---- torch.nn.Softmax2d (modules/layers.py:41) for logits with no fused producer: the probabilities DiceLoss takes when MiT + Lawin
 * trains with OHEM-CE + Dice (BASELINE configs[4]).  All tensors [B][C][H][W] with element strides (n, c, h, w).
 * backward: dx[c] = y[c] * (dy[c] - sum_k dy[k] y[k]). */
int ss_softmax2d_fwd(int dtype, const void* x, int64_t xn, int64_t xc, int64_t xh, int64_t xw, void* y, int64_t yn, int64_t yc,
                     int64_t yh, int64_t yw, int B, int C, int H, int W, hipStream_t stream);
int ss_softmax2d_bwd(int dtype, const void* y, int64_t yn, int64_t yc, int64_t yh, int64_t yw, const void* dy, int64_t gn, int64_t gc,
                     int64_t gh, int64_t gw, void* dx, int64_t xn, int64_t xc, int64_t xh, int64_t xw, int B, int C, int H, int W,
                     hipStream_t stream);
/* ---- SegmentationMetrics.update (helper/metrics.py:90-95 + torchmetrics ConfusionMatrix): cm[target][argmax] += 1, int64, exact;
 * first maximum wins ties. */
int ss_confusion_update(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const int64_t* labels,
                        int64_t* cm, int B, int C, int H, int W, hipStream_t stream);
/* the same for predictions that already are class indices (helper/metrics.py:90-95: torchmetrics' ConfusionMatrix takes scores or
 * int labels): pred, labels int64 [n]; pairs with either value outside [0, C) are not counted. */
int ss_confusion_indices(const int64_t* pred, const int64_t* labels, int64_t* cm, int64_t n, int C, hipStream_t stream);

/* ---- Next rows of SURVEY.md 8(f): the steps right after the forward in predict.py and scripts/evaluate.py.
 * ss_class_map_u8: PredictionWriter.write_on_batch_end (helper/callbacks.py:52-54), prediction.softmax(1).argmax(1): softmax is
 *   increasing in the logit, so the arg-max is taken on the logits (first maximum wins, a pixel holding a NaN -> 0, what softmax +
 *   torch.argmax give); out = uint8 [B][H][W], dense, 4-byte aligned -- one byte per pixel to the host instead of an int64.
 * ss_bilinear_class_map_u8: the same on logits still at the head's resolution, x = [B*hi*wi rows, C of ldx columns] channels-last:
 *   out[b][oy][ox] = arg-max over c of the value ss_bilinear_fwd would have stored at (oy, ox) (same taps and roundings), so
 *   core/models.py:105-115's full-size logits are never written.  out = uint8 [B][ho][wo].
 * ss_confusion_labels_u8: scripts/evaluate.py:130-135, sklearn confusion_matrix(y_true, y_pred, labels) of two class-map rasters
 *   over the pixels `valid` marks (null = all); lut = int32 [256] on the device, raster value -> index in `labels` or -1 (pixel
 *   dropped, as sklearn does for a value that is not listed); cm = int64 [C][C] (rows = truth), accumulated in place, exact. */
int ss_class_map_u8(int dtype, const void* logits, int64_t sn, int64_t sc, int64_t sh, int64_t sw, uint8_t* out, int B, int C, int H,
                    int W, hipStream_t stream);
int ss_bilinear_class_map_u8(int dtype, const void* x, int64_t ldx, uint8_t* out, int B, int hi, int wi, int ho, int wo, int C,
                             hipStream_t stream);
int ss_confusion_labels_u8(const uint8_t* truth, const uint8_t* pred, const uint8_t* valid, const int* lut, int64_t* cm, int64_t n, int C,
                           hipStream_t stream);

/* ---- data_factory/utils.py:6-37 on the device: the batch is uploaded as uint8.
 * ss_stage_image_u8: image_to_tensor, out[b][c][y][x] = (float(src) - mu[c]) / sigma[c] in fp32 with a correctly rounded division,
 *   src = uint8 [B][C][H][W] dense, out in out_dtype with element strides (on, oc, oh, ow): NCHW fp32 is the reference tensor bit
 *   for bit, channels-last compute dtype is what the first layer reads without another pass.  mu, sigma: fp32 [C] on the device.
 * ss_stage_label_u8: label_to_tensor, uint8 [n] -> int64 [n]. */
int ss_stage_image_u8(int out_dtype, const uint8_t* src, const float* mu, const float* sigma, void* out, int64_t on, int64_t oc,
                      int64_t oh, int64_t ow, int B, int C, int H, int W, hipStream_t stream);
int ss_stage_label_u8(const uint8_t* src, int64_t* out, int64_t n, hipStream_t stream);

/* ---- geometric augmentation on the device (data_factory/transforms.py:168-475; train.py:69-113: hflip, vflip, RandomAffine,
 * RandomPerspective, all NEAREST).  Planes [planes][H][W] dense of elem_bytes-sized elements (1, 2, 4, 8: uint8 images, bf16 / fp32
 * images, int64 labels -- nearest sampling moves elements, no arithmetic); src must not alias dst.
 * ss_warp_nearest: kind 0 = affine, coef[k] = 8 floats per transform on the DEVICE holding torchvision's inverse matrix
 *   [m0 m1 m2 m3 m4 m5 - -]; kind 1 = perspective, the 8 coefficients of its _get_perspective_coeffs.  Plane p uses transform
 *   p / (planes / nxform).  Output pixel -> source by the float32 grid of torchvision's tensor path and grid_sample(nearest, zeros,
 *   align_corners=False); pixels whose source falls outside take fill_host[p % nfill] (bit patterns, read on the HOST at launch;
 *   nfill = 0: zeros).  torchvision is not vendored in the reference: parity unpinned, checked against oracle/augment.py.
 * ss_flip: horizontal != 0 mirrors x (hflip), else y (vflip). */
int ss_warp_nearest(int kind, int elem_bytes, const void* src, void* dst, int planes, int H, int W, const float* coef, int nxform,
                    const uint64_t* fill_host, int nfill, hipStream_t stream);
int ss_flip(int elem_bytes, const void* src, void* dst, int planes, int H, int W, int horizontal, hipStream_t stream);

/* ---- core/decoder.py:49-59 tail on channels-last maps.  x = LawinHead output [B,H4,W4,2oc].  The host runs two GEMMs on x:
 * z9 [rows, ld9 >= 9*oc] = x . Wd (ConvTranspose2d k3 s4 weight as [(o,dy,dx)][c], modules/blocks.py:224-234) and bs [rows, oc] =
 * conv1x1(x) + b (feature_reducer, core/decoder.py:38-47; the 1x1 conv commutes with the nearest up-sampling of :52-58).
 * ss_decoder_stats: BatchNorm2d statistics of the (4H4-1)x(4W4-1) deconvolved map (stride 4 > kernel 3: gaps hold the bias only);
 * stats = [2*oc] (mean, rstd), running buffers updated as torch does.  ss_decoder_combine_fwd: prod [B,(4H4-1),(4W4-1),oc] =
 * ReLU(BN(deconv)) * nearest-up(bs).  ss_decoder_combine_bwd: dz9 (ld9 columns), dbs, dgb = [2*oc] (dgamma, dbeta), dbias_grad [oc];
 * workspace >= (1024*2*oc + 4*oc) floats.  oc <= 32, H4, W4 <= 1024. */
int ss_decoder_stats(int dtype, const void* z9, int64_t ld9, int B, int H4, int W4, int oc, const float* dbias, float eps,
                     float momentum, float* stats, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* workspace, int64_t ws_bytes, hipStream_t stream);
int ss_decoder_combine_fwd(int dtype, const void* z9, int64_t ld9, const void* bs, int64_t ldb, const float* dbias, const float* stats,
                           const float* gamma, const float* beta, void* prod, int B, int H4, int W4, int oc, hipStream_t stream);
int ss_decoder_combine_bwd(int dtype, const void* dprod, const void* z9, int64_t ld9, const void* bs, int64_t ldb, const float* dbias,
                           const float* stats, const float* gamma, const float* beta, int training, void* dz9, void* dbs, float* dgb,
                           float* dbias_grad, int B, int H4, int W4, int oc, float* workspace, int64_t ws_bytes, hipStream_t stream);
/* ---- core/dnet.py:45-51 + MultilabelClassifier (modules/layers.py:25-43): nearest resize (Hd,Wd) -> (H,W) (torch's float index
 * rule), 1x1 conv oc -> nc + bias, Softmax2d.  probs / dprobs: NCHW [B,nc,H,W]; wt [nc][oc], bt [nc] fp32.
 * ss_terminator_bwd: dprod [B,Hd,Wd,oc], dw_db = [nc*oc + nc] fp32; workspace >= 1024*(nc*oc + nc) floats.
 * probs_f32 != 0: probs / dprobs are fp32 whatever `dtype` is (torch.autocast runs softmax in fp32, so the reference's reduced-
 * precision path hands fp32 probabilities to DiceLoss; 6 channels at full resolution: +0.2 GB of traffic per 16 x 1024^2 step). */
int ss_terminator_fwd(int dtype, int probs_f32, const void* prod, const float* wt, const float* bt, void* probs, int B, int Hd, int Wd, int H,
                      int W, int oc, int nc, hipStream_t stream);
int ss_terminator_bwd(int dtype, int probs_f32, const void* dprobs, const void* prod, const float* wt, const float* bt, void* dprod,
                      float* dw_db, int B, int Hd, int Wd, int H, int W, int oc, int nc, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- TopFormer token pyramid (modules/topformer.py, modules/heads.py, core/models.py), inference.
 * ss_im2col_ex: im2col with the asymmetric AutoPadding of modules/layers.py:146-164 (3x3 / s2 stem, :213-250).
 * ss_avgpool: adaptive_avg_pool2d with exact win x win windows (PyramidPoolAgg :527-552), y may be a channel slice of the concat buffer.
 * ss_act: y = act(x) (codes as ss_gemm_nt; the 'ACN' projection :394-409).  ss_inject_fuse: InjectionDotSum.forward :640-673,
 * out = local * bilinear(gact) + bilinear(gemb).  ss_bilinear_add: y += bilinear(x) (UniRes + Sum, modules/heads.py:60-106). */
/* One inverted-residual block of the token pyramid in one kernel (modules/topformer.py InvertedResidual.forward: conv_in 1x1 + BN + act
 * -> conv_dw k x k, stride 1 / 2, AutoPadding + BN + act -> conv_out 1x1 + BN (+ x when stride 1 and inc == outc); BatchNorm folded,
 * inference).  The expanded tensor lives only in LDS.  x [B, H, W, Cin] -> y [B, Ho, Wo, Cout], channels-last bf16.
 * Host-padded weights (padding zero): w_in [hidp][cinp] bf16 (cinp = Cin rounded up to 16; identity for a block without expansion,
 * act_in 0), b_in [hidp], w_dw [k*k][hidp] fp32 tap-major, b_dw [hidp], w_out [coutp][hidp] bf16 (coutp = Cout rounded up to 32),
 * b_out [coutp].  ss_ir_block_ok: 1 when the kernel takes the block (bf16, k 3 / 5, stride 1 / 2, Cin <= 64, Cin and Cout multiples
 * of 8, hidp a multiple of 32, tile + weights within 80 KiB of LDS so two workgroups share a CU), else 0 -- callers then run the three
 * launches. */
int ss_ir_block_ok(int dtype, int Cin, int hidp, int Cout, int k, int s);
int ss_ir_block_fwd(int dtype, const void* x, void* y, int B, int H, int W, int Cin, int hidp, int Cout, int k, int s, int pad_top,
                    int pad_left, int Ho, int Wo, const void* w_in, const float* b_in, int act_in, const float* w_dw,
                    const float* b_dw, int act_dw, const void* w_out, const float* b_out, int residual, hipStream_t stream);
int ss_im2col_ex(int dtype, const void* x, void* col, int B, int H, int W, int C, int k, int s, int pad_top, int pad_left, int Ho,
                 int Wo, int Kp, hipStream_t stream);
/* Stem of the token pyramid without the staging pass and the im2col matrix: Conv(3 -> 16, k3, s2, AutoPadding) + folded BN + act
 * (modules/topformer.py:1023-1033) straight from the image.  x: fp32 (in_dtype 0) or compute-dtype image with element strides
 * (sn, sc, sh, sw) -- NCHW or channels-last; wp: [16][Kp] packed weight of ss_pack_conv_weight (columns (ky, kx, c)) in dtype;
 * y: [B*Ho*Wo][16] in dtype.  Inputs are rounded to dtype on entry, as ss_nchw_to_nhwc would have stored them. */
int ss_stem_conv3x3s2(int dtype, int in_dtype, const void* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const void* wp, int Kp,
                      const float* bias, void* y, int B, int Cin, int H, int W, int pad_top, int pad_left, int Ho, int Wo, int act,
                      hipStream_t stream);
int ss_avgpool(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, int win, hipStream_t stream);
int ss_act(int dtype, const void* x, void* y, int64_t n, int act, hipStream_t stream);
int ss_inject_fuse(int dtype, const void* local, int64_t ldl, const void* gact, int64_t lda, const void* gemb, int64_t lde, void* out,
                   int64_t ldo, int B, int hl, int wl, int hg, int wg, int C, hipStream_t stream);
int ss_bilinear_add(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int hi, int wi, int ho, int wo, int C,
                    hipStream_t stream);
/* InjectionDotSum of the finest stage (modules/topformer.py:640-673) and the SimpleHead's UniRes + Sum over the other stages
 * (modules/heads.py:60-106,120-170) in one pass: out = local * bilinear(gact) + bilinear(gemb) + sum_i bilinear(extra_i), fp32 until the
 * one rounding.  extra / extra_ld / extra_h / extra_w: HOST arrays of n_extra (0..3) device pointers, row strides and map sizes. */
int ss_inject_fuse_sum(int dtype, const void* local, int64_t ldl, const void* gact, int64_t lda, const void* gemb, int64_t lde,
                       int n_extra, const void* const* extra, const int64_t* extra_ld, const int* extra_h, const int* extra_w, void* out,
                       int64_t ldo, int B, int hl, int wl, int hg, int wg, int C, hipStream_t stream);

/* ---- TopFormer training step (train.py:45-52 trains TopFormerModel('B', head refiner); modules/topformer.py in train mode):
 * the backward-side kernels the inference path has no use for.
 * ss_dwconvk_dgrad: data gradient of ss_dwconvk_fwd for stride 1 / 2 and the asymmetric AutoPadding (modules/layers.py:146-164;
 *   InvertedResidual conv_dw, modules/topformer.py:166-184): g [B*Ho*Wo][C] -> dx [B*H*W][C]; wkkc as the forward's.  C % 8 == 0, k in {1, 3, 5, 7}.
 * ss_dwconvk_wgrad_s: the weight / bias gradient for the same geometry, dw_db = [(k*k + 1)][C] fp32 (taps tap-major, then the bias
 *   gradient); k in {3, 5}, C % 8 == 0; workspace >= 512 * (k*k+1) * C floats.
 * ss_avgpool_bwd: gradient of ss_avgpool (PyramidPoolAgg, modules/topformer.py:527-552); g may be a channel slice (row pitch ldg).
 * ss_act_bwd: dx = dy * act'(.) from the activation's OUTPUT y (2 ReLU, 3 ReLU6 -- open interval like torch's hardtanh backward --,
 *   5 sigmoid, 0 copy).  ss_mul: out = a * b (InjectionDotSum's product, modules/topformer.py:640-673, and its two gradients). */
int ss_dwconvk_dgrad(int dtype, const void* g, const float* wkkc, void* dx, int B, int H, int W, int C, int k, int stride, int pad_top,
                     int pad_left, int Ho, int Wo, hipStream_t stream);
int ss_dwconvk_wgrad_s(int dtype, const void* g, const void* x, float* dw_db, int B, int H, int W, int C, int k, int stride, int pad_top,
                       int pad_left, int Ho, int Wo, float* workspace, int64_t ws_bytes, hipStream_t stream);
int ss_avgpool_bwd(int dtype, const void* g, int64_t ldg, void* dx, int64_t ldx, int B, int H, int W, int C, int win, hipStream_t stream);
int ss_act_bwd(int dtype, const void* dy, const void* y, void* dx, int64_t n, int act, hipStream_t stream);
int ss_mul(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t stream);
/* InjectionDotSum backward in one pass over the fine map (modules/topformer.py:640-673; out = local * up(gact) + up(gemb)):
 * dlocal = g * up(gact), dgact = up^T(g * local), dgemb = up^T(g), up = the bilinear resize of ss_bilinear_fwd by an even integer factor.
 * g, local, dlocal dense [B*hl*wl][C]; gact / dgact / dgemb [B*hg*wg][C] with row pitches.  ss_inject_bwd_ok says whether the geometry is
 * taken (else: ss_bilinear_fwd + ss_mul + ss_bilinear_bwd).  workspace >= B (hg+1) (wg+1) 8 C floats. */
int ss_inject_bwd_ok(int hl, int wl, int hg, int wg, int C);
int ss_inject_bwd(int dtype, const void* g, const void* local, const void* gact, int64_t lda, void* dlocal, void* dgact, int64_t ldda,
                  void* dgemb, int64_t ldde, int B, int hl, int wl, int hg, int wg, int C, float* workspace, int64_t ws_bytes,
                  hipStream_t stream);
/* RefinerHead in train mode (modules/heads.py:22-39, :173-269).
 * ss_simam_bwd: gradient of ss_simam_apply w.r.t. x (through the mean and the variance as well) and w.r.t. the learnable k
 *   (dk, one float, += when accumulate_dk); moments as ss_image_moments(x); sums: [images][2][C] fp32 scratch;
 *   workspace >= images * 1024 * 2 * C floats.
 * ss_groupnorm_fwd / _bwd: nn.GroupNorm(groups, groups*cg) with an optional sigmoid in front (pre_act 5: the 'CAN' order of the group
 *   conv, :200-222) on fp32 maps [images * rows_per_image][ncol], channel g*cg + j in column g*pitch + j, other columns padding (written
 *   as zero).  fwd: moments = ss_image_moments(x, pre_act) -> coef [images][2][ncol], stats [images][groups][2] (mean, rstd), y.
 *   bwd: dx, dgamma_dbeta [2][groups*cg]; scratch >= images * 5 * ncol floats; workspace >= images * 1024 * 2 * ncol floats. */
int ss_simam_bwd(int dtype, const void* dy, int64_t ldd, const void* x, int64_t ldx, const float* moments, const float* k, void* dx,
                 int64_t ldo, float* dk, int accumulate_dk, float* sums, int64_t rows_per_image, int C, int images, float* workspace,
                 int64_t ws_bytes, hipStream_t stream);
int ss_groupnorm_fwd(const float* x, int64_t ldx, const float* moments, int64_t rows_per_image, int images, int groups, int cg, int pitch,
                     int ncol, float eps, const float* gamma, const float* beta, int pre_act, float* coef, float* stats, float* y,
                     int64_t ldy, hipStream_t stream);
int ss_groupnorm_bwd(const float* dy, int64_t ldd, const float* x, int64_t ldx, const float* stats, const float* gamma,
                     int64_t rows_per_image, int images, int groups, int cg, int pitch, int ncol, int pre_act, float* dx, int64_t ldo,
                     float* dgamma_dbeta, float* scratch, float* workspace, int64_t ws_bytes, hipStream_t stream);

/* ---- TopFormer RefinerHead, inference (modules/heads.py:22-39 SimpleAttention, :42-60 AttentionUniRes, :173-269 RefinerHead;
 * core/models.py:52-72,99-115).  Token matrices [images * rows_per_image, C], all reductions in a fixed order.
 * ss_image_moments: moments[img][0][c] = mean over the image's rows of f(x[:, c]), [img][1][c] = sum (f(x) - mean)^2;
 *   pre_act 0: f = identity (SimAM, :29-34), 5: f = sigmoid (the 'CAN' group conv's activation feeding GroupNorm, :200-222).
 *   workspace: images * 1024 * 2 * C floats.
 * ss_simam_apply: y = x * sigmoid((x - mean)^2 / (4 (k + M2 / (rows_per_image - 1))) + 0.5), k = uni_res.attention.k (:27-39).
 * ss_refiner_fold: nn.GroupNorm(groups, groups*cg) statistics per image from the per-column moments (columns g*pitch + j, j < cg,
 *   hold channel g*cg + j; biased variance), its affine, and the fixed linear map L [lrows][groups*cg] that follows it (the
 *   view + Conv3d(n_heads -> 1) of :262-267, or the classifier after the bilinear resize, core/models.py:103-115) folded into
 *   M [images][mrows][ncol] and cvec [images][mrows] (moments are [images][2][ncol]; rows >= lrows and padding columns zero).
 * ss_image_affine: out[p][r] = sum_c M[img][r][c] f(in[p][c]) + cvec[img][r]; in fp32 [P, cin], out of out_dtype [P, mrows]. */
int ss_image_moments(int dtype, const void* x, int64_t ld, int64_t rows_per_image, int C, int images, int pre_act, float* moments,
                     float* workspace, int64_t ws_bytes, hipStream_t stream);
int ss_simam_apply(int dtype, const void* x, int64_t ldx, const float* moments, const float* k, void* y, int64_t ldy,
                   int64_t rows_per_image, int C, int images, hipStream_t stream);
/* SimAM and the grouped 1x1 conv that consumes it in one kernel (modules/heads.py:27-39 + :200-222): out[p][n] (fp32) =
 * sum_c w[n][c] * att[p][c], att = ss_simam_apply's output rounded to the compute dtype -- which is then never written.
 * ss_simam_gemm_ok: 1 for bf16, C a power of two in 64..1024 (the LDS tile), N <= 32 and a multiple of 4; else callers run ss_simam_apply + ss_gemm_nt. */
int ss_simam_gemm_ok(int dtype, int C, int N);
int ss_simam_gemm(int dtype, const void* x, int64_t ldx, const float* moments, const float* k, const void* w, int64_t ldw, float* out,
                  int64_t ldo, int64_t rows_per_image, int C, int images, int N, hipStream_t stream);
int ss_refiner_fold(const float* moments, int64_t rows_per_image, int images, int groups, int cg, int pitch, int ncol, float eps, const float* gamma,
                    const float* beta, const float* L, int lrows, int mrows, float* M, float* cvec, hipStream_t stream);
int ss_image_affine(int out_dtype, const float* in, int64_t ldi, const float* M, const float* cvec, void* out, int64_t ldo,
                    int64_t rows_per_image, int images, int cin, int mrows, int pre_act, hipStream_t stream);

/* ---- optimizer step right after the gradient all-reduce: AdaBound as configured in train.py:53-58
 * (torch_optimizer.AdaBound, absent here: restated from Luo et al. 2019, parity unpinned) over FLAT fp32 buffers -- one launch
 * for the whole model.  step_size / lower / upper are the per-step scalars of the published algorithm (host computed);
 * grad_scale multiplies g first (1/world_size when the all-reduce summed).  The betas are doubles: 1 - beta is taken in double and
 * rounded to float once, as torch does with the python scalars of exp_avg.mul_(beta).add_(grad, alpha=1 - beta). */
int ss_adabound_step(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, float eps,
                     float step_size, float lower, float upper, float weight_decay, float grad_scale, hipStream_t stream);

/* ---- gradient delivery: dst[i][0..numel[i]) += src[i] for `count` fp32 tensors in one launch per 40 tensors.  Source i is
 * numel[i] / cols[i] rows of cols[i] contiguous floats, src_ld[i] apart (cols[i] == numel[i]: contiguous); dst[i] is contiguous.
 * src, src_ld, cols, dst and numel are HOST arrays.  Replaces autograd's per-parameter AccumulateGrad adds (what DDP's reducer
 * does when it copies gradients into its buckets, train.py:306) when the gradients live in one flat buffer (engine.FlatParams). */
int ss_multi_add_f32(int count, const float* const* src, const int64_t* src_ld, const int64_t* cols, float* const* dst,
                     const int64_t* numel, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMSEG_HIP_H */
