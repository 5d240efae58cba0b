/*
 * kmunet_hip.h -- C ABI of libkmunet_hip.so: hand-written gfx950 (MI355X) HIP kernels
 * for the KM-UNet hot path.  This is the drop-in boundary underneath the nn.Module
 * surface.  The reference (Zhou-dot9/KM-UNet) has no FFI of its
 * own -- every op is stock ATen -- so each entry point cites the reference Python it
 * replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes only, no torch types; every pointer is DEVICE memory owned
 *     by the caller (inputs, outputs, saved-for-backward, workspaces).  The library
 *     allocates nothing, keeps no mutable global state, and never synchronises: every
 *     kernel is launched asynchronously on `stream` (a hipStream_t passed as void*), so
 *     calls are hipGraph-capturable and re-entrant from the autograd thread.
 *   - tensors are contiguous fp32 in the reference's layouts (NCHW / [B,C,L]).
 *   - return value: 0 on success, KMU_ERR_ARG (-1) for a rejected argument, otherwise the
 *     hipError_t of the failed launch; kmu_last_error() returns a thread-local message.
 *   - "ws" workspaces: query the byte size with the matching *_ws_bytes() call; contents
 *     are scratch, except where a comment says they carry forward->backward state.
 */
#ifndef KMUNET_HIP_H
#define KMUNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMU_ABI_VERSION 1
#define KMU_ERR_ARG (-1)

typedef void* kmu_stream_t; /* hipStream_t */

int kmu_version(void);
const char* kmu_last_error(void);

/* ------------------------------------------------------------------------------------
 * K1  KANConv2d 3x3 / stride 1 / pad 1   (convKAN/KANConv2Dlayers.py:15-37 +
 *     convKAN/KANlayers.py:577-660: unfold -> b_splines -> two F.linear)
 *
 * Formulated as conv3x3(Phi(x), W') with Phi(x) = [SiLU(x), B_0(x)..B_7(x)] evaluated once
 * per input element into LDS (out-of-image taps see Phi(0), not 0 -- SURVEY quirk 1) and
 * contracted on the fp32 MFMA (v_mfma_f32_16x16x4_f32), K = 81*Cin.
 *
 * knots: 12 floats = one row of KANLinear.grid [in_features,12] (KANlayers.py:526-535);
 *        all rows must be identical (they are unless update_grid() was called, which the
 *        reference never does) -- the host wrapper checks this.
 * ------------------------------------------------------------------------------------ */

/* floats needed for the forward pack (wp_fwd) and the dX pack (wp_bwd) of one layer */
size_t kmu_kan_pack_fwd_elems(int Cin, int Cout);
size_t kmu_kan_pack_bwd_elems(int Cin, int Cout);

/* base_weight [Cout,Cin*9], spline_weight [Cout,Cin*9,8], spline_scaler [Cout,Cin*9]
 * (KANlayers.py:537-545; scaled_spline_weight :644-650) -> MFMA-fragment-ordered packs.
 * wp_bwd may be NULL (inference). */
int kmu_kan_pack_weights(const float* base_weight, const float* spline_weight, const float* spline_scaler,
                         float* wp_fwd, float* wp_bwd, int Cin, int Cout, kmu_stream_t stream);

/* y[B,Cout,H,W] = KANConv2d(x[B,Cin,H,W]).  Optional fused epilogue of
 * StableHybridKANConv.forward (KM_UNetV3_SH.py:92-94): y = relu?(residual + kan(x)).
 * residual may be NULL. */
int kmu_kan_conv2d_fwd(const float* x, const float* knots, const float* wp_fwd, const float* residual, float* y,
                       int B, int Cin, int Cout, int H, int W, int relu, kmu_stream_t stream);

/* bytes of scratch for kmu_kan_conv2d_bwd_weights */
size_t kmu_kan_bwd_ws_bytes(int B, int Cin, int Cout, int H, int W);

/* dx[B,Cin,H,W] = d loss / d x given dy[B,Cout,H,W] (autograd of the lines above). */
int kmu_kan_conv2d_bwd_input(const float* x, const float* dy, const float* knots, const float* wp_bwd, float* dx,
                             int B, int Cin, int Cout, int H, int W, kmu_stream_t stream);

/* d_base_weight, d_spline_weight, d_spline_scaler (same shapes as the parameters);
 * deterministic two-stage reduction through ws. */
int kmu_kan_conv2d_bwd_weights(const float* x, const float* dy, const float* knots, const float* spline_weight,
                               const float* spline_scaler, float* d_base_weight, float* d_spline_weight,
                               float* d_spline_scaler, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int H,
                               int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2  HSM-SSD mixer  (vim_block_init/efficient_vim_init.py:33-61) with its LayerNorm1D
 *     prologue (vim_block_init/vim_utils_init.py:50-59) available as a separate call.
 *
 * x [B,C,L] with L = Hs*Hs.  N = state_dim (64 in KM-UNet, KM_UNetV3_SH.py:166).
 * w_bcdt [3N,C], w_dw [3N,9], w_hz [2C,C], w_out [C,C], D [1]   (bias-free, :21-31).
 * The parameter A is not an input: softmax_L(dt + A[n]) is shift invariant (quirk 3).
 *
 * forward:  pass 1 (tile-wise 1x1 proj -> dw3x3 -> online-softmax partials (m,s,acc[C,N]),
 *           combined across workgroups), gate stage (hz_proj, SiLU gate, out_proj on the
 *           [C,N] state), pass 2 (recompute C-part tile, y = h @ Cm).
 * The `state` buffer carries forward -> backward: kmu_hsmssd_state_elems() floats.
 * ------------------------------------------------------------------------------------ */
int kmu_layernorm1d_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_mean /*[B,L,2]*/,
                        int B, int C, int L, float eps, kmu_stream_t stream);
int kmu_layernorm1d_bwd(const float* x, const float* weight, const float* rstd_mean, const float* dy, float* dx,
                        float* d_weight_partial /*[kmu_layernorm1d_partials(B,C,L), C]*/, float* d_bias_partial, int B, int C,
                        int L, kmu_stream_t stream);
int kmu_layernorm1d_partials(int B, int C, int L); /* rows of the *_partial outputs */

size_t kmu_hsmssd_state_elems(int B, int C, int N);
size_t kmu_hsmssd_fwd_ws_bytes(int B, int C, int N, int Hs);
int kmu_hsmssd_fwd(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
                   const float* D, float* y /*[B,C,Hs,Hs]*/, float* h /*[B,C,N]*/, float* state, void* ws,
                   size_t ws_bytes, int B, int C, int N, int Hs, kmu_stream_t stream);

/* the same forward, one kernel per call: stage 0 = pass 1, 1 = gate, 2 = pass 2 (call in order, same arguments);
 * used by the host wrapper so that each kernel can be timed / profiled on its own */
int kmu_hsmssd_fwd_stage(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
                         const float* D, float* y, float* h, float* state, void* ws, size_t ws_bytes, int B, int C,
                         int N, int Hs, int stage, kmu_stream_t stream);
/* The same three stages with the 1x1 projection and the depthwise 3x3 composed into one 3x3 convolution on the bf16 matrix
 * core, split-bf16 ("bf16x3") operands (csrc/hsmssd_x3.inc; ~1e-5 relative; stage 0 also packs the composite weights into
 * the workspace, which the stage-2 call of the same forward reads back: pass the same ws to all three). */
int kmu_hsmssd_fwd_stage_x3(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
                            const float* D, float* y, float* h, float* state, void* ws, size_t ws_bytes, int B, int C, int N,
                            int Hs, int stage, kmu_stream_t stream);

size_t kmu_hsmssd_bwd_ws_bytes(int B, int C, int N, int Hs);
/* number of per-workgroup partial slabs written for d_w_bcdt / d_w_dw (caller sums over dim 0) */
int kmu_hsmssd_bwd_partials(int B, int C, int Hs);
/* number of partial rows G written for d_w_hz / d_w_out / d_D by the gate stage (B x state-column groups) */
int kmu_hsmssd_gate_partials(int B);
int kmu_hsmssd_bwd(const float* x, const float* dy, const float* dh /*may be NULL*/, const float* w_bcdt,
                   const float* w_dw, const float* w_hz, const float* w_out, const float* D, const float* state,
                   float* dx, float* d_w_bcdt_partial /*[P,3N,C]*/, float* d_w_dw_partial /*[P,3N,9]*/,
                   float* d_w_hz_partial /*[G,2C,C]*/, float* d_w_out_partial /*[G,C,C]*/,
                   float* d_D_partial /*[G]*/, void* ws, size_t ws_bytes, int B, int C, int N, int Hs,
                   kmu_stream_t stream);

/* stage 0 = pass A, 1 = gate, 2 = pass B (call in order, same arguments) */
int kmu_hsmssd_bwd_stage(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                         const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                         float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial, float* d_w_out_partial,
                         float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N, int Hs, int stage,
                         kmu_stream_t stream);
/* Backward on the bf16 matrix core with split-bf16 operands (csrc/hsmssd_x3.inc), same three stages and outputs; the
 * per-tile partial rows follow the x3 tiling: allocate d_w_bcdt_partial / d_w_dw_partial with kmu_hsmssd_bwd_partials_x3 rows
 * and the workspace with kmu_hsmssd_bwd_ws_bytes_x3 (it also holds the two packed composite-weight sets written by stage 0). */
size_t kmu_hsmssd_bwd_ws_bytes_x3(int B, int C, int N, int Hs);
int kmu_hsmssd_bwd_partials_x3(int B, int C, int Hs);
int kmu_hsmssd_bwd_stage_x3(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                            const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                            float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial, float* d_w_out_partial,
                            float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N, int Hs, int stage,
                            kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K3  DySample x2, style 'lp', 4 groups  (DySample_md.py:49-68).
 *
 * conv_out [B,32,H,W] is the raw output of the 1x1 offset conv (offset.weight/bias,
 * DySample_md.py:39,67); the kernel does "*0.25 + init_pos" (:67), the coordinate
 * normalisation + pixel_shuffle (:50-59) and F.grid_sample(bilinear, border,
 * align_corners=False) (:60-61).  Index generation follows the oracle's fp32 op order
 * exactly (compiled -ffp-contract=off): ix0/iy0 are bit-exact.
 * ix0, iy0: optional int32 [B*4, 2H, 2W] outputs (NULL to skip).
 * ------------------------------------------------------------------------------------ */
int kmu_dysample_lp_fwd(const float* x, const float* conv_out, const float* init_pos /*[32]*/, float* y,
                        int32_t* ix0, int32_t* iy0, int B, int C, int H, int W, kmu_stream_t stream);
/* dx must be zero-initialised by the caller (scatter-add); d_conv_out [B,32,H,W] is fully written. */
int kmu_dysample_lp_bwd(const float* x, const float* conv_out, const float* init_pos, const float* dy, float* dx,
                        float* d_conv_out, int B, int C, int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K4  Deformable conv 3x3 / stride 1 / pad 1 / one offset group, no mask
 *     (DAGEM_md.py:46,98-101 -> torchvision.ops.DeformConv2d, torchvision 0.14.0;
 *     third-party, restated from its published semantics: PARITY UNPINNED).
 * offset [B,18,H,W] interleaved (dy,dx) per tap; weight [Cout,Cin,3,3]; bias [Cout] or NULL.
 * ------------------------------------------------------------------------------------ */
int kmu_deform_conv2d_fwd(const float* x, const float* offset, const float* weight, const float* bias, float* y,
                          int B, int Cin, int Cout, int H, int W, kmu_stream_t stream);
/* dx, d_weight, d_bias must be zero-initialised by the caller (atomic accumulation);
 * d_offset is fully written. */
int kmu_deform_conv2d_bwd(const float* x, const float* offset, const float* weight, const float* dy, float* dx,
                          float* d_offset, float* d_weight, float* d_bias, int B, int Cin, int Cout, int H, int W,
                          kmu_stream_t stream);

/* The same operator split for an im2col + GEMM formulation (what km-unet_amd/ops.py uses: the [Cout, Cin*9] contraction
 * and its two backward GEMMs run on a BLAS): cols [B, Cin*9, H*W] = the bilinear samples; bwd = adjoint of the sampling
 * (dx must be zero on entry, scatter by atomicAdd; d_offset [B,18,H,W] written in full). */
int kmu_deform_sample_fwd(const float* x, const float* offset, float* cols, int B, int Cin, int H, int W, kmu_stream_t stream);
int kmu_deform_sample_bwd(const float* x, const float* offset, const float* dcols, float* dx, float* d_offset, int B, int Cin,
                          int H, int W, kmu_stream_t stream);
/* the same adjoint without float atomics (round 3): per sample the cell -> (tap, pixel, weight) lists of the bilinear corners are
 * built once in LDS (integer atomics) and every channel's dx is gathered from them and WRITTEN (dx need not be zeroed).
 * H*W <= 256 (the bridge's 16x16 level; kmu_deform_sample_bwd_lds_supported) -- larger maps take kmu_deform_sample_bwd. */
int kmu_deform_sample_bwd_lds_supported(int B, int Cin, int H, int W);
int kmu_deform_sample_bwd_lds(const float* x, const float* offset, const float* dcols, float* dx, float* d_offset, int B, int Cin, int H,
                              int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Depthwise 3x3 / stride 1 / pad 1 convolution: EfficientViMBlock.dwconv1/dwconv2
 * (vim_block_init/efficient_vim_init.py:74-75 -> ConvLayer2D, vim_utils_init.py:62-89, groups=dim, no bias)
 * and DirectionAttention.conv (KM_UNetV3_SH.py:222, groups=dim, bias).
 * weight [C,1,3,3] (= [C,9]); bias [C] or NULL.  Weight/bias gradients are per-block partials
 * [P,C,9] / [P,C] with P = kmu_dwconv3x3_partials(B); d_bias_partial may be NULL.
 * ------------------------------------------------------------------------------------ */
int kmu_dwconv3x3_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int C, int H, int W,
                      kmu_stream_t stream);
int kmu_dwconv3x3_bwd_data(const float* dy, const float* weight, float* dx, int B, int C, int H, int W,
                           kmu_stream_t stream);
/* dx = dwconv3x3^T(dy) + addend (same purpose as kmu_pwconv_bwd_input_add) */
int kmu_dwconv3x3_bwd_data_add(const float* dy, const float* weight, const float* addend, float* dx, int B, int C, int H, int W,
                               kmu_stream_t stream);
int kmu_dwconv3x3_partials(int B);
int kmu_dwconv3x3_bwd_weight(const float* x, const float* dy, float* d_weight_partial, float* d_bias_partial, int B,
                             int C, int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused BatchNorm2d (+ReLU) (+ sigmoid-alpha blend): the elementwise chain around every conv of
 * EfficientViMBlock (vim_block_init/efficient_vim_init.py:81-97; ConvLayer2D, vim_utils_init.py:62-89):
 *     out = x + a*(f(t) - x),  a = sigmoid(alpha[c]),  f(t) = relu?(BatchNorm2d(t))   or   f(t) = t
 * t, x, out: [B,C,HW].  gamma == NULL: no BatchNorm (the mixer blend, :90).  alpha == NULL: no blend
 * (x ignored; FFN.fc1 = conv+BN+ReLU).  training != 0: batch statistics, running_mean/var updated with
 * `momentum` (unbiased variance) and *num_batches_tracked (int64 on the device, may be NULL) incremented, as
 * nn.BatchNorm2d.  stats [C,2] = (mean, rstd) saved for backward.
 * ws: forward [C,S,2], backward [C,S,3] floats with S = kmu_bn_blend_splits(B, HW).
 * backward: dt (and dx if blending) fully written; d_gamma/d_beta/d_alpha [C] (d_alpha is w.r.t. the RAW
 * alpha, i.e. includes sigmoid'(alpha)).
 * ------------------------------------------------------------------------------------ */
int kmu_bn_blend_splits(int B, int HW);
int kmu_bn_blend_fwd(const float* t, const float* x, const float* gamma, const float* beta, const float* alpha,
                     float* running_mean, float* running_var, float momentum, float eps, int relu, int training,
                     float* out, float* stats, float* ws, long long* num_batches_tracked, int B, int C, int HW,
                     kmu_stream_t stream);
int kmu_bn_blend_bwd(const float* gout, const float* t, const float* x, const float* gamma, const float* beta,
                     const float* alpha, const float* stats, int relu, int training, float* dt, float* dx,
                     float* d_gamma, float* d_beta, float* d_alpha, float* ws, int B, int C, int HW,
                     kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DirectionAttention's local gate  attn = sigmoid(q*k)*v  (KM_UNetV3_SH.py:258-261) on the packed output
 * of the qkv 1x1 conv: qkv [B,3C,HW] -> out [B,C,HW]; backward writes dqkv [B,3C,HW] in full.  HW % 4 == 0.
 * ------------------------------------------------------------------------------------ */
int kmu_qkv_gate_fwd(const float* qkv, float* out, int B, int C, int HW, kmu_stream_t stream);
int kmu_qkv_gate_bwd(const float* qkv, const float* gout, float* dqkv, int B, int C, int HW, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pointwise (1x1) convolution, NCHW fp32: EnhancedViMBlock.ffn (KM_UNetV3_SH.py:120-124), the EfficientViM FFN's
 * two ConvLayer2D 1x1 (vim_block_init/vim_utils_init.py:62-89), DirectionAttention.qkv (KM_UNetV3_SH.py:221), the
 * 'channel' projection (:174), StableHybridKANConv.residual (:59).  x [B,Ci,P], w [Co,Ci], bias [Co] or NULL.
 *   fwd        y  = W * act(x) + bias            act_in = 0: identity, 1: exact (erf) GELU applied to x on load
 *   bwd_input  dx = (W^T * gy) . act'(x_pre)     x_pre = the forward's x, only read when act_in
 *   bwd_weight dW = sum_{b,p} gy * act(x)^T, dbias = sum_{b,p} gy (dbias may be NULL); deterministic two-stage
 *              reduction through ws (kmu_pwconv_bwd_weight_ws_bytes).
 * Ci, Co positive multiples of 16 and <= 256; P = H*W a multiple of 64.
 * ------------------------------------------------------------------------------------ */
int kmu_pwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int P, int act_in,
                   kmu_stream_t stream);
int kmu_pwconv_bwd_input(const float* gy, const float* w, const float* x_pre, float* dx, int B, int Ci, int Co, int P,
                         int act_in, kmu_stream_t stream);
/* dx = W^T * gy + addend  (addend [B,Ci,P]: the gradient that reaches the same tensor along another path, e.g. the blend
 * partner of EfficientViMBlock's FFN stage -- saves autograd's separate fan-in add) */
/* dx = W^T gy + row_add[b][ci] * row_mul -- the input gradient of a 1x1 conv whose input x ALSO fed a global spatial mean
 * (DirectionAttention.forward, KM_UNetV3_SH.py:231 and :258): row_add = d mean [B,Ci], row_mul = 1 / (H*W). */
int kmu_pwconv_bwd_input_rowadd(const float* gy, const float* w, const float* row_add, float row_mul, float* dx, int B, int Ci, int Co,
                                int P, kmu_stream_t stream);
int kmu_pwconv_bwd_input_add(const float* gy, const float* w, const float* addend, float* dx, int B, int Ci, int Co, int P,
                             kmu_stream_t stream);
size_t kmu_pwconv_bwd_weight_ws_bytes(int B, int Ci, int Co, int P);
int kmu_pwconv_bwd_weight(const float* x, const float* gy, float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int Ci,
                          int Co, int P, int act_in, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DirectionAttention's  conv(attn) * gate  (KM_UNetV3_SH.py:262-263): depthwise 3x3 with a per-(b,c) plane scale folded
 * into the epilogue, y = scale[b,c] * (dwconv3x3(x) + bias).  Backward: dx = scaled_bwd_data(dy); run
 * kmu_dwconv3x3_bwd_weight on the UNSCALED dy, then scaled_finish turns its partials into d_weight [C,9], d_bias [C]
 * and d_scale [B,C].
 * ------------------------------------------------------------------------------------ */
int kmu_dwconv3x3_scaled_fwd(const float* x, const float* weight, const float* bias, const float* scale, float* y, int B, int C,
                             int H, int W, kmu_stream_t stream);
int kmu_dwconv3x3_scaled_bwd_data(const float* dy, const float* weight, const float* scale, float* dx, int B, int C, int H, int W,
                                  kmu_stream_t stream);
/* DirectionAttention's local gate folded into its stencil (KM_UNetV3_SH.py:258-263; round 3):
 *     out = s[b,c] * (dwconv3x3(sigmoid(q k) v) + bias),   q, k, v = the three C-channel chunks of qkv [B,3C,H,W],  W % 4 == 0.
 * The attn tensor is formed on the fly at the stencil's taps (never stored).  bwd: dqkv [B,3C,H,W] and the weight / bias partials
 * [kmu_dwconv3x3_partials(B)][C][9] / [..][C] that kmu_dwconv3x3_scaled_finish turns into d weight, d bias and d s. */
int kmu_qkv_dw_scaled_supported(int B, int C, int H, int W);      /* W % 4 == 0 and the per-workgroup attn tile fits 64 KB of LDS */
int kmu_qkv_dw_scaled_fwd(const float* qkv, const float* weight, const float* bias, const float* scale, float* out, int B, int C, int H,
                          int W, kmu_stream_t stream);
int kmu_qkv_dw_scaled_bwd(const float* dy, const float* qkv, const float* weight, const float* scale, float* dqkv, float* d_weight_partial,
                          float* d_bias_partial, int B, int C, int H, int W, kmu_stream_t stream);
/* kmu_dwconv3x3_scaled_bwd_data and kmu_dwconv3x3_bwd_weight (with bias partials) in one launch, W % 4 == 0 (KM_UNetV3_SH.py:262-263) */
int kmu_dwconv3x3_scaled_bwd_all(const float* dy, const float* x, const float* weight, const float* scale, float* dx,
                                 float* d_weight_partial, float* d_bias_partial, int B, int C, int H, int W, kmu_stream_t stream);
int kmu_dwconv3x3_scaled_finish(const float* d_weight_partial, const float* d_bias_partial, const float* scale,
                                const float* weight, const float* bias, float* d_weight, float* d_bias, float* d_scale, int B,
                                int C, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Separable 11-tap window filter of HybridLoss's SSIM term (train_shanghai.py:298-325; torchmetrics SSIM defaults:
 * gaussian 11x11, sigma 1.5) over N planes [H,W]: adjoint = 0 -> 'valid' filter, out [N,H-10,W-10];
 * adjoint = 1 -> its transpose (input gradient), out [N,H+10,W+10].  taps: 11 floats on the device.
 * ------------------------------------------------------------------------------------ */
int kmu_gauss11_filter(const float* in, const float* taps, float* out, int N, int H, int W, int adjoint, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * EnhancedViMBlock's branch fusion + DropPath + residual (KM_UNetV3_SH.py:141-146):
 *   out[b] = x[b] + s[b] * (g[b,0] f0[b] + g[b,1] f1[b] + g[b,2] f2[b]),   g [B,3], s [B] or NULL (= 1).
 * bwd: d_f_i = s g_i dy (dx == dy: nothing to compute); d_g_partial [kmu_mix3_blocks(n)][B*3] per-block partial dot
 * products, to be column-summed (kmu_colsum_multi) into d_g [B,3].  n_per_sample = C*H*W, a multiple of 4.
 * ------------------------------------------------------------------------------------ */
int kmu_mix3_blocks(int n_per_sample);
int kmu_mix3_fwd(const float* x, const float* f0, const float* f1, const float* f2, const float* g, const float* s, float* out,
                 int B, int n_per_sample, kmu_stream_t stream);
int kmu_mix3_bwd(const float* dy, const float* f0, const float* f1, const float* f2, const float* g, const float* s, float* d_f0,
                 float* d_f1, float* d_f2, float* d_g_partial, int B, int n_per_sample, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The squeeze-excite pools (nn.AdaptiveAvgPool2d(1): KM_UNetV3_SH.py:111 fusion_gate on cat(f0,f1,f2), :231 DirectionAttention,
 * :320 ChannelAttention, :342 LocalContrastAttention):  pooled[b][t*C + c] = mean_hw f_t[b][c][:]  for n_tensors (1..3)
 * tensors [B,C,HW] in one launch (pooling commutes with the channel concat).  Deterministic.
 * With it the fusion gate + branch mix form one autograd node whose backward is
 *   kmu_mix3_bwd_dg   : d_g_partial only (as kmu_mix3_bwd without the three stores)
 *   kmu_gate_mlp_bwd  : d_pooled [B,3C]
 *   kmu_mix3_bwd_apply: d_f_t[b][c][:] = s[b] g[b,t] dy[b][c][:] + d_pooled[b][t*C + c] / HW      (HW a multiple of 4)
 * ------------------------------------------------------------------------------------ */
int kmu_mean_rows(const float* f0, const float* f1, const float* f2, float* pooled, int B, int C, int HW, int n_tensors,
                  kmu_stream_t stream);
int kmu_mix3_bwd_dg(const float* dy, const float* f0, const float* f1, const float* f2, const float* g, const float* s,
                    float* d_g_partial, int B, int n_per_sample, kmu_stream_t stream);
int kmu_mix3_bwd_apply(const float* dy, const float* g, const float* s, const float* d_pooled, float* d_f0, float* d_f1, float* d_f2,
                       int B, int C, int HW, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Tap stacking for DirectionViM's (3,1) / (1,3) projections (KM_UNetV3_SH.py:170-172): x [B,C,H,W] ->
 * out [B,3C,H,W], out[3*c + t](p) = x[c](p + (t-1) e_axis) (zero outside), axis 0 = H, 1 = W; a 3-tap conv along
 * that axis is then kmu_pwconv_* with Ci = 3C and weight W'[co, 3*ci + t] = W[co, ci, t] -- the convolution's own
 * [Co,Ci,3,1] / [Co,Ci,1,3] weight read as [Co, 3 Ci], no permuted copy.  bwd: dx = sum of the three shifted gradient slices.
 * ------------------------------------------------------------------------------------ */
int kmu_shift3_fwd(const float* x, float* out, int B, int C, int H, int W, int axis, kmu_stream_t stream);
int kmu_shift3_bwd(const float* gout, float* dx, int B, int C, int H, int W, int axis, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * HybridLoss of the training loop (train_shanghai.py:298-325) on N = B*C planes [H,W] (H, W >= 12), fused around
 * kmu_gauss11_filter.  stats: 8 floats on the device {sum (P-T)^2, sum (P-T)^2 e^{2T}, min P, max P, min T, max T,
 * sum SSIM map, loss}; part: 6 * kmu_hybrid_loss_blocks(N,H,W) floats of scratch.
 * Forward, in order:  stats (fills stats[0..5])  ->  stack [5][N][H+10][W+10] = {p, t, p^2, t^2, pt} of the min-max
 * normalised, reflect-padded inputs  ->  kmu_gauss11_filter(stack -> filt [5][N][H][W], adjoint 0)  ->  combine
 * (SSIM map over the border-cropped interior, stats[6], loss -> stats[7]).
 * Backward: grad_maps (gmaps [3][N][H][W] = d loss / d {mu_p, E_pp, E_pt}; gout = upstream scalar on the device) ->
 * kmu_gauss11_filter(gmaps -> q [3][N][H+10][W+10], adjoint 1) -> grad_input (d loss / d pred, incl. the MSE terms;
 * the extrema are constants, the target gets no gradient).
 * ------------------------------------------------------------------------------------ */
int kmu_hybrid_loss_blocks(int N, int H, int W);
int kmu_hybrid_loss_stats(const float* pred, const float* target, float* part, float* stats, int N, int H, int W,
                          kmu_stream_t stream);
int kmu_hybrid_loss_stack(const float* pred, const float* target, const float* stats, float* stack, int N, int H, int W,
                          kmu_stream_t stream);
int kmu_hybrid_loss_combine(const float* filt, float* part, float* stats, int N, int H, int W, float alpha, kmu_stream_t stream);
int kmu_hybrid_loss_grad_maps(const float* filt, const float* gout, float* gmaps, int N, int H, int W, float alpha,
                              kmu_stream_t stream);
int kmu_hybrid_loss_grad_input(const float* pred, const float* target, const float* stats, const float* q, const float* gout,
                               float* dpred, int N, int H, int W, float alpha, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * IntelligentWaveletPoolingModule up to its fusion conv (WPL/iwp.py:124-130; Haar DWT_2D iwp.py:47-113):
 * x [B,C,H,W] (H, W even) -> out [B,Ct,H/2,W/2] = cat[LL, mean over the 3C channels of cat[LH,HL,HH], zeros]; Ct >= C+1
 * (channel padding so that fusion_conv can use kmu_pwconv_* with zero weight columns; Ct = C+1 for the exact layout).  The
 * Softmax2d attention over one channel (iwp.py:127) is identically 1, its conv receives an exactly-zero gradient.
 * bwd: gout [B,Ct,H/2,W/2] -> dx [B,C,H,W], written in full (padding channels ignored).
 * ------------------------------------------------------------------------------------ */
int kmu_iwp_front_fwd(const float* x, float* out, int B, int C, int Ct, int H, int W, kmu_stream_t stream);
int kmu_iwp_front_bwd(const float* gout, float* dx, int B, int C, int Ct, int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Squeeze-excite gates on pooled vectors:  g = act2(W2 . act1(W1 . p + b1) + b2),  p [B,I], W1 [H,I], W2 [O,H].
 * DirectionAttention.fc (KM_UNetV3_SH.py:231-236), EnhancedViMBlock.fusion_gate (:111-117, on the pooled means),
 * ChannelAttention.fc (:320-325), LocalContrastAttention.fc (:342-347).  act1: 0 GELU (erf), 1 SiLU, 2 ReLU;
 * act2: 0 sigmoid, 1 softmax over O.  fwd also returns z1 = W1 p + b1 [B,H] for the backward; bwd writes final
 * (batch-summed) dw1 [H,I], db1 [H], dw2 [O,H], db2 [O] and dp [B,I] (b1/b2/db1/db2/dp may be NULL).
 * One workgroup; B*(I+2H+O)*4 bytes must fit 144 KB of LDS.
 * ------------------------------------------------------------------------------------ */
int kmu_gate_mlp_fwd(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1, float* g,
                     int B, int I, int H, int O, int act1, int act2, kmu_stream_t stream);
int kmu_gate_mlp_bwd(const float* p, const float* w1, const float* w2, const float* z1, const float* g, const float* dg,
                     float* dp, float* dw1, float* db1, float* dw2, float* db2, int B, int I, int H, int O, int act1, int act2,
                     kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Second stage of the two-stage (deterministic) parameter-gradient reductions: column sums of up to 8 partial
 * arrays src[k] = [rows[k]][cols[k]] -> dst[k] = [cols[k]] in one launch.  srcs/dsts/rows/cols are HOST arrays of
 * length n holding device pointers / sizes.  (No reference counterpart: autograd's SumBackward does this.)
 * ------------------------------------------------------------------------------------ */
/* dsts[i][c] = sum over (b, pixel) of srcs[i][b][c][pixel], srcs[i] = [B[i], C[i], HW[i]] contiguous, up to 32 tensors per launch: the
 * bias gradients of the plain nn.Conv2d layers (KM_UNetV3_SH.py:300-306, :375, :430-446; DAGEM_md.py:43) in the backward's tail. */
int kmu_bias_sum_multi(int n, const float* const* srcs, float* const* dsts, const int* B, const int* C, const int* HW, kmu_stream_t stream);
/* out = y > 0 ? dy : 0 over numel floats: the gradient through the ReLU behind StableHybridKANConv's residual add
 * (KM_UNetV3_SH.py:91-94), one launch (ATen: gt + mul). */
int kmu_relu_mask(const float* dy, const float* y, float* out, long long numel, kmu_stream_t stream);
/* out = ((a + b) + c) + d over numel floats (c, d may be NULL): the gradient fan-in of a tensor with up to four consumers in one
 * launch -- EnhancedViMBlock's x feeds the three direction branches and the residual (KM_UNetV3_SH.py:141-146); e1 / e2 feed the
 * next encoder stage and both MultiScaleFusion pyramids (:487-509).  Autograd itself accumulates pairwise, one launch per extra use. */
int kmu_add_n(const float* a, const float* b, const float* c, const float* d, float* out, long long numel, kmu_stream_t stream);
/* Gather copy of `count` contiguous fp32 tensors: dsts[i][0..numel[i]) = srcs[i][...] (host arrays of device pointers; the triples
 * ride in the kernel arguments, 160 per launch).  Used for the parameter gradients -> flat gradient bucket copy that closes the
 * backward pass (DataParallel; train_shanghai.py:175-176 has loss.backward() fill .grad in place). */
int kmu_copy_multi(int count, const float* const* srcs, float* const* dsts, const long long* numel, kmu_stream_t stream);
int kmu_colsum_multi(int n, const float* const* srcs, float* const* dsts, const int* rows, const int* cols,
                     kmu_stream_t stream);
/* the same with a row layout per array: strides[k] >= cols[k] elements between two partial rows (NULL = packed: column sums of a column
 * range of a wider partial array); inner[k] / ostrides[k] (NULL = one run): rows in runs of inner[k], run i starting i * ostrides[k]
 * elements in (the rows of one weight group of a grouped launch's [samples / G][G][tiles] partials) */
int kmu_colsum_multi_strided(int n, const float* const* srcs, float* const* dsts, const int* rows, const int* cols, const int* strides,
                             const int* inner, const long long* ostrides, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm(G, C) over [B,C,HW]: StableHybridKANConv.pre_norm (KM_UNetV3_SH.py:57,73), TripleNorm.norm_h/w
 * (:271-273), MultiScaleFusion (:294), KM_UNetV3.output_norm (:448,516).  stats [B,G,2] = (mean, rstd);
 * ws: [B*C*S*2] floats, S = kmu_group_norm_splits(HW); d_gamma/d_beta leave as [B,C] partials.
 * ------------------------------------------------------------------------------------ */
int kmu_group_norm_splits(int HW);
int kmu_group_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* ws, int B,
                       int C, int G, int HW, float eps, kmu_stream_t stream);
int kmu_group_norm_bwd(const float* x, const float* gout, const float* gamma, const float* stats, float* dx,
                       float* d_gamma_partial, float* d_beta_partial, float* ws, int B, int C, int G, int HW,
                       kmu_stream_t stream);
/* the same with an activation folded in: act = 1: y = SiLU(GroupNorm(x)) (MultiScaleFusion's conv -> GroupNorm -> SiLU blocks,
 * KM_UNetV3_SH.py:300-306), act = 2: y = sigmoid(GroupNorm(x)) (the output head, :516-517); the backward multiplies gout by act'(u),
 * u re-derived from x, gamma, beta and the saved statistics */
int kmu_group_norm_act_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* ws, int B, int C, int G,
                           int HW, float eps, int act, kmu_stream_t stream);
int kmu_group_norm_act_bwd(const float* x, const float* gout, const float* gamma, const float* beta, const float* stats, float* dx,
                           float* d_gamma_partial, float* d_beta_partial, float* ws, int B, int C, int G, int HW, int act,
                           kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Split-bf16 ("bf16x3") matrix-core variants of the 3x3 convolutions (csrc/conv3x3_x3.hip): every fp32 operand is split
 * v = hi + lo into two bf16 and each product accumulated in fp32 as lo.hi + hi.lo + hi.hi on v_mfma_f32_16x16x32_bf16
 * (~2^-16 relative product error; 5.3x the ceiling of the exact-fp32 MFMA used by kmu_kan_conv2d_fwd).
 *   kmu_kan_conv2d_fwd_x3   convKAN/KANConv2Dlayers.py:15-37 + KANlayers.py:577-660, same contract as kmu_kan_conv2d_fwd
 *                           (residual / ReLU epilogue); needs a (near-)uniform knot vector (|U[k] - (U[0] + k h)| < h/4:
 *                           the layer's own grid, KANlayers.py:526-535) and Cin % 4 == 0
 *   kmu_conv3x3_fwd_x3      nn.Conv2d(Cin, Cout, 3, padding=1) (+ bias): KM_UNetV3_SH.py:375 (conv_f), :430-446 (dec2[1],
 *                           dec3[1], dec3[3]), :300-306 (MultiScaleFusion), DAGEM_md.py:43 (offset_conv); any Cin / Cout
 * Packed weights: kmu_conv3x3_x3_pack_elems(kan, Cin, Cout) bf16 elements (2 bytes each), written by the pack calls from
 * the layer's parameters (KAN: base_weight [Cout,9Cin], spline_weight [Cout,9Cin,8], spline_scaler [Cout,9Cin]; plain: weight
 * [Cout,Cin,3,3]); repack whenever the parameters change.
 * ------------------------------------------------------------------------------------ */
size_t kmu_conv3x3_x3_pack_elems(int kan, int Cin, int Cout);
int kmu_kan_pack_weights_x3(const float* base_weight, const float* spline_weight, const float* spline_scaler, void* wp,
                            int Cin, int Cout, kmu_stream_t stream);
int kmu_conv3x3_pack_weights_x3(const float* weight, void* wp, int Cin, int Cout, kmu_stream_t stream);
/* input gradient of the same conv (autograd of KM_UNetV3_SH.py:375,430-446,...): dx = conv3x3(dy, W2) with W2[c][o][tap] =
 * W[o][c][8 - tap]; pack W2 with this call (kmu_conv3x3_x3_pack_elems(0, Cout, Cin) elements), then run
 * kmu_conv3x3_fwd_x3(dy, wp, NULL, dx, B, Cout, Cin, H, W). */
int kmu_conv3x3_pack_weights_dgrad_x3(const float* weight, void* wp, int Cin, int Cout, kmu_stream_t stream);
int kmu_kan_conv2d_fwd_x3(const float* x, const float* knots, const void* wp, const float* residual, float* y, int B, int Cin,
                          int Cout, int H, int W, int relu, kmu_stream_t stream);
int kmu_conv3x3_fwd_x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                       kmu_stream_t stream);
/* K1 input gradient on the matrix core (autograd of KANConv2Dlayers.py:15-37 w.r.t. x; same contract as
 * kmu_kan_conv2d_bwd_input, dy already masked by the ReLU): G = conv3x3(dy, flipped W') per basis, dX = sum_j dPhi_j(x) G_j.
 * wpd: kmu_kan_dgrad_x3_pack_elems(Cin, Cout) bf16 elements written by kmu_kan_pack_weights_dgrad_x3. */
size_t kmu_kan_dgrad_x3_pack_elems(int Cin, int Cout);
int kmu_kan_pack_weights_dgrad_x3(const float* base_weight, const float* spline_weight, const float* spline_scaler, void* wpd,
                                  int Cin, int Cout, kmu_stream_t stream);
int kmu_kan_conv2d_bwd_input_x3(const float* x, const float* dy, const float* knots, const void* wpd, float* dx, int B, int Cin,
                                int Cout, int H, int W, kmu_stream_t stream);
/* The same three plain-conv calls for K x K / stride 1 / padding K/2, K in {3, 5, 7} (MultiScaleFusion's 5x5 and 7x7 convs,
 * KM_UNetV3_SH.py:300-306): weight [Cout][Cin][K][K]; dgrad = 1 packs the flipped / transposed weights whose "forward" on dy is
 * the input gradient (pack size and forward call then take (Cout, Cin) swapped). */
size_t kmu_conv2d_x3_pack_elems(int Cin, int Cout, int ksize);
int kmu_conv2d_pack_weights_x3(const float* weight, void* wp, int Cin, int Cout, int ksize, int dgrad, kmu_stream_t stream);
int kmu_conv2d_fwd_x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int ksize,
                      kmu_stream_t stream);
size_t kmu_conv2d_x3_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int ksize);
int kmu_conv2d_bwd_weight_x3(const float* x, const float* dy, float* d_weight, void* ws, size_t ws_bytes, int B, int Cin, int Cout,
                             int H, int W, int ksize, kmu_stream_t stream);

/* Weight gradients on the matrix core (autograd of the same layers w.r.t. their parameters): dW'[o][f][tap] = sum_pix
 * dY[o][pix] F[pix + tap - 1][f], both operands read transposed from (hi, lo) LDS images; ws: kmu_conv3x3_x3_wgrad_ws_bytes.
 * K1: same outputs as kmu_kan_conv2d_bwd_weights (final gradients of base_weight / spline_weight / spline_scaler).
 * Plain conv: d_weight [Cout][Cin][3][3] (the bias gradient is a plain sum over dy, left to the caller). */
size_t kmu_conv3x3_x3_wgrad_ws_bytes(int kan, int B, int Cin, int Cout, int H, int W);
int kmu_kan_conv2d_bwd_weights_x3(const float* x, const float* dy, const float* knots, const float* spline_weight,
                                  const float* spline_scaler, float* d_base_weight, float* d_spline_weight,
                                  float* d_spline_scaler, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int H, int W,
                                  kmu_stream_t stream);
int kmu_conv3x3_bwd_weight_x3(const float* x, const float* dy, float* d_weight, void* ws, size_t ws_bytes, int B, int Cin, int Cout,
                              int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True) between the encoder pyramid levels
 * (KM_UNetV3_SH.py:487-492, 503-507) and its exact adjoint in gather form (deterministic, no atomics).
 * ------------------------------------------------------------------------------------ */
int kmu_resize_bilinear_ac_fwd(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, kmu_stream_t stream);
int kmu_resize_bilinear_ac_bwd(const float* gy, float* dx, int B, int C, int Hi, int Wi, int Ho, int Wo, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Contingency counts of the reference evaluator (metrics.py:45-47 float2int: clip(x,0,1)*scale as uint16; :105-114
 * _cal_frame; :220-288 pools TP/FN/FP/TN over all frames before forming CSI / POD / FAR / HSS): one pass over
 * pred / target [n] (16-byte aligned), counts [n_thresholds][3] = {TP, FN, FP} as 64-bit integers ADDED to the
 * caller-zeroed buffer (integer atomics: exact and deterministic); TN = n - TP - FN - FP.  thresholds: HOST array.
 * ------------------------------------------------------------------------------------ */
int kmu_contingency_counts(const float* pred, const float* target, unsigned long long* counts, size_t n,
                           const int* thresholds, int n_thresholds, float scale, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Convolution + the statistics pass of the BatchNorm behind it (ConvLayer2D = conv + BatchNorm2d, vim_utils_init.py:62-89, in
 * EfficientViMBlock's dwconv1 / dwconv2 / FFN): the conv kernel leaves per-workgroup (sum, sum of squares) of its output,
 * stat_part [C][S][2] with S = kmu_*_stats_partials(...) (0 = shape not covered: use the plain entry points), and
 * kmu_bn_blend_fwd_pre is train-mode kmu_bn_blend_fwd reading those partials -- one launch and one read of the tensor less.
 * ------------------------------------------------------------------------------------ */
int kmu_dwconv3x3_stats_partials(int B, int C, int H, int W);
int kmu_dwconv3x3_fwd_stats(const float* x, const float* weight, const float* bias, float* y, float* stat_part, int B, int C, int H, int W,
                            kmu_stream_t stream);
int kmu_pwconv_stats_partials(int B, int P);
int kmu_pwconv_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* stat_part, int B, int Ci, int Co, int P,
                         int act_in, kmu_stream_t stream);
int kmu_bn_blend_fwd_pre(const float* t, const float* x, const float* gamma, const float* beta, const float* alpha, float* running_mean,
                         float* running_var, float momentum, float eps, int relu, float* out, float* stats, const float* stat_part,
                         int S_part, long long* num_batches_tracked, int B, int C, int HW, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * kmu_pwconv_bwd_weight in two halves, for callers that collect many weight gradients (a train step's weight-gradient tail is
 * bound by its launch count): the slab pass of one gradient into its workspace (kmu_pwconv_bwd_weight_ws_bytes) -- or of up to 8
 * gradients of identical dimensions in one launch (_partial_multi) -- and the slab reduction of up to 32 of them in one launch.
 * kmu_colsum_multi likewise takes up to 64 arrays.
 * ------------------------------------------------------------------------------------ */
int kmu_pwconv_bwd_weight_partial(const float* x, const float* gy, void* ws, size_t ws_bytes, int with_bias, int B, int Ci, int Co, int P,
                                  int act_in, kmu_stream_t stream);
int kmu_pwconv_bwd_weight_partial_multi(int n, const float* const* x, const float* const* gy, void* const* ws, size_t ws_bytes,
                                        int with_bias, int B, int Ci, int Co, int P, int act_in, kmu_stream_t stream);
int kmu_pwconv_bwd_weight_reduce_multi(int n, const void* const* ws, float* const* dw, float* const* dbias, const int* B, const int* Ci,
                                       const int* Co, const int* P, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * LocalContrastAttention's output (KM_UNetV3_SH.py:366-368): y = x (1 - g) + g, g [B,C] (= torch.lerp(x, ones, g[:,:,None,None]));
 * bwd: dx = dy (1 - g), dg[b,c] = sum_hw dy (1 - x) in one pass.
 * ------------------------------------------------------------------------------------ */
int kmu_lca_fwd(const float* x, const float* g, float* y, int B, int C, int HW, kmu_stream_t stream);
int kmu_lca_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, int B, int C, int HW, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DAGEM's edge features (DAGEM_md.py:56-62): edge[b,c,h,w,k] = x[b,c,h,w] * x[b,c,nbr_k(h,w)], nbr = (h-1,w), (h+1,w), (h,w-1), (h,w+1)
 * cyclic (= torch.roll by +1 / -1 along H / W, stacked, times x); edge [B,C,H,W,4].  bwd: dx from d_edge (gather form, no atomics).
 * ------------------------------------------------------------------------------------ */
int kmu_dagem_edges_fwd(const float* x, float* edge, int B, int C, int H, int W, kmu_stream_t stream);
int kmu_dagem_edges_bwd(const float* x, const float* d_edge, float* dx, int B, int C, int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DAGEM (DAGEM_md.py:56-111) without its deformable convolution, one launch per BatchNorm boundary (csrc/dagem_fused.hip).
 *   forward  stage 0  edges (:56-62), a_pre = edge_aggregation_func[0](edge rows) (:65), u_pre = edge_update_func[0]([x | edge] rows) (:74-81)
 *            stage 1  agg = ReLU(BN(a_pre)); v_pre = vertex_update_func[0]([x | agg] rows) (:68-72); r_pre = update_edge_reduce_func[0](ReLU(BN(u_pre))) (:82)
 *            stage 2  z = final_aggregation_layer[0]([dconv + x | ReLU(BN(v_pre)) * ReLU(BN(r_pre))]) (:85-103), dconv = deform_conv(x, offset)
 *            stage 3  out = ReLU(BN(z)) (:104)
 *   backward stages 4..8 (given g_out): BN sums of the final layer; dz -> g_dd (gradient of dconv, and of the residual x), d W_f partials;
 *            dv_pre / dr_pre -> d W_v, d w_r partials; da_pre / du_pre -> d w_a, d W_e partials, de; dx = edge adjoint + the other pieces.
 * BatchNorm order in the five-element arrays: 0 edge_aggregation_func[1], 1 vertex_update_func[1], 2 edge_update_func[1],
 * 3 update_edge_reduce_func[1], 4 final_aggregation_layer[1].  training != 0: batch statistics; running statistics and the batch counter
 * are updated by stages 1..3 (momentum must be a number).  Weights in their nn.Module layouts: wa / wr [1,4], wv / we [C/2, 2C],
 * wf [C, C + C/2]; biases ba / br [1], bv / be [C/2].  Saved activations (caller-allocated, forward -> backward): a_pre [B,C,P],
 * u_pre [B,C/2,P,4], v_pre / r_pre [B,C/2,P], z [B,C,P], bnstat [5,C,2] (mean, rstd), P = H W.  part / part_bwd:
 * kmu_dagem_part_floats() floats each (per-workgroup BatchNorm partial sums).  Optional outputs (may be NULL): agg_out [B,C,P],
 * u_out [B,C/2,P,4], vert_out / ue_out [B,C/2,P] -- the post-ReLU activations (parity tooling reads the branch masks off them).
 * Backward scratch: g_dd, ga, dxb [B,C,P]; gv, gr, dr_pre [B,C/2,P]; de [B,C,P,4].  Parameter gradients: d_gamma / d_beta final;
 * p_* = partial rows, kmu_dagem_tiles(B,H,W) of them, to be column-summed (kmu_colsum_multi): p_wf [.,C,C+C/2], p_wv / p_we [.,C/2,2C],
 * p_bv / p_be [.,C/2], p_wa / p_wr [.,5] = (d weight[0..3], d bias).  C in {32, 64} (kmu_dagem_supported).
 * ------------------------------------------------------------------------------------ */
typedef struct kmu_dagem_args {
    int B, C, H, W, training;
    float eps[5], momentum[5];
    const float *x, *dconv;
    const float *wa, *ba, *wv, *bv, *we, *be, *wr, *br, *wf;
    const float *gamma[5], *beta[5];
    float *running_mean[5], *running_var[5];
    long long* num_batches_tracked[5];
    float *a_pre, *u_pre, *v_pre, *r_pre, *z, *out, *bnstat, *part, *part_bwd;
    float *agg_out, *u_out, *vert_out, *ue_out;
    const float* g_out;
    float *g_dd, *gv, *gr, *ga, *dr_pre, *dxb, *de, *dx;
    float *d_gamma[5], *d_beta[5];
    float *p_wf, *p_wv, *p_bv, *p_we, *p_be, *p_wa, *p_wr;
} kmu_dagem_args;
size_t kmu_dagem_args_bytes(void); /* sizeof(kmu_dagem_args) as the library was built: a binding checks its own layout against it */
int kmu_dagem_supported(int C);
int kmu_dagem_tiles(int B, int H, int W);
size_t kmu_dagem_part_floats(int B, int C, int H, int W);
int kmu_dagem_stage(const kmu_dagem_args* args, int stage, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * TripleNorm (KM_UNetV3_SH.py:266-284): y = (GroupNorm_h(x) + GroupNorm_w(x) + LayerNorm_c(x)) / 3 on x [B,C,HW], C in {16,32,64}.
 * norm_h / norm_w = nn.GroupNorm(1, C) share their statistics (one group: invariant under the H/W transpose of the 'height' branch);
 * norm_c = nn.LayerNorm(C) on the channels-last view = per-pixel statistics over C.
 *   fwd: stats [B,2] = (mean, rstd) of each sample for the backward; ws: B*C*kmu_triple_norm_splits(HW)*2 floats
 *   bwd: dx = addend (or 0 when NULL) + d loss / d x;  d_gsum_partial [B,C] = per-sample d(gh) = d(gw),
 *        d_bsum_partial [B,C] = per-sample d(bh) = d(bw) = d(bc), d_gc_partial [kmu_triple_norm_partials(B,C,HW), C]: column sums
 *        (kmu_colsum_multi) give the parameter gradients; ws as in fwd
 * kmu_pwconv_fwd_res / kmu_pwconv_bwd_input_s: the 1x1 convs of EnhancedViMBlock's FFN tail with the residual and DropPath's
 * per-sample factor folded in (:147-150): y = addend + s[b] (W act(x) + bias);  dx = s[b] (W^T gy) act'(x_pre); s may be NULL.
 * ------------------------------------------------------------------------------------ */
int kmu_triple_norm_supported(int C, int HW);
int kmu_triple_norm_splits(int HW);
int kmu_triple_norm_partials(int B, int C, int HW);
int kmu_triple_norm_fwd(const float* x, const float* gh, const float* bh, const float* gw, const float* bw, const float* gc,
                        const float* bc, float* y, float* stats, float* ws, int B, int C, int HW, float eps_gn, float eps_ln,
                        kmu_stream_t stream);
int kmu_triple_norm_bwd(const float* x, const float* dy, const float* gh, const float* gw, const float* gc, const float* stats,
                        const float* addend, float* dx, float* d_gsum_partial, float* d_bsum_partial, float* d_gc_partial, float* ws,
                        int B, int C, int HW, float eps_ln, kmu_stream_t stream);
int kmu_pwconv_fwd_res(const float* x, const float* w, const float* bias, const float* addend, const float* bscale, float* y, int B,
                       int Ci, int Co, int P, int act_in, kmu_stream_t stream);
int kmu_pwconv_bwd_input_s(const float* gy, const float* w, const float* x_pre, const float* bscale, float* dx, int B, int Ci, int Co,
                           int P, int act_in, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Grouped variants.  EnhancedViMBlock (KM_UNetV3_SH.py:97-151) runs three DirectionViM branches (height / width / channel,
 * :154-212) that differ only in their first projection: afterwards each applies the SAME layer sequence (EfficientViMBlock,
 * efficient_vim_init.py:64-97, then DirectionAttention, :215-263) to a tensor of the same shape with its own weights.
 * Stacked along the channel axis -- x [B, G*C, H, W], parameters [G, ...] -- every layer is one launch instead of G.
 * Per-channel layers (depthwise conv, BatchNorm blend, qkv gate, pooling) run unchanged on G*C channels; the entry points
 * below are the layers that mix channels.  In all of them sample index b of a [B*G, C, ...] view uses parameter set b % G.
 *   kmu_pwconv_*_g            block-diagonal 1x1 conv: x [B, G*Ci, P] -> y [B, G*Co, P], w [G*Co, Ci], bias [G*Co];
 *                             bwd_input_g: optional GELU' (x_pre) and addend epilogues; bwd_weight_g: ONE group per call
 *   kmu_layernorm1d_*_g       LayerNorm1D over C of [B (= samples*G), C, L], weight / bias [G, C]; the partial rows of
 *                             sample b belong to group b % G
 *   kmu_hsmssd_*_stage_x3_g   HSMSSD (bf16x3 kernels) with w_bcdt [G,3N,C], w_dw [G,3N,9], w_hz [G,2C,C], w_out [G,C,C], D [G];
 *                             workspace sizes from the *_ws_bytes*_g functions (G weight packs)
 *   kmu_gate_mlp_*_g          G independent gate MLPs on p [B, G, I] with stacked weights; one workgroup per group
 *   kmu_mix3_*_stacked        the branch mix on the channel slices of ONE tensor F [B, 3C, H, W] (and its gradient dF)
 * ------------------------------------------------------------------------------------ */
int kmu_pwconv_fwd_g(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int P, int act_in, int groups,
                     kmu_stream_t stream);
int kmu_pwconv_bwd_input_g(const float* gy, const float* w, const float* x_pre, const float* addend, float* dx, int B, int Ci, int Co,
                           int P, int act_in, int groups, kmu_stream_t stream);
int kmu_pwconv_bwd_weight_g(const float* x, const float* gy, float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int Ci, int Co,
                            int P, int act_in, int groups, int g, kmu_stream_t stream);
int kmu_layernorm1d_fwd_g(const float* x, const float* weight, const float* bias, float* y, float* rstd_mean, int B, int C, int L,
                          float eps, int groups, kmu_stream_t stream);
int kmu_layernorm1d_bwd_g(const float* x, const float* weight, const float* rstd_mean, const float* dy, float* dx,
                          float* d_weight_partial, float* d_bias_partial, int B, int C, int L, int groups, kmu_stream_t stream);
/* kmu_layernorm1d_bwd_g with dx += addend [B,C,L] (NULL: none): the gradient of a second consumer of the normalised tensor --
 * EfficientViMBlock blends the mixer's output with the x it normalised (efficient_vim_init.py:88-90) -- folded into the kernel. */
int kmu_layernorm1d_bwd_add(const float* x, const float* weight, const float* rstd_mean, const float* dy, const float* addend, float* dx,
                            float* d_weight_partial, float* d_bias_partial, int B, int C, int L, int groups, kmu_stream_t stream);
size_t kmu_hsmssd_fwd_ws_bytes_g(int B, int C, int N, int Hs, int groups);
int kmu_hsmssd_fwd_stage_x3_g(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
                              const float* D, float* y, float* h, float* state, void* ws, size_t ws_bytes, int B, int C, int N, int Hs,
                              int stage, int groups, kmu_stream_t stream);
size_t kmu_hsmssd_bwd_ws_bytes_x3_g(int B, int C, int N, int Hs, int groups);
int kmu_hsmssd_bwd_stage_x3_g(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                              const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                              float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial, float* d_w_out_partial,
                              float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups,
                              kmu_stream_t stream);
/* ------------------------------------------------------------------------------------
 * Weight packs once per training step (round 3).  The split-bf16 kernels read their weights from fragment-ordered (hi, lo) packs;
 * until now every forward and every backward call packed its own (63 launches per step, each the head of a dependent chain).
 * A caller that knows the step boundary (km_unet_amd.ops.PackCache, used by train.TrainStep) keeps one persistent pack buffer per
 * weight, lists them in a job table (64-byte records built on the HOST by kmu_conv_pack_job / kmu_hsm_pack_job, then copied to
 * device memory once) and re-packs ALL of them with one launch per table at the head of the step; the consumers below take the pack
 * as an argument.  Same packs as kmu_kan_pack_weights_x3 / kmu_kan_pack_weights_dgrad_x3 / kmu_conv2d_pack_weights_x3 and the pack
 * stage of kmu_hsmssd_*_stage_x3 (KANConv2Dlayers.py:15-37, efficient_vim_init.py:39-45: pure re-layouts of the parameters).
 * which: 0 KAN forward, 1 KAN input gradient, 2 plain conv forward, 3 plain conv input gradient (w1, w2 NULL for 2 and 3). */
size_t kmu_pack_job_bytes(void);
int kmu_conv_pack_job(void* host_table, int index, int which, const float* w0, const float* w1, const float* w2, void* wp, int Cin,
                      int Cout, int ksize);
int kmu_conv_pack_multi(const void* device_table, int njobs, kmu_stream_t stream);
size_t kmu_hsmssd_pack_elems(int C, int groups);      /* bf16 elements */
int kmu_hsmssd_pack_x3(const float* w_bcdt, const float* w_dw, void* wpk, int C, int groups, kmu_stream_t stream);
int kmu_hsm_pack_job(void* host_table, int index, const float* w_bcdt, const float* w_dw, void* wpk, int C, int groups);
int kmu_hsm_pack_multi(const void* device_table, int njobs, kmu_stream_t stream);
/* ------------------------------------------------------------------------------------
 * K2 forward, round-4 form: LayerNorm1D (vim_utils_init.py:50-59) + HSMSSD.forward (efficient_vim_init.py:33-61) in TWO launches
 * (csrc/hsmssd_v2.inc).
 *   stage 0  pass 1: LayerNorm on load, B / dt rows (1x1 projection on the bf16 matrix core, depthwise 3x3 in registers), online
 *            softmax partials per 4H x 16 token tile; the workgroup that draws the LAST atomic ticket of a sample combines the
 *            sample's tiles in tile order, applies hz_proj / SiLU gate / out_proj (:52-55), writes `state` (layout of
 *            kmu_hsmssd_fwd: [M | S | hpre | hz | h2], read by kmu_hsmssd_bwd*), h, and the per-sample dense 3x3 weights
 *            M_b = h2 . Wc of stage 1 into ws.  No workgroup waits for another one.
 *   stage 1  pass 2: y = conv3x3(LayerNorm(x); M_b)  (== h2 . Cm, :57-59); optional outputs xn [B,C,L] (the normalised x: the
 *            `x` operand of kmu_mixer_bwd_stage / kmu_hsmssd_bwd*) and rstd_mean [B,L,2] (kmu_layernorm1d_bwd*).
 * ln_weight / ln_bias [groups, C] (both NULL: no LayerNorm, x is the mixer input itself, xn / rstd_mean must be NULL).
 * wpk: kmu_hsmssd_pack_x3 / kmu_hsm_pack_multi output (kmu_hsmssd_pack_elems elements).  groups as kmu_hsmssd_*_g.
 * tickets: B zero-initialised 32-bit words in device memory that no other in-flight launch uses; the kernel leaves them zero.
 * ------------------------------------------------------------------------------------ */
size_t kmu_mixer_fwd_ws_bytes(int B, int C, int N, int Hs);
int kmu_mixer_fwd_stage(const float* x, const float* ln_weight, const float* ln_bias, float eps, const float* w_dw, const float* w_hz,
                        const float* w_out, const float* D, const void* wpk, float* y, float* h, float* state, float* xn,
                        float* rstd_mean, void* ws, size_t ws_bytes, unsigned int* tickets, int B, int C, int N, int Hs, int stage,
                        int groups, kmu_stream_t stream);
/* The K2 backward behind kmu_mixer_fwd_stage (csrc/hsmssd_bwdc.inc), efficient_vim_init.py:33-61 differentiated with the C rows
 * (:57-59, y = h2 . Cm) taken as the per-sample dense 3x3 convolution y = conv3x3(x; M_b) the forward ran:
 *   stage 0  G_b[c][tap][c'] = sum_l dy[c][l] x[c'][l + tap - 1]   (split-bf16 matrix core, K = tokens; replaces pass A)
 *   stage 1  dh2 = G . (w_dw (x) W_C), partial rows of d W_C [N, C] and d w_dw[C rows] [N, 9]: kmu_mixer_bwd_partials(B, C) rows each
 *   stage 2  the gate stage (kmu_hsmssd_bwd_stage's stage 1)
 *   stage 3  pass B on the {B, dt} rows + dx of the C rows (conv3x3^T(dy; M_b), M_b read from `state` as the forward left it);
 *            the C-row sections of d_w_bcdt_partial / d_w_dw_partial ([.][N..2N)) are NOT written.
 * x = the mixer's input (the NORMALISED tensor kmu_mixer_fwd_stage returns in xn when it folds a LayerNorm); `state` must come from
 * kmu_mixer_fwd_stage.  Partial-row counts as kmu_hsmssd_bwd_stage_x3.  wpk = the weight pack (kmu_hsmssd_pack_x3 / kmu_hsm_pack_multi) or
 * NULL: with it stage 3 may run its contractions on the bf16 matrix core (csrc/hsmssd_bwdb.inc; kmu_mixer_debug_passb forces the choice). */
size_t kmu_mixer_bwd_ws_bytes(int B, int C, int N, int Hs);
int kmu_mixer_bwd_partials(int B, int C);
int kmu_mixer_bwd_stage(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw, const float* w_hz,
                        const float* w_out, const float* D, const float* state, float* dx, float* d_w_bcdt_partial, float* d_w_dw_partial,
                        float* d_w_hz_partial, float* d_w_out_partial, float* d_D_partial, float* d_wc_partial, float* d_dwc_partial, void* ws,
                        size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups, const void* wpk, kmu_stream_t stream);
void kmu_mixer_debug_passb(int mode);
void kmu_mixer_debug_rows(int rows); /* tools only: force pass 1's configuration (H | 16: 8 waves | 32: wide tiles; 0 = automatic) */
void kmu_conv_debug_split(int mode); /* tools only: K1 / KxK forward at small images: 0 automatic, 1 never split Cout tiles over workgroups, 2 always */

/* kmu_hsmssd_{fwd,bwd}_stage_x3_g with the pack handed in (wpk NULL: pack inside stage 0 as before) */
int kmu_hsmssd_fwd_stage_x3_pk(const float* x, const float* w_bcdt, const float* w_dw, const float* w_hz, const float* w_out,
                               const float* D, float* y, float* h, float* state, void* ws, size_t ws_bytes, int B, int C, int N, int Hs,
                               int stage, int groups, const void* wpk, kmu_stream_t stream);
int kmu_hsmssd_bwd_stage_x3_pk(const float* x, const float* dy, const float* dh, const float* w_bcdt, const float* w_dw,
                               const float* w_hz, const float* w_out, const float* D, const float* state, float* dx,
                               float* d_w_bcdt_partial, float* d_w_dw_partial, float* d_w_hz_partial, float* d_w_out_partial,
                               float* d_D_partial, void* ws, size_t ws_bytes, int B, int C, int N, int Hs, int stage, int groups,
                               const void* wpk, kmu_stream_t stream);
int kmu_gate_mlp_fwd_g(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1, float* g, int B,
                       int I, int H, int O, int act1, int act2, int groups, kmu_stream_t stream);
int kmu_gate_mlp_bwd_g(const float* p, const float* w1, const float* w2, const float* z1, const float* g, const float* dg, float* dp,
                       float* dw1, float* db1, float* dw2, float* db2, int B, int I, int H, int O, int act1, int act2, int groups,
                       kmu_stream_t stream);
int kmu_mix3_fwd_stacked(const float* x, const float* F, const float* g, const float* s, float* out, int B, int n_per_sample,
                         kmu_stream_t stream);
int kmu_mix3_bwd_dg_stacked(const float* dy, const float* F, const float* g, const float* s, float* d_g_partial, int B,
                            int n_per_sample, kmu_stream_t stream);
int kmu_mix3_bwd_apply_stacked(const float* dy, const float* g, const float* s, const float* d_pooled, float* dF, int B, int C, int HW,
                               kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * EfficientViMBlock's FFN stage as recompute kernels (round 3):
 *     out = x + sigmoid(alpha) * ( BN2( W2 relu( BN1( W1 x ) ) ) - x )
 * efficient_vim_init.py:96 (the blend), vim_utils_init.py:122-130 (FFN = fc1, fc2) and :62-89 (ConvLayer2D = bias-free 1x1 conv +
 * BatchNorm2d [+ ReLU]).  x [B,C,P] with C in {16, 32, 64}, W1 [4C,C], W2 [C,4C], P = H*W a multiple of 64 (kmu_ffn_fused_supported;
 * other shapes: kmu_pwconv_* + kmu_bn_blend_*).  Nothing 4C wide reaches memory: every pass re-derives W1 x from x on the matrix core.
 *   fwd stages: 0 statistics of W1 x (train only), 1 z2 = W2 relu(BN1(W1 x)) + its statistics, 2 out; stage -1 = all.
 *               Saved for backward: x, z2, stats1 [4C][2], stats2 [C][2] (mean, rstd).  Running statistics / num_batches_tracked are
 *               updated as nn.BatchNorm2d does (train).  h_tap (NULL in production): the hidden activation [B,4C,P], for tests.
 *   bwd stages: 0 BN2's backward sums, 1 BN1's backward sums + dW2 slabs, 2 dx + dW1 slabs; stage -1 = all.  training = 0: BatchNorm
 *               backward without the batch-mean terms (eval-mode statistics).
 *               slab_w1 [rows1][4C][C], slab_w2 [rows2][C][4C] with rows = kmu_ffn_fused_rows(.., 1 / 2): partial weight gradients,
 *               d W = their column sum (kmu_colsum_multi).
 * ------------------------------------------------------------------------------------ */
int kmu_ffn_fused_supported(int C, int hid, int P);
int kmu_ffn_fused_rows(int B, int C, int P, int which); /* which: 1 = rows of slab_w1, 2 = rows of slab_w2 */
size_t kmu_ffn_fused_fwd_ws_bytes(int B, int C, int P);
size_t kmu_ffn_fused_bwd_ws_bytes(int B, int C, int P);
int kmu_ffn_fused_fwd(const float* x, const float* w1, const float* gamma1, const float* beta1, float* running_mean1,
                      float* running_var1, long long* nbt1, float momentum1, float eps1, const float* w2, const float* gamma2,
                      const float* beta2, float* running_mean2, float* running_var2, long long* nbt2, float momentum2, float eps2,
                      const float* alpha, int training, float* z2, float* out, float* stats1, float* stats2, float* h_tap, void* ws,
                      size_t ws_bytes, int B, int C, int P, int stage, kmu_stream_t stream);
int kmu_ffn_fused_bwd(const float* g, const float* x, const float* z2, const float* w1, const float* gamma1, const float* beta1,
                      const float* stats1, const float* w2, const float* gamma2, const float* beta2, const float* stats2,
                      const float* alpha, int training, float* dx, float* d_gamma1, float* d_beta1, float* d_gamma2, float* d_beta2,
                      float* d_alpha, float* slab_w1, float* slab_w2, void* ws, size_t ws_bytes, int B, int C, int P, int stage,
                      kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * EfficientViMBlock's dwconv stage backward with the BatchNorm folded into the stencil (round 3): x + a (BN(dwconv3x3(x)) - x),
 * efficient_vim_init.py:85,93 with ConvLayer2D = conv + BatchNorm2d, vim_utils_init.py:62-89.
 *   kmu_bn_blend_bwd_partials   the reduction half of kmu_bn_blend_bwd: part [C][S][3], S = kmu_bn_blend_splits
 *   kmu_dwconv3x3_bn_bwd_data   dx = dwconv^T(dt) + (1 - a) g with dt (BatchNorm's input gradient) formed on the fly from (g, t) and the
 *                               folded partials; also d_gamma, d_beta, d_alpha and the per-channel constants cst [C][4]
 *   kmu_dwconv3x3_bn_bwd_weight d weight partials [kmu_dwconv3x3_partials][C][9] from (x, g, t, cst): dt is never materialised
 * ------------------------------------------------------------------------------------ */
int kmu_bn_blend_bwd_partials(const float* gout, const float* t, const float* x, const float* gamma, const float* beta, const float* alpha,
                              const float* stats, int relu, float* part, int B, int C, int HW, kmu_stream_t stream);
int kmu_dwconv3x3_bn_bwd_data(const float* g, const float* t, const float* weight, const float* gamma, const float* alpha, const float* stats,
                              const float* part, int S, int training, float* dx, float* d_gamma, float* d_beta, float* d_alpha, float* cst,
                              int B, int C, int H, int W, kmu_stream_t stream);
/* both of them in one launch (round 3): dx, the BatchNorm / blend parameter gradients AND the weight-gradient partials from one pass
 * over (g, t, x); results bit-identical to the two-kernel path */
int kmu_dwconv3x3_bn_bwd_all(const float* g, const float* t, const float* x, const float* weight, const float* gamma, const float* alpha,
                             const float* stats, const float* part, int S, int training, float* dx, float* d_gamma, float* d_beta,
                             float* d_alpha, float* d_weight_partial, int B, int C, int H, int W, kmu_stream_t stream);
int kmu_dwconv3x3_bn_bwd_weight(const float* x, const float* g, const float* t, const float* cst, float* d_weight_partial, int B, int C,
                                int H, int W, kmu_stream_t stream);

/* ------------------------------------------------------------------------------------
 * EnhancedViMBlock's FFN tail as recompute kernels (round 3): out = x + s[b] (W2 gelu(W0 nrm + b0) + b2) with nrm = TripleNorm(x)
 * made by kmu_triple_norm_fwd (KM_UNetV3_SH.py:120-124 ffn, :147-150; s = DropPath's per-sample factor or NULL).  One launch each
 * way, the 4C-wide hidden tensor is never stored: bwd re-derives W0 nrm, returns dn = d loss / d nrm and per-workgroup partials
 * slab_w0 [rows][4C][C], slab_w2 [rows][C][4C], rows_b [rows][4C + C] (d b0 | d b2), rows = kmu_ffn_fused_rows(B, C, P, 1).
 * Shapes: kmu_ffn_fused_supported.
 * ------------------------------------------------------------------------------------ */
int kmu_tail_ffn_fwd(const float* nrm, const float* x, const float* w0, const float* b0, const float* w2, const float* b2,
                     const float* sdp, float* out, int B, int C, int P, kmu_stream_t stream);
int kmu_tail_ffn_bwd(const float* nrm, const float* g, const float* w0, const float* b0, const float* w2, const float* sdp, float* dn,
                     float* slab_w0, float* slab_w2, float* rows_b, int B, int C, int P, kmu_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KMUNET_HIP_H */
