/*
 * pesr_hip.h - C ABI of libpesr_hip.so: the MI355X (gfx950) kernels behind PESR's x4 SR GAN train step.
 *
 * The reference (thangvubk/PESR) has no FFI of its own: every kernel it runs is a PyTorch built-in
 * reached from model/basic.py, model/pesr.py, model/vgg.py, model/focal_loss.py and train.py.  Each
 * entry point below therefore names the ATen op / reference line it stands in for.  The Python side
 * (pesr_amd/_lib.py) binds these with ctypes; see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - all tensors are fp32 device pointers owned by the caller (torch's caching allocator);
 *   - activations are NHWC ("channels_last" physical layout of a logical NCHW tensor);
 *   - 3x3 conv weights are passed PRE-PACKED (pesr_pack_conv3x3) - the nn.Module keeps OIHW;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and
 *     returns 0 on success, a positive hipError_t, or a negative PESR_E* code; nothing throws;
 *   - workspaces are caller-allocated; their byte sizes come from the *_workspace_bytes helpers.
 */
#ifndef PESR_HIP_H
#define PESR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PESR_OK 0
#define PESR_EINVAL (-1)
#define PESR_EWORKSPACE (-2)

#define PESR_ACT_NONE 0
#define PESR_ACT_RELU 1
#define PESR_ACT_LRELU 2

int pesr_abi_version(void);

/* ---- weight packing (build-owned layout transform; no reference counterpart) --------------- */
/* mode 0: forward packing [9][I/16][O][16]; mode 1: dgrad packing [9][O/16][I][16].
 * ps=1: output channels ordered sub-pixel-major for a conv feeding nn.PixelShuffle(2)
 * (reference model/basic.py:56-59). w is OIHW [O][I][3][3]. */
int pesr_pack_conv3x3(const float* w, float* out, int O, int I, int mode, int ps, void* stream);
/* many packs in one launch: desc is a DEVICE array of `count` rows of 8 int64 {w ptr, out ptr, O, I, mode, ps, R, Nn}
 * with R = ceil16(reduction channels), Nn = 16 if n-channels <= 16 else ceil64 (the sizes pesr_pack_conv3x3 derives itself);
 * modes 2 / 3: the F(2,3) Winograd packing of mode 0 / 1 (pesr_pack_conv3x3_wino), modes 4 / 5: the F(4,3) one */
int pesr_pack_conv3x3_batched(const long long* desc, int count, void* stream);
int pesr_pack_bias_ps(const float* b, float* out, int O, void* stream);

/* ---- 3x3 conv, pad 1 (reference model/basic.py:4-7 `Conv`; ATen conv2d / convolution_backward) */
/* y = act( alpha * (conv(x, w) + bias) [masked by mask > 0] + skip ).
 * x [N][H][W][Cin], y [N][OH][OW][Cout] with OH = (H-1)/stride+1.  bias/skip/mask may be NULL.
 * ps_out=1 writes y pixel-shuffled: [N][2*OH][2*OW][Cout/4] (fuses nn.PixelShuffle(2)).
 * Fused epilogues stand in for relu_ (model/basic.py:43), .mul(res_scale) and `res += x`
 * (model/basic.py:49-50, model/pesr.py:33). */
/* workspace (may be NULL / 0): scratch for split-K on layers too small to fill 256 CUs; size from
 * pesr_conv3x3_workspace_bytes(N, OH, OW, Cout) (0 for large layers). */
size_t pesr_conv3x3_workspace_bytes(int N, int OH, int OW, int Cout);
int pesr_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                     float* y, int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act, float slope,
                     int ps_out, void* workspace, size_t ws_bytes, void* stream);

/* dx = alpha * conv_transpose(dy, w) [masked by mask > 0] + skip.  dx [N][H][W][Cin], dy [N][OH][OW][Cout].
 * w_packed_dgrad from pesr_pack_conv3x3(mode 1).  mask fuses ReLU's threshold_backward (the conv's
 * own input was a ReLU output); skip fuses the residual fan-in add.  ps_in=1: dy is the gradient of
 * the pixel-shuffled output, [N][2*OH][2*OW][Cout/4] (fuses pixel_unshuffle; stride 1 only). */
int pesr_conv3x3_dgrad(const float* dy, const float* w_packed_dgrad, const float* mask, const float* skip, float* dx,
                       int N, int H, int W, int Cin, int Cout, int stride, float alpha, int ps_in, void* workspace,
                       size_t ws_bytes, void* stream);

/* dw[O][I][3][3] (OIHW, the parameter's own layout) = alpha * sum_pixels dy (x) x ;  db[O] = alpha * sum dy.
 * db may be NULL.  ps_in as above.  Workspace: pesr_conv3x3_wgrad_workspace_bytes (same algo).
 * algo: PESR_WGRAD_AUTO = the transposed Winograd kernel (on v_mfma_f32_32x32x2_f32; F(4,3) along x nested with F(2,3) along y: a third of
 * the multiplies) where it applies (stride 1, 64-multiple channels, width % 4 == 0 and >= 48 - or a width of 24 / 16 / 12 / 8 pixels, whose images are
 * laid side by side in the kernel's 48-pixel strips), else the F(2,3) one (even width >= 48; 2/3), else the direct kernel;
 * PESR_WGRAD_DIRECT = the direct kernel everywhere; PESR_WGRAD_WINO23 = F(2,3) where it applies, else direct.  All produce
 * the same gradient up to fp32 rounding (measured vs fp64: <= 2e-6 of the gradient's maximum); the choice is an argument,
 * never process state.
 * accumulate = 1: dw (and db) are ADDED TO instead of overwritten - the second contribution of a layer that is used twice in one
 * backward pass (the Discriminator sees hr and sr in one graph, reference train.py:205-214), written straight into the
 * gradient buffer the first use filled; stands in for autograd's fan-in add_.  (The F(2,3) form has no such mode: an
 * accumulating call runs on the F(4,3) or the direct kernel.) */
#define PESR_WGRAD_AUTO 0
#define PESR_WGRAD_DIRECT 1
#define PESR_WGRAD_WINO23 2
#define PESR_WGRAD_WINO4_16X16 3 /* as AUTO, but the F(4,3) kernel in round 2's v_mfma_f32_16x16x4_f32 form (8 waves) instead of the
                                  * 32x32x2 form AUTO uses since round 3: same transform, same results up to rounding (cross-checks, A/B) */
#define PESR_WGRAD_WINO4_1D 4    /* (ABI 13) as AUTO, but with round 3's 1-D F(4,3) transform (half the multiplies) on the 12-wave 32x32x2 kernel (cross-checks, A/B) */
#define PESR_WGRAD_WINO4_12W 5   /* (ABI 16) as AUTO, but on round 4's 12-wave kernel (every wave stages and multiplies) instead of the 16-wave one with
                                  * four producer waves AUTO uses since round 5: same transform, same order of additions, bit-identical results */
size_t pesr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride, int algo);
int pesr_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                       int stride, float alpha, int ps_in, int algo, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* Stride-1 3x3 conv (pad 1) with a 1-D Winograd F(2,3) transform along x: 2/3 of the multiplies of pesr_conv3x3_fwd, same
 * tensors and fused epilogue (y = act(alpha * (conv + bias) [masked] + skip)), for even W, Cin % 16 == 0, Cout % 128 == 0
 * (pesr_conv3x3_wino_supported).  Stands in for the same ATen conv2d / convolution_backward(input) calls of the reference
 * `Conv` (model/basic.py:4-7).  w_packed: 12 * Cin * Cout floats from pesr_pack_conv3x3_wino (mode 0: forward weights from
 * OIHW [Cout][Cin][3][3]; mode 1: the input-gradient weights - then call with Cin/Cout of the gradient problem swapped).
 * ps / ps_out / ps_in: as for pesr_pack_conv3x3 / pesr_conv3x3_fwd / pesr_conv3x3_dgrad (fused PixelShuffle store, fused
 * pixel-unshuffle load).  workspace (optional, pesr_conv3x3_workspace_bytes): lets layers with few tiles split K. */
int pesr_conv3x3_wino_supported(int N, int H, int W, int Cin, int Cout);
int pesr_pack_conv3x3_wino(const float* w, float* w_packed, int Cout, int Cin, int mode, int ps, void* stream);
int pesr_conv3x3_wino(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask, float* y,
                      int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out, int ps_in,
                      void* workspace, size_t ws_bytes, void* stream);

/* Stride-1 3x3 conv (pad 1) with a 1-D Winograd F(4,3) transform along x (interpolation points 0, +-1, +-2, inf): HALF of
 * the multiplies of pesr_conv3x3_fwd, same tensors and fused epilogue as pesr_conv3x3_wino, for W % 4 == 0, Cin % 16 == 0,
 * Cout % 64 == 0 (Cout % 256 == 0 with ps_out).  Same ATen calls of the reference `Conv` (model/basic.py:4-7).
 * Measured relative error vs fp64: 1.2e-6 .. 1.7e-6 of the output maximum at 256 input channels (direct kernel: 3e-7).
 * w_packed: 18 * Cin * Cout floats from pesr_pack_conv3x3_wino4 (mode 0 forward / mode 1 input gradient, as above).
 * pesr_conv3x3_wino4_score: per-mille of the kernel's 576-pixel tiles that lies inside the image, or 0 if the shape is not
 * supported or gives fewer than 192 workgroups; allow_split = 1 when the split-K workspace (pesr_conv3x3_workspace_bytes) is passed. */
int pesr_conv3x3_wino4_score(int N, int H, int W, int Cin, int Cout, int allow_split);
int pesr_pack_conv3x3_wino4(const float* w, float* w_packed, int Cout, int Cin, int mode, int ps, void* stream);
int pesr_conv3x3_wino4(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask, float* y,
                       int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out, int ps_in,
                       void* workspace, size_t ws_bytes, void* stream);
/* ---- OPTIONAL bf16-operand mode (never the default) ---------------------------------------------------------------
 * Stride-1 3x3 conv (pad 1) on v_mfma_f32_16x16x32_bf16: same tensors (fp32 NHWC in HBM), same fused epilogue and PixelShuffle
 * options as pesr_conv3x3_wino4, but BOTH operands of every product are rounded to bf16 (round to nearest even) and summed in
 * fp32.  Not the reference's arithmetic (model/basic.py:4-7 runs nn.Conv2d in fp32): checked against its own oracle
 * (oracle/ops.py conv3x3_bf16: the same rounding, float64 sums) with its own tolerance.  Cin % 32 == 0, Cout % 64 == 0
 * (with ps_out: Cout a multiple of four workgroup widths - 256 / 128 / 64 channels for Cout % 256 / 128 / 64 == 0; Cin % 128 == 0 with ps_in).
 * w_packed: 9 * Cin * Cout bf16 (2 bytes each) from pesr_pack_conv3x3_bf16 (mode 0 forward / mode 1 input gradient - then call
 * with Cin / Cout of the gradient problem swapped).  pesr_conv3x3_bf16_score: per-mille of the kernel's 144-pixel tiles inside
 * the image, 0 for unsupported shapes or fewer than min_wgs workgroups (the host-side dispatch passes 64). */
int pesr_conv3x3_bf16_score(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_pack_conv3x3_bf16(const float* w, void* w_packed, int Cout, int Cin, int mode, int ps, void* stream);
int pesr_conv3x3_bf16(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask, float* y,
                      int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out, int ps_in,
                      void* stream);
/* ---- OPTIONAL split-bf16 mode (SURVEY 8 f4 "bf16/split-bf16 MFMA"; never the default, never the headline row; ABI 12) --------
 * The same stride-1 3x3 conv with every fp32 operand written as hi + lo (hi = bf16(v), lo = bf16(v - hi)) and every product
 * replaced by a_hi*b_lo + a_lo*b_hi + a_hi*b_hi on v_mfma_f32_16x16x32_bf16, summed in fp32: 3.6 .. 4.7e-6 of the output maximum
 * against an fp64 conv (profiles/r04_split_bf16_numerics.txt) - inside the tolerances the fp32 kernels are held to, so its oracle
 * is the reference's fp32 arithmetic itself (model/basic.py:4-7), not a restatement of the mode.  Cin % 32 == 0, Cout % 128 == 0
 * (ps_out: Cout a multiple of four workgroup widths, 256 or 128 channels; ps_in: Cin % 128 == 0).  w_packed: 2 * 9 * Cin * Cout
 * bf16 from pesr_pack_conv3x3_bf16x3 (hi plane, then lo plane; mode 0 forward / mode 1 input gradient).  Weight gradients stay on
 * the fp32 kernels.  _score: per-mille of the 144-pixel tiles inside the image, 0 if unsupported or fewer than min_wgs workgroups. */
int pesr_conv3x3_bf16x3_score(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_pack_conv3x3_bf16x3(const float* w, void* w_packed, int Cout, int Cin, int mode, int ps, void* stream);
int pesr_conv3x3_bf16x3(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask, float* y,
                        int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out, int ps_in, void* stream);

/* Stride-2 forward in the bf16 mode (the Discriminator's down-sampling convs, reference model/pesr.py:56-64: Conv(k=3, stride=2,
 * padding=1)): x [N][H][W][Cin] -> y [N][(H-1)/2+1][(W-1)/2+1][Cout], skip / mask shaped like y; weights packed by
 * pesr_pack_conv3x3_bf16 mode 0 (the stride-1 forward's packing).  Its weight gradient stays on the fp32 kernels.
 * pesr_conv3x3_bf16_s2_score: as above, but half of min_wgs workgroups already qualify (these layers are small). */
int pesr_conv3x3_bf16_s2_score(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_conv3x3_bf16_s2(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask, float* y,
                         int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, void* stream);
/* Input gradient of the same stride-2 conv in the bf16 mode: dy [N][(H-1)/2+1][(W-1)/2+1][Cout_fwd] -> dx [N][H][W][Cin_fwd] =
 * alpha * grad [zeroed where mask <= 0] + skip (mask / skip shaped like dx, may be null); the four output parity classes are
 * four small stride-1 problems with 1 / 2 / 2 / 4 taps inside ONE launch.  w_packed: pesr_pack_conv3x3_bf16 mode 1 of the forward
 * weights (the stride-1 input gradient's packing).  Cout_fwd % 32 == 0, Cin_fwd % 64 == 0. */
int pesr_conv3x3_bf16_s2_dgrad_score(int N, int H, int W, int Cout_fwd, int Cin_fwd, int min_wgs);
int pesr_conv3x3_bf16_s2_dgrad(const float* dy, const void* w_packed, const float* mask, const float* skip, float* dx, int N, int H,
                               int W, int Cout_fwd, int Cin_fwd, float alpha, void* stream);
/* Weight / bias gradient of the same conv in the bf16 mode: dw = alpha * sum bf16(dy) * bf16(x) (fp32 sums; fixed-order split-K
 * reduce, bit-reproducible), db = alpha * sum dy (fp32, un-rounded).  Same tensors and ps_in / accumulate meaning as
 * pesr_conv3x3_wgrad.  Stride 1, W % 48 == 0, Cin % 64 == 0, Cout % 128 == 0 (Cout % 512 == 0 with ps_in); workspace_bytes
 * returns 0 for shapes it does not cover. */
size_t pesr_conv3x3_wgrad_bf16_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int pesr_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                            float alpha, int ps_in, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* Forward 3x3 conv from a 3-channel input, stride 1 (reference `embed` model/pesr.py:23, Discriminator features.0
 * model/pesr.py:53, vgg19 features.0): x [N][H][W][3], w OIHW [Cout][3][3][3] (NOT packed), y [N][H][W][Cout]. */
int pesr_conv3x3_rgb_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                         float slope, void* stream);

/* Forward 3x3 conv to 3 output channels, stride 1 (the Generator's last conv, reference model/pesr.py:38 `Conv(num_channels, 3)`
 * via model/basic.py:4-7): x [N][H][W][C] (C a multiple of 64, <= 512), w OIHW [3][C][3][3] (NOT packed), bias [3] or null,
 * y [N][H][W][3].  HBM-bound: every input row is read once per band of 12 output rows. */
int pesr_conv3x3_rgb_out_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                             float slope, void* stream);

/* Input gradient of a C -> 3 conv (stride 1): reference Upsampler's last conv (model/basic.py:60), i.e. ATen
 * convolution_backward(input) for it.  dy [N][H][W][3], w OIHW [3][C][3][3] (NOT packed), dx [N][H][W][C]. */
int pesr_conv3x3_rgb_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, void* stream);

/* Input gradient of a 3 -> C conv (stride 1): Discriminator features.0 (reference model/pesr.py:53) and vgg19 features.0
 * (model/vgg.py:8-10) on the Generator's backward path, i.e. ATen convolution_backward(input) for them.  dy [N][H][W][C]
 * (C a multiple of 64, <= 512), w = the forward conv's OIHW [C][3][3][3] (NOT packed), dx [N][H][W][3].  HBM-bound (reads dy
 * once): the kernel of pesr_conv3x3_rgb_out_fwd with a transposing weight index.  (ABI 12) */
int pesr_conv3x3_rgb_in_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, void* stream);

/* Weight gradient of the RGB-boundary convs (one operand has 3 channels): reference `embed`
 * (model/pesr.py:23), Discriminator features.0 (model/pesr.py:53), Upsampler's last conv (model/basic.py:60).
 * a: the C-channel tensor [N][H][W][C], b3: the 3-channel tensor [N][H][W][3].
 * mode 0 (3 -> C): a = dy, b3 = x, dw [C][3][3][3], db [C].  mode 1 (C -> 3): a = x, b3 = dy, dw [3][C][3][3], db [3].
 * accumulate = 1: dw is added to (db must then be NULL), as for pesr_conv3x3_wgrad. */
size_t pesr_conv3x3_wgrad_rgb_workspace_bytes(int N, int H, int W, int C);
int pesr_conv3x3_wgrad_rgb(const float* a, const float* b3, float* dw, float* db, int N, int H, int W, int C, int mode,
                           float alpha, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* ---- MeanShift: trainable 1x1 conv 3->3 (reference model/basic.py:9-17) ----------- */
/* x_nchw / y_nchw: the 3-channel tensor is stored NCHW-contiguous instead of NHWC (folds the layout
 * change of the network's RGB input/output into this kernel). */
int pesr_meanshift_fwd(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int x_nchw, int y_nchw,
                       void* stream);
/* dy, dx NHWC; dx may be NULL.  dw [3][3], db [3].  workspace >= 1024*12*4 + 128 bytes. */
int pesr_meanshift_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int N, int H, int W,
                       int x_nchw, void* workspace, size_t ws_bytes, void* stream);

/* ---- nn.PixelShuffle(2) standalone (reference model/basic.py:57,59), bit-exact indexing ------- */
/* x [N][H][W][4C] -> y [N][2H][2W][C]:  y[n][2h+i][2w+j][c] = x[n][h][w][4c+2i+j];  bwd is the inverse. */
int pesr_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int pesr_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* out = (ref > 0 ? alpha*g : slope*alpha*g) + add : ReLU (slope 0) / LeakyReLU backward (+ residual fan-in);
 * ref/add may be NULL */
int pesr_relu_mask(const float* g, const float* ref, const float* add, float* out, long n, float alpha, float slope,
                   void* stream);

/* ---- 2x2/2 max-pool (torchvision vgg19 features, reference model/vgg.py:8-10) ------------------ */
int pesr_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* relu_in=1 also applies the mask of the ReLU that produced x (gradient only where x > 0) */
int pesr_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int relu_in, void* stream);

/* ---- BatchNorm2d(train) + LeakyReLU (reference model/basic.py:29-30, model/pesr.py:47) --------- */
size_t pesr_bn_workspace_bytes(long M, int C);
/* mean_invstd [2][C] is saved for backward; running stats / num_batches (int64) updated as nn.BatchNorm2d
 * does, may be NULL.  y_nchw=1 writes y NCHW-contiguous (the flatten of reference model/pesr.py:79). */
int pesr_bn_lrelu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_invstd,
                      float* running_mean, float* running_var, long long* num_batches, int N, int H, int W, int C, float eps,
                      float momentum, float slope, int y_nchw, void* workspace, size_t ws_bytes, void* stream);
/* (ABI 14) conv 3 -> C (pad 1, stride 1, NO bias) -> BatchNorm2d(train) -> LeakyReLU in one call, for the layer where the statistics
 * pass costs most (the Discriminator's features.0, reference model/pesr.py:53 + model/basic.py:26-30: 16 x 192 x 192 x 64): the conv
 * kernel (that of pesr_conv3x3_rgb_fwd, OIHW weights [C][3][3][3]) leaves per-workgroup sums / sums of squares of its result, the
 * finalize step reduces them in a fixed order in double - no pass over z for the statistics (SURVEY K10's first half).  Writes z
 * (the conv output, saved for backward) AND y; everything else as pesr_bn_lrelu_fwd.  C % 4 == 0 and 256 % (C / 4) == 0. */
size_t pesr_conv3x3_rgb_bn_workspace_bytes(int N, int H, int W, int C);
int pesr_conv3x3_rgb_bn_lrelu_fwd(const float* x, const float* w, float* z, const float* gamma, const float* beta, float* y,
                                  float* mean_invstd, float* running_mean, float* running_var, long long* num_batches, int N, int H,
                                  int W, int C, float eps, float momentum, float slope, int y_nchw, void* workspace, size_t ws_bytes,
                                  void* stream);
/* accumulate = 1: dgamma / dbeta are added to instead of overwritten (second use of the layer in one backward pass). */
int pesr_bn_lrelu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                      float* dx, float* dgamma, float* dbeta, int N, int H, int W, int C, float slope, int dy_nchw,
                      int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* Eval-mode BatchNorm2d (+ activation): the statistics are GIVEN - mean_invstd [2][C] = {running_mean, 1/sqrt(running_var + eps)}
 * (nn.BatchNorm2d in .eval(), a constructor branch of reference model/basic.py:29 the reference's own scripts never take).
 * slope: 0.2 LeakyReLU, 0 ReLU, 1 no activation (also accepted by the training-mode entry points above).
 * bwd: dx = gamma * invstd * dz, dgamma = sum dz * xhat, dbeta = sum dz with dz = dy * act'(z); workspace as pesr_bn_workspace_bytes. */
int pesr_bn_lrelu_eval_fwd(const float* x, const float* gamma, const float* beta, const float* mean_invstd, float* y, int N, int H,
                           int W, int C, float slope, int y_nchw, void* stream);
int pesr_bn_lrelu_eval_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd, float* dx,
                           float* dgamma, float* dbeta, int N, int H, int W, int C, float slope, int dy_nchw, void* workspace,
                           size_t ws_bytes, void* stream);

/* Backward OF the training-mode BatchNorm backward - what the gradient penalty (reference train.py:216-226, autograd.grad(...,
 * create_graph=True) through D's eight BatchNorm2d layers) needs: for dz = gamma * invstd * (du - mean(du) - xhat * mean(du * xhat))
 * and g = dL/d(dz) it returns l_du = dL/d(du), l_z = dL/dz (through xhat and invstd) and l_gamma = dL/dgamma; any of the three
 * outputs may be NULL.  All tensors [N][H][W][C]; mean_invstd as saved by pesr_bn_lrelu_fwd. */
size_t pesr_bn_bwd_bwd_workspace_bytes(long M, int C);
int pesr_bn_bwd_bwd(const float* z, const float* du, const float* g, const float* gamma, const float* mean_invstd, float* l_du,
                    float* l_z, float* l_gamma, int N, int H, int W, int C, void* workspace, size_t ws_bytes, void* stream);

/* ---- skinny-batch Linear (reference model/pesr.py:69-74; ATen addmm/mm), M <= 32 per call ------- */
/* (the Python binding walks larger batches in chunks of 32 rows: pesr_amd/ops.py linear_*).
 * pesr_linear_wgrad: accumulate = 1 adds to dw / db instead of overwriting them (the second use of a layer inside one
 * backward pass - the Discriminator sees hr and sr in the same graph, reference train.py:205-214 - and batch chunks). */
size_t pesr_linear_workspace_bytes(int M, int N, long K);
int pesr_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, long K, int act, float slope,
                    void* workspace, size_t ws_bytes, void* stream);
int pesr_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, long K, void* workspace, size_t ws_bytes,
                      void* stream);
int pesr_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, long K, int accumulate, void* stream);

/* ---- losses, fused forward + gradient (reference train.py:131-140) --------------------------------- */
/* sr, hr, grad: [N][H][W][3].  out2[0] = mean|sr-hr|, out2[1] = TV sum.  grad = g_l1*sign(sr-hr) + g_tv*dTV/dsr
 * (caller folds alpha_l1/numel and alpha_tv into g_l1, g_tv); grad may be NULL.  workspace >= 8 KiB + 64 B. */
int pesr_loss_l1_tv_fwd_bwd(const float* sr, const float* hr, float* grad, float* out2, int N, int H, int W, float g_l1,
                            float g_tv, void* workspace, size_t ws_bytes, void* stream);
/* out1[0] = mean (a-b)^2 ; grad = gscale*(a-b) (caller passes 2*alpha/numel); grad may be NULL */
int pesr_mse_fwd_bwd(const float* a, const float* b, float* grad, float* out1, long n, float gscale, void* workspace,
                     size_t ws_bytes, void* stream);

/* ---- training-sample assembly on the GPU (reference data.py:79-126: _crop, _aug_data, _to_tensor) ---- */
/* pool: device uint8 HWC images back to back.  desc: B rows of 3 int64 {byte offset of the image, stride_w | y0 << 32,
 * x0 | aug << 32} (stride_w = image width in pixels; (y0, x0) crop origin; aug bit 0 hflip, bit 1 vflip, bit 2
 * transpose, applied transpose -> vflip -> hflip as the reference).  out: [B][3][P][P] fp32 (nhwc = 0) or [B][P][P][3]. */
int pesr_crop_augment(const unsigned char* pool, const long long* desc, float* out, int B, int P, int nhwc, void* stream);

/* ---- validation PSNR on the Y channel (reference utils.py:32-41 compute_PSNR), one image pair [1][3][H][W] ------- */
/* a_nhwc / b_nhwc: the tensor is stored [H][W][3] instead of [3][H][W].  out2 (device doubles): {mse, psnr dB}; all
 * arithmetic in double on integer-valued terms -> bit-identical to the reference's numpy path.  workspace >= 2 KiB. */
int pesr_psnr_y(const float* a, const float* b, double* out2, int H, int W, int a_nhwc, int b_nhwc, void* workspace,
                size_t ws_bytes, void* stream);

/* ---- GAN losses on the [B][1] logits (reference train.py:132-133,210-213,244-253; model/focal_loss.py:9-13), value and both
 * gradients in one launch.  gan_type 0 SGAN, 1 RSGAN, 2 RaSGAN (an extension: batch means over the B samples given);
 * side 0 = discriminator loss (BCE), 1 = generator loss (BCE, or the reference's FocalLoss when focal = 1, with the torch-0.4
 * gradient through its weight AND the BCE term).  out[0] = scale * loss; d_real / d_fake [B] = scale * dloss/dlogit, may be NULL.
 * B <= 1024. */
int pesr_gan_loss_fwd_bwd(const float* pred_real, const float* pred_fake, int B, int gan_type, int side, int focal, float gamma,
                          float scale, float* out, float* d_real, float* d_fake, void* stream);

/* ---- fused Adam on one flat buffer (reference train.py:124-125; torch.optim.Adam) ---------------- */
/* g is multiplied by grad_scale first (1/world_size after a sum all-reduce). step is 1-based. */
int pesr_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   int step, float grad_scale, void* stream);

/* The same Adam step with its step-dependent scalars in DEVICE memory, for hipGraph replay (a captured launch replays its kernel
 * arguments): state = 6 floats {lr, unused, lr/bc1, sqrt(bc2), step count low 32 bits, high 32 bits (bit patterns)}.  The call
 * first advances the step count and recomputes state[2..3] in double (reference train.py:124-125 / torch.optim.Adam's
 * bias_correction1/2), then applies the update.  The host writes state[0] (StepLR, reference train.py:127-128) between replays. */
int pesr_adam_step_dev(float* p, const float* g, float* m, float* v, long n, float* state, float beta1, float beta2, float eps,
                       float grad_scale, void* stream);

/* ---- convs whose epilogue leaves the BatchNorm sums (round 6, ABI 18): SURVEY K10 for the Discriminator's BasicBlocks (reference
 * model/basic.py:26-30: conv -> BatchNorm2d(train) -> LeakyReLU).  Replaces aten::native_batch_norm's statistics pass and
 * aten::native_batch_norm_backward's reduction pass: the conv kernel that WRITES a tensor also leaves, per pixel tile, the per-channel sums the
 * BatchNorm needs of it - no pass of its own over the tensor, no atomics (one row of [2][C] floats per pixel tile, added up in a fixed order
 * by pesr_bn_finalize / pesr_bn_lrelu_bwd_fused).
 *   mode PESR_BN_FWD_STATS      rows hold sum(y), sum(y^2) of the stored output y (a conv in front of a BatchNorm);
 *   mode PESR_BN_BWD_MASK_SUMS  the kernel is the input gradient that produces the gradient g of a BatchNorm + LeakyReLU OUTPUT; it stores
 *                               g' = g * lrelu'(gamma * xhat(z) + beta) instead of g and the rows hold sum(g'), sum(g' * xhat).
 * pesr_conv3x3_bn_rows(which, ...): rows the call will write - 0 when the fused form does not cover the shape (split-K layers, odd channel
 * counts): the caller then uses the un-fused calls.  which: 0 = pesr_conv3x3_fwd_bn, 1 = pesr_conv3x3_dgrad_bn, 2 = pesr_conv3x3_wino4_bn
 * (arguments as that call takes them).  Workspace sizes: as for the un-fused calls. */
#define PESR_BN_FWD_STATS 1
#define PESR_BN_BWD_MASK_SUMS 2
typedef struct PesrBnFuse {
    int mode;
    int rows;                     /* capacity of `part` in rows of 2 * C floats */
    float* part;                  /* [rows][2][C], written by the conv kernel */
    const float* z;               /* PESR_BN_BWD_MASK_SUMS: the BatchNorm's input, shape of the conv's output */
    const float* mean_invstd;     /* PESR_BN_BWD_MASK_SUMS: [2][C] saved by the forward */
    const float* gamma;
    const float* beta;
    float slope;                  /* negative slope of the activation behind the BatchNorm */
} PesrBnFuse;
long pesr_conv3x3_bn_rows(int which, int N, int H, int W, int Cin, int Cout, int stride);
int pesr_conv3x3_fwd_bn(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int stride,
                        void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream);
int pesr_conv3x3_dgrad_bn(const float* dy, const float* w_packed_dgrad, float* dx, int N, int H, int W, int Cin, int Cout, int stride,
                          void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream);
int pesr_conv3x3_wino4_bn(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                          void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream);
/* rows of sums -> mean / invstd [2][C] (+ running statistics, num_batches_tracked as nn.BatchNorm2d updates them); M = N * H * W */
int pesr_bn_finalize(const float* part, int rows, int C, long M, float eps, float momentum, float* mean_invstd, float* running_mean,
                     float* running_var, long long* num_batches, void* stream);
/* rows of (sum g', sum g' xhat) + the masked gradient g' -> dz = gamma invstd (g' - mean g' - xhat mean(g' xhat)), dgamma, dbeta (NULL: not
 * wanted; accumulate: added to).  workspace: 2 * C floats. */
int pesr_bn_lrelu_bwd_fused(const float* z, const float* g_masked, const float* part, int rows, const float* gamma, const float* beta,
                            const float* mean_invstd, float* dz, float* dgamma, float* dbeta, int N, int H, int W, int C, int accumulate,
                            void* workspace, size_t ws_bytes, void* stream);

/* ---- gradient exchange over peer memory (round 5, ABI 15): the replacement of nn.DataParallel's reduce_add (reference
 * train.py:114-118) as a reduce-scatter + all-gather that keeps no workgroup resident - stream wait / write-value operations and
 * peer copies on one stream, ONE small kernel over 1/N of the bytes (pesr_amd/csrc/peer_exchange.hip).
 * pesr_peer_alloc: hipMalloc + zero + the allocation's IPC handle (64 bytes).  pesr_peer_export: IPC handle of the allocation that
 * CONTAINS ptr, ptr's byte offset in it and the allocation's size.  pesr_peer_open / _close: map / unmap another process's
 * allocation.  pesr_peer_allreduce: SUM over the ranks, in place, of args->mine (numel % 4 == 0), enqueued on `stream`; all ranks
 * must call it in the same order with the same numel and the next `epoch` (1, 2, 3, ...). */
typedef struct PesrPeerArgs {
    int rank, world;
    unsigned epoch, pad_;
    float* mine;
    float* peer[16];              /* the same tensor in every rank's buffer as mapped in THIS process (peer[rank] == mine) */
    unsigned* my_flags;           /* [3][16] zero-initialised words of this rank (pesr_peer_alloc), written by the peers */
    unsigned* peer_flags[16];     /* every rank's flag block as mapped in this process */
    float* scratch;               /* (world - 1) * ((numel / world rounded up to a multiple of 4)) floats of this rank */
    size_t numel;
    void* ctx;                    /* pesr_peer_ctx_create(world): one side stream + event per peer, so that a phase's copies use all links at once */
} PesrPeerArgs;
int pesr_peer_ctx_create(int world, void** ctx);
int pesr_peer_ctx_destroy(void* ctx);
int pesr_peer_alloc(size_t bytes, void** ptr, unsigned char* handle64);
int pesr_peer_free(void* ptr);
int pesr_peer_release(void* my_flags, size_t bytes);   /* (ABI 17) abandon a stuck exchange: fill this rank's own flag block with 0xffffffff so that every wait of its streams runs out; the transport is unusable afterwards */
int pesr_peer_export(const void* ptr, unsigned char* handle64, size_t* offset, size_t* alloc_bytes);
int pesr_peer_open(const unsigned char* handle64, void** base);
int pesr_peer_close(void* base);
int pesr_peer_allreduce(const void* args, void* stream);
/* (ABI 18) measurement helper: which engine moves a copy out of a peer mapping?  Times hipMemcpyAsync(dst <- src, bytes) alone, then started
 * while a kernel holds every wave slot of this GPU for hog_us microseconds (100 .. 50000), and that kernel: out_ms[3] = {alone, under the
 * hog, hog}.  A blit kernel waits for the hog, a copy engine does not.  Synchronises its own two streams. */
int pesr_peer_copy_probe(const void* src, void* dst, size_t bytes, int hog_us, float* out_ms);

/* ---- generic k x k conv, odd k != 3 (reference `Conv(in, out, kernel_size, stride, bias)`, model/basic.py:4-7, accepts any
 * kernel size; its networks only use 3): padding k/2, NHWC activations, w OIHW [Cout][Cin][k][k] (not packed).  Plain VALU
 * kernels for completeness - untuned, deterministic.  fwd: y = conv(x, w) + bias (bias may be NULL); dgrad: dx [N][H][W][Cin] from
 * dy [N][OH][OW][Cout]; wgrad: dw OIHW and db (may be NULL). */
int pesr_conv_kxk_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int k,
                      int stride, void* stream);
int pesr_conv_kxk_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int k, int stride,
                        void* stream);
int pesr_conv_kxk_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout, int k, int stride,
                        void* stream);

/* ---- spectral normalisation of a conv weight (reference model/basic.py:25 `spectral_norm(Conv(...))`: an undefined name there;
 * the evident intent is torch.nn.utils.spectral_norm, whose algorithm - one power iteration per training forward - this is) ---- */
/* w [O][K] (the OIHW tensor as it lies in memory, K = Cin * 9).  update = 1 (training): v <- normalize(W^T u), u <- normalize(W v)
 * in place (x / max(||x||, eps)); always: sigma[0] = u^T W v, w_hat = w / sigma.  workspace: pesr_spectral_norm_workspace_bytes. */
size_t pesr_spectral_norm_workspace_bytes(int O, int K);
int pesr_spectral_norm_fwd(const float* w, float* u, float* v, float* w_hat, float* sigma, int O, int K, int update, float eps,
                           void* workspace, size_t ws_bytes, void* stream);
/* dw = (g - <g, w_hat> u v^T) / sigma for g = dL/d(w_hat), with the u, v, sigma of that forward (constants, as in torch);
 * accumulate = 1 adds to dw. */
int pesr_spectral_norm_bwd(const float* g, const float* w_hat, const float* u, const float* v, const float* sigma, float* dw, int O,
                           int K, int accumulate, void* workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PESR_HIP_H */
