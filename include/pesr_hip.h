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
int pesr_pack_bias_ps(const float* b, float* out, int O, void* stream);

/* ---- 3x3 conv, pad 1 (reference model/basic.py:4-7 `Conv`; ATen conv2d / convolution_backward) */
/* y = act( alpha * (conv(x, w) + bias) [masked by mask > 0] + skip ).
 * x [N][H][W][Cin], y [N][OH][OW][Cout] with OH = (H-1)/stride+1.  bias/skip/mask may be NULL.
 * ps_out=1 writes y pixel-shuffled: [N][2*OH][2*OW][Cout/4] (fuses nn.PixelShuffle(2)).
 * Fused epilogues stand in for relu_ (model/basic.py:43), .mul(res_scale) and `res += x`
 * (model/basic.py:49-50, model/pesr.py:33). */
int pesr_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                     float* y, int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act, float slope,
                     int ps_out, void* stream);

/* dx = alpha * conv_transpose(dy, w) [masked by mask > 0] + skip.  dx [N][H][W][Cin], dy [N][OH][OW][Cout].
 * w_packed_dgrad from pesr_pack_conv3x3(mode 1).  mask fuses ReLU's threshold_backward (the conv's
 * own input was a ReLU output); skip fuses the residual fan-in add.  ps_in=1: dy is the gradient of
 * the pixel-shuffled output, [N][2*OH][2*OW][Cout/4] (fuses pixel_unshuffle; stride 1 only). */
int pesr_conv3x3_dgrad(const float* dy, const float* w_packed_dgrad, const float* mask, const float* skip, float* dx,
                       int N, int H, int W, int Cin, int Cout, int stride, float alpha, int ps_in, void* stream);

/* dw[O][I][3][3] (OIHW, the parameter's own layout) = alpha * sum_pixels dy (x) x ;  db[O] = alpha * sum dy.
 * db may be NULL.  ps_in as above.  Workspace: pesr_conv3x3_wgrad_workspace_bytes. */
size_t pesr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride);
int pesr_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                       int stride, float alpha, int ps_in, void* workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PESR_HIP_H */
