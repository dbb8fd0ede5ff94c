// extern "C" surface of libpesr_hip.so (declared in include/pesr_hip.h).  Thin, exception-free
// wrappers over the per-family launchers; no torch types cross this boundary.
#include "common.h"
#include "launchers.h"
#include "../../include/pesr_hip.h"

static inline int pad16(int c) { return (c + 15) / 16 * 16; }
// "n" channel padding of the packed weights: <= 16 channels (the C -> 3 layers) pad to 16, everything else to a multiple of 64
static inline int pad64(int c) { return c <= 16 ? 16 : (c + 63) / 64 * 64; }

PESR_API int pesr_abi_version(void) { return 18; }

PESR_API int pesr_pack_conv3x3(const float* w, float* out, int O, int I, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_launch(w, out, O, I, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_pack_conv3x3_batched(const long long* desc, int count, void* stream) {
    return pesr_pack_conv3x3_batched_launch(desc, count, (hipStream_t)stream);
}
PESR_API int pesr_pack_bias_ps(const float* b, float* out, int O, void* stream) {
    return pesr_pack_bias_ps_launch(b, out, O, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                              float* y, int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act,
                              float slope, int ps_out, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_launch(x, w_packed, bias, skip, mask, y, N, H, W, pad16(Cin), pad64(Cout), stride, alpha, act, slope,
                               ps_out, 0, 0, Cin, Cout, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_dgrad(const float* dy, const float* w_packed_dgrad, const float* mask, const float* skip, float* dx,
                                int N, int H, int W, int Cin, int Cout, int stride, float alpha, int ps_in, void* workspace,
                                size_t ws_bytes, void* stream) {
    if (stride == 1)  // a stride-1 conv over dy with Cin/Cout swapped and the taps flipped
        return pesr_conv3x3_launch(dy, w_packed_dgrad, nullptr, skip, mask, dx, N, H, W, pad16(Cout), pad64(Cin), 1, alpha,
                                   PESR_ACT_NONE, 0.f, 0, ps_in, 1, Cout, Cin, workspace, ws_bytes, (hipStream_t)stream);
    if (stride == 2 && !ps_in && !skip)
        return pesr_conv3x3_s2_dgrad_launch(dy, w_packed_dgrad, mask, dx, N, H, W, Cout, Cin, alpha, (hipStream_t)stream);
    return PESR_EINVAL;
}

// split-K scratch for layers whose tiles cannot fill the chip: up to 8 partial copies of a (small) output
PESR_API size_t pesr_conv3x3_workspace_bytes(int N, int OH, int OW, int Cout) {
    const long tiles = ((long)N * OH * OW + 143) / 144 * ((Cout + 63) / 64);
    return tiles < 640 ? (size_t)8 * N * OH * OW * Cout * sizeof(float) : 0;
}

// ---- convs whose epilogue leaves BatchNorm sums (round 6, ABI 18; common.h BnEpi) ------------------------------------------------
static PesrBnFuseArgs fuse_args(const PesrBnFuse* f, int dry) {
    PesrBnFuseArgs a{};
    a.dry = dry;
    if (f) { a.mode = f->mode; a.rows = f->rows; a.part = f->part; a.z = f->z; a.mean_invstd = f->mean_invstd; a.gamma = f->gamma; a.beta = f->beta; a.slope = f->slope; }
    return a;
}
PESR_API long pesr_conv3x3_bn_rows(int which, int N, int H, int W, int Cin, int Cout, int stride) {
    PesrBnFuseArgs a = fuse_args(nullptr, 1);
    int rc = PESR_EINVAL;
    // the planners decide split-K from the workspace the caller would pass: the same sizes as ops.py / pesr_conv3x3_workspace_bytes
    if (which == 0) {
        const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
        const size_t wsb = pesr_conv3x3_workspace_bytes(N, OH, OW, Cout);
        rc = pesr_conv3x3_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, pad16(Cin), pad64(Cout), stride, 1.f, PESR_ACT_NONE, 0.f,
                                 0, 0, 0, Cin, Cout, wsb ? (void*)8 : nullptr, wsb, nullptr, &a);
    } else if (which == 1) {
        if (stride == 1) {
            const size_t wsb = pesr_conv3x3_workspace_bytes(N, H, W, Cin);
            rc = pesr_conv3x3_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, pad16(Cout), pad64(Cin), 1, 1.f, PESR_ACT_NONE, 0.f, 0, 0,
                                     1, Cout, Cin, wsb ? (void*)8 : nullptr, wsb, nullptr, &a);
        } else if (stride == 2) {
            rc = pesr_conv3x3_s2_dgrad_launch(nullptr, nullptr, nullptr, nullptr, N, H, W, Cout, Cin, 1.f, nullptr, &a);
        }
    } else if (which == 2) {
        rc = pesr_conv3x3_wino4_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, 1.f, PESR_ACT_NONE, 0.f, 0, 0, nullptr, 0,
                                       nullptr, &a);
    }
    return rc ? 0 : a.rows_out;
}
PESR_API int pesr_conv3x3_fwd_bn(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                                 int stride, void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream) {
    if (!fuse) return PESR_EINVAL;
    PesrBnFuseArgs a = fuse_args(fuse, 0);
    return pesr_conv3x3_launch(x, w_packed, bias, nullptr, nullptr, y, N, H, W, pad16(Cin), pad64(Cout), stride, 1.f, PESR_ACT_NONE, 0.f, 0, 0, 0, Cin,
                               Cout, workspace, ws_bytes, (hipStream_t)stream, &a);
}
PESR_API int pesr_conv3x3_dgrad_bn(const float* dy, const float* w_packed_dgrad, float* dx, int N, int H, int W, int Cin, int Cout, int stride,
                                   void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream) {
    if (!fuse) return PESR_EINVAL;
    PesrBnFuseArgs a = fuse_args(fuse, 0);
    if (stride == 1)
        return pesr_conv3x3_launch(dy, w_packed_dgrad, nullptr, nullptr, nullptr, dx, N, H, W, pad16(Cout), pad64(Cin), 1, 1.f, PESR_ACT_NONE, 0.f, 0, 0,
                                   1, Cout, Cin, workspace, ws_bytes, (hipStream_t)stream, &a);
    if (stride == 2) return pesr_conv3x3_s2_dgrad_launch(dy, w_packed_dgrad, nullptr, dx, N, H, W, Cout, Cin, 1.f, (hipStream_t)stream, &a);
    return PESR_EINVAL;
}
PESR_API int pesr_conv3x3_wino4_bn(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                                   void* workspace, size_t ws_bytes, const PesrBnFuse* fuse, void* stream) {
    if (!fuse) return PESR_EINVAL;
    PesrBnFuseArgs a = fuse_args(fuse, 0);
    return pesr_conv3x3_wino4_launch(x, w_packed, bias, nullptr, nullptr, y, N, H, W, Cin, Cout, 1.f, PESR_ACT_NONE, 0.f, 0, 0, workspace, ws_bytes,
                                     (hipStream_t)stream, &a);
}
PESR_API int pesr_bn_finalize(const float* part, int rows, int C, long M, float eps, float momentum, float* mean_invstd, float* running_mean,
                              float* running_var, long long* num_batches, void* stream) {
    return pesr_bn_finalize_launch(part, rows, C, M, eps, momentum, mean_invstd, running_mean, running_var, num_batches, (hipStream_t)stream);
}
PESR_API int pesr_bn_lrelu_bwd_fused(const float* z, const float* g_masked, const float* part, int rows, const float* gamma, const float* beta,
                                     const float* mean_invstd, float* dz, float* dgamma, float* dbeta, int N, int H, int W, int C, int accumulate,
                                     void* workspace, size_t ws_bytes, void* stream) {
    return pesr_bn_lrelu_bwd_fused_launch(z, g_masked, part, rows, gamma, beta, mean_invstd, dz, dgamma, dbeta, (long)N * H * W, C, (long)H * W,
                                          accumulate, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API size_t pesr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride, int algo) {
    return pesr_conv3x3_wgrad_ws_bytes(N, H, W, Cin, Cout, stride, algo);
}
PESR_API int pesr_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                int stride, float alpha, int ps_in, int algo, int accumulate, void* workspace, size_t ws_bytes,
                                void* stream) {
    return pesr_conv3x3_wgrad_launch(x, dy, dw, db, N, H, W, Cin, Cout, stride, alpha, ps_in, algo, accumulate, workspace, ws_bytes,
                                     (hipStream_t)stream);
}

PESR_API size_t pesr_conv3x3_wgrad_rgb_workspace_bytes(int N, int H, int W, int C) {
    return pesr_conv3x3_wgrad_rgb_ws_bytes(N, H, W, C);
}
PESR_API int pesr_conv3x3_wgrad_rgb(const float* a, const float* b3, float* dw, float* db, int N, int H, int W, int C, int mode,
                                    float alpha, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_wgrad_rgb_launch(a, b3, dw, db, N, H, W, C, mode, alpha, accumulate, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_meanshift_fwd(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int x_nchw,
                                int y_nchw, void* stream) {
    const long HW = (long)H * W;
    const long xsn = 3 * HW, xsc = x_nchw ? HW : 1, xsp = x_nchw ? 1 : 3;
    const long ysn = 3 * HW, ysc = y_nchw ? HW : 1, ysp = y_nchw ? 1 : 3;
    return pesr_meanshift_fwd_launch(x, w, b, y, N, H, W, xsn, xsc, xsp, ysn, ysc, ysp, (hipStream_t)stream);
}
PESR_API int pesr_meanshift_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int N, int H,
                                int W, int x_nchw, void* workspace, size_t ws_bytes, void* stream) {
    const long HW = (long)H * W;
    const long xsn = 3 * HW, xsc = x_nchw ? HW : 1, xsp = x_nchw ? 1 : 3;
    return pesr_meanshift_bwd_launch(dy, x, w, dx, dw, db, N, H, W, xsn, xsc, xsp, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    return pesr_pixel_shuffle_launch(x, y, N, H, W, C, 0, (hipStream_t)stream);
}
PESR_API int pesr_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    return pesr_pixel_shuffle_launch(dy, dx, N, H, W, C, 1, (hipStream_t)stream);
}
PESR_API int pesr_relu_mask(const float* g, const float* ref, const float* add, float* out, long n, float alpha, float slope,
                            void* stream) {
    return pesr_relu_mask_launch(g, ref, add, out, n, alpha, slope, (hipStream_t)stream);
}
PESR_API int pesr_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    return pesr_maxpool2x2_fwd_launch(x, y, N, H, W, C, (hipStream_t)stream);
}
PESR_API int pesr_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int relu_in, void* stream) {
    return pesr_maxpool2x2_bwd_launch(x, dy, dx, N, H, W, C, relu_in, (hipStream_t)stream);
}

PESR_API size_t pesr_bn_workspace_bytes(long M, int C) { return pesr_bn_ws_bytes(M, C); }
PESR_API int pesr_bn_lrelu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_invstd,
                               float* running_mean, float* running_var, long long* num_batches, int N, int H, int W, int C,
                               float eps, float momentum, float slope, int y_nchw, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_bn_lrelu_fwd_launch(x, gamma, beta, y, mean_invstd, running_mean, running_var, num_batches, (long)N * H * W, C,
                                    (long)H * W, eps, momentum, slope, y_nchw, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API size_t pesr_conv3x3_rgb_bn_workspace_bytes(int N, int H, int W, int C) { return pesr_conv_rgb_bn_ws_bytes(N, H, W, C); }
PESR_API int pesr_conv3x3_rgb_bn_lrelu_fwd(const float* x, const float* w, float* z, const float* gamma, const float* beta, float* y,
                                           float* mean_invstd, float* running_mean, float* running_var, long long* num_batches, int N,
                                           int H, int W, int C, float eps, float momentum, float slope, int y_nchw, void* workspace,
                                           size_t ws_bytes, void* stream) {
    return pesr_conv_rgb_bn_lrelu_fwd_launch(x, w, z, gamma, beta, y, mean_invstd, running_mean, running_var, num_batches, N, H, W, C, eps,
                                             momentum, slope, y_nchw, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_bn_lrelu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                               float* dx, float* dgamma, float* dbeta, int N, int H, int W, int C, float slope, int dy_nchw,
                               int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_bn_lrelu_bwd_launch(x, dy, gamma, beta, mean_invstd, dx, dgamma, dbeta, (long)N * H * W, C, (long)H * W, slope,
                                    dy_nchw, accumulate, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_bn_lrelu_eval_fwd(const float* x, const float* gamma, const float* beta, const float* mean_invstd, float* y, int N,
                                    int H, int W, int C, float slope, int y_nchw, void* stream) {
    return pesr_bn_lrelu_apply_launch(x, gamma, beta, mean_invstd, y, (long)N * H * W, C, (long)H * W, slope, y_nchw, (hipStream_t)stream);
}
PESR_API int pesr_bn_lrelu_eval_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                                    float* dx, float* dgamma, float* dbeta, int N, int H, int W, int C, float slope, int dy_nchw,
                                    void* workspace, size_t ws_bytes, void* stream) {
    return pesr_bn_lrelu_bwd_eval_launch(x, dy, gamma, beta, mean_invstd, dx, dgamma, dbeta, (long)N * H * W, C, (long)H * W, slope,
                                         dy_nchw, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API size_t pesr_bn_bwd_bwd_workspace_bytes(long M, int C) { return pesr_bn_bwd_bwd_ws_bytes(M, C); }
PESR_API int pesr_bn_bwd_bwd(const float* z, const float* du, const float* g, const float* gamma, const float* mean_invstd, float* l_du,
                             float* l_z, float* l_gamma, int N, int H, int W, int C, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_bn_bwd_bwd_launch(z, du, g, gamma, mean_invstd, l_du, l_z, l_gamma, (long)N * H * W, C, workspace, ws_bytes,
                                  (hipStream_t)stream);
}

PESR_API size_t pesr_linear_workspace_bytes(int M, int N, long K) { return pesr_linear_ws_bytes(M, N, K); }
PESR_API int pesr_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, long K, int act, float slope,
                             void* workspace, size_t ws_bytes, void* stream) {
    return pesr_linear_fwd_launch(x, w, b, y, M, N, K, act, slope, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, long K, void* workspace, size_t ws_bytes,
                               void* stream) {
    return pesr_linear_dgrad_launch(dy, w, dx, M, N, K, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, long K, int accumulate,
                               void* stream) {
    return pesr_linear_wgrad_launch(dy, x, dw, db, M, N, K, accumulate, (hipStream_t)stream);
}

PESR_API int pesr_loss_l1_tv_fwd_bwd(const float* sr, const float* hr, float* grad, float* out2, int N, int H, int W, float g_l1,
                                     float g_tv, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_loss_l1_tv_launch(sr, hr, grad, out2, N, H, W, g_l1, g_tv, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_mse_fwd_bwd(const float* a, const float* b, float* grad, float* out1, long n, float gscale, void* workspace,
                              size_t ws_bytes, void* stream) {
    return pesr_loss_mse_launch(a, b, grad, out1, n, gscale, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                            int step, float grad_scale, void* stream) {
    return pesr_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream);
}

PESR_API int pesr_adam_step_dev(float* p, const float* g, float* m, float* v, long n, float* state, float beta1, float beta2, float eps,
                                float grad_scale, void* stream) {
    return pesr_adam_dev_launch(p, g, m, v, n, state, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_rgb_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                                  int act, float slope, void* stream) {
    return pesr_conv_rgb_in_launch(x, w, bias, y, N, H, W, Cout, act, slope, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_rgb_out_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                                      float slope, void* stream) {
    return pesr_conv_rgb_out_fwd_launch(x, w, bias, y, N, H, W, C, act, slope, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_rgb_in_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, void* stream) {
    return pesr_conv_rgb_in_dgrad_launch(dy, w, dx, N, H, W, C, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_rgb_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, void* stream) {
    return pesr_conv_rgb_out_dgrad_launch(dy, w, dx, N, H, W, C, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_wino_supported(int N, int H, int W, int Cin, int Cout) {
    return pesr_conv3x3_wino_supported_impl(N, H, W, Cin, Cout);
}
PESR_API int pesr_pack_conv3x3_wino(const float* w, float* w_packed, int Cout, int Cin, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_wino_launch(w, w_packed, Cout, Cin, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_wino(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                               float* y, int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out,
                               int ps_in, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_wino_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, alpha, act, slope, ps_out, ps_in,
                                    workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_wino4_score(int N, int H, int W, int Cin, int Cout, int allow_split) {
    return pesr_conv3x3_wino4_score_impl(N, H, W, Cin, Cout, allow_split);
}
PESR_API int pesr_pack_conv3x3_wino4(const float* w, float* w_packed, int Cout, int Cin, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_wino4_launch(w, w_packed, Cout, Cin, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_wino4(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                                float* y, int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out,
                                int ps_in, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_wino4_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, alpha, act, slope, ps_out, ps_in,
                                     workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_bf16_score(int N, int H, int W, int Cin, int Cout, int min_wgs) {
    return pesr_conv3x3_bf16_score_impl(N, H, W, Cin, Cout, min_wgs);
}
PESR_API int pesr_pack_conv3x3_bf16(const float* w, void* w_packed, int Cout, int Cin, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_bf16_launch(w, w_packed, Cout, Cin, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_bf16(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask,
                               float* y, int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out,
                               int ps_in, void* stream) {
    return pesr_conv3x3_bf16_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, alpha, act, slope, ps_out, ps_in,
                                    (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_bf16x3_score(int N, int H, int W, int Cin, int Cout, int min_wgs) {
    return pesr_conv3x3_bf16x3_score_impl(N, H, W, Cin, Cout, min_wgs);
}
PESR_API int pesr_pack_conv3x3_bf16x3(const float* w, void* w_packed, int Cout, int Cin, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_bf16x3_launch(w, w_packed, Cout, Cin, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_bf16x3(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask,
                                 float* y, int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps_out,
                                 int ps_in, void* stream) {
    return pesr_conv3x3_bf16x3_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, alpha, act, slope, ps_out, ps_in,
                                      (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_bf16_s2_score(int N, int H, int W, int Cin, int Cout, int min_wgs) {
    return pesr_conv3x3_bf16_s2_score_impl(N, H, W, Cin, Cout, min_wgs);
}
PESR_API int pesr_conv3x3_bf16_s2(const float* x, const void* w_packed, const float* bias, const float* skip, const float* mask,
                                  float* y, int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, void* stream) {
    return pesr_conv3x3_bf16_s2_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, alpha, act, slope, (hipStream_t)stream);
}
PESR_API int pesr_conv3x3_bf16_s2_dgrad_score(int N, int H, int W, int Cout_fwd, int Cin_fwd, int min_wgs) {
    return pesr_conv3x3_bf16_s2_dgrad_score_impl(N, H, W, Cout_fwd, Cin_fwd, min_wgs);
}
PESR_API int pesr_conv3x3_bf16_s2_dgrad(const float* dy, const void* w_packed, const float* mask, const float* skip, float* dx, int N,
                                        int H, int W, int Cout_fwd, int Cin_fwd, float alpha, void* stream) {
    return pesr_conv3x3_bf16_s2_dgrad_launch(dy, w_packed, mask, skip, dx, N, H, W, Cout_fwd, Cin_fwd, alpha, (hipStream_t)stream);
}
PESR_API size_t pesr_conv3x3_wgrad_bf16_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    return pesr_conv3x3_wgrad_bf16_ws_bytes(N, H, W, Cin, Cout);
}
PESR_API int pesr_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                     float alpha, int ps_in, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_wgrad_bf16_launch(x, dy, dw, db, N, H, W, Cin, Cout, alpha, ps_in, accumulate, workspace, ws_bytes,
                                          (hipStream_t)stream);
}

PESR_API int pesr_crop_augment(const unsigned char* pool, const long long* desc, float* out, int B, int P, int nhwc, void* stream) {
    return pesr_crop_augment_launch(pool, desc, out, B, P, nhwc, (hipStream_t)stream);
}

PESR_API int pesr_psnr_y(const float* a, const float* b, double* out2, int H, int W, int a_nhwc, int b_nhwc, void* workspace,
                         size_t ws_bytes, void* stream) {
    return pesr_psnr_y_launch(a, b, out2, H, W, a_nhwc, b_nhwc, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API size_t pesr_spectral_norm_workspace_bytes(int O, int K) { return pesr_spectral_norm_ws_bytes(O, K); }
PESR_API int pesr_spectral_norm_fwd(const float* w, float* u, float* v, float* w_hat, float* sigma, int O, int K, int update, float eps,
                                    void* workspace, size_t ws_bytes, void* stream) {
    return pesr_spectral_norm_fwd_launch(w, u, v, w_hat, sigma, O, K, update, eps, workspace, ws_bytes, (hipStream_t)stream);
}
PESR_API int pesr_spectral_norm_bwd(const float* g, const float* w_hat, const float* u, const float* v, const float* sigma, float* dw,
                                    int O, int K, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_spectral_norm_bwd_launch(g, w_hat, u, v, sigma, dw, O, K, accumulate, workspace, ws_bytes, (hipStream_t)stream);
}

PESR_API int pesr_conv_kxk_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                               int k, int stride, void* stream) {
    return pesr_conv_kxk_fwd_launch(x, w, bias, y, N, H, W, Cin, Cout, k, stride, (hipStream_t)stream);
}
PESR_API int pesr_conv_kxk_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int k, int stride,
                                 void* stream) {
    return pesr_conv_kxk_dgrad_launch(dy, w, dx, N, H, W, Cin, Cout, k, stride, (hipStream_t)stream);
}
PESR_API int pesr_conv_kxk_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout, int k,
                                 int stride, void* stream) {
    return pesr_conv_kxk_wgrad_launch(x, dy, dw, db, N, H, W, Cin, Cout, k, stride, (hipStream_t)stream);
}

PESR_API int pesr_gan_loss_fwd_bwd(const float* pred_real, const float* pred_fake, int B, int gan_type, int side, int focal, float gamma,
                                   float scale, float* out, float* d_real, float* d_fake, void* stream) {
    return pesr_gan_loss_launch(pred_real, pred_fake, B, gan_type, side, focal, gamma, scale, out, d_real, d_fake, (hipStream_t)stream);
}
