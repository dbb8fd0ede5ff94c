// Internal C++ launchers (one per kernel family); wrapped by api.hip into the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

int pesr_pack_conv3x3_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_pack_conv3x3_batched_launch(const long long* desc, int count, hipStream_t stream);
int pesr_pack_bias_ps_launch(const float* b, float* out, int O, hipStream_t stream);

// BatchNorm sums out of a conv kernel's epilogue (round 6; include/pesr_hip.h PesrBnFuse + the launcher's own in / out fields)
struct PesrBnFuseArgs {
    int mode;                 // 1: sum / sum of squares of the stored output; 2: the kernel writes g' = v * lrelu'(bn(z)) and sums g', g' * xhat
    int rows;                 // capacity of `part` in rows of [2][C]
    float* part;
    const float* z;           // mode 2
    const float* mean_invstd; // mode 2: [2][C]
    const float* gamma;
    const float* beta;
    float slope;
    int dry;                  // plan only: report rows_out, launch nothing
    long rows_out;            // rows the launch writes (0: this shape is not covered - split-K, packed-shuffle stores, odd channel counts)
};

int pesr_conv3x3_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                        int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act, float slope, int ps,
                        int ps_in, int flip, int cin_real, int cout_store, void* ws, size_t ws_bytes, hipStream_t stream,
                        PesrBnFuseArgs* fuse = nullptr);
int pesr_conv3x3_s2_dgrad_launch(const float* dy, const float* wp, const float* mask, float* dx, int N, int H, int W,
                                 int Cout_fwd, int Cin_fwd, float alpha, hipStream_t stream, PesrBnFuseArgs* fuse = nullptr);

size_t pesr_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int stride, int algo);
int pesr_conv3x3_wgrad_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                              int stride, float alpha, int ps_in, int algo, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_bias_grad_launch(const float* dy, float* db, long pixels, int Cout, int OW, float alpha, int ps_in, float* part,
                          size_t part_bytes, hipStream_t stream);

size_t pesr_conv3x3_wgrad_rgb_ws_bytes(int N, int H, int W, int C);
int pesr_conv3x3_wgrad_rgb_launch(const float* A, const float* b3, float* dw, float* db, int N, int H, int W, int C, int mode,
                                  float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

int pesr_meanshift_fwd_launch(const float* x, const float* w, const float* b, float* y, int N, int H, int W, long xsn, long xsc,
                              long xsp, long ysn, long ysc, long ysp, hipStream_t stream);
int pesr_meanshift_bwd_launch(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int N, int H,
                              int W, long xsn, long xsc, long xsp, void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_pixel_shuffle_launch(const float* in, float* out, int N, int H, int W, int C, int inverse, hipStream_t stream);
int pesr_relu_mask_launch(const float* g, const float* ref, const float* add, float* out, long n, float alpha, float slope, hipStream_t stream);
int pesr_maxpool2x2_fwd_launch(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream);
int pesr_maxpool2x2_bwd_launch(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int relu_in, hipStream_t stream);

size_t pesr_bn_ws_bytes(long M, int C);
size_t pesr_conv_rgb_bn_ws_bytes(int N, int H, int W, int C);
int pesr_conv_rgb_bn_lrelu_fwd_launch(const float* x, const float* w, float* z, const float* gamma, const float* beta, float* y,
                                      float* mean_invstd, float* running_mean, float* running_var, long long* num_batches, int N, int H,
                                      int W, int C, float eps, float momentum, float slope, int y_nchw, void* ws, size_t ws_bytes,
                                      hipStream_t stream);
int pesr_bn_lrelu_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean_invstd,
                             float* running_mean, float* running_var, long long* num_batches, long M, int C, long HW, float eps,
                             float momentum, float slope, int y_nchw, void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_bn_lrelu_bwd_launch(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                             float* dx, float* dgamma, float* dbeta, long M, int C, long HW, float slope, int dy_nchw, int accumulate,
                             void* ws, size_t ws_bytes, hipStream_t stream);

int pesr_bn_finalize_launch(const float* part, int rows, int C, long M, float eps, float momentum, float* mean_invstd, float* running_mean,
                            float* running_var, long long* num_batches, hipStream_t stream);
int pesr_bn_lrelu_bwd_fused_launch(const float* z, const float* gmasked, const float* part, int rows, const float* gamma, const float* beta,
                                   const float* mean_invstd, float* dz, float* dgamma, float* dbeta, long M, int C, long HW, int accumulate,
                                   void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_bn_lrelu_apply_launch(const float* x, const float* gamma, const float* beta, const float* mean_invstd, float* y, long M,
                               int C, long HW, float slope, int y_nchw, hipStream_t stream);
int pesr_bn_lrelu_bwd_eval_launch(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean_invstd,
                                  float* dx, float* dgamma, float* dbeta, long M, int C, long HW, float slope, int dy_nchw, void* ws,
                                  size_t ws_bytes, hipStream_t stream);

size_t pesr_bn_bwd_bwd_ws_bytes(long M, int C);
int pesr_bn_bwd_bwd_launch(const float* z, const float* du, const float* g, const float* gamma, const float* mean_invstd, float* l_du,
                           float* l_z, float* l_gamma, long M, int C, void* ws, size_t ws_bytes, hipStream_t stream);

size_t pesr_linear_ws_bytes(int M, int N, long K);
int pesr_linear_fwd_launch(const float* x, const float* W, const float* b, float* y, int M, int N, long K, int act, float slope,
                           void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_linear_dgrad_launch(const float* dy, const float* W, float* dx, int M, int N, long K, void* ws, size_t ws_bytes,
                             hipStream_t stream);
int pesr_linear_wgrad_launch(const float* dy, const float* x, float* dW, float* db, int M, int N, long K, int accumulate,
                             hipStream_t stream);

int pesr_loss_l1_tv_launch(const float* sr, const float* hr, float* grad, float* out2, int N, int H, int W, float g_l1, float g_tv,
                           void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_loss_mse_launch(const float* a, const float* b, float* grad, float* out1, long n, float gscale, void* ws, size_t ws_bytes,
                         hipStream_t stream);
int pesr_gan_loss_launch(const float* pred_real, const float* pred_fake, int B, int gan, int side, int focal, float gamma, float scale,
                         float* out, float* d_real, float* d_fake, hipStream_t stream);
int pesr_adam_dev_launch(float* p, const float* g, float* m, float* v, long n, float* state, float b1, float b2, float eps, float gscale,
                         hipStream_t stream);
int pesr_adam_launch(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                     float gscale, hipStream_t stream);

// dsum[col] = sum_k part[k*ncols + col] in double, fixed order (reduce.hip)
int pesr_reduce_rows_launch(const float* part, double* dsum, int nb, int ncols, hipStream_t stream);

// fixed-order split-K reduce of the direct wgrad kernel: slab [split][9][Cout][Cin] -> dw OIHW (+ bias partials -> db)
int pesr_wgrad_reduce_launch(const float* slab, float* dw, int split, int Cout, int Cin, float alpha, int ps, const float* bias_part,
                             int bias_rows, float* db, int accumulate, hipStream_t stream);
// transposed Winograd F(4,3) weight gradient (conv3x3_wgrad_wino4.hip); PESR_EINVAL for shapes it does not cover
size_t pesr_conv3x3_wgrad_wino4_ws_bytes(int N, int H, int W, int Cin, int Cout);
int pesr_conv3x3_wgrad_wino4_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                    float alpha, int ps_in, int accumulate, int variant, void* ws, size_t ws_bytes, hipStream_t stream);
// transposed Winograd weight gradient (conv3x3_wgrad_wino.hip); the launch returns PESR_EINVAL for shapes it does not cover
size_t pesr_conv3x3_wgrad_wino_ws_bytes(int N, int H, int W, int Cin, int Cout);
int pesr_conv3x3_wgrad_wino_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                   float alpha, int ps_in, void* ws, size_t ws_bytes, hipStream_t stream);
// 1-D Winograd F(2,3) variant of the stride-1 conv (conv3x3_wino.hip)
int pesr_conv3x3_wino_supported_impl(int N, int H, int W, int Cin, int Cout);
int pesr_pack_conv3x3_wino_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_conv3x3_wino_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                             int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                             void* ws, size_t ws_bytes, hipStream_t stream);
// bf16-operand mode (conv3x3_bf16.hip, conv3x3_wgrad_bf16.hip): operands rounded to bf16, fp32 accumulation
int pesr_conv3x3_bf16_score_impl(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_pack_conv3x3_bf16_launch(const float* w, void* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_conv3x3_bf16_launch(const float* x, const void* wp, const float* bias, const float* skip, const float* mask, float* y,
                             int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                             hipStream_t stream);
// split-bf16 mode (conv3x3_bf16x3.hip): every operand as hi + lo bf16 terms, three products, fp32 accumulation
int pesr_conv3x3_bf16x3_score_impl(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_pack_conv3x3_bf16x3_launch(const float* w, void* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_conv3x3_bf16x3_launch(const float* x, const void* wp, const float* bias, const float* skip, const float* mask, float* y,
                               int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                               hipStream_t stream);
int pesr_conv3x3_bf16_s2_score_impl(int N, int H, int W, int Cin, int Cout, int min_wgs);
int pesr_conv3x3_bf16_s2_launch(const float* x, const void* wp, const float* bias, const float* skip, const float* mask, float* y,
                                int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, hipStream_t stream);
int pesr_conv3x3_bf16_s2_dgrad_score_impl(int N, int H, int W, int Cout_fwd, int Cin_fwd, int min_wgs);
int pesr_conv3x3_bf16_s2_dgrad_launch(const float* dy, const void* wp, const float* mask, const float* skip, float* dx, int N, int H, int W,
                                      int Cout_fwd, int Cin_fwd, float alpha, hipStream_t stream);
size_t pesr_conv3x3_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout);
int pesr_conv3x3_wgrad_bf16_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                   float alpha, int ps_in, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

// 1-D Winograd F(4,3) variant (conv3x3_wino4.hip): half of the direct conv's multiplies
int pesr_conv3x3_wino4_score_impl(int N, int H, int W, int Cin, int Cout, int allow_split);
int pesr_pack_conv3x3_wino4_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_conv3x3_wino4_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                              int N, int H, int W, int Cin, int Cout, float alpha, int act, float slope, int ps, int ps_in,
                              void* ws, size_t ws_bytes, hipStream_t stream, PesrBnFuseArgs* fuse = nullptr);
struct BnEpi;
long pesr_conv_splitk_finish_bn_rows(int C, int ksplit);
int pesr_conv_splitk_finish_bn_launch(const float* slab, const float* bias, float* y, long total, int C, int ksplit, float alpha, const BnEpi& bn,
                                      hipStream_t stream);
int pesr_conv_splitk_finish_launch(const float* slab, const float* bias, const float* skip, const float* mask, float* y, long total,
                                   int C, int ksplit, float alpha, int act, float slope, hipStream_t stream);
int pesr_conv_rgb_in_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                            float slope, hipStream_t stream);
int pesr_conv_rgb_out_fwd_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int act,
                                 float slope, hipStream_t stream);
int pesr_conv_rgb_in_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int C, hipStream_t stream);
// the same conv (no bias) leaving BatchNorm partial sums per workgroup: part[rows][2][C] (conv_rgb_in.hip)
int pesr_conv_rgb_in_stats_rows(int N, int H, int W, int C);
int pesr_conv_rgb_in_stats_launch(const float* x, const float* w, float* y, float* part, int N, int H, int W, int C, hipStream_t stream);
int pesr_conv_rgb_out_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int C, hipStream_t stream);

int pesr_conv_kxk_fwd_launch(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int Cin, int Cout, int k, int s,
                             hipStream_t stream);
int pesr_conv_kxk_dgrad_launch(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int k, int s,
                               hipStream_t stream);
int pesr_conv_kxk_wgrad_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout, int k, int s,
                               hipStream_t stream);

size_t pesr_spectral_norm_ws_bytes(int O, int K);
int pesr_spectral_norm_fwd_launch(const float* W, float* u, float* v, float* w_hat, float* sigma, int O, int K, int update, float eps,
                                  void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_spectral_norm_bwd_launch(const float* G, const float* w_hat, const float* u, const float* v, const float* sigma, float* dW, int O,
                                  int K, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

int pesr_crop_augment_launch(const unsigned char* pool, const long long* desc, float* out, int B, int P, int nhwc, hipStream_t stream);

int pesr_psnr_y_launch(const float* a, const float* b, double* out2, int H, int W, int a_nhwc, int b_nhwc, void* ws, size_t ws_bytes,
                       hipStream_t stream);
