// Internal C++ launchers (one per kernel family); wrapped by api.hip into the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

int pesr_pack_conv3x3_launch(const float* w, float* out, int O, int I, int mode, int ps, hipStream_t stream);
int pesr_pack_bias_ps_launch(const float* b, float* out, int O, hipStream_t stream);

int pesr_conv3x3_launch(const float* x, const float* wp, const float* bias, const float* skip, const float* mask, float* y,
                        int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act, float slope, int ps,
                        int ps_in, int flip, hipStream_t stream);
int pesr_conv3x3_s2_dgrad_launch(const float* dy, const float* wp, const float* mask, float* dx, int N, int H, int W,
                                 int Cout_fwd, int Cin_fwd, float alpha, hipStream_t stream);

size_t pesr_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int stride);
int pesr_conv3x3_wgrad_launch(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                              int stride, float alpha, int ps_in, void* ws, size_t ws_bytes, hipStream_t stream);
int pesr_bias_grad_launch(const float* dy, float* db, long pixels, int Cout, int OW, float alpha, int ps_in, float* part,
                          size_t part_bytes, hipStream_t stream);
