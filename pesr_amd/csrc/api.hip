// extern "C" surface of libpesr_hip.so (declared in include/pesr_hip.h).  Thin, exception-free
// wrappers over the per-family launchers; no torch types cross this boundary.
#include "common.h"
#include "launchers.h"
#include "../../include/pesr_hip.h"

PESR_API int pesr_abi_version(void) { return 1; }

PESR_API int pesr_pack_conv3x3(const float* w, float* out, int O, int I, int mode, int ps, void* stream) {
    return pesr_pack_conv3x3_launch(w, out, O, I, mode, ps, (hipStream_t)stream);
}
PESR_API int pesr_pack_bias_ps(const float* b, float* out, int O, void* stream) {
    return pesr_pack_bias_ps_launch(b, out, O, (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, const float* skip, const float* mask,
                              float* y, int N, int H, int W, int Cin, int Cout, int stride, float alpha, int act,
                              float slope, int ps_out, void* stream) {
    return pesr_conv3x3_launch(x, w_packed, bias, skip, mask, y, N, H, W, Cin, Cout, stride, alpha, act, slope, ps_out, 0, 0,
                               (hipStream_t)stream);
}

PESR_API int pesr_conv3x3_dgrad(const float* dy, const float* w_packed_dgrad, const float* mask, const float* skip, float* dx,
                                int N, int H, int W, int Cin, int Cout, int stride, float alpha, int ps_in, void* stream) {
    if (stride == 1)  // a stride-1 conv over dy with Cin/Cout swapped and the taps flipped
        return pesr_conv3x3_launch(dy, w_packed_dgrad, nullptr, skip, mask, dx, N, H, W, Cout, Cin, 1, alpha, PESR_ACT_NONE,
                                   0.f, 0, ps_in, 1, (hipStream_t)stream);
    if (stride == 2 && !ps_in && !skip)
        return pesr_conv3x3_s2_dgrad_launch(dy, w_packed_dgrad, mask, dx, N, H, W, Cout, Cin, alpha, (hipStream_t)stream);
    return PESR_EINVAL;
}

PESR_API size_t pesr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride) {
    return pesr_conv3x3_wgrad_ws_bytes(N, H, W, Cin, Cout, stride);
}
PESR_API int pesr_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                int stride, float alpha, int ps_in, void* workspace, size_t ws_bytes, void* stream) {
    return pesr_conv3x3_wgrad_launch(x, dy, dw, db, N, H, W, Cin, Cout, stride, alpha, ps_in, workspace, ws_bytes,
                                     (hipStream_t)stream);
}
