// GPU-side training-sample assembly: random-crop + 8-way flip/transpose augmentation + uint8 -> fp32,
// straight from a device-resident uint8 HWC image pool - the work of reference data.py:79-126 (_crop, _aug_data,
// _to_tensor) that 4 DataLoader worker processes do on the host in the reference.  Pure indexing: bit-exact.
//   out[b][:, oy, ox] = pool[desc[b].offset + ((y0 + iy) * stride_w + (x0 + ix)) * 3 + c]     (values 0..255 as floats)
// where (iy, ix) is (oy, ox) mapped back through hflip (bit 0), vflip (bit 1), transpose (bit 2) - the reference applies
// transpose first, then the vertical, then the horizontal flip.
#include "common.h"
#include "launchers.h"



__global__ void crop_augment_kernel(const unsigned char* __restrict__ pool, const long long* __restrict__ desc, float* __restrict__ out,
                                    int B, int P, int nhwc) {
    const long total = (long)B * P * P;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(e % P);
        const int oy = (int)((e / P) % P);
        const int b = (int)(e / ((long)P * P));
        const long long off = desc[b * 3];
        const long long w1 = desc[b * 3 + 1], w2 = desc[b * 3 + 2];
        const int stride_w = (int)(w1 & 0xffffffff), y0 = (int)(w1 >> 32);
        const int x0 = (int)(w2 & 0xffffffff), aug = (int)(w2 >> 32);
        const int cx = (aug & 1) ? P - 1 - ox : ox;
        const int cy = (aug & 2) ? P - 1 - oy : oy;
        const int iy = (aug & 4) ? cx : cy, ix = (aug & 4) ? cy : cx;
        const unsigned char* src = pool + off + ((long long)(y0 + iy) * stride_w + (x0 + ix)) * 3;
        const float r = (float)src[0], g = (float)src[1], bl = (float)src[2];
        if (nhwc) {
            float* o = out + e * 3;
            o[0] = r; o[1] = g; o[2] = bl;
        } else {
            float* o = out + ((long)b * 3 * P + oy) * P + ox;
            o[0] = r; o[(long)P * P] = g; o[2L * P * P] = bl;
        }
    }
}

int pesr_crop_augment_launch(const unsigned char* pool, const long long* desc, float* out, int B, int P, int nhwc, hipStream_t stream) {
    if (B < 1 || P < 1) return PESR_EINVAL;
    const long total = (long)B * P * P;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(crop_augment_kernel, dim3(grid), dim3(256), 0, stream, pool, desc, out, B, P, nhwc);
    return pesr_launch_status();
}
