// dataaug.hip — the synchronised augmentation of a decoded training sample on the device:
//   SynchronizedTransform_Hyper   /root/reference/training/dataloaders/load.py:67-101  h-flip, resize rgb / normals (PIL bilinear), depth (PIL nearest), ToTensor
//   SynchronizedTransform_VKITTI  load.py:104-152                                      h-flip, ToTensor, KITTI benchmark crop 352 x 1216 (:112-131)
// The reference runs these per sample on the host through PIL (torchvision.transforms.Resize on a PIL image IS Image.resize).  Here a batch
// of decoded images (uint8 HWC rgb / normals, fp32 depth) is flipped, resized or cropped and converted in three kernels.  The bilinear
// resize is PIL's own algorithm, bit for bit: separable triangle filter whose support grows with the down-scaling factor, coefficients
// in 22-bit fixed point (built on the host exactly as Pillow's precompute_coeffs does, data.pil_bilinear_coeffs), a horizontal pass
// rounded to uint8, then a vertical pass rounded to uint8 (ImagingResampleHorizontal_8bpc / Vertical_8bpc) — so the tensors equal what
// the reference's loader would have produced, not an approximation of it.  The flip is an index reversal on the input side (with
// x -> 255 - x on the normals' first channel, load.py:80-82), nearest / crop are index tables.  All HBM-bound, one read + one write per pixel.
#include "common.h"

namespace e2eft {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t pil_clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [B][h0][w0][3] uint8 (flipped on the fly) -> mid [B][h0][w][3] uint8
__global__ __launch_bounds__(256) void aug_resample_h_kernel(int h0, int w0, int w, int ksize, const uint8_t* __restrict__ in, const int32_t* __restrict__ bounds,
                                                             const int32_t* __restrict__ kk, const uint8_t* __restrict__ flip, int invert_x,
                                                             uint8_t* __restrict__ mid) {
    const int b = blockIdx.z, y = blockIdx.y;
    const bool fl = flip && flip[b];
    const uint8_t* row = in + ((long)b * h0 + y) * w0 * 3;
    for (int xo = blockIdx.x * 256 + threadIdx.x; xo < w; xo += gridDim.x * 256) {
        const int xmin = bounds[xo * 2], n = bounds[xo * 2 + 1];
        const int32_t* k = kk + (long)xo * ksize;
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < n; ++i) {
            const int xi = fl ? w0 - 1 - (xmin + i) : xmin + i;
            const uint8_t* px = row + xi * 3;
            const int c0 = (fl && invert_x) ? 255 - px[0] : px[0];
            s0 += c0 * k[i]; s1 += px[1] * k[i]; s2 += px[2] * k[i];
        }
        uint8_t* o = mid + (((long)b * h0 + y) * w + xo) * 3;
        o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
    }
}

// vertical pass + ToTensor: mid [B][h0][w][3] uint8 -> out [B][3][h][w] fp32 = uint8 / 255
__global__ __launch_bounds__(256) void aug_resample_v_kernel(int h0, int w, int h, int ksize, const uint8_t* __restrict__ mid, const int32_t* __restrict__ bounds,
                                                             const int32_t* __restrict__ kk, float* __restrict__ out) {
    const int b = blockIdx.z, yo = blockIdx.y;
    const int ymin = bounds[yo * 2], n = bounds[yo * 2 + 1];
    const int32_t* k = kk + (long)yo * ksize;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < n; ++i) {
            const uint8_t* px = mid + (((long)b * h0 + ymin + i) * w + x) * 3;
            s0 += px[0] * k[i]; s1 += px[1] * k[i]; s2 += px[2] * k[i];
        }
        const long o = ((long)b * 3 * h + yo) * w + x;
        const long plane = (long)h * w;
        out[o] = pil_clip8(s0) / 255.0f;
        out[o + plane] = pil_clip8(s1) / 255.0f;
        out[o + 2 * plane] = pil_clip8(s2) / 255.0f;
    }
}

// index-table gathers (nearest resize, crop), flipped on the input side:  out[b][y][x] = in[b][ymap[y]][xmap[x] or w0-1-xmap[x]]
__global__ __launch_bounds__(256) void aug_gather_f32_kernel(int h0, int w0, int h, int w, const float* __restrict__ in, const int32_t* __restrict__ ymap,
                                                             const int32_t* __restrict__ xmap, const uint8_t* __restrict__ flip, float* __restrict__ out) {
    const int b = blockIdx.z, y = blockIdx.y;
    const bool fl = flip && flip[b];
    const float* row = in + ((long)b * h0 + ymap[y]) * w0;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const int xi = fl ? w0 - 1 - xmap[x] : xmap[x];
        out[((long)b * h + y) * w + x] = row[xi];
    }
}

// uint8 HWC -> fp32 planar / 255 through the same index tables (VKITTI: ToTensor + crop; also nearest resizing of 8-bit images)
__global__ __launch_bounds__(256) void aug_gather_u8_kernel(int h0, int w0, int h, int w, const uint8_t* __restrict__ in, const int32_t* __restrict__ ymap,
                                                            const int32_t* __restrict__ xmap, const uint8_t* __restrict__ flip, int invert_x,
                                                            float* __restrict__ out) {
    const int b = blockIdx.z, y = blockIdx.y;
    const bool fl = flip && flip[b];
    const uint8_t* row = in + ((long)b * h0 + ymap[y]) * w0 * 3;
    const long plane = (long)h * w;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const int xi = fl ? w0 - 1 - xmap[x] : xmap[x];
        const uint8_t* px = row + xi * 3;
        const long o = ((long)b * 3 * h + y) * w + x;
        out[o] = ((fl && invert_x) ? 255 - px[0] : px[0]) / 255.0f;
        out[o + plane] = px[1] / 255.0f;
        out[o + 2 * plane] = px[2] / 255.0f;
    }
}

static unsigned aug_gx(int w) { return (unsigned)((w + 255) / 256); }

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_aug_resample_bilinear_u8(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const uint8_t* in, const uint8_t* flip,
                                              int32_t invert_x_on_flip, const int32_t* xbounds, const int32_t* xcoef, int32_t xksize,
                                              const int32_t* ybounds, const int32_t* ycoef, int32_t yksize, uint8_t* mid, float* out, void* stream) {
    E2EFT_REQUIRE(in && xbounds && xcoef && ybounds && ycoef && mid && out, "aug_resample: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && h0 > 0 && w0 > 0 && h > 0 && w > 0 && h0 <= 65535 && h <= 65535 && xksize > 0 && yksize > 0, "aug_resample: shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(aug_resample_h_kernel, dim3(aug_gx(w), h0, batch), dim3(256), 0, s, h0, w0, w, xksize, in, xbounds, xcoef, flip, invert_x_on_flip, mid);
    hipLaunchKernelGGL(aug_resample_v_kernel, dim3(aug_gx(w), h, batch), dim3(256), 0, s, h0, w, h, yksize, (const uint8_t*)mid, ybounds, ycoef, out);
    return check_launch("aug_resample");
}

extern "C" int e2eft_aug_gather_f32(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const float* in, const int32_t* ymap, const int32_t* xmap,
                                    const uint8_t* flip, float* out, void* stream) {
    E2EFT_REQUIRE(in && ymap && xmap && out, "aug_gather: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && h0 > 0 && w0 > 0 && h > 0 && w > 0 && h <= 65535, "aug_gather: shape");
    hipLaunchKernelGGL(aug_gather_f32_kernel, dim3(aug_gx(w), h, batch), dim3(256), 0, (hipStream_t)stream, h0, w0, h, w, in, ymap, xmap, flip, out);
    return check_launch("aug_gather_f32");
}

extern "C" int e2eft_aug_gather_u8(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const uint8_t* in, const int32_t* ymap, const int32_t* xmap,
                                   const uint8_t* flip, int32_t invert_x_on_flip, float* out, void* stream) {
    E2EFT_REQUIRE(in && ymap && xmap && out, "aug_gather: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && h0 > 0 && w0 > 0 && h > 0 && w > 0 && h <= 65535, "aug_gather: shape");
    hipLaunchKernelGGL(aug_gather_u8_kernel, dim3(aug_gx(w), h, batch), dim3(256), 0, (hipStream_t)stream, h0, w0, h, w, in, ymap, xmap, flip, invert_x_on_flip, out);
    return check_launch("aug_gather_u8");
}
