// prepost.hip — pre- / post-processing of `MarigoldPipeline.__call__` / `DepthNormalEstimationPipeline.__call__` on the device:
//   resize_max_res                 /root/reference/Marigold/marigold/util/image_util.py:79-108 (torchvision `resize(..., BILINEAR, antialias=True)`)
//   uint8 -> [-1, 1]               marigold_pipeline.py:245
//   min-max to [0, 1]              marigold_pipeline.py:301-306
//   resize back to the input size  marigold_pipeline.py:312-321
// The resize is torch's ANTIALIASED bilinear interpolation (aten `_upsample_bilinear2d_aa`, align_corners = False): a separable triangle filter
// whose support grows with the down-scaling factor, weights normalised per output position — built on the host in aten's float arithmetic
// (pipeline.aa_bilinear_tables: `_compute_indices_min_size_weights_aa`), applied here as a horizontal then a vertical pass over planar [P][H][W]
// images in fp32 (uint8 or fp32 in).  `round_u8`: the result is rounded half-to-even and clamped to [0, 255] — what torchvision does for an
// integer image — before `y = v * mul + add` (the [-1, 1] normalisation rides on the same store).  All HBM-bound, launch-latency-sized.
#include "common.h"

namespace e2eft {

// horizontal pass: in [P][h0][w0] (uint8 or fp32) -> mid [P][h0][w] fp32
template <typename TI>
__global__ __launch_bounds__(256) void aa_resample_h_kernel(int h0, int w0, int w, int ksize, const TI* __restrict__ in, const int32_t* __restrict__ bounds,
                                                            const float* __restrict__ wt, float* __restrict__ mid) {
    const int pl = blockIdx.z, y = blockIdx.y;
    const TI* row = in + ((long)pl * h0 + y) * w0;
    for (int xo = blockIdx.x * 256 + threadIdx.x; xo < w; xo += gridDim.x * 256) {
        const int xmin = bounds[xo * 2], n = bounds[xo * 2 + 1];
        const float* k = wt + (long)xo * ksize;
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += (float)row[xmin + i] * k[i];
        mid[((long)pl * h0 + y) * w + xo] = s;
    }
}

// vertical pass: mid [P][h0][w] fp32 -> out [P][h][w] fp32, optionally rounded to the uint8 grid, then v * mul + add
__global__ __launch_bounds__(256) void aa_resample_v_kernel(int h0, int w, int h, int ksize, const float* __restrict__ mid, const int32_t* __restrict__ bounds,
                                                            const float* __restrict__ wt, int round_u8, float mul, float add, float* __restrict__ out) {
    const int pl = blockIdx.z, yo = blockIdx.y;
    const int ymin = bounds[yo * 2], n = bounds[yo * 2 + 1];
    const float* k = wt + (long)yo * ksize;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += mid[((long)pl * h0 + ymin + i) * w + x] * k[i];
        if (round_u8) s = fminf(fmaxf(rintf(s), 0.f), 255.f);
        out[((long)pl * h + yo) * w + x] = s * mul + add;
    }
}

// min / max of a flat fp32 buffer in two fixed-order stages (no atomics: reproducible), then out = (x - min) / (max - min) (zeros when max == min)
__global__ __launch_bounds__(256) void minmax_partial_kernel(long n, const float* __restrict__ x, float* __restrict__ part /* [256][2] */) {
    __shared__ float smn[4], smx[4];
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ mm /* [2] */) {
    __shared__ float smn[4], smx[4];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        mn = fminf(mn, part[i * 2]);
        mx = fmaxf(mx, part[i * 2 + 1]);
    }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mm[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}
__global__ __launch_bounds__(256) void minmax_apply_kernel(long n, const float* __restrict__ x, const float* __restrict__ mm, float* __restrict__ out) {
    const float mn = mm[0], mx = mm[1];
    const float d = mx - mn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = d == 0.f ? 0.f : (x[i] - mn) / d;
}

static unsigned pp_gx(int w) { return (unsigned)((w + 255) / 256); }

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_resample_bilinear_aa(int32_t planes, int32_t h0, int32_t w0, int32_t h, int32_t w, int32_t in_is_u8, const void* in,
                                          const int32_t* xbounds, const float* xweights, int32_t xksize, const int32_t* ybounds, const float* yweights,
                                          int32_t yksize, int32_t round_u8, float mul, float add, float* mid, float* out, void* stream) {
    E2EFT_REQUIRE(in && xbounds && xweights && ybounds && yweights && mid && out, "resample_aa: null pointer");
    E2EFT_REQUIRE(planes > 0 && planes <= 65535 && h0 > 0 && w0 > 0 && h > 0 && w > 0 && h0 <= 65535 && h <= 65535 && xksize > 0 && yksize > 0, "resample_aa: shape");
    hipStream_t s = (hipStream_t)stream;
    if (in_is_u8) hipLaunchKernelGGL((aa_resample_h_kernel<uint8_t>), dim3(pp_gx(w), h0, planes), dim3(256), 0, s, h0, w0, w, xksize, (const uint8_t*)in, xbounds, xweights, mid);
    else hipLaunchKernelGGL((aa_resample_h_kernel<float>), dim3(pp_gx(w), h0, planes), dim3(256), 0, s, h0, w0, w, xksize, (const float*)in, xbounds, xweights, mid);
    hipLaunchKernelGGL(aa_resample_v_kernel, dim3(pp_gx(w), h, planes), dim3(256), 0, s, h0, w, h, yksize, (const float*)mid, ybounds, yweights, round_u8, mul, add, out);
    return check_launch("resample_aa");
}

extern "C" size_t e2eft_minmax_unit_workspace_bytes(void) { return (256 * 2 + 2) * sizeof(float); }

extern "C" int e2eft_minmax_unit(int64_t n, const float* x, float* out, float* minmax, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(x && out && workspace && n > 0, "minmax_unit: bad args");
    if (ws_bytes < e2eft_minmax_unit_workspace_bytes()) return fail(E2EFT_ERR_WORKSPACE, "minmax_unit: workspace %zu < %zu", ws_bytes, e2eft_minmax_unit_workspace_bytes());
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    float* mm = minmax ? minmax : part + 512;
    long nb = (n + 255) / 256;
    const int nparts = (int)(nb > 256 ? 256 : nb);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, s, (long)n, x, part);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, s, nparts, (const float*)part, mm);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(minmax_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)n, x, (const float*)mm, out);
    return check_launch("minmax_unit");
}
