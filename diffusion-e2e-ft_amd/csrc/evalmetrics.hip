// evalmetrics.hip — the acceptance arithmetic of BASELINE.json ("depth AbsRel within 1e-3 of the reference") on the device: least-squares
// scale / shift alignment of an affine-invariant prediction to metric depth and the ten depth metrics of the reference's evaluation loop
// (SURVEY.md §8 f4).
//   alignment   /root/reference/Marigold/src/util/alignment.py:8-56 (align_depth_least_square, optional nearest down-sampling to
//               max_resolution; disparity-space variant Marigold/eval.py:180-201)
//   clipping    Marigold/eval.py:203-209 (dataset min / max depth, then >= 1e-6)
//   metrics     Marigold/src/util/metric.py:34-158: abs_relative_difference, squared_relative_difference, rmse_linear, rmse_log, log10,
//               delta1/2/3_acc, i_rmse, silog_rmse — the reference evaluates them image by image and averages the per-image values
// HBM-bound: two passes over (prediction, ground truth, mask) per image — 9 B/pixel each — with fp64 accumulators; every reduction is
// two-stage over NBLK fixed partials (no atomics) and therefore bit-reproducible.
#include "common.h"

namespace e2eft {

constexpr int EV_NBLK = 128;      // partial blocks per image
constexpr int EV_NSUM = 5;        // alignment sums: n, sum p, sum p^2, sum g, sum p g
constexpr int EV_NMET = 12;       // metric sums (below)

__device__ __forceinline__ double ev_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int N> __device__ __forceinline__ void ev_block_store(double (&v)[N], double* dst) {
    __shared__ double red[4][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        v[i] = ev_wave_sum(v[i]);
        if (lane == 0) red[wave][i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < N) dst[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// pass 1: per-image sums of the (optionally nearest-down-sampled) valid pixels.  disparity != 0: the target is 1 / gt where gt > 0 and
// only pixels with gt > 0 and pred > 0 take part (eval.py:182-190).  Down-sampling: torch.nn.Upsample(scale_factor, "nearest") reads source
// index floor(dst * (1 / scale_factor)) (alignment.py:23-33; the host passes oh == h: see e2eft_depth_eval); oh == h, ow == w means none.
__global__ __launch_bounds__(256) void ev_align_sums_kernel(int h, int w, int oh, int ow, float inv_scale, int disparity, const float* __restrict__ pred,
                                                            const float* __restrict__ gt, const uint8_t* __restrict__ mask, double* __restrict__ part) {
    const int b = blockIdx.y;
    const long img = (long)b * h * w;
    double v[EV_NSUM] = {0, 0, 0, 0, 0};
    const int total = oh * ow;
    const bool ds = oh != h || ow != w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        long src = i;
        if (ds) {
            const int oy = i / ow, ox = i - oy * ow;
            const int sy = oh == h ? oy : min((int)floorf(oy * inv_scale), h - 1), sx = min((int)floorf(ox * inv_scale), w - 1);
            src = (long)sy * w + sx;
        }
        if (!mask[img + src]) continue;
        const float p = pred[img + src];
        float g = gt[img + src];
        if (disparity) {
            if (!(g > 0.f) || !(p > 0.f)) continue;
            g = 1.0f / g;
        }
        v[0] += 1.0; v[1] += (double)p; v[2] += (double)p * p; v[3] += (double)g; v[4] += (double)p * g;
    }
    ev_block_store<EV_NSUM>(v, part + ((long)b * gridDim.x + blockIdx.x) * EV_NSUM);
}

// closed-form least squares  [sum p^2, sum p; sum p, n] [s; t] = [sum p g; sum g]  in fp64 (numpy.linalg.lstsq solves the same normal
// problem through an SVD); a singular system (fewer than two distinct valid values) gives scale 0, shift mean(g)
__global__ void ev_solve_kernel(int batch, int nblk, const double* __restrict__ part, float* __restrict__ ss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double s[EV_NSUM] = {0, 0, 0, 0, 0};
    for (int k = 0; k < nblk; ++k)
        for (int i = 0; i < EV_NSUM; ++i) s[i] += part[((long)b * nblk + k) * EV_NSUM + i];
    const double n = s[0], sp = s[1], spp = s[2], sg = s[3], spg = s[4];
    const double det = spp * n - sp * sp;
    double sc = 0.0, sh = n > 0 ? sg / n : 0.0;
    if (n > 1 && fabs(det) > 1e-30 * fmax(spp * n, 1e-300)) {
        sc = (spg * n - sp * sg) / det;
        sh = (spp * sg - sp * spg) / det;
    }
    ss[b * 2] = (float)sc;
    ss[b * 2 + 1] = (float)sh;
}

// pass 2: aligned prediction (optionally written) and the metric sums over the valid pixels of the FULL-resolution image
//   0 n  1 sum |a-g|/g  2 sum (a-g)^2/g  3 sum (a-g)^2  4 sum d^2 (d = ln a - ln g)  5 sum d  6 sum |log10 a - log10 g|
//   7..9 counts max(a/g, g/a) < 1.25^k  10 sum (1/a - 1/g)^2  11 unused
__global__ __launch_bounds__(256) void ev_metric_sums_kernel(int hw, int disparity, float dmin, float dmax, const float* __restrict__ pred,
                                                             const float* __restrict__ gt, const uint8_t* __restrict__ mask, const float* __restrict__ ss,
                                                             float* __restrict__ aligned_out, double* __restrict__ part) {
    const int b = blockIdx.y;
    const long img = (long)b * hw;
    const float sc = ss[b * 2], sh = ss[b * 2 + 1];
    double v[EV_NMET] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float a = pred[img + i] * sc + sh;                       // alignment.py:50 (float32 arithmetic as numpy's)
        if (disparity) {                                         // eval.py:193-201: clip the disparity at 1e-3, back to depth
            a = fmaxf(a, 1e-3f);
            a = 1.0f / a;
        }
        a = fminf(fmaxf(a, dmin), dmax);                         // eval.py:203-206
        a = fmaxf(a, 1e-6f);                                     // eval.py:209
        if (aligned_out) aligned_out[img + i] = a;
        if (!mask[img + i]) continue;
        const float g = gt[img + i];
        const float df = a - g;
        const float la = logf(a), lg = logf(g);
        const float d = la - lg;
        const float r = fmaxf(a / g, g / a);
        const float id = 1.0f / a - 1.0f / g;
        v[0] += 1.0;
        v[1] += (double)(fabsf(df) / g);
        v[2] += (double)(df * df / g);
        v[3] += (double)(df * df);
        v[4] += (double)(d * d);
        v[5] += (double)d;
        v[6] += (double)fabsf(log10f(a) - log10f(g));
        v[7] += r < 1.25f ? 1.0 : 0.0;
        v[8] += r < 1.5625f ? 1.0 : 0.0;
        v[9] += r < 1.953125f ? 1.0 : 0.0;
        v[10] += (double)(id * id);
    }
    ev_block_store<EV_NMET>(v, part + ((long)b * gridDim.x + blockIdx.x) * EV_NMET);
}

// per image: abs_rel, sq_rel, rmse, rmse_log, log10, delta1, delta2, delta3, i_rmse, silog, scale, shift  (metric.py; order of eval.py's table)
__global__ void ev_finalize_kernel(int batch, int nblk, const double* __restrict__ part, const float* __restrict__ ss, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double s[EV_NMET];
    for (int i = 0; i < EV_NMET; ++i) s[i] = 0;
    for (int k = 0; k < nblk; ++k)
        for (int i = 0; i < EV_NMET; ++i) s[i] += part[((long)b * nblk + k) * EV_NMET + i];
    const double n = s[0];
    float* o = out + (long)b * 12;
    o[0] = (float)(s[1] / n);
    o[1] = (float)(s[2] / n);
    o[2] = (float)sqrt(s[3] / n);
    o[3] = (float)sqrt(s[4] / n);
    o[4] = (float)(s[6] / n);
    o[5] = (float)(s[7] / n);
    o[6] = (float)(s[8] / n);
    o[7] = (float)(s[9] / n);
    o[8] = (float)sqrt(s[10] / n);
    o[9] = (float)(sqrt(s[4] / n - (s[5] * s[5]) / (n * n)) * 100.0);
    o[10] = ss[b * 2];
    o[11] = ss[b * 2 + 1];
}

}  // namespace e2eft

using namespace e2eft;

extern "C" size_t e2eft_depth_eval_workspace_bytes(int32_t batch) {
    return batch > 0 ? (size_t)batch * EV_NBLK * EV_NMET * sizeof(double) + (size_t)batch * 2 * sizeof(float) : 0;
}

extern "C" int e2eft_depth_eval(int32_t batch, int32_t height, int32_t width, const float* pred, const float* gt, const uint8_t* mask, int32_t disparity,
                                int32_t align_max_res, float min_depth, float max_depth, float* out_metrics, float* aligned_out, void* workspace,
                                size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(pred && gt && mask && out_metrics && workspace, "depth_eval: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && height > 0 && width > 0 && (long)height * width < 0x7fffffffL, "depth_eval: shape");
    E2EFT_REQUIRE(min_depth > 0.f && max_depth > min_depth, "depth_eval: depth range");
    const size_t need = e2eft_depth_eval_workspace_bytes(batch);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "depth_eval: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(((uintptr_t)workspace & 7) == 0, "depth_eval: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* ss = (float*)(part + (size_t)batch * EV_NBLK * EV_NMET);
    int oh = height, ow = width;
    float inv_scale = 1.0f;
    if (align_max_res > 0) {   // alignment.py:23-33: scale = min(max_resolution / (H, W)), applied when < 1 through torch.nn.Upsample(scale_factor,
        // "nearest") on a [1, H, W] tensor — a 3-D input, i.e. (N, C, L): the reference down-samples the WIDTH only, rows are kept.
        // Reproduced as is: ow = floor(W * scale), source column floor(x / scale), oh = H.
        const double sf = fmin((double)align_max_res / height, (double)align_max_res / width);
        if (sf < 1.0) {
            ow = (int)floor(width * sf);
            inv_scale = (float)(1.0 / sf);
            E2EFT_REQUIRE(ow > 0, "depth_eval: align_max_res too small");
        }
    }
    const int hw = height * width;
    hipLaunchKernelGGL(ev_align_sums_kernel, dim3(EV_NBLK, batch), dim3(256), 0, s, height, width, oh, ow, inv_scale, disparity, pred, gt, mask, part);
    hipLaunchKernelGGL(ev_solve_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, s, batch, EV_NBLK, part, ss);
    hipLaunchKernelGGL(ev_metric_sums_kernel, dim3(EV_NBLK, batch), dim3(256), 0, s, hw, disparity, min_depth, max_depth, pred, gt, mask, ss, aligned_out, part);
    hipLaunchKernelGGL(ev_finalize_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, s, batch, EV_NBLK, part, ss, out_metrics);
    return check_launch("depth_eval");
}
