// dataprep.hip — training-sample preparation on the GPU: what Hypersim.__getitem__ / VirtualKITTI2.__getitem__ do to a
// decoded, augmented sample (/root/reference/training/dataloaders/load.py:236-283 and :342-375, the same arithmetic twice):
//   valid = near < depth < far;  (lo, hi) = 2 % / 98 % quantiles of the valid depths (torch.quantile, linear interpolation);
//   depth -> clamp to [lo, hi], invalid pixels = hi (the metric target), then mapped to [-1, 1] and stacked x3 for the VAE;
//   rgb, normals -> [-1, 1], normals renormalised, invalid pixels zeroed; lo == hi or no valid pixel -> zeros and an empty mask.
// The quantiles are EXACT order statistics: a three-level radix select (11 + 11 + 10 key bits) over integer histograms, four ranks
// per image at once (below / above neighbour of both quantiles) — integer atomics only, so the result is deterministic.
#include "common.h"
#include <math.h>

namespace e2eft {

constexpr int DP_RANKS = 4;
constexpr int DP_BINS = 2048;

struct DpSel {            // per (image, rank) selection state carried between levels
    unsigned prefix;      // key bits fixed so far (left-aligned)
    unsigned remaining;   // rank inside the elements that share the prefix
};
struct DpHead {           // per image
    unsigned count;       // number of valid pixels
    float w_lo, w_hi;     // interpolation weights of the two quantiles
};

__device__ __forceinline__ unsigned dp_key(float v) {   // order-preserving float -> uint32
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dp_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// level 0: count + histogram of the top 11 key bits of the valid pixels.  grid (blocks, B)
__global__ __launch_bounds__(256) void dp_hist0(long n, float near, float far, const float* __restrict__ depth, unsigned* __restrict__ hist /* [B][2048] */) {
    __shared__ unsigned h[DP_BINS];
    for (int i = threadIdx.x; i < DP_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const float* d = depth + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = d[i];
        if (v > near && v < far) atomicAdd(&h[dp_key(v) >> 21], 1u);
    }
    __syncthreads();
    unsigned* g = hist + (long)blockIdx.y * DP_BINS;
    for (int i = threadIdx.x; i < DP_BINS; i += 256)
        if (h[i]) atomicAdd(&g[i], h[i]);
}

// one block per image: total count, the four ranks (torch.quantile: rank = q * (n - 1) in float32, floor / ceil neighbours, weight =
// fractional part), then the bin of every rank in the level-0 histogram
__global__ __launch_bounds__(256) void dp_scan0(float q_lo, float q_hi, const unsigned* __restrict__ hist, DpHead* __restrict__ head,
                                                DpSel* __restrict__ sel /* [B][4] */) {
    __shared__ unsigned cum[DP_BINS];
    const unsigned* g = hist + (long)blockIdx.x * DP_BINS;
    for (int i = threadIdx.x; i < DP_BINS; i += 256) cum[i] = g[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < DP_BINS; ++i) { const unsigned c = cum[i]; cum[i] = run; run += c; }   // exclusive prefix (2048 adds, once per image)
        DpHead hd;
        hd.count = run;
        unsigned ranks[DP_RANKS] = {0, 0, 0, 0};
        hd.w_lo = hd.w_hi = 0.f;
        if (run > 0) {
            const float last = (float)(run - 1);
            const float r_lo = q_lo * last, r_hi = q_hi * last;
            const float f_lo = floorf(r_lo), f_hi = floorf(r_hi);
            hd.w_lo = r_lo - f_lo;
            hd.w_hi = r_hi - f_hi;
            ranks[0] = (unsigned)f_lo; ranks[1] = (unsigned)ceilf(r_lo);
            ranks[2] = (unsigned)f_hi; ranks[3] = (unsigned)ceilf(r_hi);
        }
        head[blockIdx.x] = hd;
        for (int r = 0; r < DP_RANKS; ++r) {
            int lo = 0, hi = DP_BINS - 1;            // last bin whose exclusive prefix <= rank
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cum[mid] <= ranks[r]) lo = mid; else hi = mid - 1;
            }
            DpSel s;
            s.prefix = (unsigned)lo << 21;
            s.remaining = ranks[r] - cum[lo];
            sel[blockIdx.x * DP_RANKS + r] = s;
        }
    }
}

// levels 1 / 2: histogram of the next SHIFT-window of key bits among the pixels matching each rank's prefix.  grid (blocks, B)
template <int LEVEL>
__global__ __launch_bounds__(256) void dp_hist(long n, float near, float far, const float* __restrict__ depth, const DpSel* __restrict__ sel,
                                               unsigned* __restrict__ hist /* [B][4][2048] */) {
    constexpr unsigned PMASK = LEVEL == 1 ? 0xFFE00000u : 0xFFFFFC00u;   // 11 / 22 fixed bits
    constexpr int SHIFT = LEVEL == 1 ? 10 : 0;
    constexpr unsigned BMASK = LEVEL == 1 ? 0x7FFu : 0x3FFu;
    __shared__ unsigned h[DP_RANKS][DP_BINS];
    for (int i = threadIdx.x; i < DP_RANKS * DP_BINS; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    unsigned pre[DP_RANKS];
#pragma unroll
    for (int r = 0; r < DP_RANKS; ++r) pre[r] = sel[blockIdx.y * DP_RANKS + r].prefix;
    const float* d = depth + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = d[i];
        if (v > near && v < far) {
            const unsigned k = dp_key(v);
#pragma unroll
            for (int r = 0; r < DP_RANKS; ++r)
                if ((k & PMASK) == pre[r]) atomicAdd(&h[r][(k >> SHIFT) & BMASK], 1u);
        }
    }
    __syncthreads();
    unsigned* g = hist + (long)blockIdx.y * DP_RANKS * DP_BINS;
    for (int i = threadIdx.x; i < DP_RANKS * DP_BINS; i += 256) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&g[i], c);
    }
}

// one block per image, one wave per rank: locate the rank inside its histogram, extend the prefix.  The last level turns the four keys
// into the two interpolated quantiles (torch.lerp's two-branch formula) and the ok flag.
template <int LEVEL>
__global__ __launch_bounds__(256) void dp_scan(const unsigned* __restrict__ hist, DpSel* __restrict__ sel, const DpHead* __restrict__ head,
                                               float* __restrict__ out /* [B][4] = lo, hi, count, ok */) {
    constexpr int SHIFT = LEVEL == 1 ? 10 : 0;
    constexpr int NB = LEVEL == 1 ? 2048 : 1024;
    __shared__ float val[DP_RANKS];
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned* g = hist + ((long)blockIdx.x * DP_RANKS + r) * DP_BINS;
    DpSel s = sel[blockIdx.x * DP_RANKS + r];
    if (lane == 0) {
        unsigned run = 0;
        int b = 0;
        for (; b < NB; ++b) {
            const unsigned c = g[b];
            if (run + c > s.remaining) break;
            run += c;
        }
        if (b == NB) b = NB - 1;   // empty image: nothing selected, value unused
        s.prefix |= (unsigned)b << SHIFT;
        s.remaining -= run;
        sel[blockIdx.x * DP_RANKS + r] = s;
        if (LEVEL == 2) val[r] = dp_unkey(s.prefix);
    }
    if (LEVEL == 2) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const DpHead hd = head[blockIdx.x];
            auto lerp = [](float a, float b, float w) { return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w); };
            float lo = 0.f, hi = 0.f;
            if (hd.count > 0) {
                lo = lerp(val[0], val[1], hd.w_lo);
                hi = lerp(val[2], val[3], hd.w_hi);
            }
            float* o = out + blockIdx.x * 4;
            o[0] = lo; o[1] = hi; o[2] = (float)hd.count; o[3] = (hd.count > 0 && lo != hi) ? 1.f : 0.f;
        }
    }
}

// the fused elementwise pass.  grid (blocks, B)
__global__ __launch_bounds__(256) void dp_prepare(long hw, float near, float far, const float* __restrict__ rgb, const float* __restrict__ depth,
                                                  const float* __restrict__ normal, const float* __restrict__ quant, float* __restrict__ o_rgb,
                                                  float* __restrict__ o_depth3, float* __restrict__ o_metric, float* __restrict__ o_normal,
                                                  uint8_t* __restrict__ o_mask) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const float lo = quant[b * 4], hi = quant[b * 4 + 1];
    const bool ok = quant[b * 4 + 3] != 0.f;
    const float range = hi - lo;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long)gridDim.x * 256) {
        const float d = depth[(long)b * hw + p];
        const bool valid = ok && d > near && d < far;                       // load.py:236, :246 (mask emptied when lo == hi)
        float metric = 0.f, dn = 0.f;
        if (ok) {
            metric = fminf(fmaxf(d, lo), hi);                              // :249  clamp
            if (!valid) metric = hi;                                       // :250  invalid -> relative far plane
            dn = ((metric - lo) / range) * 2.0f - 1.0f;                    // :252
            dn = fminf(fmaxf(dn, -1.f), 1.f);
        }
        o_metric[(long)b * hw + p] = metric;
        o_mask[(long)b * hw + p] = valid ? 1 : 0;
        float n[3];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long i = ((long)b * 3 + c) * hw + p;
            o_rgb[i] = rgb[i] * 2.0f - 1.0f;                               // :239
            o_depth3[i] = dn;                                              // :256  three identical channels
            n[c] = normal[i] * 2.0f - 1.0f;                                // :259
            ss += n[c] * n[c];
        }
        const float inv = fmaxf(sqrtf(ss), 1e-12f);                        // F.normalize(p=2, dim=0), eps = 1e-12
#pragma unroll
        for (int c = 0; c < 3; ++c) o_normal[((long)b * 3 + c) * hw + p] = valid ? n[c] / inv : 0.f;   // :260-264
    }
}

// Hypersim.__getitem__'s normal orientation fix (load.py:225-232 with align_normals / creat_uv_mesh, :185-204; Hypersim's stored normals do not always
// face the camera), on the DECODED uint8 normal image and the metric depth, in the reference's float64 arithmetic operation for operation (no contraction):
//   n = u8 / 255 * 2 - 1;  n.y, n.z *= -1;  P = depth * (invK [x, y, 1]^T);  if (n . P > 0) n = -n;  n = -n;  u8' = trunc((n + 1) / 2 * 255).
// One thread per pixel; 7 B read, 3 B written.
struct DpInvK { double m[9]; };
__global__ __launch_bounds__(256) void dp_align_normals(int h, int w, const uint8_t* __restrict__ nrm, const float* __restrict__ depth, DpInvK ik,
                                                        uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
    const long hw = (long)h * w;
    const long b = blockIdx.y;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long)gridDim.x * 256) {
        const int y = (int)(p / w), x = (int)(p - (long)y * w);
        const uint8_t* src = nrm + (b * hw + p) * 3;
        double n[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) n[c] = ((double)src[c] / 255.0) * 2.0 - 1.0;
        n[1] *= -1.0;
        n[2] *= -1.0;
        const double d = (double)depth[b * hw + p];
        double dot = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double pr = (ik.m[3 * r] * (double)x + ik.m[3 * r + 1] * (double)y) + ik.m[3 * r + 2] * 1.0;      // (inv_K @ xy)[r], k in order
            const double t = n[r] * (d * pr);
            dot = r == 0 ? t : dot + t;
        }
        const double sgn = dot > 0.0 ? 1.0 : -1.0;          // flipped by the mask, then the whole map times -1: net +1 where the mask is set, -1 elsewhere
        uint8_t* dst = out + (b * hw + p) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = n[c] * sgn;
            dst[c] = (uint8_t)(int)(((v + 1.0) / 2.0) * 255.0);      // .astype(np.uint8): truncation
        }
    }
}

static size_t dp_ws_bytes(int batch) {
    return (size_t)batch * (DP_BINS + 2 * DP_RANKS * DP_BINS) * sizeof(unsigned) + (size_t)batch * (sizeof(DpHead) + DP_RANKS * sizeof(DpSel));
}

}  // namespace e2eft

using namespace e2eft;

extern "C" size_t e2eft_masked_quantiles_workspace_bytes(int32_t batch) {
    if (batch < 1) {
        fail(E2EFT_ERR_BAD_ARG, "masked_quantiles: batch = %d", (int)batch);
        return 0;
    }
    return dp_ws_bytes(batch);
}

extern "C" int e2eft_masked_quantiles(int32_t batch, int64_t n, const float* depth, float near_plane, float far_plane, float q_lo, float q_hi,
                                      float* out, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(batch >= 1 && n >= 1 && n < 2147483647L && depth && out && workspace, "masked_quantiles: bad arguments");
    E2EFT_REQUIRE(q_lo >= 0.f && q_lo <= 1.f && q_hi >= 0.f && q_hi <= 1.f, "masked_quantiles: quantiles must lie in [0, 1]");
    E2EFT_REQUIRE(ws_bytes >= dp_ws_bytes(batch), "masked_quantiles: workspace %zu < %zu bytes", ws_bytes, dp_ws_bytes(batch));
    hipStream_t s = (hipStream_t)stream;
    unsigned* h0 = (unsigned*)workspace;
    unsigned* h1 = h0 + (size_t)batch * DP_BINS;
    unsigned* h2 = h1 + (size_t)batch * DP_RANKS * DP_BINS;
    DpHead* head = (DpHead*)(h2 + (size_t)batch * DP_RANKS * DP_BINS);
    DpSel* sel = (DpSel*)(head + batch);
    if (hipMemsetAsync(workspace, 0, (size_t)batch * (DP_BINS + 2 * DP_RANKS * DP_BINS) * sizeof(unsigned), s) != hipSuccess)
        return fail(E2EFT_ERR_LAUNCH, "masked_quantiles: memset failed");
    const int blocks = cdiv(n, 256 * 16) < 256 ? cdiv(n, 256 * 16) : 256;
    hipLaunchKernelGGL(dp_hist0, dim3(blocks, batch), dim3(256), 0, s, (long)n, near_plane, far_plane, depth, h0);
    hipLaunchKernelGGL(dp_scan0, dim3(batch), dim3(256), 0, s, q_lo, q_hi, (const unsigned*)h0, head, sel);
    hipLaunchKernelGGL((dp_hist<1>), dim3(blocks, batch), dim3(256), 0, s, (long)n, near_plane, far_plane, depth, (const DpSel*)sel, h1);
    hipLaunchKernelGGL((dp_scan<1>), dim3(batch), dim3(256), 0, s, (const unsigned*)h1, sel, (const DpHead*)head, out);
    hipLaunchKernelGGL((dp_hist<2>), dim3(blocks, batch), dim3(256), 0, s, (long)n, near_plane, far_plane, depth, (const DpSel*)sel, h2);
    hipLaunchKernelGGL((dp_scan<2>), dim3(batch), dim3(256), 0, s, (const unsigned*)h2, sel, (const DpHead*)head, out);
    return check_launch("masked_quantiles");
}

extern "C" int e2eft_prepare_sample(int32_t batch, int64_t hw, const float* rgb01, const float* depth, const float* normal01, float near_plane,
                                    float far_plane, const float* quantiles, float* rgb, float* depth3, float* metric, float* normals,
                                    uint8_t* val_mask, void* stream) {
    E2EFT_REQUIRE(batch >= 1 && hw >= 1 && rgb01 && depth && normal01 && quantiles && rgb && depth3 && metric && normals && val_mask,
                  "prepare_sample: bad arguments");
    const int blocks = cdiv(hw, 256 * 4) < 1024 ? cdiv(hw, 256 * 4) : 1024;
    hipLaunchKernelGGL(dp_prepare, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, (long)hw, near_plane, far_plane, rgb01, depth, normal01,
                       quantiles, rgb, depth3, metric, normals, val_mask);
    return check_launch("prepare_sample");
}

extern "C" int e2eft_align_normals_u8(int32_t batch, int32_t h, int32_t w, const uint8_t* normal_u8, const float* depth, const double* inv_k,
                                      uint8_t* out, void* stream) {
    E2EFT_REQUIRE(batch >= 1 && batch <= 65535 && h >= 1 && w >= 1 && normal_u8 && depth && inv_k && out, "align_normals: bad arguments");
    DpInvK ik;
    for (int i = 0; i < 9; ++i) ik.m[i] = inv_k[i];
    const long hw = (long)h * w;
    const int blocks = cdiv(hw, 256) < 4096 ? (int)cdiv(hw, 256) : 4096;
    hipLaunchKernelGGL(dp_align_normals, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, (int)h, (int)w, normal_u8, depth, ik, out);
    return check_launch("align_normals");
}
