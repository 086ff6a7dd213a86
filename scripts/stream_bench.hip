// stream_bench.hip — how fast can a GroupNorm-apply-shaped stream (read fp16 NHWC, per-channel affine + SiLU, write fp16) go on MI355X,
// next to a plain 16-byte copy of the same bytes?  Standalone: hipcc --offload-arch=gfx950 -O3 -o scripts/bin/stream_bench scripts/stream_bench.hip
// Prints one line per variant: milliseconds (median of 20) and TB/s counting bytes read + written.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float silu_f(float t) { return t / (1.f + __expf(-t)); }

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(long n16, const u32x4* __restrict__ x, u32x4* __restrict__ y) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) y[i] = x[i];
}

// GroupNorm-apply stream, C = 128 halves per pixel (16 chunks of 16 B): thread <-> (chunk column tid & 15, pixel lane tid >> 4).
// MODE 0: the product's geometry (grid = slabs x images, 4 loads in flight).  MODE 1: grid-stride over 16-pixel groups, U loads in flight.
template <int U, bool NT, bool SILU, int MODE>
__global__ __launch_bounds__(256) void gn_kernel(int batch, int hw, int slab, const __half* __restrict__ x, const float* __restrict__ ad,
                                                 const __half* __restrict__ beta, __half* __restrict__ y) {
    constexpr int C = 128;
    const int chl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = chl * 8;
    auto params = [&](int b, float* a, float* mu, float* be) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = ad[((long)b * C + c + e) * 2];
            mu[e] = ad[((long)b * C + c + e) * 2 + 1];
            be[e] = __half2float(beta[c + e]);
        }
    };
    auto one = [&](const u32x4& v, const float* a, const float* mu, const float* be) {
        u32x4 o;
        const __half* hv = reinterpret_cast<const __half*>(&v);
        __half* ho = reinterpret_cast<__half*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = fmaf(__half2float(hv[e]) - mu[e], a[e], be[e]);
            if (SILU) t = silu_f(t);
            ho[e] = __float2half(t);
        }
        return o;
    };
    if (MODE == 0) {
        const int b = blockIdx.y;
        float a[8], mu[8], be[8];
        params(b, a, mu, be);
        const int p0 = blockIdx.x * slab, p1 = min(p0 + slab, hw);
        int pix = p0 + pl;
        for (; pix + (U - 1) * 16 < p1; pix += U * 16) {
            const long row = (long)b * hw + pix;
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(x + (row + u * 16) * C + c);
#pragma unroll
            for (int u = 0; u < U; ++u) *reinterpret_cast<u32x4*>(y + (row + u * 16) * C + c) = one(v[u], a, mu, be);
        }
        for (; pix < p1; pix += 16) {
            const long row = (long)b * hw + pix;
            *reinterpret_cast<u32x4*>(y + row * C + c) = one(*reinterpret_cast<const u32x4*>(x + row * C + c), a, mu, be);
        }
    } else {
        // persistent: image by image (parameters reloaded per image), 16-pixel groups strided by the grid
        const int groups = hw / 16;
        for (int b = 0; b < batch; ++b) {
            float a[8], mu[8], be[8];
            params(b, a, mu, be);
            int g = blockIdx.x;
            for (; g + (U - 1) * (int)gridDim.x < groups; g += U * gridDim.x) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long row = (long)b * hw + (long)(g + u * gridDim.x) * 16 + pl;
                    const u32x4* src = reinterpret_cast<const u32x4*>(x + row * C + c);
                    v[u] = NT ? __builtin_nontemporal_load(src) : *src;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long row = (long)b * hw + (long)(g + u * gridDim.x) * 16 + pl;
                    u32x4* dst = reinterpret_cast<u32x4*>(y + row * C + c);
                    const u32x4 o = one(v[u], a, mu, be);
                    if (NT) __builtin_nontemporal_store(o, dst); else *dst = o;
                }
            }
            for (; g < groups; g += gridDim.x) {
                const long row = (long)b * hw + (long)g * 16 + pl;
                *reinterpret_cast<u32x4*>(y + row * C + c) = one(*reinterpret_cast<const u32x4*>(x + row * C + c), a, mu, be);
            }
        }
    }
}

template <typename F> static float time_ms(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int B = 8, HW = 768 * 768, C = 128;
    const long n = (long)B * HW * C;
    const double bytes = 2.0 * n * 2;
    __half *x, *y, *beta; float* ad;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&beta, C * 2)); CK(hipMalloc(&ad, (long)B * C * 2 * 4));
    CK(hipMemset(x, 0x3c, n * 2)); CK(hipMemset(beta, 0, C * 2));
    std::vector<float> had(B * C * 2, 1.0f);
    CK(hipMemcpy(ad, had.data(), had.size() * 4, hipMemcpyHostToDevice));
    auto report = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    const long n16 = n / 8;
    for (int grid : {2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, 96, "copy 16B U=1 grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((copy_kernel<1, false>), dim3(grid), dim3(256), 0, 0, n16, (const u32x4*)x, (u32x4*)y); }));
        snprintf(nm, 96, "copy 16B U=4 grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(grid), dim3(256), 0, 0, n16, (const u32x4*)x, (u32x4*)y); }));
        snprintf(nm, 96, "copy 16B U=8 grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((copy_kernel<8, false>), dim3(grid), dim3(256), 0, 0, n16, (const u32x4*)x, (u32x4*)y); }));
        snprintf(nm, 96, "copy 16B U=4 nontemporal grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(grid), dim3(256), 0, 0, n16, (const u32x4*)x, (u32x4*)y); }));
    }
    {
        const int pl = 16; long ns = (HW + pl * 16 - 1) / (pl * 16); if (ns > 1024) ns = 1024;
        const int slab = (int)((HW + ns - 1) / ns), nslabs = (HW + slab - 1) / slab;
        report("gn product geometry (1024 slabs x 8) silu", time_ms([&] { hipLaunchKernelGGL((gn_kernel<4, false, true, 0>), dim3(nslabs, B), dim3(256), 0, 0, B, HW, slab, x, ad, beta, y); }));
        report("gn product geometry no silu", time_ms([&] { hipLaunchKernelGGL((gn_kernel<4, false, false, 0>), dim3(nslabs, B), dim3(256), 0, 0, B, HW, slab, x, ad, beta, y); }));
        report("gn product geometry U=8 silu", time_ms([&] { hipLaunchKernelGGL((gn_kernel<8, false, true, 0>), dim3(nslabs, B), dim3(256), 0, 0, B, HW, slab, x, ad, beta, y); }));
        for (int sl : {144, 288, 2304, 9216}) {
            char nm[96];
            const int nsl = (HW + sl - 1) / sl;
            snprintf(nm, 96, "gn slab=%d px (%d slabs x 8) U=4 silu", sl, nsl);
            report(nm, time_ms([&] { hipLaunchKernelGGL((gn_kernel<4, false, true, 0>), dim3(nsl, B), dim3(256), 0, 0, B, HW, sl, x, ad, beta, y); }));
        }
    }
    for (int grid : {1024, 2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, 96, "gn grid-stride U=4 silu grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((gn_kernel<4, false, true, 1>), dim3(grid), dim3(256), 0, 0, B, HW, 0, x, ad, beta, y); }));
        snprintf(nm, 96, "gn grid-stride U=8 silu grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((gn_kernel<8, false, true, 1>), dim3(grid), dim3(256), 0, 0, B, HW, 0, x, ad, beta, y); }));
        snprintf(nm, 96, "gn grid-stride U=8 silu nontemporal grid=%d", grid); report(nm, time_ms([&] { hipLaunchKernelGGL((gn_kernel<8, true, true, 1>), dim3(grid), dim3(256), 0, 0, B, HW, 0, x, ad, beta, y); }));
    }
    return 0;
}
