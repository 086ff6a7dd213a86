// clock_probe.hip — which shader clock does MI355X deliver under a sustained MFMA load?  Every workgroup times itself with BOTH counters
// (s_memtime = shader cycles, s_memrealtime = constant 100 MHz), so the answer needs no profiler (the 1.5 GHz figure in
// DESIGN.md came from a PMC-profiled run).  Variants: v_mfma_f32_32x32x16_f16 only; the same plus ds_read_b128 fragment traffic.
// hipcc --offload-arch=gfx950 -O3 -o scripts/bin/clock_probe scripts/clock_probe.hip ; run: scripts/bin/clock_probe [milliseconds per launch ~]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(512) void probe(int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned int*>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 fa = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, fb = fa;
    const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            fa = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + it * 1024) & (64 * 1024 - 16)));
            fb = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + it * 1024 + 32768) & (64 * 1024 - 16)));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), acc[a], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) s += acc[a][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    long long* d; float* sink;
    hipMalloc(&d, 256 * 2 * sizeof(long long)); hipMalloc(&sink, 4);
    std::vector<long long> h(512);
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL((probe<false>), dim3(256), dim3(512), 0, 0, iters, d, sink);
            else hipLaunchKernelGGL((probe<true>), dim3(256), dim3(512), 0, 0, iters, d, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, 512 * sizeof(long long), hipMemcpyDeviceToHost);
            std::vector<double> mhz;
            for (int b = 0; b < 256; ++b) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
            std::sort(mhz.begin(), mhz.end());
            const double flops = 256.0 * 8 * iters * 4 * 2.0 * 32 * 32 * 16;
            printf("%s  launch %d: %.1f ms, %.0f TFLOP/s, shader clock median %.0f MHz (min %.0f max %.0f)\n", variant ? "mfma + ds_read_b128" : "mfma only          ", rep, ms,
                   flops / (ms * 1e-3) / 1e12, mhz[128], mhz[0], mhz[255]);
        }
    }
    return 0;
}
