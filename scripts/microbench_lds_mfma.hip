// Microbenchmark (experiment, not part of the library): does ds_read_b128 fragment traffic overlap with MFMA execution?
// One workgroup per CU, 8 waves (2 per SIMD), loop of "k-tiles": [barrier] + R fragment reads (ds_read_b128) + M MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench_lds_mfma.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int READS, int MFMAS, int BARRIER, int WTM, int DMA>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc, const char* gsrc) {
    __shared__ __attribute__((aligned(16))) char smem[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147456 / 4; i += 512) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    floatx16 acc[WTM][2];
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int l31 = lane & 31, h = lane >> 5, sw = (l31 >> 1) & 7;
    int off[4];
    for (int g = 0; g < 4; ++g) off[g] = ((wave >> 1) * 64 + l31) * 128 + (((g * 2 + h) ^ sw) * 16);
    u32x4 stg[8];
    for (int q = 0; q < 8; ++q) stg[q] = u32x4{0u, 0u, 0u, 0u};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char* sb = smem + (it % 3) * 49152;
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (DMA < 0) {   // register staging: write last iteration's loads to LDS, issue this iteration's loads
            char* dst = smem + ((it + 2) % 3) * 49152 + wave * 1024 + lane * 16;
            const char* src = gsrc + (size_t)blockIdx.x * 65536 + wave * 1024 + lane * 16 + (it & 1) * 64;
#pragma unroll
            for (int q = 0; q < -DMA; ++q) *(u32x4*)(dst + q * 8192) = stg[q];
#pragma unroll
            for (int q = 0; q < -DMA; ++q) stg[q] = *(const u32x4*)(src + q * 8192);
        }
        if (DMA > 0) {   // DMA pieces per wave per iteration (1 KiB each) from an L2-resident source into the ring
            char* dst = smem + ((it + 2) % 3) * 49152 + __builtin_amdgcn_readfirstlane(wave) * 1024;
            const char* src = gsrc + (size_t)blockIdx.x * 49152 + __builtin_amdgcn_readfirstlane(wave) * 1024 + lane * 16;
#pragma unroll
            for (int q = 0; q < DMA; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 8192),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        u32x4 fa[4][WTM], fb[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (READS) {
#pragma unroll
                for (int a = 0; a < WTM; ++a) fa[g][a] = *(const u32x4*)(sb + off[g] + a * 4096);
                fb[g][0] = *(const u32x4*)(sb + 32768 + off[g] % 8192);
                fb[g][1] = *(const u32x4*)(sb + 32768 + off[g] % 8192 + 4096);
            } else {
#pragma unroll
                for (int a = 0; a < WTM; ++a) fa[g][a] = u32x4{(unsigned)it, 1u, 2u, 3u};
                fb[g][0] = u32x4{1u, (unsigned)it, 2u, 3u}; fb[g][1] = fb[g][0];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MFMAS) {
#pragma unroll
                for (int a = 0; a < WTM; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[g][a]), __builtin_bit_cast(half8, fb[g][b]), acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int a = 0; a < WTM; ++a) asm volatile("" ::"v"(fa[g][a]), "v"(fb[g][0]), "v"(fb[g][1]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int M, int B, int W, int DM = 0> void run(const char* name, float* out, long long* cyc, const char* gsrc = nullptr) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<R, M, B, W, DM>), dim3(blocks), dim3(512), 0, 0, out, 10, cyc, gsrc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, M, B, W, DM>), dim3(blocks), dim3(512), 0, 0, out, iters, cyc, gsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
    const double mf = (double)W * 2 * 4;  // MFMAs per wave per iteration
    printf("%-34s %8.3f ms  %7.0f cycles/iter (wave tile %dx64: %g MFMA/wave/iter = %g pipe cycles per SIMD)  eff clock %.2f GHz\n", name, ms, avg / iters,
           W * 32, mf, mf * 32 * 2, avg / (ms * 1e6));
}

// 16-wave workgroup: waves 0-7 consume (fragment reads one group ahead + MFMAs, <= 128 VGPRs), waves 8-15 only issue the
// LDS-DMA (NDMA pieces each) — the split the igemm kernel would use.
template <int NDMA, int NLOAD>
__global__ __launch_bounds__((8 + NLOAD) * 64) void k16(float* out, int iters, long long* cyc, const char* gsrc) {
    __shared__ __attribute__((aligned(16))) char smem[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 147456 / 4; i += (8 + NLOAD) * 64) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (wave >= 8) {
        const int lw = wave - 8;
        for (int it = 0; it < iters; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            char* dst = smem + ((it + 2) % 3) * 49152 + lw * 1024;
            const char* src = gsrc + (size_t)blockIdx.x * 65536 + lw * 1024 + lane * 16;
#pragma unroll
            for (int q = 0; q < NDMA; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (q % 6) * 8192),
                                                 (__attribute__((address_space(3))) void*)(dst + (q % 6) * 8192), 16, 0, 0);
        }
        if (tid == 512) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
        return;
    }
    floatx16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int l31 = lane & 31, h = lane >> 5, sw = (l31 >> 1) & 7;
    int off[4];
    for (int g = 0; g < 4; ++g) off[g] = ((wave >> 1) * 64 + l31) * 128 + (((g * 2 + h) ^ sw) * 16);
    for (int it = 0; it < iters; ++it) {
        const char* sb = smem + (it % 3) * 49152;
        __builtin_amdgcn_s_barrier();
        u32x4 fa[2][2], fb[2][2];
        auto rd = [&](int g, int s) {
            fa[s][0] = *(const u32x4*)(sb + off[g]); fa[s][1] = *(const u32x4*)(sb + off[g] + 4096);
            fb[s][0] = *(const u32x4*)(sb + 32768 + off[g] % 8192); fb[s][1] = *(const u32x4*)(sb + 32768 + off[g] % 8192 + 4096);
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) rd(g + 1, (g + 1) & 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[g & 1][a]), __builtin_bit_cast(half8, fb[g & 1][b]), acc[a][b], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NDMA, int NLOAD> void run16(const char* name, float* out, long long* cyc, const char* gsrc) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k16<NDMA, NLOAD>), dim3(blocks), dim3((8 + NLOAD) * 64), 0, 0, out, 10, cyc, gsrc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NDMA, NLOAD>), dim3(blocks), dim3((8 + NLOAD) * 64), 0, 0, out, iters, cyc, gsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
    printf("%-40s %8.3f ms  %7.0f cycles/iter  (1024 MFMA pipe cycles per SIMD per iter)\n", name, ms, avg / iters);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<1, 1, 1, 2>("reads+mfma+barrier  (64x64)", out, cyc);
    run<1, 0, 1, 2>("reads only+barrier  (64x64)", out, cyc);
    run<0, 1, 1, 2>("mfma only+barrier   (64x64)", out, cyc);
    run<1, 1, 0, 2>("reads+mfma no barrier (64x64)", out, cyc);
    run<0, 1, 0, 2>("mfma only no barrier (64x64)", out, cyc);
    run<1, 0, 0, 2>("reads only no barrier (64x64)", out, cyc);
    run<1, 1, 1, 4>("reads+mfma+barrier  (128x64)", out, cyc);
    run<0, 1, 1, 4>("mfma only+barrier   (128x64)", out, cyc);
    run<1, 0, 1, 4>("reads only+barrier  (128x64)", out, cyc);
    char* gsrc; hipMalloc(&gsrc, 257 * 65536); hipMemset(gsrc, 0, 257 * 65536);
    run<0, 0, 1, 2, 6>("dma(6/wave) only+barrier", out, cyc, gsrc);
    run<1, 0, 1, 2, 6>("dma+reads+barrier (64x64)", out, cyc, gsrc);
    run<1, 1, 1, 2, 6>("dma+reads+mfma+barrier (64x64)", out, cyc, gsrc);
    run<0, 1, 1, 2, 6>("dma+mfma+barrier (64x64)", out, cyc, gsrc);
    run<1, 1, 1, 4, 4>("dma(4)+reads+mfma+barrier (128x64)", out, cyc, gsrc);
    run<1, 1, 1, 4, 8>("dma(8)+reads+mfma+barrier (128x64)", out, cyc, gsrc);
    run<0, 0, 1, 2, -6>("regstage(6) only+barrier", out, cyc, gsrc);
    run<1, 1, 1, 2, -6>("regstage(6)+reads+mfma+barrier (64x64)", out, cyc, gsrc);
    run<1, 1, 1, 4, -8>("regstage(8)+reads+mfma+barrier (128x64)", out, cyc, gsrc);
    run16<6, 8>("8 consumers + 8 loaders x 6 DMA", out, cyc, gsrc);
    run16<12, 4>("8 consumers + 4 loaders x 12 DMA", out, cyc, gsrc);
    run16<0, 8>("8 consumers + 8 idle loaders (no DMA)", out, cyc, gsrc);
    return 0;
}
