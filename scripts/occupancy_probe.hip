// occupancy_probe.hip — how much dynamic LDS (and which VGPR budget) still lets TWO 512-thread workgroups share a CU on gfx950?
// Every workgroup spins for a fixed number of cycles; 512 workgroups on 256 CUs take T if two are co-resident, 2T otherwise.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void spin(long long cycles, int* sink) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {}
    if (lds[(threadIdx.x + 1) & 511] == 99 && cycles < 0) sink[0] = 1;
}
// same, with a dynamically indexed private array (forces scratch) and a 128-VGPR budget
__global__ __launch_bounds__(512, 4) void spin_scratch(long long cycles, int* sink, int idx) {
    extern __shared__ char lds[];
    volatile int priv[24];
    for (int i = 0; i < 24; ++i) priv[i] = i * idx;
    lds[threadIdx.x] = (char)priv[(threadIdx.x + idx) % 24];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {}
    if (lds[(threadIdx.x + 1) & 511] == 99 && cycles < 0) sink[0] = priv[idx % 24];
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  multiProcessorCount %d\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.multiProcessorCount);
    const int kbs[] = {32, 64, 72, 74, 76, 78, 80, 81, 96};
    for (int kb : kbs) {
        float ms[2];
        for (int rep = 0; rep < 2; ++rep) {
            const int blocks = rep == 0 ? 256 : 512;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), kb * 1024, 0, 1000LL, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), kb * 1024, 0, 2000000LL, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[rep], e0, e1);
        }
        printf("LDS %3d KB: 256 WGs %.3f ms, 512 WGs %.3f ms -> %s\n", kb, ms[0], ms[1], ms[1] < 1.5f * ms[0] ? "two per CU" : "ONE per CU");
    }
    hipFuncSetAttribute((const void*)spin_scratch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {32, 76}) {
        float ms[2];
        for (int rep = 0; rep < 2; ++rep) {
            const int blocks = rep == 0 ? 256 : 512;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(spin_scratch, dim3(blocks), dim3(512), kb * 1024, 0, 1000LL, sink, 3);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin_scratch, dim3(blocks), dim3(512), kb * 1024, 0, 2000000LL, sink, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[rep], e0, e1);
        }
        printf("scratch kernel, LDS %3d KB: 256 WGs %.3f ms, 512 WGs %.3f ms -> %s\n", kb, ms[0], ms[1], ms[1] < 1.5f * ms[0] ? "two per CU" : "ONE per CU");
    }
    return 0;
}
