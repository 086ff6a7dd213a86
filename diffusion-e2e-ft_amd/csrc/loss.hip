// loss.hip — E2E-FT task losses (forward), fp32 I/O with fp64 reductions.
//   ScaleAndShiftInvariantLoss  — /root/reference/training/util/loss.py:13-47
//   AngularLoss                 — /root/reference/training/util/loss.py:51-67
#include "common.h"

namespace e2eft {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-reduce NV doubles and atomically add to dst[0..NV)
template <int NV> __device__ __forceinline__ void block_atomic_add(double (&v)[NV], double* dst) {
    __shared__ double red[4][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = wave_sum_d(v[i]);
        if (lane == 0) red[wave][i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(dst + threadIdx.x, t);
    }
}

// pass 1: per-image masked sums a00=sum m p^2, a01=sum m p, a11=sum m, b0=sum m p t, b1=sum m t   (loss.py:33-38)
__global__ __launch_bounds__(256) void ssi_sums_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       const uint8_t* __restrict__ mask, double* __restrict__ sums /* [B][5] */) {
    const int b = blockIdx.y;
    const float* p = pred + (long)b * hw;
    const float* t = tgt + (long)b * hw;
    const uint8_t* m = mask + (long)b * hw;
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        if (m[i]) {
            const double pp = p[i], tt = t[i];
            v[0] += pp * pp; v[1] += pp; v[2] += 1.0; v[3] += pp * tt; v[4] += tt;
        }
    }
    block_atomic_add<5>(v, sums + b * 5);
}

// solve the 2x2 system per image in fp32 exactly as loss.py:39-46 (det > 0 guard, zeros otherwise)
__global__ void ssi_solve_kernel(int batch, const double* __restrict__ sums, float* __restrict__ ss /* [B][2] */) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float a00 = (float)sums[b * 5], a01 = (float)sums[b * 5 + 1], a11 = (float)sums[b * 5 + 2];
    const float b0 = (float)sums[b * 5 + 3], b1 = (float)sums[b * 5 + 4];
    const float det = a00 * a11 - a01 * a01;
    float x0 = 0.f, x1 = 0.f;
    if (det > 0.f) {
        x0 = (a11 * b0 - a01 * b1) / det;
        x1 = (-a01 * b0 + a00 * b1) / det;
    }
    ss[b * 2] = x0;
    ss[b * 2 + 1] = x1;
}

// pass 2: sum over valid pixels of |scale_b p + shift_b - t| and the valid count (loss.py:26-28)
__global__ __launch_bounds__(256) void ssi_l1_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                     const uint8_t* __restrict__ mask, const float* __restrict__ ss,
                                                     double* __restrict__ acc /* [2] */) {
    const int b = blockIdx.y;
    const float sc = ss[b * 2], sh = ss[b * 2 + 1];
    const float* p = pred + (long)b * hw;
    const float* t = tgt + (long)b * hw;
    const uint8_t* m = mask + (long)b * hw;
    double v[2] = {0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        if (m[i]) {
            v[0] += (double)fabsf(sc * p[i] + sh - t[i]);
            v[1] += 1.0;
        }
    }
    block_atomic_add<2>(v, acc);
}

// mean over the valid pixels.  No valid pixel -> 0 (the reference skips the loss term: `if val_mask.any()`, training/train.py:504);
// a NaN sum -> 0 as well (`if not torch.isnan(...)`, train.py:548,552): the term contributes neither loss nor gradient (bwd.hip).
__global__ void mean_kernel(const double* __restrict__ acc, float* __restrict__ out) {
    out[0] = (acc[1] > 0.0 && !isnan(acc[0])) ? (float)(acc[0] / acc[1]) : 0.f;
}

__global__ __launch_bounds__(256) void angular_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      const uint8_t* __restrict__ mask, double* __restrict__ acc) {
    const int b = blockIdx.y;
    const float* p = pred + (long)b * 3 * hw;
    const float* t = tgt + (long)b * 3 * hw;
    const uint8_t* m = mask + (long)b * hw;
    double v[2] = {0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        if (m[i]) {
            float d = p[i] * t[i] + p[hw + i] * t[hw + i] + p[2 * hw + i] * t[2 * hw + i];
            d = isnan(d) ? d : fminf(fmaxf(d, -1.f), 1.f);   // torch.clamp propagates NaN (fminf / fmaxf would swallow it, loss.py:62)
            v[0] += (double)acosf(d);
            v[1] += 1.0;
        }
    }
    block_atomic_add<2>(v, acc);
}

}  // namespace e2eft

using namespace e2eft;

extern "C" size_t e2eft_ssi_loss_workspace_bytes(int32_t batch) { return batch > 0 ? ((size_t)batch * 5 + 2) * sizeof(double) + (size_t)batch * 2 * sizeof(float) : 0; }

extern "C" int e2eft_ssi_loss_fwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                                  float* out_loss, float* out_scale_shift, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(pred && target && mask && out_loss && workspace, "ssi_loss: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && hw > 0, "ssi_loss: shape");
    const size_t need = e2eft_ssi_loss_workspace_bytes(batch);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "ssi_loss: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(((uintptr_t)workspace & 7) == 0, "ssi_loss: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* sums = (double*)workspace;
    double* acc = sums + (size_t)batch * 5;
    float* ss = out_scale_shift ? out_scale_shift : (float*)(acc + 2);
    if (hipMemsetAsync(workspace, 0, ((size_t)batch * 5 + 2) * sizeof(double), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "ssi_loss: memset failed");
    int nb = cdiv(hw, 256 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(ssi_sums_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, sums);
    hipLaunchKernelGGL(ssi_solve_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, s, batch, sums, ss);
    hipLaunchKernelGGL(ssi_l1_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, ss, acc);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1), 0, s, acc, out_loss);
    return check_launch("ssi_loss");
}

extern "C" size_t e2eft_angular_loss_workspace_bytes(int32_t batch) { return batch > 0 ? 2 * sizeof(double) : 0; }

extern "C" int e2eft_angular_loss_fwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                                      float* out_loss, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(pred && target && mask && out_loss && workspace, "angular_loss: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && hw > 0, "angular_loss: shape");
    if (ws_bytes < 2 * sizeof(double)) return fail(E2EFT_ERR_WORKSPACE, "angular_loss: workspace too small");
    E2EFT_REQUIRE(((uintptr_t)workspace & 7) == 0, "angular_loss: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* acc = (double*)workspace;
    if (hipMemsetAsync(workspace, 0, 2 * sizeof(double), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "angular_loss: memset failed");
    int nb = cdiv(hw, 256 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(angular_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, acc);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1), 0, s, acc, out_loss);
    return check_launch("angular_loss");
}
