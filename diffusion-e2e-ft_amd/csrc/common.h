// common.h — shared device/host helpers for libe2eft (gfx950 only; wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/e2eft.h"
#include "../../include/e2eft_debug.h"

namespace e2eft {

// ---- error plumbing (thread-local message, no exceptions across the ABI) -------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
int option(int key);   // process-wide tuning options (e2eft_set_option), api.hip
void tag_kernel(const char* fmt, ...);   // thread-local name of the kernel a launcher just enqueued (e2eft_debug_last_kernel: bench.py --detail)

#define E2EFT_REQUIRE(cond, ...)                                     \
    do {                                                             \
        if (!(cond)) return ::e2eft::fail(E2EFT_ERR_BAD_ARG, __VA_ARGS__); \
    } while (0)

// ---- scalar types ----------------------------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bhalf8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = E2EFT_F32; };
template <> struct DT<f16> { static constexpr int id = E2EFT_F16; };
template <> struct DT<bf16> { static constexpr int id = E2EFT_BF16; };

static inline size_t dtype_size(int dt) { return dt == E2EFT_F32 ? 4 : 2; }

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// 16-byte vector of T: 4 floats or 8 halves.
template <typename T> struct Vec16 {
    static constexpr int N = 16 / sizeof(T);
    union {
        u32x4 raw;
        T e[16 / sizeof(T)];
    };
};

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const u32x4*>(p);
    return v;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
    *reinterpret_cast<u32x4*>(p) = v.raw;
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // v_rcp_f32: 1 ulp, no division sequence
// GroupNorm(+SiLU) apply on 16-bit tensors, in the exp2 domain (round 6): with a = rstd * gamma,  a2 = a * log2(e),  d2 = fma(-mean, a, beta) * log2(e)
//     u = fma(x, a2, d2) = log2(e) * ((x - mean) * a + beta)
//     SiLU:  u * rcp(fma(exp2(-u), log2(e), log2(e))) = t / (1 + e^-t)            no activation:  u * ln(2) = t
// — 4 operations per value after the fma (exp2 with a negated operand, fma, rcp, mul) instead of 6 (sub, mul, exp2, add, rcp, mul): the fused form in igemm6's patch
// staging is VALU-bound, every instruction there is matrix-pipe time.  gn_apply_kernel (norm.hip), the halo staging of narrow.hip and igemm6's
// in-LDS form use EXACTLY these operations in this order: their results are bit-identical (tests/test_fused_norm_conv_gpu.py).  x * a2 - mean * a2 cancels to
// ~|mean| / sigma * 2^-24 relative — four decimal orders below a 16-bit output's rounding; fp32 tensors keep (x - mean) * a + beta.
constexpr float GN_L2E = 1.4426950408889634f, GN_LN2 = 0.6931471805599453f;
__device__ __forceinline__ void gn_fold(const float a, const float mean, const float beta, float& a2, float& d2) {
    a2 = a * GN_L2E;
    d2 = __builtin_fmaf(-mean, a, beta) * GN_L2E;
}
__device__ __forceinline__ float gn_act_u(const float u, const bool silu) {
    const float r = silu ? __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-u), GN_L2E, GN_L2E)) : GN_LN2;
    return u * r;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// dispatch helper: calls F<T>(args...) for the runtime dtype
#define E2EFT_DISPATCH_DTYPE(dt, T, ...)                                  \
    switch (dt) {                                                         \
        case E2EFT_F32: { typedef float T; __VA_ARGS__; break; }          \
        case E2EFT_F16: { typedef ::e2eft::f16 T; __VA_ARGS__; break; }   \
        case E2EFT_BF16: { typedef ::e2eft::bf16 T; __VA_ARGS__; break; } \
        default: return ::e2eft::fail(E2EFT_ERR_BAD_ARG, "bad dtype %d", (int)(dt)); \
    }

}  // namespace e2eft
