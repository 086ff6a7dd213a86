// elementwise.hip — layout / scaling glue of the path (all HBM-bound, tiny next to the convolutions).
#include "common.h"

namespace e2eft {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(int batch, int c, int hw, int cpad, int ldy, float mul, float add,
                                                           const TI* __restrict__ x, TO* __restrict__ y) {
    const long total = (long)batch * hw;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long b = it / hw;
        const long pix = it - b * hw;
        TO* dst = y + it * ldy;
        for (int ch = 0; ch < cpad; ++ch) {
            float v = 0.f;
            if (ch < c) v = fmaf(to_f(x[(b * c + ch) * hw + pix]), mul, add);
            dst[ch] = from_f<TO>(v);
        }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(int batch, int c, int hw, int ldx, float mul, float add,
                                                           const TI* __restrict__ x, TO* __restrict__ y) {
    const long total = (long)batch * hw;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long b = it / hw;
        const long pix = it - b * hw;
        const TI* src = x + it * ldx;
        for (int ch = 0; ch < c; ++ch) y[(b * c + ch) * hw + pix] = from_f<TO>(fmaf(to_f(src[ch]), mul, add));
    }
}

// y[p, 0:c] = a[p, 0:c] * mul + add (+ b[p, 0:c] if b)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void axpb_kernel(long pixels, int c, int lda, int ldb, int ldy, float mul, float add,
                                                   const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y) {
    constexpr int EPC = VEC ? 16 / (int)sizeof(T) : 1;
    const int cch = c / EPC;
    const long total = pixels * cch;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pix = it / cch;
        const int ch = (int)(it - pix * cch) * EPC;
        if constexpr (VEC) {
            Vec16<T> va = ld16(a + pix * lda + ch), o;
            if (b) {
                Vec16<T> vb = ld16(b + pix * ldb + ch);
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(fmaf(to_f(va.e[e]), mul, add) + to_f(vb.e[e]));
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(fmaf(to_f(va.e[e]), mul, add));
            }
            st16(y + pix * ldy + ch, o);
        } else {
            float v = fmaf(to_f(a[pix * lda + ch]), mul, add);
            if (b) v += to_f(b[pix * ldb + ch]);
            y[pix * ldy + ch] = from_f<T>(v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void timestep_embedding_kernel(int batch, int dim, const int64_t* __restrict__ t, T* __restrict__ out) {
    const int half = dim / 2;
    const int total = batch * half;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < total; it += gridDim.x * 256) {
        const int b = it / half, i = it - b * half;
        const float freq = expf(-9.210340371976184f * (float)i / (float)half);  // ln(10000)
        const float arg = (float)t[b] * freq;
        out[(long)b * dim + i] = from_f<T>(cosf(arg));          // flip_sin_to_cos: cos first
        out[(long)b * dim + half + i] = from_f<T>(sinf(arg));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(long n, const T* __restrict__ x, T* __restrict__ y) {
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) y[it] = from_f<T>(silu_f(to_f(x[it])));
}

// y = act(x) on a flat buffer, 16-byte vectors + scalar tail.  kind: 0 quick_gelu x*sigmoid(1.702x) (CLIP), 1 gelu (erf), 2 silu, 3 sigmoid (the two-key softmax of the folded cross-attention)
template <typename T>
__global__ __launch_bounds__(256) void act_kernel(int kind, long n, const T* __restrict__ x, T* __restrict__ y) {
    constexpr int EPC = 16 / (int)sizeof(T);
    auto f = [kind](float v) {
        if (kind == 0) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
        if (kind == 1) return gelu_erf_f(v);
        if (kind == 3) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        return silu_f(v);
    };
    const long nv = n / EPC;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < nv; it += (long)gridDim.x * 256) {
        Vec16<T> v = ld16(x + it * EPC), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(f(to_f(v.e[e])));
        st16(y + it * EPC, o);
    }
    for (long it = nv * EPC + (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) y[it] = from_f<T>(f(to_f(x[it])));
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void depth_head_kernel(long pixels, int ldx, int to_unit, const TI* __restrict__ x, TO* __restrict__ y) {
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < pixels; it += (long)gridDim.x * 256) {
        const TI* s = x + it * ldx;
        // torch: stacked.mean(dim=1) computed in the tensor dtype; we average the three values in fp32
        float v = (to_f(s[0]) + to_f(s[1]) + to_f(s[2])) / 3.0f;
        if (to_unit != 2) v = fminf(fmaxf(v, -1.f), 1.f);     // 2: the bare channel mean `decode_depth` returns (marigold_pipeline.py:517-519)
        if (to_unit == 1) v = (v + 1.0f) * 0.5f;
        y[it] = from_f<TO>(v);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void normal_head_kernel(int batch, int hw, int ldx, int clampv, float sign, const TI* __restrict__ x, TO* __restrict__ y) {
    const long total = (long)batch * hw;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long b = it / hw, pix = it - b * hw;
        const TI* s = x + it * ldx;
        const float n0 = to_f(s[0]), n1 = to_f(s[1]), n2 = to_f(s[2]);
        const float inv = 1.0f / (sqrtf(n0 * n0 + n1 * n1 + n2 * n2) + 1e-5f);
        float o[3] = {n0 * inv, n1 * inv, n2 * inv};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = o[ch];
            if (clampv) v = fminf(fmaxf(v, -1.f), 1.f);
            y[(b * 3 + ch) * hw + pix] = from_f<TO>(v * sign);
        }
    }
}

static unsigned grid_for(long total) {
    long nb = (total + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 16384) nb = 16384;
    return (unsigned)nb;
}

#define E2EFT_DISPATCH2(dti, dto, TI, TO, ...)                                     \
    E2EFT_DISPATCH_DTYPE(dti, TI, { E2EFT_DISPATCH_DTYPE(dto, TO, __VA_ARGS__); })

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_nchw_to_nhwc(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t c, int32_t hw, int32_t cpad,
                                  int32_t ldy, float mul, float add, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y, "nchw_to_nhwc: null pointer");
    E2EFT_REQUIRE(batch > 0 && c > 0 && hw > 0 && cpad >= c && ldy >= cpad, "nchw_to_nhwc: shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((long)batch * hw);
    E2EFT_DISPATCH2(dt_in, dt_out, TI, TO,
                    hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, TO>), dim3(g), dim3(256), 0, s, batch, c, hw, cpad, ldy, mul, add, (const TI*)x, (TO*)y));
    return check_launch("nchw_to_nhwc");
}

extern "C" int e2eft_nhwc_to_nchw(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t c, int32_t hw, int32_t ldx,
                                  float mul, float add, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y, "nhwc_to_nchw: null pointer");
    E2EFT_REQUIRE(batch > 0 && c > 0 && hw > 0 && ldx >= c, "nhwc_to_nchw: shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((long)batch * hw);
    E2EFT_DISPATCH2(dt_in, dt_out, TI, TO,
                    hipLaunchKernelGGL((nhwc_to_nchw_kernel<TI, TO>), dim3(g), dim3(256), 0, s, batch, c, hw, ldx, mul, add, (const TI*)x, (TO*)y));
    return check_launch("nhwc_to_nchw");
}

static int axpb(int32_t dtype, int64_t pixels, int32_t c, int32_t lda, int32_t ldb, int32_t ldy, float mul, float add,
                const void* a, const void* b, void* y, void* stream, const char* what) {
    E2EFT_REQUIRE(a && y, "%s: null pointer", what);
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "%s: bad dtype", what);
    E2EFT_REQUIRE(pixels > 0 && c > 0 && lda >= c && ldy >= c && (!b || ldb >= c), "%s: shape", what);
    const int epc = 16 / (int)dtype_size(dtype);
    const bool vec = c % epc == 0 && lda % epc == 0 && ldy % epc == 0 && (!b || ldb % epc == 0) &&
                     (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b) & 15) == 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for(pixels * (vec ? c / epc : c));
    E2EFT_DISPATCH_DTYPE(dtype, T, {
        if (vec) hipLaunchKernelGGL((axpb_kernel<T, true>), dim3(g), dim3(256), 0, s, (long)pixels, c, lda, ldb, ldy, mul, add, (const T*)a, (const T*)b, (T*)y);
        else hipLaunchKernelGGL((axpb_kernel<T, false>), dim3(g), dim3(256), 0, s, (long)pixels, c, lda, ldb, ldy, mul, add, (const T*)a, (const T*)b, (T*)y);
    });
    return check_launch(what);
}

extern "C" int e2eft_copy_scale(int32_t dtype, int64_t pixels, int32_t c, int32_t ldx, int32_t ldy, float mul, float add,
                                const void* x, void* y, void* stream) {
    return axpb(dtype, pixels, c, ldx, 0, ldy, mul, add, x, nullptr, y, stream, "copy_scale");
}

extern "C" int e2eft_add(int32_t dtype, int64_t pixels, int32_t c, int32_t lda, int32_t ldb, int32_t ldy, const void* a,
                         const void* b, void* y, void* stream) {
    E2EFT_REQUIRE(b, "add: null pointer");
    return axpb(dtype, pixels, c, lda, ldb, ldy, 1.0f, 0.0f, a, b, y, stream, "add");
}

extern "C" int e2eft_timestep_embedding(int32_t dtype, int32_t batch, int32_t dim, const int64_t* t, void* out, void* stream) {
    E2EFT_REQUIRE(t && out, "timestep_embedding: null pointer");
    E2EFT_REQUIRE(batch > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((long)batch * dim / 2);
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((timestep_embedding_kernel<T>), dim3(g), dim3(256), 0, s, batch, dim, t, (T*)out));
    return check_launch("timestep_embedding");
}

extern "C" int e2eft_silu(int32_t dtype, int64_t n, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y && n > 0, "silu: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for(n);
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((silu_kernel<T>), dim3(g), dim3(256), 0, s, (long)n, (const T*)x, (T*)y));
    return check_launch("silu");
}

extern "C" int e2eft_activation(int32_t dtype, int32_t kind, int64_t n, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y && n > 0, "activation: bad args");
    E2EFT_REQUIRE(kind >= 0 && kind <= 3, "activation: kind %d (0 quick_gelu, 1 gelu, 2 silu, 3 sigmoid)", (int)kind);
    E2EFT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "activation: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for(n / 4 + 1);
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((act_kernel<T>), dim3(g), dim3(256), 0, s, (int)kind, (long)n, (const T*)x, (T*)y));
    return check_launch("activation");
}

extern "C" int e2eft_depth_head(int32_t dt_in, int32_t dt_out, int64_t pixels, int32_t ldx, int32_t to_unit, const void* x,
                                void* y, void* stream) {
    E2EFT_REQUIRE(x && y && pixels > 0 && ldx >= 3, "depth_head: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for(pixels);
    E2EFT_DISPATCH2(dt_in, dt_out, TI, TO,
                    hipLaunchKernelGGL((depth_head_kernel<TI, TO>), dim3(g), dim3(256), 0, s, (long)pixels, ldx, to_unit, (const TI*)x, (TO*)y));
    return check_launch("depth_head");
}

extern "C" int e2eft_normal_head(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t hw, int32_t ldx, int32_t clampv,
                                 float sign, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y && batch > 0 && hw > 0 && ldx >= 3, "normal_head: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((long)batch * hw);
    E2EFT_DISPATCH2(dt_in, dt_out, TI, TO,
                    hipLaunchKernelGGL((normal_head_kernel<TI, TO>), dim3(g), dim3(256), 0, s, batch, hw, ldx, clampv, sign, (const TI*)x, (TO*)y));
    return check_launch("normal_head");
}
