// f32split.hip — fp32 convolutions on the f16 matrix pipe (round 6).
//
// The reference trains and evaluates in fp32 (`--mixed_precision "no"`, training/scripts/train_marigold_e2e_ft_depth.sh:15); on gfx950 the fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32) peaks at 157 TF/s, a sixteenth of the f16 pipe, and the 3x3 VAE convolutions are three quarters of a strict-fp32 training step.
// An fp32 value x with |x| * s < 2^15 (s a power of two) splits EXACTLY into two f16 terms up to 2^-22 |x|:
//     x0 = f16(x s),   x1 = f16(x s - x0)          (x s - x0 is representable in fp32: Sterbenz; both roundings to nearest even)
// and a product of two such values is  x0 w0 + x0 w1 + x1 w0  up to 3 * 2^-22 |x w| (the dropped x1 w1 term is 2^-22) — each term is an exact f16 x f16 product
// accumulated in fp32 by v_mfma_f32_32x32x16_f16.  That error is below what the fp32 accumulation of a K = 1152 ... 4608 reduction adds on EITHER pipe
// (~sqrt(K) 2^-24), and it is measured, not argued: tests/test_f32split_gpu.py compares both routes with a float64 convolution.  Range: the scale comes from the
// tensor's own maximum (one reduction pass, on the device: no host synchronisation), so nothing overflows; values more than 2^17 below the maximum lose RELATIVE
// precision (their x1 is an f16 subnormal) but keep an ABSOLUTE error of 2^-39 of the maximum — invisible in a sum of products.
//   e2eft_f32_split2            x fp32 [pixels][c]  ->  planes f16 [pixels][x0 (c) | x1 (c)],  scale[1] = s, scale[2] = 1 / s
//   e2eft_conv2d_fwd_f32split   the 3x3 / stride-1 / pad-1 convolution of such planes with weights split the same way on the host (once per weight):
//                               igemm6_kernel<f16, ..., F32O> — K runs over the blocks (x0, w0), (x0, w1), (x1, w0); fp32 bias / residual / output / statistics.
// Three f16 MFMAs per fp32 product against sixteen times the rate: 2.9 ms instead of 11.8 ms for 128 -> 128 at 16 x 576^2, plus the two split passes (12 B / element).
#include "igemm.h"

namespace e2eft {

// |x| maximum of an fp32 [pixels][ldx] tensor's first c columns (c % 4 == 0): non-negative floats order like their bit patterns
__global__ __launch_bounds__(256) void absmax_f32_kernel(long pixels, int c, int ldx, const float* __restrict__ x, unsigned* __restrict__ amax_bits) {
    const int c4 = c >> 2;
    const long total = pixels * c4;
    float m = 0.f;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pix = it / c4;
        const int ch = (int)(it - pix * c4) * 4;
        const floatx4 v = *reinterpret_cast<const floatx4*>(x + pix * ldx + ch);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));      // (fmaxf drops NaNs: a NaN input shows in the output through x0 anyway)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __float_as_uint(m));
}

// power-of-two scale that brings amax into [2^14, 2^15): (s, 1 / s); 1 for an all-zero or non-finite tensor
__device__ __forceinline__ void split_scale(const unsigned amax_bits, float& s, float& inv) {
    const int ex = (int)((amax_bits >> 23) & 0xffu);        // biased exponent: amax in [2^(ex - 127), 2^(ex - 126))
    int k = 0;
    if (amax_bits != 0u && ex != 0xff) k = 14 - (ex == 0 ? -126 : ex - 127);
    k = k > 100 ? 100 : k < -100 ? -100 : k;
    s = __uint_as_float((unsigned)(127 + k) << 23);
    inv = __uint_as_float((unsigned)(127 - k) << 23);
}

// one thread: eight channels of one pixel -> one 16-byte unit of each plane
__global__ __launch_bounds__(256) void split2_f16_kernel(long pixels, int c, int ldx, int ldp, const float* __restrict__ x, f16* __restrict__ planes,
                                                         const unsigned* __restrict__ amax_bits, float* __restrict__ scale_out) {
    float s, inv;
    split_scale(*amax_bits, s, inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[1] = s; scale_out[2] = inv; }
    const int c8 = c >> 3;
    const long total = pixels * c8;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pix = it / c8;
        const int ch = (int)(it - pix * c8) * 8;
        const floatx4 a = *reinterpret_cast<const floatx4*>(x + pix * ldx + ch);
        const floatx4 b = *reinterpret_cast<const floatx4*>(x + pix * ldx + ch + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        Vec16<f16> p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = v[e] * s;
            const f16 h0 = (f16)t;                      // round to nearest even
            p0.e[e] = h0;
            p1.e[e] = (f16)(t - (float)h0);             // exact difference, rounded once
        }
        st16(planes + pix * ldp + ch, p0);
        st16(planes + pix * ldp + c + ch, p1);
    }
}

bool igemm_patch_eligible(int dtype, int mode, IgemmParams& p, int nz);              // igemm6.hip
int launch_igemm_patch(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);

static bool f32split_params(const E2eftConvDesc* d, IgemmParams& p) {
    if (!d || d->dtype != E2EFT_F32 || d->c2 != 0 || !option(E2EFT_OPT_F32_SPLIT)) return false;
    if (d->batch <= 0 || d->hin <= 0 || d->win <= 0 || d->cout <= 0 || d->c1 <= 0) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->hl != d->hin || d->wl != d->win || d->hout != d->hin || d->wout != d->win) return false;
    if (d->c1 % 64 != 0 || d->ldx1 < 2 * d->c1 || d->ldw < 27 * d->c1 || (long)d->batch * d->hout * d->wout >= 2147483647L) return false;
    p = IgemmParams{};
    p.M = d->batch * d->hout * d->wout; p.N = d->cout; p.K = 27 * d->c1;
    p.ldx1 = d->ldx1; p.c1 = 3 * d->c1; p.cin = 3 * d->c1; p.split_c = d->c1;
    p.hin = d->hin; p.win = d->win; p.hl = d->hl; p.wl = d->wl;
    p.kh = 3; p.kw = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
    p.hout = d->hout; p.wout = d->wout;
    p.up_sh = p.up_sw = 1.f;
    p.ldw = d->ldw; p.ldr = d->ldr > 0 ? d->ldr : d->ldo; p.ldo = d->ldo;
    p.rows_per_img = d->hout * d->wout;
    p.alpha = d->alpha;
    p.nzi = 1;
    return true;
}

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_f32_split2(const float* x, int64_t pixels, int32_t c, int32_t ldx, void* planes, int32_t ldp, float* scale, void* stream) {
    E2EFT_REQUIRE(x && planes && scale && pixels > 0, "f32_split2: null pointer / empty tensor");
    E2EFT_REQUIRE(c > 0 && c % 8 == 0 && ldx >= c && ldx % 4 == 0 && ldp >= 2 * c && ldp % 8 == 0, "f32_split2: c=%d ldx=%d ldp=%d (c %% 8, ldx %% 4, ldp %% 8, ldp >= 2 c)", c, ldx, ldp);
    E2EFT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0 && ((uintptr_t)scale & 3) == 0, "f32_split2: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scale, 0, sizeof(float), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "f32_split2: memset failed");
    const long units = pixels * (c / 8);
    long nb = (units + 255) / 256;
    const long nbm = nb > 4096 ? 4096 : nb;          // the reduction: few atomics
    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)nbm), dim3(256), 0, s, (long)pixels, c, ldx, x, reinterpret_cast<unsigned*>(scale));
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(split2_f16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)pixels, c, ldx, ldp, x, (f16*)planes, reinterpret_cast<const unsigned*>(scale), scale);
    tag_kernel("split2_f16_kernel");
    return check_launch("f32_split2");
}

// pure host arithmetic: would e2eft_conv2d_fwd_f32split take this launch (with 16-byte aligned pointers)?  desc: dtype E2EFT_F32, c1 = channels of the fp32
// tensor, ldx1 = pixel stride of the PLANES in f16 elements (>= 2 c1), ldw = weight row length in f16 elements (>= 27 c1), ldo / ldr in fp32 elements
extern "C" int e2eft_conv2d_fwd_f32split_supported(const E2eftConvDesc* d) {
    IgemmParams p;
    if (!f32split_params(d, p)) return 0;
    void* const al = (void*)(uintptr_t)256;
    p.x1 = al; p.w = al; p.out = al;
    return igemm_patch_eligible(E2EFT_F16, 1, p, 1) ? 1 : 0;
}

extern "C" int e2eft_conv2d_fwd_f32split(const E2eftConvDesc* d, const void* planes, const float* scale, const void* w_split, const float* bias,
                                         const float* residual, float* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && planes && scale && w_split && out, "conv2d_fwd_f32split: null pointer");
    IgemmParams p;
    if (!f32split_params(d, p)) return fail(E2EFT_ERR_UNSUPPORTED, "conv2d_fwd_f32split: not a 3x3 / stride-1 / pad-1 fp32 convolution of 64-channel multiples (ask e2eft_conv2d_fwd_f32split_supported)");
    p.x1 = planes; p.w = w_split; p.bias = bias; p.residual = residual; p.out = out;
    p.alpha_dev = scale + 2;
    if (gn_partial && slab_rows) {
        const size_t need = (size_t)d->batch * (size_t)cdiv(p.rows_per_img, 128) * (size_t)d->cout * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "conv2d_fwd_f32split: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
        p.gn_partial = gn_partial;
    }
    const int rc = launch_igemm_patch(E2EFT_F16, 1, p, 1, (hipStream_t)stream);
    if (rc < 0) return fail(E2EFT_ERR_UNSUPPORTED, "conv2d_fwd_f32split: this launch is not eligible for the halo-patch kernel");
    if (rc == 0 && slab_rows && p.gn_partial) *slab_rows = p.rows_per_img / p.gn_nslabs;
    return rc;
}
