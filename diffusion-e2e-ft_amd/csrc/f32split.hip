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
    const long stride = (long)gridDim.x * 256;
    auto at = [&](const long it) -> floatx4 {
        const long pix = it / c4;
        const int ch = (int)(it - pix * c4) * 4;
        return *reinterpret_cast<const floatx4*>(x + pix * ldx + ch);
    };
    auto fold = [&](const floatx4& v) { m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3]))); };   // (fmaxf drops NaNs: a NaN input shows in the output through x0 anyway)
    long it = (long)blockIdx.x * 256 + threadIdx.x;
    for (; it + 3 * stride < total; it += 4 * stride) {      // four independent 16-byte loads in flight per thread (1024 workgroups have to cover the HBM latency)
        const floatx4 v0 = at(it), v1 = at(it + stride), v2 = at(it + 2 * stride), v3 = at(it + 3 * stride);
        fold(v0); fold(v1); fold(v2); fold(v3);
    }
    for (; it < total; it += stride) fold(at(it));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // ONE atomic per workgroup, at most 1024 workgroups: 16 k same-address atomics (one per wave of 4096 workgroups) cost 0.25 ms — more than the pass itself on a 100 MB tensor
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(amax_bits, __float_as_uint(m));
    }
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
// (coff, ctot: this source's first column inside a plane and the plane's width — two sources side by side are the split of their channel concatenation)
// third != 0: the first plane is written a second time behind the second one — the [w0 | w1 | w0] rows of a split WEIGHT (e2eft_f32_split_weight)
__global__ __launch_bounds__(256) void split2_f16_kernel(long pixels, int c, int ldx, int ldp, int coff, int ctot, int third, const float* __restrict__ x, f16* __restrict__ planes,
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
        st16(planes + pix * ldp + coff + ch, p0);
        st16(planes + pix * ldp + ctot + coff + ch, p1);
        if (third) st16(planes + pix * ldp + 2 * ctot + coff + ch, p0);
    }
}

bool igemm_patch_eligible(int dtype, int mode, IgemmParams& p, int nz);              // igemm6.hip
int launch_igemm_patch(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);
bool igemm_persistent_eligible(int dtype, int mode, IgemmParams& p, int nz);        // igemm5.hip
int launch_igemm_persistent(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);

// the implicit-GEMM parameter block of an fp32 convolution whose input arrives as split planes: any filter / stride / padding of e2eft_conv2d_fwd without a second
// source and without a fused upsample; K = taps * 3 * c1
static bool f32split_params(const E2eftConvDesc* d, IgemmParams& p) {
    if (!d || d->dtype != E2EFT_F32 || d->c2 != 0 || !option(E2EFT_OPT_F32_SPLIT)) return false;
    if (d->batch <= 0 || d->hin <= 0 || d->win <= 0 || d->cout <= 0 || d->c1 <= 0 || d->hout <= 0 || d->wout <= 0) return false;
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->hl != d->hin || d->wl != d->win) return false;
    if ((d->hout - 1) * d->stride - d->pad_t >= d->hl || (d->wout - 1) * d->stride - d->pad_l >= d->wl) return false;
    const int taps = d->kh * d->kw;
    if (d->c1 % 64 != 0 || d->ldx1 < 2 * d->c1 || d->ldx1 % 8 != 0 || d->ldw < taps * 3 * d->c1 || d->ldw % 8 != 0 || (long)d->batch * d->hout * d->wout >= 2147483647L) return false;
    p = IgemmParams{};
    p.M = d->batch * d->hout * d->wout; p.N = d->cout; p.K = taps * 3 * d->c1;
    p.ldx1 = d->ldx1; p.c1 = 3 * d->c1; p.cin = 3 * d->c1; p.split_c = d->c1;
    p.hin = d->hin; p.win = d->win; p.hl = d->hl; p.wl = d->wl;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
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
    const long nbm = nb > 1024 ? 1024 : nb;          // the reduction: few atomics
    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)nbm), dim3(256), 0, s, (long)pixels, c, ldx, x, reinterpret_cast<unsigned*>(scale));
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(split2_f16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)pixels, c, ldx, ldp, 0, c, 0, x, (f16*)planes, reinterpret_cast<const unsigned*>(scale), scale);
    tag_kernel("split2_f16_kernel");
    return check_launch("f32_split2");
}

// a packed fp32 weight [rows = cout * taps][c] -> the operand e2eft_conv2d_fwd_f32split / e2eft_gemm_f32split multiply with: f16 [rows][w0 (c) | w1 (c) | w0 (c)] with
// w s_w = w0 + w1, scale[1] = s_w, scale[2] = 1 / s_w (the consumers' w_inv_scale = scale + 2).  Two launches on the device: trainable weights are split once per
// optimizer step and parameter, several hundred times per step — as a dozen tensor-library calls each that was the largest host-side cost of the fp32 step.
extern "C" int e2eft_f32_split_weight(const float* w, int64_t rows, int32_t c, void* w_split, float* scale, void* stream) {
    E2EFT_REQUIRE(w && w_split && scale && rows > 0, "f32_split_weight: null pointer / empty weight");
    E2EFT_REQUIRE(c > 0 && c % 8 == 0, "f32_split_weight: c=%d must be a multiple of 8", c);
    E2EFT_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)scale & 3) == 0, "f32_split_weight: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scale, 0, sizeof(float), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "f32_split_weight: memset failed");
    const long units = rows * (c / 8);
    long nb = (units + 255) / 256;
    const long nbm = nb > 1024 ? 1024 : nb;
    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)nbm), dim3(256), 0, s, (long)rows, c, c, w, reinterpret_cast<unsigned*>(scale));
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(split2_f16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)rows, c, c, 3 * c, 0, c, 1, w, (f16*)w_split, reinterpret_cast<const unsigned*>(scale), scale);
    return check_launch("f32_split_weight");
}

// the channel concatenation [x1 (c1) | x2 (c2)] of two fp32 tensors as ONE pair of planes under ONE scale (the two-source convolutions of the UNet's up blocks:
// unet_2d_blocks.py:2328,2456 `torch.cat([hidden_states, res_hidden_states], dim=1)`): planes [pixels][x1_0 | x2_0 | x1_1 | x2_1], ldp >= 2 (c1 + c2)
extern "C" int e2eft_f32_split2_cat(const float* x1, int32_t c1, int32_t ldx1, const float* x2, int32_t c2, int32_t ldx2, int64_t pixels, void* planes, int32_t ldp,
                                    float* scale, void* stream) {
    E2EFT_REQUIRE(x1 && x2 && planes && scale && pixels > 0, "f32_split2_cat: null pointer / empty tensor");
    E2EFT_REQUIRE(c1 > 0 && c1 % 8 == 0 && c2 > 0 && c2 % 8 == 0 && ldx1 >= c1 && ldx1 % 4 == 0 && ldx2 >= c2 && ldx2 % 4 == 0 && ldp >= 2 * (c1 + c2) && ldp % 8 == 0,
                  "f32_split2_cat: c1=%d c2=%d ldx1=%d ldx2=%d ldp=%d", c1, c2, ldx1, ldx2, ldp);
    E2EFT_REQUIRE(((uintptr_t)x1 & 15) == 0 && ((uintptr_t)x2 & 15) == 0 && ((uintptr_t)planes & 15) == 0 && ((uintptr_t)scale & 3) == 0, "f32_split2_cat: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scale, 0, sizeof(float), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "f32_split2_cat: memset failed");
    const float* xs[2] = {x1, x2};
    const int cs[2] = {c1, c2}, lds[2] = {ldx1, ldx2};
    for (int i = 0; i < 2; ++i) {      // one maximum over both sources
        long nb = (pixels * (cs[i] / 8) + 255) / 256;
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)pixels, cs[i], lds[i], xs[i], reinterpret_cast<unsigned*>(scale));
    }
    for (int i = 0; i < 2; ++i) {
        long nb = (pixels * (cs[i] / 8) + 255) / 256;
        if (nb > 65536) nb = 65536;
        hipLaunchKernelGGL(split2_f16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (long)pixels, cs[i], lds[i], ldp, i ? c1 : 0, c1 + c2, 0, xs[i], (f16*)planes,
                           reinterpret_cast<const unsigned*>(scale), scale);
    }
    tag_kernel("split2_f16_kernel");
    return check_launch("f32_split2_cat");
}

// pure host arithmetic: would e2eft_conv2d_fwd_f32split take this launch (with 16-byte aligned pointers)?  desc: dtype E2EFT_F32, c1 = channels of the fp32
// tensor, ldx1 = pixel stride of the PLANES in f16 elements (>= 2 c1), ldw = weight row length in f16 elements (>= kh kw 3 c1), ldo / ldr in fp32 elements
extern "C" int e2eft_conv2d_fwd_f32split_supported(const E2eftConvDesc* d) {
    IgemmParams p;
    if (!f32split_params(d, p)) return 0;
    void* const al = (void*)(uintptr_t)256;
    p.x1 = al; p.w = al; p.out = al;
    IgemmParams q = p;
    if (igemm_patch_eligible(E2EFT_F16, 1, q, 1)) return 1;
    q = p;
    return igemm_persistent_eligible(E2EFT_F16, 1, q, 1) ? 1 : 0;
}

extern "C" int e2eft_conv2d_fwd_f32split(const E2eftConvDesc* d, const void* planes, const float* scale, const void* w_split, const float* w_inv_scale, const float* bias,
                                         const float* residual, float* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && planes && w_split && out, "conv2d_fwd_f32split: null pointer");
    IgemmParams p;
    if (!f32split_params(d, p)) return fail(E2EFT_ERR_UNSUPPORTED, "conv2d_fwd_f32split: not an fp32 convolution of 64-channel multiples from split planes (ask e2eft_conv2d_fwd_f32split_supported)");
    E2EFT_REQUIRE(((uintptr_t)planes & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)out & 15) == 0, "conv2d_fwd_f32split: pointers must be 16-byte aligned");
    p.x1 = planes; p.w = w_split; p.bias = bias; p.residual = residual; p.out = out;
    p.alpha_dev = scale ? scale + 2 : nullptr;
    p.alpha_dev2 = w_inv_scale;
    if (gn_partial && slab_rows) {
        const size_t need = (size_t)d->batch * (size_t)cdiv(p.rows_per_img, 128) * (size_t)d->cout * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "conv2d_fwd_f32split: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
        p.gn_partial = gn_partial;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_igemm_patch(E2EFT_F16, 1, p, 1, s);                         // 3x3 / stride 1 / pad 1 on a 32 x 8 grid: the halo-patch kernel
    if (rc < 0) rc = launch_igemm_persistent(E2EFT_F16, 1, p, 1, s);            // everything else in whole 256-row tiles: igemm5
    if (rc < 0 && p.gn_partial) {                                               // statistics need whole tiles inside one image: without them (the consumer runs its own pass)
        p.gn_partial = nullptr;
        rc = launch_igemm_persistent(E2EFT_F16, 1, p, 1, s);
    }
    if (rc < 0) return fail(E2EFT_ERR_UNSUPPORTED, "conv2d_fwd_f32split: this launch is not eligible for the persistent kernels");
    if (rc == 0 && slab_rows && p.gn_partial) *slab_rows = p.rows_per_img / p.gn_nslabs;
    return rc;
}

// ---- nn.Linear in fp32 from split planes: out[m][n] = alpha / (s s_w) * sum_k a[m][k] w[n][k] + bias[n] + residual[m][n] on igemm5's GEMM mode.  `d`: dtype E2EFT_F32, k = the
// fp32 operand's columns (% 64 == 0), lda = row stride of the PLANES in f16 elements (>= 2 k), ldw = row length of w_split [n][w0 (k) | w1 (k) | w0 (k)] (>= 3 k), ldo / ldr
// in fp32 elements, one problem (nzo = nzi = 1), bias along n.  m > 256 (a ragged last 256-row tile is masked): ask e2eft_gemm_f32split_supported first.
static bool f32split_gemm_params(const E2eftGemmDesc* d, IgemmParams& p) {
    if (!d || d->dtype != E2EFT_F32 || !option(E2EFT_OPT_F32_SPLIT) || d->nzo != 1 || d->nzi != 1 || d->bias_along_m) return false;
    if (d->m <= 0 || d->n <= 0 || d->k <= 0 || d->k % 64 != 0 || d->lda < 2 * d->k || d->lda % 8 != 0 || d->ldw < 3 * d->k || d->ldw % 8 != 0 || d->ldo < d->n) return false;
    p = IgemmParams{};
    p.M = d->m; p.N = d->n; p.K = 3 * d->k;
    p.ldx1 = d->lda; p.c1 = 3 * d->k; p.cin = 3 * d->k; p.split_c = d->k;
    p.ldw = d->ldw; p.ldr = d->ldr > 0 ? d->ldr : d->ldo; p.ldo = d->ldo;
    p.rows_per_img = 1;
    p.alpha = d->alpha;
    p.nzi = 1;
    return true;
}

extern "C" int e2eft_gemm_f32split_supported(const E2eftGemmDesc* d) {
    IgemmParams p;
    if (!f32split_gemm_params(d, p)) return 0;
    void* const al = (void*)(uintptr_t)256;
    p.x1 = al; p.w = al; p.out = al;
    return igemm_persistent_eligible(E2EFT_F16, 0, p, 1) ? 1 : 0;
}

extern "C" int e2eft_gemm_f32split(const E2eftGemmDesc* d, const void* planes, const float* scale, const void* w_split, const float* w_inv_scale, const float* bias,
                                   const float* residual, float* out, void* stream) {
    E2EFT_REQUIRE(d && planes && w_split && out, "gemm_f32split: null pointer");
    IgemmParams p;
    if (!f32split_gemm_params(d, p)) return fail(E2EFT_ERR_UNSUPPORTED, "gemm_f32split: not an fp32 GEMM of 64-column multiples from split planes (ask e2eft_gemm_f32split_supported)");
    E2EFT_REQUIRE(((uintptr_t)planes & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)out & 15) == 0, "gemm_f32split: pointers must be 16-byte aligned");
    p.x1 = planes; p.w = w_split; p.bias = bias; p.residual = residual; p.out = out;
    p.alpha_dev = scale ? scale + 2 : nullptr;
    p.alpha_dev2 = w_inv_scale;
    const int rc = launch_igemm_persistent(E2EFT_F16, 0, p, 1, (hipStream_t)stream);
    if (rc < 0) return fail(E2EFT_ERR_UNSUPPORTED, "gemm_f32split: this launch is not eligible for the persistent kernel");
    return rc;
}

// ---- nearest-2x upsample + 3x3 / stride-1 / pad-1 convolution in fp32 as four 2x2 phase convolutions of the split planes (e2eft_upconv2x_fwd's algebra, igemm.hip).
// `d` as e2eft_upconv2x_fwd's (dtype E2EFT_F32, hl = 2 hin, ...) with ldx1 = the planes' pixel stride in f16 elements; w_phase_split: f16 [4][cout][2][2][w0 | w1 | w0]
// (the fp32 phase weights of autograd.phase_conv_weight split as ONE tensor: one scale).  The halo-patch kernel's 2x2-tap variant where the grid allows, else igemm5.
static bool f32split_upconv_params(const E2eftConvDesc* d, int ph, IgemmParams& p) {
    if (!d || d->dtype != E2EFT_F32 || d->c2 != 0 || !option(E2EFT_OPT_F32_SPLIT) || !option(E2EFT_OPT_UPCONV_PHASES)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->hl != 2 * d->hin || d->wl != 2 * d->win || d->hout != d->hl || d->wout != d->wl) return false;
    if (d->batch <= 0 || d->hin <= 0 || d->win <= 0 || d->c1 <= 0 || d->c1 % 64 != 0 || d->cout % 8 != 0 || d->ldo % 8 != 0 || d->alpha != 1.0f) return false;
    if (d->ldx1 < 2 * d->c1 || d->ldx1 % 8 != 0 || d->win % 16 != 0 || ((long)d->hin * d->win) % 256 != 0 || (long)d->batch * d->hin * d->win >= 2147483647L) return false;
    const int py = ph >> 1, px = ph & 1;
    const int rows_img = d->hin * d->win;
    p = IgemmParams{};
    p.M = d->batch * rows_img; p.N = d->cout; p.K = 4 * 3 * d->c1;
    p.ldx1 = d->ldx1; p.c1 = 3 * d->c1; p.cin = 3 * d->c1; p.split_c = d->c1;
    p.hin = d->hin; p.win = d->win; p.hl = d->hin; p.wl = d->win;
    p.kh = 2; p.kw = 2; p.stride = 1; p.pad_t = 1 - py; p.pad_l = 1 - px;
    p.hout = d->hin; p.wout = d->win;
    p.up_sh = p.up_sw = 1.f;
    p.ldw = 12 * d->c1; p.ldo = 2 * d->ldo; p.ldr = p.ldo;
    p.out_seg = d->win;
    p.rows_per_img = rows_img;
    p.alpha = 1.f;
    p.nzi = 1;
    return true;
}

extern "C" int e2eft_upconv2x_fwd_f32split_supported(const E2eftConvDesc* d) {
    IgemmParams p;
    if (!f32split_upconv_params(d, 0, p)) return 0;
    void* const al = (void*)(uintptr_t)256;
    p.x1 = al; p.w = al; p.out = al;
    IgemmParams q = p;
    if (igemm_patch_eligible(E2EFT_F16, 1, q, 1)) return 1;
    q = p;
    return igemm_persistent_eligible(E2EFT_F16, 1, q, 1) ? 1 : 0;
}

extern "C" int e2eft_upconv2x_fwd_f32split(const E2eftConvDesc* d, const void* planes, const float* scale, const void* w_phase_split, const float* w_inv_scale,
                                           const float* bias, float* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && planes && w_phase_split && out, "upconv2x_f32split: null pointer");
    if (!e2eft_upconv2x_fwd_f32split_supported(d)) return fail(E2EFT_ERR_UNSUPPORTED, "upconv2x_f32split: this launch is not eligible (ask e2eft_upconv2x_fwd_f32split_supported)");
    E2EFT_REQUIRE(((uintptr_t)planes & 15) == 0 && ((uintptr_t)w_phase_split & 15) == 0 && ((uintptr_t)out & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "upconv2x_f32split: pointers must be 16-byte aligned");
    const int rows_img = d->hin * d->win, slabs = rows_img / 256;
    const bool stats = gn_partial && slab_rows;
    if (stats) {
        const size_t need = (size_t)d->batch * (size_t)cdiv(4 * rows_img, 128) * (size_t)d->cout * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "upconv2x_f32split: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
    }
    for (int ph = 0; ph < 4; ++ph) {
        IgemmParams p;
        f32split_upconv_params(d, ph, p);
        const int py = ph >> 1, px = ph & 1;
        p.x1 = planes; p.w = (const char*)w_phase_split + (size_t)ph * d->cout * 12 * d->c1 * 2; p.bias = bias;
        p.out = (char*)out + ((size_t)py * d->wl + px) * d->ldo * 4;
        p.alpha_dev = scale ? scale + 2 : nullptr;
        p.alpha_dev2 = w_inv_scale;
        if (stats) {
            p.gn_partial = gn_partial + (size_t)ph * slabs * d->cout * 3;
            p.gn_islabs = 4 * slabs;
        }
        p.mtiles = p.M / 256; p.ntiles = cdiv(p.N, 128);
        int rc = launch_igemm_patch(E2EFT_F16, 1, p, 1, (hipStream_t)stream);
        if (rc < 0) rc = launch_igemm_persistent(E2EFT_F16, 1, p, 1, (hipStream_t)stream);
        if (rc < 0) return fail(E2EFT_ERR_UNSUPPORTED, "upconv2x_f32split: the persistent kernels declined phase %d", ph);
        if (rc) return rc;
    }
    if (stats) *slab_rows = 256;
    return E2EFT_OK;
}
