// igemm.hip — host side of the implicit-GEMM convolution / NT-GEMM family for gfx950 (MI355X): argument checks, the split-K plan and its
// finish kernel, kernel choice.
//
//   out[m, n] = alpha * ( sum_k A[m, k] * W[n, k] + bias + rowadd[img(m), n] ) + residual[m, n]
//
// One kernel family serves every dense contraction of the path: conv3x3 (s1/s2, symmetric or VAE-asymmetric padding,
// fused nearest upsample, fused channel concat of two sources), conv1x1, nn.Linear and the batched attention matmuls of the unfused
// path.  The kernels live in igemm5.hip (persistent workgroups, the big 16-bit launches) and igemm2.hip (everything else, incl. exact
// fp32 on v_mfma_f32_32x32x2_f32).  (Round 1's register-staged 128x128 kernel and round 2's row-strip kernel igemm4 were retired in
// round 3: neither was on any production path; their text is in the history: git show 964b305:scripts/experiments/.)
#include "common.h"
#include "igemm.h"

namespace e2eft {

constexpr int BM = 128, BN = 128;   // tile counts reported to the kernels' launchers (each re-derives its own tile shape)

// split-K finish: out = alpha * (sum_z part[z] + bias + rowadd[img(m)]) + residual, 16-byte vectors; thread per (row, chunk)
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(int M, int N, int nz, const T* __restrict__ part, long pstride, const T* __restrict__ bias,
                                                            const T* __restrict__ rowadd, int rows_per_img, float alpha,
                                                            const T* __restrict__ res, int ldr, T* __restrict__ out, int ldo) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cch = N / EPC;
    const long total = (long)M * cch;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long m = it / cch;
        const int n = (int)(it - m * cch) * EPC;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int z = 0; z < nz; ++z) {
            const Vec16<T> v = ld16(part + z * pstride + m * N + n);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += to_f(v.e[e]);
        }
        if (bias) {
            const Vec16<T> v = ld16(bias + n);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += to_f(v.e[e]);
        }
        if (rowadd) {
            const Vec16<T> v = ld16(rowadd + (m / rows_per_img) * N + n);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += to_f(v.e[e]);
        }
        Vec16<T> o;
        if (res) {
            const Vec16<T> v = ld16(res + m * ldr + n);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(fmaf(acc[e], alpha, to_f(v.e[e])));
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(acc[e] * alpha);
        }
        st16(out + m * ldo + n, o);
    }
}

// split-K plan of a convolution: by rows of filter taps, for problems that leave most CUs idle (the 9^2-24^2 UNet layers: ~90
// 128x128 tiles with K up to 23040).  0 = do not split.
static int conv_splitk_plan(const E2eftConvDesc* d) {
    const int epc = 16 / (int)dtype_size(d->dtype);
    const int bk = 128 / (int)dtype_size(d->dtype);
    const int cin = d->c1 + d->c2;
    const long M = (long)d->batch * d->hout * d->wout;
    const long tiles = ((M + 127) / 128) * ((d->cout + 127) / 128);
    if (d->kh < 2 || tiles >= 160 || (long)d->kh * d->kw * cin < 2304) return 0;
    if (cin % bk != 0 || d->c1 % bk != 0 || d->cout % epc != 0) return 0;          // FAST path + vector finish
    const long img_bytes = (long)d->hin * d->win * (d->ldx1 > d->ldx2 ? d->ldx1 : d->ldx2) * (long)dtype_size(d->dtype);
    if (img_bytes * (256 / ((long)d->hout * d->wout) + 2) >= 0xD0000000L) return 0;
    return d->kh;
}

static int run_igemm(int dtype, int mode, IgemmParams& p, int nz, void* stream) {
    p.mtiles = cdiv(p.M, BM);
    p.ntiles = cdiv(p.N, BN);
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return fail(E2EFT_ERR_BAD_ARG, "igemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    if ((long)p.mtiles * p.ntiles > 2000000000L) return fail(E2EFT_ERR_BAD_ARG, "igemm: grid too large");
    if (nz > 65535) return fail(E2EFT_ERR_BAD_ARG, "igemm: batch %d > 65535", nz);
    hipStream_t s = (hipStream_t)stream;
    if (dtype < 0 || dtype > 2) return fail(E2EFT_ERR_BAD_ARG, "igemm: bad dtype %d", dtype);
    if (p.nrm_ad) {   // the fused-normalisation route exists on igemm6 only: the caller asked e2eft_conv2d_fwd_normed_supported first
        const int rcn = launch_igemm_patch(dtype, mode, p, nz, s);
        return rcn >= 0 ? rcn : fail(E2EFT_ERR_UNSUPPORTED, "conv2d_fwd_normed: this launch is not eligible for the fused-normalisation kernel");
    }
    const int rc7 = launch_conv_thin_in(dtype, mode, p, nz, s);      // conv_in: eight input channels (convin.hip)
    if (rc7 >= 0) return rc7;
    const int rc6 = launch_igemm_patch(dtype, mode, p, nz, s);        // big 16-bit 3x3 convolutions: halo patch in LDS (igemm6.hip)
    if (rc6 >= 0) return rc6;
    const int rc5 = launch_igemm_persistent(dtype, mode, p, nz, s);   // big 16-bit problems: persistent workgroups (igemm5.hip)
    if (rc5 >= 0) return rc5;
    return launch_igemm_v2(dtype, mode, p, nz, s);                    // everything else: igemm2.hip (256- or 128-row tiles)
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int launch_conv3x3_narrow(const E2eftConvDesc* d, const void* x1, const void* w, const void* bias, void* out, void* stream, const float* nrm_ad = nullptr,
                          const void* nrm_beta = nullptr, int nrm_silu = 0);   // narrow.hip
bool conv3x3_narrow_eligible(const E2eftConvDesc* d, bool normed);

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_conv2d_fwd(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w,
                                const void* bias, const void* rowadd, const void* residual, void* out, void* stream) {
    return e2eft_conv2d_fwd_gnstats(d, x1, x2, w, bias, rowadd, residual, out, nullptr, 0, nullptr, stream);
}

extern "C" size_t e2eft_conv2d_splitk_workspace_bytes(const E2eftConvDesc* d) {
    if (!d || d->dtype < 0 || d->dtype > 2) return 0;
    const int ns = conv_splitk_plan(d);
    return ns ? (size_t)ns * d->batch * d->hout * d->wout * d->cout * dtype_size(d->dtype) : 0;
}

struct ConvNorm { const float* ad; const void* beta; int silu; };   // the input is read through GroupNorm(+SiLU): e2eft_conv2d_fwd_normed
static int conv2d_core(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w, const void* bias, const void* rowadd,
                       const void* residual, void* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* workspace,
                       size_t ws_bytes, void* stream, const ConvNorm* norm = nullptr);

extern "C" int e2eft_conv2d_fwd_splitk(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w, const void* bias,
                                       const void* rowadd, const void* residual, void* out, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(d && workspace, "conv2d_splitk: null pointer");
    const size_t need = e2eft_conv2d_splitk_workspace_bytes(d);
    E2EFT_REQUIRE(need > 0, "conv2d_splitk: this problem is not split (e2eft_conv2d_splitk_workspace_bytes == 0)");
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "conv2d_splitk: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(((uintptr_t)workspace & 15) == 0, "conv2d_splitk: workspace must be 16-byte aligned");
    return conv2d_core(d, x1, x2, w, bias, rowadd, residual, out, nullptr, 0, nullptr, workspace, ws_bytes, stream);
}

extern "C" int e2eft_conv2d_fwd_gnstats(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w,
                                        const void* bias, const void* rowadd, const void* residual, void* out,
                                        float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    return conv2d_core(d, x1, x2, w, bias, rowadd, residual, out, gn_partial, gn_partial_bytes, slab_rows, nullptr, 0, stream);
}

extern "C" int e2eft_conv2d_fwd_normed(const E2eftConvDesc* d, const void* x1, const float* coeff, const void* beta, int32_t silu, const void* w,
                                      const void* bias, const void* rowadd, const void* residual, void* out, float* gn_partial,
                                      size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    E2EFT_REQUIRE(coeff, "conv2d_fwd_normed: null coefficients");
    E2EFT_REQUIRE(d && d->c2 == 0, "conv2d_fwd_normed: one source only");
    const ConvNorm n = {coeff, beta, silu ? 1 : 0};
    return conv2d_core(d, x1, nullptr, w, bias, rowadd, residual, out, gn_partial, gn_partial_bytes, slab_rows, nullptr, 0, stream, &n);
}

// pure host arithmetic: would e2eft_conv2d_fwd_normed take this launch (with 16-byte aligned pointers)?
extern "C" int e2eft_conv2d_fwd_normed_supported(const E2eftConvDesc* d) {
    if (!d || d->c2 != 0 || d->dtype < 1 || d->dtype > 2 || !option(E2EFT_OPT_FUSED_NORM)) return 0;
    if (d->batch <= 0 || d->hin <= 0 || d->win <= 0 || d->cout <= 0 || d->c1 <= 0) return 0;
    if (conv3x3_narrow_eligible(d, true)) return 1;   // conv_norm_out -> conv_out
    IgemmParams p = {};
    void* const al = (void*)(uintptr_t)256;
    p.x1 = al; p.w = al; p.out = al; p.nrm_ad = (const float*)al;
    p.M = d->batch * d->hout * d->wout; p.N = d->cout; p.K = d->kh * d->kw * d->c1;
    p.ldx1 = d->ldx1; p.c1 = d->c1; p.cin = d->c1;
    p.hin = d->hin; p.win = d->win; p.hl = d->hl; p.wl = d->wl;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.hout = d->hout; p.wout = d->wout;
    p.ldw = d->ldw; p.ldr = d->ldr > 0 ? d->ldr : d->ldo; p.ldo = d->ldo;
    p.rows_per_img = d->hout * d->wout;
    p.nzi = 1;
    return igemm_patch_eligible(d->dtype, 1, p, 1) ? 1 : 0;
}

static int conv2d_core(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w, const void* bias, const void* rowadd,
                       const void* residual, void* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* workspace,
                       size_t ws_bytes, void* stream, const ConvNorm* norm) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && x1 && w && out, "conv2d: null pointer");
    const int epc = 16 / (int)dtype_size(d->dtype);
    const int cin = d->c1 + d->c2;
    E2EFT_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "conv2d: bad dtype %d", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->hin > 0 && d->win > 0 && d->hout > 0 && d->wout > 0 && d->cout > 0, "conv2d: bad geometry");
    E2EFT_REQUIRE(d->c1 > 0 && d->c1 % epc == 0 && d->c2 >= 0 && d->c2 % epc == 0, "conv2d: channels (%d,%d) must be multiples of %d", d->c1, d->c2, epc);
    E2EFT_REQUIRE(d->ldx1 >= d->c1 && d->ldx1 % epc == 0, "conv2d: ldx1=%d", d->ldx1);
    E2EFT_REQUIRE(d->c2 == 0 || (x2 && d->ldx2 >= d->c2 && d->ldx2 % epc == 0), "conv2d: second source");
    E2EFT_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0, "conv2d: kernel/stride");
    E2EFT_REQUIRE(d->ldw >= d->kh * d->kw * cin && d->ldw % epc == 0, "conv2d: ldw=%d", d->ldw);
    E2EFT_REQUIRE(d->ldo >= d->cout, "conv2d: ldo=%d < cout=%d", d->ldo, d->cout);
    E2EFT_REQUIRE(d->hl >= d->hin && d->wl >= d->win, "conv2d: logical size smaller than physical");
    E2EFT_REQUIRE(al16(x1) && al16(w) && (d->c2 == 0 || al16(x2)), "conv2d: pointers must be 16-byte aligned");
    E2EFT_REQUIRE((long)d->batch * d->hout * d->wout < 2147483647L, "conv2d: M overflows int32");
    // the last output row/col must read at least one in-range tap row/col origin
    E2EFT_REQUIRE((d->hout - 1) * d->stride - d->pad_t < d->hl && (d->wout - 1) * d->stride - d->pad_l < d->wl, "conv2d: output larger than padded input");

    if (!x2 && !rowadd && !residual && !gn_partial && !workspace) {   // <= 4 output channels: LDS-halo kernel instead of a 128-wide MFMA tile
        const int rn = norm ? launch_conv3x3_narrow(d, x1, w, bias, out, stream, norm->ad, norm->beta, norm->silu) : launch_conv3x3_narrow(d, x1, w, bias, out, stream);
        if (rn >= 0) return rn;
    }

    IgemmParams p = {};
    p.x1 = x1; p.x2 = x2; p.w = w; p.bias = bias; p.rowadd = rowadd; p.residual = residual; p.out = out;
    p.M = d->batch * d->hout * d->wout;
    p.N = d->cout;
    p.K = d->kh * d->kw * cin;
    p.ldx1 = d->ldx1; p.ldx2 = d->ldx2; p.c1 = d->c1; p.cin = cin;
    p.hin = d->hin; p.win = d->win; p.hl = d->hl; p.wl = d->wl;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.hout = d->hout; p.wout = d->wout;
    p.up_sh = (float)d->hin / (float)d->hl;
    p.up_sw = (float)d->win / (float)d->wl;
    p.ldw = d->ldw; p.ldr = d->ldr; p.ldo = d->ldo;
    p.bias_along_m = 0;
    p.rows_per_img = d->hout * d->wout;
    p.alpha = d->alpha;
    p.nzi = 1;
    if (norm) { p.nrm_ad = norm->ad; p.nrm_beta = norm->beta; p.nrm_silu = norm->silu; }
    const bool plain = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->c2 == 0 &&
                       d->hl == d->hin && d->wl == d->win && d->hout == d->hin && d->wout == d->win;
    if (gn_partial && slab_rows) {
        const size_t need = (size_t)d->batch * (size_t)cdiv(p.rows_per_img, 128) * (size_t)d->cout * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "conv2d: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
        p.gn_partial = gn_partial;
    }
    if (workspace) {   // split-K over rows of filter taps: partial sums in the workspace, then one finish pass with the whole epilogue
        const int ns = conv_splitk_plan(d);
        const int epc2 = 16 / (int)dtype_size(d->dtype);
        E2EFT_REQUIRE(ns > 0 && !plain, "conv2d_splitk: not a split problem");
        E2EFT_REQUIRE(d->ldo % epc2 == 0 && ((uintptr_t)out & 15) == 0 && (!residual || (d->ldr % epc2 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                          (!bias || ((uintptr_t)bias & 15) == 0) && (!rowadd || ((uintptr_t)rowadd & 15) == 0), "conv2d_splitk: vector alignment");
        IgemmParams q = p;
        q.bias = nullptr; q.rowadd = nullptr; q.residual = nullptr; q.gn_partial = nullptr;
        q.out = workspace; q.ldo = d->cout; q.alpha = 1.f;
        q.ksplit_taps = d->kw;
        q.nzi = ns; q.sa_o = q.sa_i = q.sw_o = q.sw_i = 0; q.so_o = 0; q.so_i = (long)p.M * d->cout; q.sr_o = q.sr_i = 0;
        const int rc2 = run_igemm(d->dtype, 1, q, ns, stream);
        if (rc2) return rc2;
        const long total = (long)p.M * (d->cout / epc2);
        long nb = (total + 255) / 256;
        if (nb > 16384) nb = 16384;
        hipStream_t s = (hipStream_t)stream;
        E2EFT_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3((unsigned)nb), dim3(256), 0, s, p.M, d->cout, ns, (const T*)workspace,
                                                          (long)p.M * d->cout, (const T*)bias, (const T*)rowadd, p.rows_per_img, d->alpha, (const T*)residual,
                                                          d->ldr, (T*)out, d->ldo));
        return check_launch("conv2d_splitk");
    }
    const int rc = run_igemm(d->dtype, plain ? 0 : 1, p, 1, stream);
    if (rc == 0 && slab_rows && p.gn_partial) *slab_rows = p.rows_per_img / p.gn_nslabs;
    return rc;
}


// ---- nearest-2x upsample + 3x3 / stride-1 / pad-1 convolution as FOUR 2x2 convolutions of the low-resolution input -------------------------------------
// (diffusers Upsample2D: F.interpolate(scale_factor=2, mode="nearest") then conv; unet_2d_blocks.py via the reference's upsamplers, the VAE decoder's UpDecoderBlock2D.)
// Output pixel (2Y + py, 2X + px) reads the upsampled rows 2Y + py - 1 .. 2Y + py + 1, i.e. the SOURCE rows {Y - 1, Y, Y} (py = 0) or {Y, Y, Y + 1} (py = 1): two distinct
// rows, the 3x3 taps that fall on the same source pixel can be added up front.  Per parity phase (py, px) the layer is a 2x2 convolution (pad_t = 1 - py, pad_l = 1 - px)
// with weights w_phase[2 py + px][co][(i, j, ci)] = sum of the 3x3 taps that map to source offset (i, j) — 4/9 of the multiply-adds of the fused-upsample form, same
// zero padding (the upsampled image's border IS the source's border).  The phases write interleaved pixels of the full-resolution output: IgemmParams.out_seg.
namespace e2eft { int device_cus(); }   // api.hip
static bool upconv2x_shape_ok(const E2eftConvDesc* d) {     // the layer is a 2x-upsample + 3x3 / stride-1 / pad-1 convolution whose phases any igemm kernel can run
    if (!d || d->dtype < 0 || d->dtype > 2) return false;
    const int bk = 128 / (int)dtype_size(d->dtype);          // one k-tile of the FAST operand path
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->c2 == 0 && d->hl == 2 * d->hin && d->wl == 2 * d->win &&
           d->hout == d->hl && d->wout == d->wl && d->batch > 0 && d->hin > 0 && d->win > 0 && d->c1 > 0 && d->c1 % bk == 0 && d->cout % (bk / 8) == 0 &&
           d->ldo % (bk / 8) == 0 && d->alpha == 1.0f && (long)d->batch * d->hin * d->win < 2147483647L &&
           (long)d->hin * d->win * d->ldx1 * (long)dtype_size(d->dtype) * 3 < 0xD0000000L && (long)128 * 4 * d->c1 * (long)dtype_size(d->dtype) < 0x40000000L;
}
// the persistent kernel takes the phases (and emits GroupNorm statistics): 16-bit, whole 256-row tiles inside one image, segments of a multiple of 16 rows, >= 2 tiles per CU
static bool upconv2x_persistent(const E2eftConvDesc* d) {
    if (d->dtype == E2EFT_F32 || !option(E2EFT_OPT_PERSISTENT)) return false;
    if (d->c1 % 64 != 0 || d->win % 16 != 0 || ((long)d->batch * d->hin * d->win) % 256 != 0 || ((long)d->hin * d->win) % 256 != 0) return false;
    int cus = device_cus();
    const int gopt = option(E2EFT_OPT_PERSISTENT_GRID);
    if (gopt >= 8 && gopt < cus) cus = gopt;
    if (cus <= 0) return false;
    return ((long)d->batch * d->hin * d->win / 256) * cdiv(d->cout, 128) >= 2L * cus;
}

extern "C" int e2eft_upconv2x_fwd_supported(const E2eftConvDesc* d) {
    if (!upconv2x_shape_ok(d) || !option(E2EFT_OPT_UPCONV_PHASES)) return 0;
    if (upconv2x_persistent(d)) return 1;
    // igemm2 runs the phases as four plain launches: that only beats ONE fused-upsample launch (on igemm6 / igemm5 for 16-bit) when every phase fills the machine —
    // at 12^2 -> 24^2 with 8 images a phase is 90 workgroups and four of them cost 0.34 ms against 0.14 ms (r05c); same bar as the persistent kernel: two tiles per CU
    int cus = device_cus();
    const int gopt = option(E2EFT_OPT_PERSISTENT_GRID);
    if (gopt >= 8 && gopt < cus) cus = gopt;
    return cus > 0 && cdiv((long)d->batch * d->hin * d->win, 256) * cdiv(d->cout, 128) >= 2L * cus ? 1 : 0;
}

extern "C" int e2eft_upconv2x_fwd(const E2eftConvDesc* d, const void* x, const void* w_phase, const void* bias, void* out, float* gn_partial,
                                  size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && x && w_phase && out, "upconv2x: null pointer");
    if (!e2eft_upconv2x_fwd_supported(d)) return fail(E2EFT_ERR_UNSUPPORTED, "upconv2x: this launch is not eligible (ask e2eft_upconv2x_fwd_supported; e2eft_conv2d_fwd serves it)");
    E2EFT_REQUIRE(al16(x) && al16(w_phase) && al16(out) && (!bias || al16(bias)), "upconv2x: pointers must be 16-byte aligned");
    const bool pers = upconv2x_persistent(d);                // all four phases take the same kernel: the statistics slabs must agree
    const int rows_img = d->hin * d->win;                    // GEMM rows of one image in ONE phase
    const int slabs = rows_img / 256;
    const bool stats = pers && gn_partial && slab_rows;      // (igemm2 serves the phases without statistics: the consuming GroupNorm runs its own pass)
    if (stats) {
        const size_t need = (size_t)d->batch * (size_t)cdiv(4 * rows_img, 128) * (size_t)d->cout * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "upconv2x: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
    }
    const size_t es = dtype_size(d->dtype);
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        IgemmParams p = {};
        p.x1 = x; p.w = (const char*)w_phase + (size_t)ph * d->cout * 4 * d->c1 * es; p.bias = bias;
        p.out = (char*)out + ((size_t)py * d->wl + px) * d->ldo * es;
        p.M = d->batch * rows_img; p.N = d->cout; p.K = 4 * d->c1;
        p.ldx1 = d->ldx1; p.c1 = d->c1; p.cin = d->c1;
        p.hin = d->hin; p.win = d->win; p.hl = d->hin; p.wl = d->win;
        p.kh = 2; p.kw = 2; p.stride = 1; p.pad_t = 1 - py; p.pad_l = 1 - px;
        p.hout = d->hin; p.wout = d->win;
        p.up_sh = p.up_sw = 1.f;
        p.ldw = 4 * d->c1; p.ldo = 2 * d->ldo;
        p.out_seg = d->win;
        p.rows_per_img = rows_img;
        p.alpha = 1.f;
        p.nzi = 1;
        if (stats) {       // the four phases deposit into disjoint slab ranges of one [image][4 * slabs][cout][3] buffer
            p.gn_partial = gn_partial + (size_t)ph * slabs * d->cout * 3;
            p.gn_islabs = 4 * slabs;
        }
        int rc;
        if (pers) {
            p.mtiles = p.M / 256; p.ntiles = cdiv(p.N, 128);
            rc = launch_igemm_patch(d->dtype, 1, p, 1, (hipStream_t)stream);      // round 6: 2x2-tap halo-patch kernel (8 x 32-pixel tiles; same 256-row statistics slabs)
            if (rc < 0) rc = launch_igemm_persistent(d->dtype, 1, p, 1, (hipStream_t)stream);
            if (rc < 0) return fail(E2EFT_ERR_UNSUPPORTED, "upconv2x: the persistent kernel declined phase %d", ph);
        } else {
            rc = launch_igemm_v2(d->dtype, 1, p, 1, (hipStream_t)stream);
        }
        if (rc) return rc;
    }
    if (stats) *slab_rows = 256;
    return E2EFT_OK;
}


// Data gradient of the convolution `fwd` describes (forward: out = alpha*(conv(x) + ...) + residual):
//   dx[b, i, j, ci] = alpha * sum_{ky,kx,co} dy_z[b, i + pad_t - ky, j + pad_l - kx, co] * w[co][ky][kx][ci]
// i.e. a stride-1 convolution of dy (read through a zero-insertion grid when fwd->stride > 1) with the spatially flipped,
// channel-transposed weights w_dgrad[ci][(kh-1-ky, kw-1-kx, co)] — the same implicit-GEMM kernel with K = kh*kw*cout.
// dx is the gradient w.r.t. the LOGICAL input [B, hl, wl, c1+c2] (before the fused nearest upsample; reduce with
// e2eft_upsample_nearest_bwd), both concat sources in one buffer.
extern "C" int e2eft_conv2d_dgrad(const E2eftConvDesc* d, const void* dy, int32_t lddy, int32_t cout_pad, const void* w_dgrad, int32_t ldwd, void* dx,
                                  int32_t lddx, void* stream) {
    E2EFT_REQUIRE(d && dy && w_dgrad && dx, "conv2d_dgrad: null pointer");
    E2EFT_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "conv2d_dgrad: bad dtype %d", d->dtype);
    const int epc = 16 / (int)dtype_size(d->dtype);
    const int cin = d->c1 + d->c2;
    E2EFT_REQUIRE(d->batch > 0 && d->hl > 0 && d->wl > 0 && d->hout > 0 && d->wout > 0 && d->cout > 0 && cin > 0, "conv2d_dgrad: bad geometry");
    E2EFT_REQUIRE(cout_pad >= d->cout && cout_pad % epc == 0 && lddy >= cout_pad && lddy % epc == 0, "conv2d_dgrad: dy channels %d (ld %d) must be padded to %d", cout_pad, lddy, epc);
    E2EFT_REQUIRE(ldwd >= d->kh * d->kw * cout_pad && ldwd % epc == 0, "conv2d_dgrad: ldwd=%d", ldwd);
    E2EFT_REQUIRE(lddx >= cin, "conv2d_dgrad: lddx=%d < cin=%d", lddx, cin);
    E2EFT_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad_t < d->kh && d->pad_l < d->kw, "conv2d_dgrad: kernel/stride/padding");
    E2EFT_REQUIRE(al16(dy) && al16(w_dgrad), "conv2d_dgrad: pointers must be 16-byte aligned");
    E2EFT_REQUIRE((long)d->batch * d->hl * d->wl < 2147483647L, "conv2d_dgrad: M overflows int32");
    IgemmParams p = {};
    p.x1 = dy; p.w = w_dgrad; p.out = dx;
    p.M = d->batch * d->hl * d->wl;
    p.N = cin;
    p.K = d->kh * d->kw * cout_pad;
    p.ldx1 = lddy; p.c1 = cout_pad; p.cin = cout_pad;
    p.hin = d->hout; p.win = d->wout;
    p.zins = d->stride;
    p.hl = (d->hout - 1) * d->stride + 1; p.wl = (d->wout - 1) * d->stride + 1;
    p.kh = d->kh; p.kw = d->kw; p.stride = 1; p.pad_t = d->kh - 1 - d->pad_t; p.pad_l = d->kw - 1 - d->pad_l;
    p.hout = d->hl; p.wout = d->wl;
    p.up_sh = p.up_sw = 1.f;
    p.ldw = ldwd; p.ldo = lddx;
    p.rows_per_img = d->hl * d->wl;
    p.alpha = d->alpha;
    p.nzi = 1;
    const bool plain = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->hout == d->hl && d->wout == d->wl;
    return run_igemm(d->dtype, plain ? 0 : 1, p, 1, stream);
}

extern "C" int e2eft_gemm(const E2eftGemmDesc* d, const void* a, const void* w, const void* bias,
                          const void* residual, void* out, void* stream) {
    return e2eft_gemm_gnstats(d, a, w, bias, residual, out, 0, nullptr, 0, nullptr, stream);
}

extern "C" int e2eft_gemm_gnstats(const E2eftGemmDesc* d, const void* a, const void* w, const void* bias,
                                  const void* residual, void* out, int32_t rows_per_image, float* gn_partial,
                                  size_t gn_partial_bytes, int32_t* slab_rows, void* stream) {
    if (slab_rows) *slab_rows = 0;
    E2EFT_REQUIRE(d && a && w && out, "gemm: null pointer");
    E2EFT_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "gemm: bad dtype %d", d->dtype);
    const int epc = 16 / (int)dtype_size(d->dtype);
    E2EFT_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0, "gemm: empty problem");
    E2EFT_REQUIRE(d->k % epc == 0, "gemm: k=%d must be a multiple of %d", d->k, epc);
    E2EFT_REQUIRE(d->lda >= d->k && d->lda % epc == 0 && d->ldw >= d->k && d->ldw % epc == 0, "gemm: lda=%d ldw=%d", d->lda, d->ldw);
    E2EFT_REQUIRE(d->ldo >= d->n, "gemm: ldo=%d < n=%d", d->ldo, d->n);
    E2EFT_REQUIRE(d->nzo >= 1 && d->nzi >= 1, "gemm: batch counts");
    E2EFT_REQUIRE(d->sa_o % epc == 0 && d->sa_i % epc == 0 && d->sw_o % epc == 0 && d->sw_i % epc == 0, "gemm: batch strides must keep 16-byte alignment");
    E2EFT_REQUIRE(al16(a) && al16(w), "gemm: pointers must be 16-byte aligned");
    IgemmParams p = {};
    p.x1 = a; p.x2 = nullptr; p.w = w; p.bias = bias; p.rowadd = nullptr; p.residual = residual; p.out = out;
    p.M = d->m; p.N = d->n; p.K = d->k;
    p.ldx1 = d->lda; p.c1 = d->k; p.cin = d->k;
    p.ldw = d->ldw; p.ldr = d->ldr; p.ldo = d->ldo;
    p.bias_along_m = d->bias_along_m;
    p.rows_per_img = 1;
    p.alpha = d->alpha;
    p.nzi = d->nzi;
    p.sa_o = d->sa_o; p.sa_i = d->sa_i; p.sw_o = d->sw_o; p.sw_i = d->sw_i;
    p.so_o = d->so_o; p.so_i = d->so_i; p.sr_o = d->sr_o; p.sr_i = d->sr_i;
    if (gn_partial && slab_rows && rows_per_image > 0 && d->nzo * d->nzi == 1 && !d->bias_along_m && d->m % rows_per_image == 0) {
        const size_t need = (size_t)(d->m / rows_per_image) * (size_t)cdiv(rows_per_image, 128) * (size_t)d->n * 3 * sizeof(float);
        if (gn_partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "gemm: gn_partial %zu < %zu bytes", gn_partial_bytes, need);
        p.gn_partial = gn_partial;
        p.rows_per_img = rows_per_image;
    }
    const int rc = run_igemm(d->dtype, 0, p, d->nzo * d->nzi, stream);
    if (rc == 0 && slab_rows && p.gn_partial) *slab_rows = p.rows_per_img / p.gn_nslabs;
    return rc;
}
