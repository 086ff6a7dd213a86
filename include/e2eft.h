/*
 * e2eft.h — C ABI of libe2eft.so: the MI355X (gfx950) kernels behind the single-step
 * Marigold / GeoWizard denoising path.
 *
 * The reference (VisualComputingInstitute/diffusion-e2e-ft) is pure Python; the arithmetic of its hot path
 * is executed by torch/diffusers leaf modules.  Each entry point below replaces one op class that those leaf
 * modules dispatch (reference call sites are cited per function, paths relative to /root/reference).
 *
 * Conventions
 *   - Every function returns 0 on success or an E2EFT_ERR_* code; e2eft_last_error() gives the message
 *     (thread-local).  Nothing throws, nothing aborts.
 *   - The caller owns ALL device memory (inputs, outputs, workspaces).  The library allocates nothing on
 *     the device, keeps no pointers after return and never synchronises the host: every kernel is enqueued
 *     on the hipStream_t passed as `void* stream` (0 = the legacy default stream).
 *   - Activations are NHWC ("channels_last"): element (b,y,x,c) of a [B,H,W,C] tensor lives at
 *     ((b*H + y)*W + x)*ld + c, where the pixel stride `ld` >= C lets a tensor be a channel slice of a wider
 *     buffer (zero-copy concat).  Token tensors [B,N,C] are the same thing with H*W = N.
 *   - dtype is one of E2EFT_F32 / E2EFT_F16 / E2EFT_BF16 for all tensors of a call; accumulation, statistics
 *     and softmax are always fp32.
 *   - Alignment: all base pointers 16-byte aligned; channel counts and pixel strides multiples of
 *     16 bytes / sizeof(dtype) unless a function says otherwise.
 *   - State: the thread-local error string and the process-wide tuning options below (e2eft_set_option) are the
 *     ONLY mutable state of the library.  The library never reads the environment.
 */
#ifndef E2EFT_H
#define E2EFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header (and e2eft_debug.h) declares is ALL it exports (tests/test_abi.py) */
#pragma GCC visibility push(default)

#define E2EFT_VERSION 119 /* 0.1.1: backward entry points; 111: test-time ensembling, CLIP towers, sample preparation */

enum {
    E2EFT_OK = 0,
    E2EFT_ERR_BAD_ARG = 1,     /* shape / dtype / alignment rejected */
    E2EFT_ERR_WORKSPACE = 2,   /* workspace too small */
    E2EFT_ERR_LAUNCH = 3,      /* hipGetLastError() != hipSuccess after a launch */
    E2EFT_ERR_UNSUPPORTED = 4, /* valid request this build has no kernel for */
};

enum { E2EFT_F32 = 0, E2EFT_F16 = 1, E2EFT_BF16 = 2 };

int e2eft_version(void);
const char* e2eft_last_error(void);
/* hex id of the sources + flags this library was built from (build.py: sha256 over csrc, include and the hipcc flags); "unstamped" for a hand build */
const char* e2eft_build_id(void);

/* Process-wide tuning options (A/B measurements and tests; the defaults are what the product path runs with).  Setting an
 * option affects launches issued AFTER the call, on every thread.  e2eft_set_option returns E2EFT_ERR_BAD_ARG for an
 * unknown key or an out-of-range value; e2eft_get_option returns -1 for an unknown key. */
enum {
    E2EFT_OPT_PERSISTENT = 0,        /* 1 (default): big 16-bit GEMM / conv launches run on the persistent kernel (igemm5); 0: never */
    E2EFT_OPT_PERSISTENT_GRID = 1,   /* 0 (default): one workgroup per CU of the launch's device; n >= 8 (multiple of 8): at most n */
    E2EFT_OPT_NARROW_CONV = 2,       /* 1 (default): <= 4-output-channel 3x3 convs on the LDS-halo kernel (narrow.hip); 0: MFMA tile */
    E2EFT_OPT_NARROW_MFMA = 3,       /* 1 (default): MFMA 16x16x32 form of that kernel; 0: v_dot2 form */
    E2EFT_OPT_IGEMM_GENERAL_OPERANDS = 4, /* 0 (default); 1: igemm2 takes its general (per-lane gather) operand path for every launch */
    E2EFT_OPT_IGEMM2_WAVES = 5,      /* 0 (default): 8-wave 256-row tiles when >= 128 of them exist (fp32: 256), else 4-wave 128-row; 4 / 8: forced */
    E2EFT_OPT_PATCH_CONV = 6,        /* 1 (default): big 16-bit 3x3 / stride-1 / pad-1 convolutions on the halo-patch kernel (igemm6); 0: igemm5 */
    E2EFT_OPT_THIN_INPUT_CONV = 7,   /* 1 (default): big 16-bit 3x3 / stride-1 / pad-1 convolutions with EIGHT input channels on convin.hip; 0: igemm2 */
    E2EFT_OPT_FUSED_NORM = 8,        /* 1 (default): e2eft_conv2d_fwd_normed_supported may answer 1; 0: it answers 0 (GroupNorm applied by its own pass) */
    E2EFT_OPT_ATTN_DMA = 9,          /* 1 (default since round 5): e2eft_attn_fwd delivers K / V tiles by LDS-DMA into a two-stage ring (K / V below 3.5 GB); 0: staged through registers (bit-identical results) */
    E2EFT_OPT_UPCONV_PHASES = 10,    /* 1 (default): e2eft_upconv2x_fwd_supported may answer 1 (2x-upsample + 3x3 convolutions as four 2x2 phase convolutions); 0: it answers 0 */
    E2EFT_OPT_PATCH_CONV_2X2 = 11,   /* 1 (default, round 6): the 2x2 parity phases of e2eft_upconv2x_fwd (and any eligible 2x2 / stride-1 convolution) on the halo-patch kernel's 2x2-tap variant (igemm6); 0: igemm5 */
    E2EFT_OPT_PERSISTENT_MIN_QROUNDS = 12, /* 2 (default since round 6, was 8): the persistent kernels (igemm5 / igemm6) take a launch of at least n / 4 tiles per CU (2 = half a round of the machine, 8 = two rounds); 1 .. 64 */
    E2EFT_OPT_GN_APPLY_ITERS = 13,   /* 0 (default): the GroupNorm apply pass picks its pixels per thread (4 x n sixteen-byte loads) by tensor size; 1 .. 16: forced n */
    E2EFT_OPT_F32_SPLIT = 14,        /* 1 (default, round 6): e2eft_conv2d_fwd_f32split_supported may answer 1 (fp32 3x3 convolutions from two-term f16 splits on the f16 matrix pipe); 0: it answers 0 (v_mfma_f32_32x32x2_f32 on igemm2) */
    E2EFT_OPT_COUNT = 15
};
int e2eft_set_option(int32_t key, int32_t value);
int e2eft_get_option(int32_t key);

/* ------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / GEMM (MFMA).
 *   out[m, n] = alpha * ( sum_k A[m, k] * W[n, k] + bias[n] + rowadd[img(m), n] ) + residual[m, n]
 *   (fp32 arithmetic; with alpha != 1 the vector epilogues evaluate fma(rowadd, alpha, fma(acc, alpha, bias * alpha)), the scalar
 *   edge-tile path (acc + bias + rowadd) * alpha: the last bit of a result may depend on the tile path the shape selects)
 * conv mode (kh*kw > 1 or stride/upsample/two sources): m = (b, oy, ox), k = ((ky*kw + kx)*cin + c),
 *   A gathered on the fly from the NHWC input(s) with zero padding; weights pre-packed OHWI = W[n][k].
 * Replaces: nn.Conv2d in diffusers ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D.conv,
 *   Upsample2D (F.interpolate nearest + conv), conv_in/conv_out, VAE quant convs
 *   (composition sites GeoWizard/geowizard/models/unet_2d_blocks.py:1064,1109,1211,2242,2285,2400;
 *   unet_2d_condition.py:294,617,1084,1212; Marigold/marigold/marigold_pipeline.py:493-494,515-516),
 *   and nn.Linear / batched matmul (attention.py:208-217,239-248,755,765; transformer_2d.py:153,215).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct E2eftConvDesc {
    int32_t dtype;
    /* input geometry: physical tensor [B, hin, win, *]; logical (virtually nearest-upsampled) size hl x wl
     * (hl == hin and wl == win when there is no fused upsample; Upsample2D, unet_2d_blocks.py:2285). */
    int32_t batch, hin, win, hl, wl;
    int32_t c1, ldx1;         /* channels taken from x1 and its pixel stride */
    int32_t c2, ldx2;         /* channels taken from x2 (0 = none): fused torch.cat([x1, x2], dim=1)
                                 (unet_2d_blocks.py:2328,2456) */
    int32_t kh, kw, stride;   /* kernel size, stride (same in y and x) */
    int32_t pad_t, pad_l;     /* zero padding top/left; bottom/right implied by hout/wout
                                 (VAE Downsample2D pads (0,1,0,1): pad_t = pad_l = 0) */
    int32_t hout, wout;
    int32_t cout, ldo;        /* output channels and output pixel stride */
    int32_t ldr;              /* residual pixel stride (if residual != NULL) */
    int32_t ldw;              /* weight row stride in elements (>= kh*kw*(c1+c2)) */
    float alpha;
} E2eftConvDesc;

/* Results are deterministic run to run for a fixed option set (e2eft_set_option).  The fp32 summation order over k depends on the kernel the
 * launch is routed to: tap-major (ky, kx, channel) on igemm2 / igemm5, 64-channel-chunk-major on igemm6 (E2EFT_OPT_PATCH_CONV; 16-bit 3x3 / stride-1 /
 * pad-1 launches with >= 128 input channels, width % 32 == 0, height % 8 == 0 and at least two 256-pixel tiles per CU) — outputs of the two routes differ
 * in the last bits of the 16-bit result. */
int e2eft_conv2d_fwd(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w,
                     const void* bias /* [cout] or NULL */,
                     const void* rowadd /* [batch, cout] per-image vector (time embedding) or NULL */,
                     const void* residual /* [B,hout,wout,ldr] or NULL */, void* out, void* stream);

/* Batched GEMM with two-level batch index z = zo*nzi + zi:
 *   out_z[m, n] = alpha * ( sum_k A_z[m, k] * W_z[n, k] + bias ) + residual_z[m, n]
 * bias is indexed by n (bias_along_m = 0) or by m (bias_along_m = 1). K-contiguous operands ("NT" GEMM). */
typedef struct E2eftGemmDesc {
    int32_t dtype;
    int32_t m, n, k;
    int32_t lda, ldw, ldo, ldr;
    int32_t nzo, nzi;                      /* batch counts (>= 1) */
    int64_t sa_o, sa_i, sw_o, sw_i, so_o, so_i, sr_o, sr_i; /* element strides per batch level */
    int32_t bias_along_m;
    float alpha;
} E2eftGemmDesc;

int e2eft_gemm(const E2eftGemmDesc* d, const void* a, const void* w, const void* bias, const void* residual,
               void* out, void* stream);

/* Same as e2eft_conv2d_fwd / e2eft_gemm, and the epilogue additionally emits the GroupNorm partial statistics of the tensor
 * it writes (of the ROUNDED values), so that the GroupNorm consuming `out` can skip its statistics pass over HBM:
 *   gn_partial [images][nslabs][cout][3] float triples (count, mean, M2) per row slab and channel,
 *   gn_partial_bytes >= images * ceil(rows_per_image / 128) * cout * 12,
 *   *slab_rows receives the rows per slab the kernel used (nslabs = rows_per_image / *slab_rows), or 0 when this launch could
 *   not emit statistics (rows_per_image not a multiple of the tile height, cout % 8 != 0, unaligned output): the caller then
 *   simply runs e2eft_groupnorm_fwd.  rows_per_image for the GEMM form = tokens per image (m % rows_per_image == 0). */
int e2eft_conv2d_fwd_gnstats(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w, const void* bias,
                             const void* rowadd, const void* residual, void* out, float* gn_partial,
                             size_t gn_partial_bytes, int32_t* slab_rows, void* stream);
/* out = conv(act(GroupNorm(x1))) [+ ...]: e2eft_conv2d_fwd_gnstats whose input is read THROUGH a GroupNorm(+SiLU) that was never applied — the
 * GroupNorm -> SiLU -> conv3x3 pairs of ResnetBlock2D (resnet.py: norm1 / conv1, norm2 / conv2) without the normalised tensor's round trip through
 * HBM.  coeff: the (a, mean) pairs of e2eft_groupnorm_fwd_stats for x1 ([batch][c1][2] fp32), beta: the norm's bias [c1] or NULL, silu != 0: SiLU.
 * The values that enter the convolution are those e2eft_groupnorm_fwd would have written (same arithmetic, same rounding to the 16-bit type), so the
 * result equals e2eft_groupnorm_fwd followed by e2eft_conv2d_fwd_gnstats bit for bit.  Inference only, served by two kernels: ask
 * e2eft_conv2d_fwd_normed_supported(d) (pure host arithmetic; 1 = yes) before skipping the apply pass — 16-bit, 3x3 / stride 1 / pad 1, one source, and
 * either 128 <= c1 <= 640, cout <= 128, width % 32 == 0, height % 8 == 0, at least two 256-pixel tiles per CU (igemm6.hip) or cout <= 4, c1 <= 128,
 * c1 % 32 == 0, at least 16384 output pixels (conv_norm_out -> conv_out: narrow.hip); otherwise E2EFT_ERR_UNSUPPORTED. */
int e2eft_conv2d_fwd_normed_supported(const E2eftConvDesc* d);
/* Nearest-2x upsample followed by a 3x3 / stride-1 / pad-1 convolution (diffusers Upsample2D: the UNet's and the VAE decoder's upsamplers) evaluated as FOUR 2x2
 * convolutions of the LOW-resolution input, one per output parity (py, px): output pixel (2Y + py, 2X + px) only ever reads two distinct source rows / columns, so the 3x3
 * taps that fall on the same source pixel are added up front — 4/9 of the multiply-adds of e2eft_conv2d_fwd's fused-upsample form.  `d` is THAT call's descriptor
 * (hin x win the source, hl = hout = 2 hin, wl = wout = 2 win, kh = kw = 3, stride 1, pads 1, c2 = 0, alpha 1).  w_phase: [4][cout][2 * 2 * c1] in the compute dtype,
 * phase = 2 py + px, row layout (i, j, c) with source offset (i - (1 - py), j - (1 - px)):
 *   w_phase[ph][co][i][j][c] = sum over ky in R(py, i), kx in R(px, j) of w[co][ky][kx][c],   R(0, 0) = {0}, R(0, 1) = {1, 2}, R(1, 0) = {0, 1}, R(1, 1) = {2}
 * (summed in fp32, rounded once: results differ from the fused-upsample form by that rounding and the summation order — inside the 16-bit bar of the path).
 * gn_partial / slab_rows as e2eft_conv2d_fwd_gnstats: statistics of the full-resolution output, emitted when the persistent kernel serves the phases (16-bit,
 * c1 % 64 == 0, win % 16 == 0, batch * hin * win and hin * win multiples of 256, at least two tiles per CU: *slab_rows = 256); other shapes and fp32 run the
 * phases on igemm2 without statistics (*slab_rows = 0).  Eligible: fp32 / fp16 / bf16 with c1 a multiple of one 128-byte k-tile (32 floats / 64 halves), cout and
 * ldo multiples of 16 bytes; ask e2eft_upconv2x_fwd_supported (pure host arithmetic) first; E2EFT_ERR_UNSUPPORTED otherwise. */
int e2eft_upconv2x_fwd_supported(const E2eftConvDesc* d);
int e2eft_upconv2x_fwd(const E2eftConvDesc* d, const void* x, const void* w_phase, const void* bias, void* out, float* gn_partial,
                       size_t gn_partial_bytes, int32_t* slab_rows, void* stream);

/* fp32 convolutions on the f16 matrix pipe (round 6; csrc/f32split.hip).  The reference runs its training recipe in fp32 (training/scripts/train_marigold_e2e_ft_depth.sh:15,
 * `--mixed_precision "no"`); the fp32 matrix instruction of gfx950 is a sixteenth of the f16 one.  An fp32 tensor is split EXACTLY (to 2^-22 relative) into two f16 terms after a
 * power-of-two scaling taken from its own maximum, x s = x0 + x1; a product is x0 w0 + x0 w1 + x1 w0 (three exact f16 products, fp32 accumulation) — error 3 * 2^-22 per
 * product, below the fp32 accumulation error of the reduction itself (tests/test_f32split_gpu.py measures both routes against float64).
 *   e2eft_f32_split2: x fp32 [pixels][ldx] (c channels, c % 8 == 0, 16-byte aligned) -> planes f16 [pixels][ldp], channels [0, c) = x0, [c, 2c) = x1 (ldp >= 2c, % 8);
 *     scale: three device floats of workspace, on return (stream-ordered) scale[1] = s, scale[2] = 1 / s.  No host synchronisation.
 *   e2eft_conv2d_fwd_f32split: out = alpha / (s s_w) * conv(planes, w_split) + bias + residual in fp32.  `d`: dtype E2EFT_F32, any filter / stride / pads of
 *     e2eft_conv2d_fwd with one source and no fused upsample (c2 = 0, hl = hin, wl = win), c1 = c (% 64 == 0), ldx1 = ldp (f16 elements), ldw = row length of w_split in
 *     f16 elements (>= kh kw 3 c), ldo / ldr in fp32 elements.  3x3 / stride 1 / pads 1 on a width % 32 == 0, height % 8 == 0 grid runs on the halo-patch kernel
 *     (igemm6), everything else in whole 256-row tiles on igemm5.  w_split: f16 [cout][kh][kw][w0 (c) | w1 (c) | w0 (c)] with w s_w = w0 + w1 built once per weight
 *     version by the caller (power-of-two s_w); w_inv_scale: device scalar 1 / s_w, or null when alpha already carries it.  scale: the workspace e2eft_f32_split2 / e2eft_groupnorm_fwd_split filled for these planes (null: the planes' scale is already folded into alpha).  gn_partial / slab_rows as e2eft_conv2d_fwd_gnstats (statistics of
 *     the fp32 output).  Ask e2eft_conv2d_fwd_f32split_supported first (pure host arithmetic; E2EFT_OPT_F32_SPLIT = 0 makes it answer 0); E2EFT_ERR_UNSUPPORTED otherwise.
 *     The data gradient of such a convolution is the same call on dY with the flipped, transposed weights (e2eft_conv2d_dgrad's w_dgrad) split the same way. */
int e2eft_f32_split2(const float* x, int64_t pixels, int32_t c, int32_t ldx, void* planes, int32_t ldp, float* scale, void* stream);
/* a packed fp32 weight [rows = cout * taps][c] (16-byte aligned, c % 8 == 0) as the multiplied operand of the calls below: f16 [rows][w0 (c) | w1 (c) | w0 (c)] with
 * w s_w = w0 + w1; scale: three device floats, on return scale[1] = s_w, scale[2] = 1 / s_w (pass scale + 2 as w_inv_scale).  Two launches, no host read. */
int e2eft_f32_split_weight(const float* w, int64_t rows, int32_t c, void* w_split, float* scale, void* stream);
/* the channel concatenation [x1 (c1) | x2 (c2)] as one pair of planes under one scale: planes [pixels][x1_0 | x2_0 | x1_1 | x2_1] (the UNet's two-source convolutions) */
int e2eft_f32_split2_cat(const float* x1, int32_t c1, int32_t ldx1, const float* x2, int32_t c2, int32_t ldx2, int64_t pixels, void* planes, int32_t ldp,
                         float* scale, void* stream);
int e2eft_conv2d_fwd_f32split_supported(const E2eftConvDesc* d);
int e2eft_conv2d_fwd_f32split(const E2eftConvDesc* d, const void* planes, const float* scale, const void* w_split, const float* w_inv_scale, const float* bias,
                              const float* residual, float* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream);
/* nn.Linear in fp32 the same way (igemm5's GEMM mode): out[m][n] = alpha / (s s_w) * sum_k a[m][k] w[n][k] + bias[n] + residual[m][n].  `d`: dtype E2EFT_F32, k = columns of the
 * fp32 operand (% 64 == 0), lda = row stride of its planes [a0 (k) | a1 (k)] in f16 elements (e2eft_f32_split2 with pixels = m, c = k), ldw = row length of
 * w_split [n][w0 (k) | w1 (k) | w0 (k)], ldo / ldr in fp32 elements, one problem (nzo = nzi = 1), m > 256 (a ragged last 256-row tile is masked; the same holds for e2eft_conv2d_fwd_f32split on igemm5).  Ask e2eft_gemm_f32split_supported first. */
/* e2eft_upconv2x_fwd in fp32 the same way: `d` as that call's (dtype E2EFT_F32) with ldx1 = the planes' pixel stride in f16 elements; w_phase_split: f16
 * [4][cout][2][2][w0 (c1) | w1 (c1) | w0 (c1)] — the fp32 phase weights split as ONE tensor (one s_w).  c1 % 64 == 0, win % 16 == 0, hin * win % 256 == 0, alpha 1. */
int e2eft_upconv2x_fwd_f32split_supported(const E2eftConvDesc* d);
int e2eft_upconv2x_fwd_f32split(const E2eftConvDesc* d, const void* planes, const float* scale, const void* w_phase_split, const float* w_inv_scale,
                                const float* bias, float* out, float* gn_partial, size_t gn_partial_bytes, int32_t* slab_rows, void* stream);
int e2eft_gemm_f32split_supported(const E2eftGemmDesc* d);
int e2eft_gemm_f32split(const E2eftGemmDesc* d, const void* planes, const float* scale, const void* w_split, const float* w_inv_scale, const float* bias,
                        const float* residual, float* out, void* stream);
int e2eft_conv2d_fwd_normed(const E2eftConvDesc* d, const void* x1, const float* coeff, const void* beta, int32_t silu, const void* w,
                            const void* bias, const void* rowadd, const void* residual, void* out, float* gn_partial,
                            size_t gn_partial_bytes, int32_t* slab_rows, void* stream);
/* Split-K variant for convolutions with few output tiles and a long reduction (the 9x9 - 24x24 UNet layers: M = B*H*W ~ 10^3,
 * K = 9*Cin up to 23040): the reduction is split by rows of filter taps across workgroups, partial sums go to `workspace`, a finish
 * pass applies bias / rowadd / alpha / residual.  e2eft_conv2d_splitk_workspace_bytes returns 0 when the library would not split. */
size_t e2eft_conv2d_splitk_workspace_bytes(const E2eftConvDesc* d);
int e2eft_conv2d_fwd_splitk(const E2eftConvDesc* d, const void* x1, const void* x2, const void* w, const void* bias,
                            const void* rowadd, const void* residual, void* out, void* workspace, size_t ws_bytes, void* stream);
int e2eft_gemm_gnstats(const E2eftGemmDesc* d, const void* a, const void* w, const void* bias, const void* residual,
                       void* out, int32_t rows_per_image, float* gn_partial, size_t gn_partial_bytes,
                       int32_t* slab_rows, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NHWC, fp32 statistics (shifted/Welford-merged).
 * Replaces nn.GroupNorm(+nn.SiLU) in ResnetBlock2D.norm1/norm2, conv_norm_out, Transformer2DModel.norm,
 * VAE attention group_norm (unet_2d_condition.py:606; transformer_2d.py:151; unet_2d_blocks.py:589-601).
 * The input may be the fused concat of two sources (c2 > 0), matching e2eft_conv2d_fwd.
 * workspace: e2eft_groupnorm_workspace_bytes().  Output y has c1+c2 channels, pixel stride ldy.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct E2eftGroupNormDesc {
    int32_t dtype;
    int32_t batch, hw;        /* pixels per image */
    int32_t c1, ldx1, c2, ldx2;
    int32_t groups;
    int32_t ldy;
    int32_t silu;             /* 1 = fuse SiLU */
    float eps;
} E2eftGroupNormDesc;

size_t e2eft_groupnorm_workspace_bytes(const E2eftGroupNormDesc* d);
int e2eft_groupnorm_fwd(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma,
                        const void* beta, void* y, void* workspace, size_t ws_bytes, void* stream);

/* fp32 GroupNorm(+SiLU) whose result leaves as split planes for e2eft_conv2d_fwd_f32split (one source, d->dtype E2EFT_F32): y * s = y0 + y1.  s is the power of two
 * with s * (max|gamma| * sqrt(hw * channels per group) + max|beta|) in [2^14, 2^15) — a bound of |y| that needs no pass over the data (a normalised value among n is
 * at most sqrt(n - 1), |SiLU(t)| <= |t|); it is computed ON THE DEVICE from the parameters (they may be training; no host read), conversions saturate.  `scale`: three
 * device floats of workspace, on return scale[1] = s, scale[2] = 1 / s — the `scale` argument of the consuming e2eft_conv2d_fwd_f32split.
 * partial1 / nslabs1 / workspace as e2eft_groupnorm_fwd_pre; afterwards the workspace serves e2eft_groupnorm_bwd like the plain forward's. */
int e2eft_groupnorm_fwd_split(const E2eftGroupNormDesc* d, const float* x, const float* gamma, const float* beta, void* planes, int32_t ldp, float* scale,
                              const float* partial1, int32_t nslabs1, void* workspace, size_t ws_bytes, void* stream);

/* GroupNorm whose statistics pass was (partly) done by the producer (e2eft_*_gnstats): partialK / nslabsK describe source K
 * ([batch][nslabsK][cK][3]); a NULL partial is computed here.  Same workspace size as e2eft_groupnorm_fwd. */
int e2eft_groupnorm_fwd_pre(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma,
                            const void* beta, void* y, const float* partial1, int32_t nslabs1, const float* partial2,
                            int32_t nslabs2, void* workspace, size_t ws_bytes, void* stream);
/* The statistics half of e2eft_groupnorm_fwd_pre alone: leaves, inside the workspace at byte offset e2eft_groupnorm_coeff_offset(d), the fp32 pairs
 * (a = gamma * rstd, mean) [batch][c1 + c2][2] that the apply pass would use (y = (x - mean) * a + beta), and applies nothing: the consumer does
 * (e2eft_conv2d_fwd_normed).  Same workspace size, same arguments. */
size_t e2eft_groupnorm_coeff_offset(const E2eftGroupNormDesc* d);
int e2eft_groupnorm_fwd_stats(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const float* partial1,
                              int32_t nslabs1, const float* partial2, int32_t nslabs2, void* workspace, size_t ws_bytes, void* stream);

/* LayerNorm over the last dim of [rows, c] (row stride ldx / ldy), eps, affine.
 * Replaces nn.LayerNorm norm1/2/3 of BasicTransformerBlock (attention.py:205,237,264). */
int e2eft_layernorm_fwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldx, int32_t ldy, float eps,
                        const void* x, const void* gamma, const void* beta, void* y, void* stream);

/* GEGLU gate: y[r, j] = h[r, j] * gelu_erf(h[r, c + j]) for j < c, h = [rows, 2c] (row stride ldh).
 * Replaces diffusers GEGLU activation after ff.net.0.proj (attention.py:755). */
int e2eft_geglu_fwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldh, int32_t ldy, const void* h, void* y,
                    void* stream);

/* Row softmax in place: s[r, :n] = softmax(scale * s[r, :n]) for r < rows (row stride lds).
 * Used by the unfused attention path (VAE mid-block d=512 attention and the strict-fp32 path). */
int e2eft_softmax_rows(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, void* s, void* stream);
/* The same with a causal mask: row r is query r % nq and attends to keys 0 .. r % nq, the rest of the row is written as zeros
 * (the CLIP text tower behind encode_empty_text, /root/reference/Marigold/marigold/marigold_pipeline.py:356-369 and
 * training/train.py:455-458: transformers CLIPTextModel builds a causal attention mask). */
int e2eft_softmax_rows_causal(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, int32_t nq, void* s, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused (flash-style) attention forward, head dim 64, MFMA + online softmax: fp16 / bf16 (attn.hip) and strict fp32 on v_mfma_f32_32x32x2_f32
 * (attn32.hip, round 5: the reference's training recipe is fp32 with xformers' fused attention, training/train.py:308-318; q / k / v / out row strides
 * multiples of 4 floats, 16-byte aligned pointers).
 *   out[b, i, h*64:(h+1)*64] = softmax_j( scale * q[b,i,h,:] . k[kb(b),j,h,:] ) v[kb(b),j,h,:]
 * q/out: [batch, nq, heads*64] (row strides ldq/ldo), k/v: [kv_batch, nk_seg, heads*64] (ldk/ldv).
 * Joint (GeoWizard) attention: kv_nseg = 2 concatenates, along the key axis, the rows of kv batches
 * (b % kv_bmod) and (b % kv_bmod) + kv_bmod  — XFormersJointAttnProcessor, attention.py:482-491.
 * Plain attention: kv_nseg = 1, kv_bmod = batch.
 * Replaces F.scaled_dot_product_attention / xformers.memory_efficient_attention
 * (attention.py:338-343,375-380,497).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct E2eftAttnDesc {
    int32_t dtype;            /* E2EFT_F16, E2EFT_BF16 or E2EFT_F32 (e2eft_attn_fwd / _lse / e2eft_attn_bwd; e2eft_attn512_fwd: 16-bit only) */
    int32_t batch, heads, nq, nk_seg;
    int32_t kv_nseg, kv_bmod;
    int32_t ldq, ldk, ldv, ldo;
    float scale;
} E2eftAttnDesc;

int e2eft_attn_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* stream);
/* same, also storing lse[batch][heads][nq] (fp32): the base-2 log-sum-exp of the scaled scores of every query row, which
 * e2eft_attn_bwd needs to recompute the probabilities (lse may be NULL) */
int e2eft_attn_fwd_lse(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream);
/* Fused attention forward for ONE head of width 512 (fp16 / bf16): the mid-block attention of AutoencoderKL
 * (GeoWizard/geowizard/models/unet_2d_blocks.py:589-601 — `Attention(heads = 1, dim_head = 512)` through AttnProcessor2_0:
 * F.scaled_dot_product_attention on [B, 1, H*W, 512]).  Descriptor as above with heads = 1, kv_nseg = 1, kv_bmod = batch;
 * q / k / v / out rows of 512 elements (row strides ld*, so slices of one fused q|k|v projection are fine).
 * workspace (optional, e2eft_attn512_workspace_bytes; 0 = not needed for this shape on this device): lets the launch cut the
 * query blocks of its last, partially filled round of workgroups along the keys (fp32 partials + a merge kernel). */
size_t e2eft_attn512_workspace_bytes(const E2eftAttnDesc* d);
int e2eft_attn512_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* workspace, size_t ws_bytes,
                      void* stream);
/* Weight gradient of the convolution `d` describes (and of nn.Linear = its 1x1 case), 16-bit, straight from the NHWC tensors:
 *   dW[co][(ky, kx, ci)] = alpha * sum over output pixels p of dy[p][co] * x[input pixel of (p, ky, kx)][ci]     (OHWI rows = the packed weight layout)
 * — torch autograd of nn.Conv2d / nn.Linear w.r.t. the weight (training/train.py:563).  dy: [batch*hout*wout][lddy] (cout columns used), x1 / x2: the
 * forward's input(s).  The pixel sum is split over the grid; `partial` receives fp32 [nsplit][cout][kh*kw*(c1+c2)] and *nsplit_out the number of
 * splits to add up (e2eft_colsum).  e2eft_conv2d_wgrad_workspace_bytes: the size of `partial`, or 0 when this kernel does not serve the problem
 * (channel counts not multiples of 64, fused upsample, >= 4 GB tensors; fp32 is served since round 6, on v_mfma_f32_32x32x2_f32): the caller then uses e2eft_transpose + e2eft_conv2d_im2col_t +
 * e2eft_gemm; e2eft_conv2d_wgrad itself returns E2EFT_ERR_UNSUPPORTED for those. */
size_t e2eft_conv2d_wgrad_workspace_bytes(const E2eftConvDesc* d, int32_t lddy);
int e2eft_conv2d_wgrad(const E2eftConvDesc* d, const void* dy, int32_t lddy, const void* x1, const void* x2, float* partial, size_t partial_bytes,
                       int32_t* nsplit_out, void* stream);
/* Fused attention backward (head dim 64, fp16 / bf16 / fp32, kv_nseg == 1; autograd of F.scaled_dot_product_attention /
 * xformers.memory_efficient_attention in diffusers Attention processors, attention.py:338-343,375-380): dq [B,Nq,heads*64],
 * dk / dv [B,Nk,heads*64] (row strides lddq / lddk / lddv), from q, k, v, the forward output `out` (desc ldo), its gradient
 * dout and the forward's lse.  No Nq x Nk matrix is materialised.  workspace: batch*heads*nq floats. */
size_t e2eft_attn_bwd_workspace_bytes(const E2eftAttnDesc* d);
int e2eft_attn_bwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, const void* out, const void* dout,
                   int32_t lddo, const float* lse, void* dq, int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv,
                   void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Elementwise / layout glue on the path.
 * ---------------------------------------------------------------------------------------------------- */

/* NCHW (any of the three dtypes, contiguous) -> NHWC dtype `dt_out`, channels zero-padded from c to cpad,
 * y = x * mul + add.  Used for rgb_in -> encoder input and latent assembly. */
int e2eft_nchw_to_nhwc(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t c, int32_t hw, int32_t cpad,
                       int32_t ldy, float mul, float add, const void* x, void* y, void* stream);
/* NHWC (pixel stride ldx, first c channels) -> NCHW contiguous, y = x * mul + add. */
int e2eft_nhwc_to_nchw(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t c, int32_t hw, int32_t ldx,
                       float mul, float add, const void* x, void* y, void* stream);
/* Strided channel-slice copy/scale: y[p, 0:c] = x[p, 0:c] * mul + add  (concat assembly, latent scaling
 * `mean * 0.18215` marigold_pipeline.py:495-497, v->x0 `-sqrt(1-abar) * v` :457-465 / train.py:509-512). */
int e2eft_copy_scale(int32_t dtype, int64_t pixels, int32_t c, int32_t ldx, int32_t ldy, float mul, float add,
                     const void* x, void* y, void* stream);
/* y = a + b (same shape [pixels, c], strides lda/ldb/ldy). */
int e2eft_add(int32_t dtype, int64_t pixels, int32_t c, int32_t lda, int32_t ldb, int32_t ldy, const void* a,
              const void* b, void* y, void* stream);
/* Sinusoidal timestep embedding Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0):
 * out[b, :] = [cos(t_b f_i) | sin(t_b f_i)], f_i = exp(-ln(1e4) i / (dim/2)) (unet_2d_condition.py:310,974-979).
 * t is int64 [batch] on the device. */
int e2eft_timestep_embedding(int32_t dtype, int32_t batch, int32_t dim, const int64_t* t, void* out, void* stream);
/* SiLU elementwise on [n] contiguous (time embedding act before time_emb_proj). */
int e2eft_silu(int32_t dtype, int64_t n, const void* x, void* y, void* stream);
/* y = act(x) on [n] contiguous, 16-byte aligned.  kind 0: quick_gelu x * sigmoid(1.702 x) — the MLP activation of the CLIP ViT-L/14
 * image encoder on GeoWizard's per-image path (/root/reference/GeoWizard/geowizard/models/geowizard_pipeline.py:232-248, module
 * transformers CLIPVisionModelWithProjection); 1: gelu (erf); 2: silu; 3: sigmoid (softmax over TWO keys = sigmoid of the score difference: the cross-attention
 * to the two-token empty prompt, modules.Attention._fold). */
int e2eft_activation(int32_t dtype, int32_t kind, int64_t n, const void* x, void* y, void* stream);
/* Depth head: decoder output NHWC [pixels, ldx>=3] -> depth[pixels].  to_unit 0: clip(mean_c(x), -1, 1) (train.py:533-534);
 * 1: the same mapped to [0, 1] by * 0.5 + 0.5 (marigold_pipeline.py:518,476-477); 2: the bare channel mean that
 * `decode_depth` returns (marigold_pipeline.py:517-519).  Output fp32 or dtype (dt_out). */
int e2eft_depth_head(int32_t dt_in, int32_t dt_out, int64_t pixels, int32_t ldx, int32_t to_unit, const void* x,
                     void* y, void* stream);
/* Normal head: NHWC [B*hw, ldx>=3] -> NCHW [B,3,hw]: n / (||n||_2 + 1e-5), optional clamp to [-1,1] and sign
 * (marigold_pipeline.py:471; train.py:537-539; geowizard_pipeline.py:341-342). */
int e2eft_normal_head(int32_t dt_in, int32_t dt_out, int32_t batch, int32_t hw, int32_t ldx, int32_t clamp,
                      float sign, const void* x, void* y, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Task losses (fp32), forward.  Replace training/util/loss.py:13-47 (ScaleAndShiftInvariantLoss) and
 * :51-67 (AngularLoss).  pred/target fp32 NCHW contiguous, mask uint8 [B,1,H,W].
 * ssi workspace: 8 floats per image + 2; out_loss: 1 float on the device.
 * ---------------------------------------------------------------------------------------------------- */
size_t e2eft_ssi_loss_workspace_bytes(int32_t batch);
int e2eft_ssi_loss_fwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                       float* out_loss, float* out_scale_shift /* [batch,2] or NULL */, void* workspace,
                       size_t ws_bytes, void* stream);
size_t e2eft_angular_loss_workspace_bytes(int32_t batch);
int e2eft_angular_loss_fwd(int32_t batch, int32_t hw, const float* pred /* [B,3,hw] */,
                           const float* target /* [B,3,hw] */, const uint8_t* mask /* [B,hw] */, float* out_loss,
                           void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Backward pass of the E2E-FT training step (training/train.py:470-568: loss.backward() through the loss hooks, the
 * frozen VAE decoder, the x0 conversion and the UNet; accelerator.clip_grad_norm_ + AdamW at :561-566).  In the reference
 * these gradients come from torch autograd over diffusers modules; here every autograd.Function.backward of the host layer
 * (diffusion-e2e-ft_amd/autograd.py) calls one of these.
 *
 * Convolution (replaces the cudnn/miopen dgrad + wgrad autograd dispatches for nn.Conv2d in diffusers ResnetBlock2D /
 * Downsample2D / Upsample2D / conv_in / conv_out):
 *   dgrad — the forward implicit-GEMM kernel run on dy with flipped, channel-transposed weights
 *           w_dgrad[ci][(kh-1-ky, kw-1-kx, co)] (co padded to cout_pad with zeros, dy pad channels must be finite);
 *           strided convolutions read dy through a zero-insertion grid; dx is w.r.t. the LOGICAL (upsampled) input
 *           [B, hl, wl, c1+c2] and covers both concat sources.
 *   wgrad — e2eft_transpose(dy) [cout, P] x e2eft_conv2d_im2col_t(x) [kh*kw*cin, P] through e2eft_gemm; the result
 *           [cout, kh*kw*cin] is the OHWI weight layout.  P = batch*hout*wout; im2col_t writes zeros in [P, ldcol).
 * ---------------------------------------------------------------------------------------------------- */
int e2eft_conv2d_dgrad(const E2eftConvDesc* fwd, const void* dy, int32_t lddy, int32_t cout_pad, const void* w_dgrad,
                       int32_t ldwd, void* dx, int32_t lddx, void* stream);
int e2eft_conv2d_im2col_t(const E2eftConvDesc* fwd, const void* x1, const void* x2, void* col /* [kh*kw*cin][ldcol] */,
                          int64_t ldcol, void* stream);
/* out[z][c][r] = in[z][r][c] for r < rows, 0 for rows <= r < rows_pad (batched 2-D transpose, 16-byte vectors both ways) */
int e2eft_transpose(int32_t dtype, int32_t batch, int64_t rows, int32_t cols, int64_t ld_in, int64_t batch_stride_in,
                    int64_t rows_pad, int64_t ld_out, int64_t batch_stride_out, const void* in, void* out, void* stream);
/* out[g][c] = alpha * sum over the g-th group of rows_per_group consecutive rows of x[:, c] (fp32 out): bias gradients
 * (groups = 1) and the per-image time-embedding gradient of ResnetBlock2D (groups = batch).  Deterministic (no atomics). */
size_t e2eft_colsum_workspace_bytes(int32_t groups, int64_t rows_per_group, int32_t cols);
int e2eft_colsum(int32_t dtype, int32_t groups, int64_t rows_per_group, int32_t cols, int64_t ld, float alpha,
                 const void* x, float* out, void* workspace, size_t ws_bytes, void* stream);
/* nearest-upsample backward: dx[B,hin,win,c] = sum of dy[B,hl,wl,c] over the forward gather's preimage */
int e2eft_upsample_nearest_bwd(int32_t dtype, int32_t batch, int32_t hin, int32_t win, int32_t hl, int32_t wl, int32_t c,
                               int32_t lddy, int32_t lddx, const void* dy, void* dx, void* stream);
/* GroupNorm(+SiLU) backward (F.group_norm / F.silu autograd in diffusers ResnetBlock2D, Transformer2DModel.norm, VAE):
 * fwd_workspace is the workspace e2eft_groupnorm_fwd[_pre] filled for the same desc (it holds mean / rstd);
 * dx [B*hw, lddx] covers both concat sources; dgamma / dbeta fp32 [C] or NULL (frozen VAE); dx may be NULL. */
size_t e2eft_groupnorm_bwd_workspace_bytes(const E2eftGroupNormDesc* d);
int e2eft_groupnorm_bwd(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const void* beta,
                        const void* dy, int32_t lddy, void* dx, int32_t lddx, float* dgamma, float* dbeta,
                        const void* fwd_workspace, void* workspace, size_t ws_bytes, void* stream);
/* same, with dx += dx_add (row stride ldadd): the gradient that reaches the same tensor over a skip connection (ResnetBlock2D /
 * Transformer2DModel residuals) is folded into the apply pass instead of a separate elementwise add */
int e2eft_groupnorm_bwd_add(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const void* beta,
                            const void* dy, int32_t lddy, const void* dx_add, int32_t ldadd, void* dx, int32_t lddx, float* dgamma,
                            float* dbeta, const void* fwd_workspace, void* workspace, size_t ws_bytes, void* stream);
/* LayerNorm backward: dx (may be NULL) and dgamma_dbeta fp32 [2][c] */
size_t e2eft_layernorm_bwd_workspace_bytes(int64_t rows, int32_t c);
int e2eft_layernorm_bwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldx, int32_t lddy, int32_t lddx, float eps,
                        const void* x, const void* gamma, const void* dy, void* dx, float* dgamma_dbeta, void* workspace,
                        size_t ws_bytes, void* stream);
/* GEGLU backward: h [rows, 2c] (value | gate), dy [rows, c] -> dh [rows, 2c] */
int e2eft_geglu_bwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldh, int32_t lddy, int32_t lddh, const void* h,
                    const void* dy, void* dh, void* stream);
/* softmax backward in place on dp: ds = p * (dp - rowsum(dp * p)) * scale (attention backward, p = softmax(scale * s)) */
int e2eft_softmax_bwd_rows(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, const void* p, void* dp,
                           void* stream);
int e2eft_silu_bwd(int32_t dtype, int64_t n, const void* x, const void* dy, void* dx, void* stream);
/* head gradients (train.py:531-538): dx NHWC [pixels, lddx] with cpad >= 3 channels written (zeros beyond 3) */
int e2eft_depth_head_bwd(int32_t dt_x, int32_t dt_y, int64_t pixels, int32_t ldx, int32_t lddx, int32_t cpad,
                         int32_t to_unit, const void* x, const void* dy, void* dx, void* stream);
int e2eft_normal_head_bwd(int32_t dt_x, int32_t dt_y, int32_t batch, int32_t hw, int32_t ldx, int32_t lddx, int32_t cpad,
                          int32_t clamp, float sign, const void* x, const void* dy, void* dx, void* stream);
/* loss gradients w.r.t. pred (training/util/loss.py autograd, including the chain through the closed-form scale/shift):
 * fwd_workspace = the workspace the matching *_fwd call filled; grad_out = 1 float on the device; workspace: 2 doubles/image */
int e2eft_ssi_loss_bwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                       const float* scale_shift, const void* fwd_workspace, const float* grad_out, float* dpred,
                       void* workspace, size_t ws_bytes, void* stream);
int e2eft_angular_loss_bwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                           const void* fwd_workspace, const float* grad_out, float* dpred, void* stream);
/* Training-sample preparation on the GPU: the arithmetic of Hypersim.__getitem__ / VirtualKITTI2.__getitem__ after decode and
 * augmentation (/root/reference/training/dataloaders/load.py:236-283, :342-375), batched.
 *   e2eft_masked_quantiles  per image: valid = near < depth < far; out[b] = (q_lo quantile, q_hi quantile, #valid, ok) with
 *                           torch.quantile's linear interpolation (rank = q * (n - 1) in float32) over EXACT order statistics
 *                           (three-level radix select on integer histograms: deterministic); ok = (#valid > 0 && lo != hi).
 *   e2eft_prepare_sample    rgb01 / normal01 [B][3][hw] in [0,1], depth [B][hw] in metres -> rgb in [-1,1]; metric = clamp(depth, lo, hi)
 *                           with invalid pixels = hi; depth3 = three copies of clamp(2 (metric - lo) / (hi - lo) - 1, -1, 1);
 *                           normals = normalize(2 n - 1) with invalid pixels zeroed; val_mask (uint8); everything zero / empty when !ok. */
size_t e2eft_masked_quantiles_workspace_bytes(int32_t batch);
int e2eft_masked_quantiles(int32_t batch, int64_t n, const float* depth, float near_plane, float far_plane, float q_lo, float q_hi, float* out,
                           void* workspace, size_t ws_bytes, void* stream);
int e2eft_prepare_sample(int32_t batch, int64_t hw, const float* rgb01, const float* depth, const float* normal01, float near_plane,
                         float far_plane, const float* quantiles, float* rgb, float* depth3, float* metric, float* normals,
                         uint8_t* val_mask, void* stream);
/* Hypersim's normal-orientation fix on decoded samples (/root/reference/training/dataloaders/load.py:185-204,225-232): normal_u8 [B][h][w][3] as decoded,
 * depth [B][h][w] metres (fp32), inv_k = the nine doubles of inv([[fx,0,cx],[0,fy,cy],[0,0,1]]) row-major (a HOST pointer, read during the call);
 * out [B][h][w][3] uint8 = the re-quantised image the reference hands to its transforms (float64 arithmetic, truncating cast). */
int e2eft_align_normals_u8(int32_t batch, int32_t h, int32_t w, const uint8_t* normal_u8, const float* depth, const double* inv_k, uint8_t* out,
                           void* stream);
/* Synchronised augmentation of decoded training samples on the device (/root/reference/training/dataloaders/load.py:67-152): horizontal flip
 * (per-image flags; `invert_x_on_flip`: first channel -> 255 - x, the normals' x component, :80-82), PIL-exact bilinear resize of uint8 HWC
 * images (torchvision.transforms.Resize on a PIL image = Image.resize: separable triangle filter, 22-bit fixed-point coefficients, uint8
 * rounding after each pass; coefficient tables in Pillow's precompute_coeffs layout: bounds [out][2] = (first tap, tap count), coef
 * [out][ksize]), ToTensor (uint8 / 255 -> planar fp32), nearest resize / KITTI benchmark crop as index tables.
 *   e2eft_aug_resample_bilinear_u8  in [B][h0][w0][3] uint8 -> out [B][3][h][w] fp32 in [0,1]; mid: [B][h0][w][3] uint8 scratch
 *   e2eft_aug_gather_f32            out[b][y][x] = in[b][ymap[y]][xmap[x]] (x mirrored when flip[b]): depth maps, fp32
 *   e2eft_aug_gather_u8             the same for uint8 HWC -> planar fp32 / 255 */
/* Pre- / post-processing of the pipelines' `__call__` on the device (Marigold/marigold/marigold_pipeline.py:221-247,301-321,
 * Marigold/marigold/util/image_util.py:79-108): torch's antialiased bilinear resize (aten `_upsample_bilinear2d_aa`, align_corners = False) of
 * planar [planes][h0][w0] images (uint8 or fp32) to [planes][h][w] fp32 — separable, float weight tables built on the host in aten's arithmetic:
 * bounds [out][2] = (first tap, tap count), weights [out][ksize]; mid: [planes][h0][w] fp32 scratch; round_u8: round half-to-even and clamp to
 * [0, 255] first (torchvision's treatment of an integer image); then y = v * mul + add.
 * e2eft_minmax_unit: out = (x - min x) / (max x - min x), zeros when max == min (marigold_pipeline.py:301-306); two-stage fixed-order reduction;
 * minmax (optional): the two values for the caller; workspace: e2eft_minmax_unit_workspace_bytes(). */
int e2eft_resample_bilinear_aa(int32_t planes, int32_t h0, int32_t w0, int32_t h, int32_t w, int32_t in_is_u8, const void* in,
                               const int32_t* xbounds, const float* xweights, int32_t xksize, const int32_t* ybounds, const float* yweights,
                               int32_t yksize, int32_t round_u8, float mul, float add, float* mid, float* out, void* stream);
size_t e2eft_minmax_unit_workspace_bytes(void);
int e2eft_minmax_unit(int64_t n, const float* x, float* out, float* minmax, void* workspace, size_t ws_bytes, void* stream);
int e2eft_aug_resample_bilinear_u8(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const uint8_t* in, const uint8_t* flip,
                                   int32_t invert_x_on_flip, const int32_t* xbounds, const int32_t* xcoef, int32_t xksize,
                                   const int32_t* ybounds, const int32_t* ycoef, int32_t yksize, uint8_t* mid, float* out, void* stream);
int e2eft_aug_gather_f32(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const float* in, const int32_t* ymap,
                         const int32_t* xmap, const uint8_t* flip, float* out, void* stream);
int e2eft_aug_gather_u8(int32_t batch, int32_t h0, int32_t w0, int32_t h, int32_t w, const uint8_t* in, const int32_t* ymap,
                        const int32_t* xmap, const uint8_t* flip, int32_t invert_x_on_flip, float* out, void* stream);
/* Evaluation arithmetic of the acceptance metric on the device, per image of a batch (fp32 [B][height][width], mask uint8):
 * least-squares scale / shift of pred to gt over the valid pixels (/root/reference/Marigold/src/util/alignment.py:8-56; align_max_res > 0:
 * on the nearest-down-sampled grid of :23-33 — the reference's 3-D Upsample call shrinks the width only, reproduced as is; disparity != 0: in 1 / depth space with the validity rule and the 1e-3 disparity clip of
 * Marigold/eval.py:180-201), clipping to [min_depth, max_depth] and >= 1e-6 (eval.py:203-209), then the ten metrics of
 * Marigold/src/util/metric.py:34-158.  out_metrics [B][12] = abs_relative_difference, squared_relative_difference, rmse_linear, rmse_log,
 * log10, delta1_acc, delta2_acc, delta3_acc, i_rmse, silog_rmse, scale, shift.  aligned_out (optional) receives the aligned, clipped
 * prediction.  fp64 two-stage reductions over fixed partials: bit-reproducible. */
size_t e2eft_depth_eval_workspace_bytes(int32_t batch);
int e2eft_depth_eval(int32_t batch, int32_t height, int32_t width, const float* pred, const float* gt, const uint8_t* mask, int32_t disparity,
                     int32_t align_max_res, float min_depth, float max_depth, float* out_metrics, float* aligned_out, void* workspace,
                     size_t ws_bytes, void* stream);
/* Test-time ensembling of the n_img (<= 32) predictions of ONE image, fp32, replacing
 *   ensemble_depths   /root/reference/Marigold/marigold/util/ensemble.py:40-132 (called from marigold_pipeline.py:293-297;
 *                     twin GeoWizard/geowizard/utils/depth_ensemble.py:21-115)
 *   ensemble_normals  /root/reference/Marigold/marigold/marigold_pipeline.py:58-71 (twin GeoWizard/geowizard/utils/normal_ensemble.py:6-23)
 * x is the [n_img][npix] stack.  Every reduction is two-stage over a fixed number of partials (bit-reproducible); one workspace of
 * e2eft_ensemble_workspace_bytes(n_img) serves all entries.
 *   _minmax        out[i] = (min, max) of image i                                    (the initial guess, ensemble.py:68-71)
 *   _gram          gram[i][j] = sum_p x_i x_j, sums[i] = sum_p x_i (fp64): sufficient statistics of the pairwise term of the
 *                  alignment objective (ensemble.py:85-86), so that one evaluation costs no pass over the stack
 *   _depth_reduce  a_i = x_i * scale[i] + shift[i]; pred = lower median over i (use_mean = 0, torch.median) with the median
 *                  absolute deviation as uncertainty, or mean with the unbiased standard deviation (use_mean = 1);
 *                  minmax = (min, max) of pred.  pred / uncertainty may be NULL (objective evaluation: ensemble.py:88-96)
 *   _depth_finish  pred = (pred - min) / (max - min), uncertainty /= (max - min), range read from device memory (ensemble.py:129-133)
 *   _normals       x, unit: [n_img][3][hw].  unit = x / (|x| + 1e-5); err[i] = sum_p acos(clip(cos(unit_i, mean direction))), the
 *                  mean direction built from the mean azimuth / polar angles; the caller returns unit[argmin err]. */
size_t e2eft_ensemble_workspace_bytes(int32_t n_img);
int e2eft_ensemble_minmax(int32_t n_img, int64_t npix, const float* x, float* out, void* workspace, size_t ws_bytes, void* stream);
int e2eft_ensemble_gram(int32_t n_img, int64_t npix, const float* x, double* gram, double* sums, void* workspace, size_t ws_bytes,
                        void* stream);
int e2eft_ensemble_depth_reduce(int32_t n_img, int64_t npix, const float* x, const float* scale, const float* shift, int32_t use_mean,
                                float* pred, float* uncertainty, float* minmax, void* workspace, size_t ws_bytes, void* stream);
int e2eft_ensemble_depth_finish(int64_t npix, const float* minmax, float* pred, float* uncertainty, void* stream);
int e2eft_ensemble_normals(int32_t n_img, int64_t hw, const float* x, float* unit, double* err, void* workspace, size_t ws_bytes,
                           void* stream);
/* Flat-buffer optimizer step (torch.optim.AdamW + accelerator.clip_grad_norm_, train.py:561-566): all trainable
 * parameters / gradients / moments are single fp32 buffers.  e2eft_sumsq: out[0] = sum g^2 (fp64).  e2eft_adamw_step
 * scales the gradient by grad_scale * min(1, max_norm / (sqrt(grad_sumsq) * grad_scale + 1e-6)) when grad_sumsq != NULL
 * and max_norm > 0 (no host synchronisation), then applies decoupled-decay Adam with bias correction for `step` (>= 1). */
int e2eft_sumsq(int64_t n, const float* g, double* out, void* stream);
int e2eft_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t step, const double* grad_sumsq, float grad_scale,
                     float max_norm, void* stream);
/* The same update with the step counter on the device and a guard against a non-finite gradient norm.  state: int64[2] device words
 * {applied steps, skipped steps} (zero them once; save / restore them with the optimizer state); coef: float[4] device scratch.
 * grad_sumsq must be given (e2eft_sumsq of `grad`).  Finite norm: state[0] += 1 and the AdamW update with the bias correction of
 * step = state[0], gradient scaled by grad_scale * min(1, max_norm / (norm * grad_scale + 1e-6)) (max_norm <= 0: no clipping).
 * Non-finite norm: state[1] += 1, parameters and moments untouched — the bias-correction step does NOT advance.  (The reference,
 * training/train.py:548-566, drops a NaN loss term but still steps AdamW on whatever gradient is there; a NaN gradient would
 * poison weights and moments for good, so this library skips and counts instead.)  No host synchronisation. */
int e2eft_adamw_step_guarded(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t* state, float* coef, const double* grad_sumsq,
                             float grad_scale, float max_norm, void* stream);
/* y = (accumulate ? y : 0) + x * mul with dtype conversion (fp32 master weights -> 16-bit compute copies, 16-bit gradients
 * accumulated into the fp32 flat gradient buffer) */
/* Exponential moving average of the parameters (diffusers `EMAModel.step`; GeoWizard/geowizard/training/train_depth_normal.py:352-353,785-786):
 * shadow[i] -= one_minus_decay * (shadow[i] - param[i]) in torch's operation order; both buffers fp32, 16-byte aligned (the flat buffers of FlatAdamW). */
int e2eft_ema_step(int64_t n, float* shadow, const float* param, float one_minus_decay, void* stream);
int e2eft_cast(int32_t dt_in, int32_t dt_out, int64_t n, float mul, int32_t accumulate, const void* x, void* y, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* E2EFT_H */
