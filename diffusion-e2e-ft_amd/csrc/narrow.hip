// narrow.hip — 3x3 / stride-1 / pad-1 convolution with at most 4 output channels (VAE decoder conv_out 128 -> 3 at full resolution,
// UNet conv_out 320 -> 4): the implicit-GEMM kernel pads N to its 128-column tile (97 % of the MFMA work wasted) and, worse, streams the
// im2col A operand through L2 -> LDS nine times (10.9 GB for 8 x 768^2 x 128 channels: 1.45 ms, ~7.5 TB/s of L2 traffic, 22 TF/s).
// Here every workgroup stages a 18 x 18 pixel halo of its 16 x 16 output tile in LDS ONCE per 64-channel chunk, so each input byte
// leaves HBM/L2 ~1.27 times, and a thread finishes one output pixel with packed 2-way dot products (v_dot2_f32_f16 / _bf16, fp32
// accumulate).  Measured (8 x 768^2, 128 -> 3, fp16): 0.68 ms vs 1.36 ms on the implicit-GEMM kernel; the HBM floor is 0.2 ms — the
// remaining time is the weight stream (scalar loads share lgkmcnt with the LDS reads, so every halo read waits for both).
#include "common.h"

namespace e2eft {

constexpr int NR_T = 16;                 // output tile edge
constexpr int NR_H = NR_T + 2;           // halo edge
constexpr int NR_CH = 64;                // channels per LDS chunk
constexpr int NR_PIX = NR_CH * 2 + 16;   // bytes per halo pixel: 128 B of data + 16 B pad (odd multiple of 16 B -> conflict-free b128 reads)
constexpr int NR_CO = 4;                 // output channels held per thread

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c);
template <> __device__ __forceinline__ float dot2<f16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}
template <> __device__ __forceinline__ float dot2<bf16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), c, false);
}
// 16 bytes of activations times 16 bytes of weights: four 2-way dot products (16-bit) or four fp32 FMAs (round 6: the strict-fp32 training recipe's decoder
// conv_out 128 -> 3 ran on a 128-wide MFMA tile at 3.3 TF/s, 22.5 ms of the 2.95 s step — training/train.py:241-242 through the frozen decoder)
template <typename T> __device__ __forceinline__ float dot16(const u32x4& a, const u32x4& b, float s) {
    if constexpr (sizeof(T) == 4) {
        const floatx4 fa = __builtin_bit_cast(floatx4, a), fb = __builtin_bit_cast(floatx4, b);
        s = fmaf(fa[0], fb[0], s); s = fmaf(fa[1], fb[1], s); s = fmaf(fa[2], fb[2], s); s = fmaf(fa[3], fb[3], s);
    } else {
        s = dot2<T>(a[0], b[0], s); s = dot2<T>(a[1], b[1], s); s = dot2<T>(a[2], b[2], s); s = dot2<T>(a[3], b[3], s);
    }
    return s;
}

struct NarrowParams {
    const void* x;      // [B][H][W][ldx]
    const void* w;      // [cout][ldw] rows = (ky, kx, c)
    const void* bias;   // [cout] or null
    void* out;          // [B][H][W][ldo]
    int H, W, cin, ldx, ldw, cout, ldo, tiles_x, tiles_y;
    float alpha;
    const float* nrm_ad;   // optional (MFMA form): the input is read through GroupNorm(+SiLU) — (a, mean) pairs [image][cin][2] of e2eft_groupnorm_fwd_stats,
    const void* nrm_beta;  // the norm's beta [cin] or null; value = (x - mean) * a + beta, then SiLU if nrm_silu: gn_apply_kernel's arithmetic (norm.hip)
    int nrm_silu;
};

// grid (tiles_x * tiles_y, B), block 256 (thread = output pixel (ty, tx) of the tile).  The weights are wave-uniform: they are read with
// SCALAR loads straight from global memory (constant address space -> s_load_dwordx4, served by the scalar cache) and enter the dot
// products as SGPR operands; staging them in LDS cost four broadcast ds_read_b128 per halo read and made the kernel LDS-issue-bound
// (0.76 ms for 8 x 768^2 x 128 -> 3; the implicit-GEMM kernel: 1.37 ms).
typedef const __attribute__((address_space(4))) u32x4* cptr_t;

template <typename T, int CO>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const NarrowParams p, const T* __restrict__ wg) {
    __shared__ __attribute__((aligned(16))) char halo[NR_H * NR_H * NR_PIX];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tile = blockIdx.x;
    const int oy0 = (tile / p.tiles_x) * NR_T, ox0 = (tile % p.tiles_x) * NR_T;
    const T* xb = (const T*)p.x + (long)blockIdx.y * p.H * p.W * p.ldx;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
    constexpr int EPC = 16 / (int)sizeof(T), CH = 128 / (int)sizeof(T);      // channels per 16 bytes / per LDS chunk (128 data bytes per halo pixel: 64 halves or 32 floats)

    for (int c0 = 0; c0 < p.cin; c0 += CH) {
        const int nch = min(CH, p.cin - c0);          // multiple of EPC
        const int ngr = nch / EPC;                    // 16-byte groups per pixel in this chunk
        // ---- stage the halo: 324 pixels x ngr 16-byte groups, zeros outside the image (the convolution's zero padding).  Two passes (round 6, as the MFMA form
        // below): all eleven loads of a thread in flight, then the stores — the one-pass loop waited out one memory latency per unit
        constexpr int NIT = (NR_H * NR_H * 8 + 255) / 256;
        u32x4 raw[NIT];
        const int g = tid & 7;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 3;
            const int hy = pix / NR_H, hx = pix - hy * NR_H;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            raw[it] = u32x4{0u, 0u, 0u, 0u};
            if (pix < NR_H * NR_H && g < ngr && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                raw[it] = *reinterpret_cast<const u32x4*>(xb + ((long)iy * p.W + ix) * p.ldx + c0 + g * EPC);
        }
        __syncthreads();                              // the previous chunk is consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 3;
            if (pix < NR_H * NR_H && g < ngr) *reinterpret_cast<u32x4*>(halo + pix * NR_PIX + g * 16) = raw[it];
        }
        __syncthreads();
        // ---- 9 taps x ngr groups x CO outputs, 4 dot2 per (group, output)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const char* px = halo + ((ty + ky) * NR_H + tx + kx) * NR_PIX;
                const long wofs = (long)(ky * 3 + kx) * p.cin + c0;           // uniform
                for (int g = 0; g < ngr; ++g) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(px + g * 16);
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const u32x4 b = *(cptr_t)(uintptr_t)(wg + (long)co * p.ldw + wofs + g * EPC);
                        acc[co] = dot16<T>(a, b, acc[co]);
                    }
                }
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < p.H && ox < p.W) {
        const T* bias = (const T*)p.bias;
        T* o = (T*)p.out + (((long)blockIdx.y * p.H + oy) * p.W + ox) * p.ldo;
#pragma unroll
        for (int co = 0; co < CO; ++co) o[co] = from_f<T>(p.alpha * (acc[co] + (bias ? to_f(bias[co]) : 0.f)));   // out = alpha * (conv + bias), as the igemm epilogue
    }
}

// MFMA form of the same kernel (cin a multiple of 32): the 16 pixels of one tile row are the M rows of v_mfma_f32_16x16x32, the (<= 4, padded
// to 16) output channels its N columns, 32 input channels of one tap its K; A fragments are the same 16-byte halo reads as above (lane =
// pixel, lane >> 4 = 8-channel group), the weight fragments of a 64-channel chunk (9 taps x 2 slabs, zero rows beyond cout) sit in 72
// registers.  The dot-product form issues 864 v_dot2 per thread and chunk (~7k VALU cycles per wave) and streams the weights through the
// scalar cache; here a wave issues 72 MFMAs (1.2k matrix-pipe cycles) per chunk and the kernel is bound by the halo traffic.
template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    __device__ static __forceinline__ floatx4 run(const u32x4& a, const u32x4& b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16> {
    __device__ static __forceinline__ floatx4 run(const u32x4& a, const u32x4& b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_narrow_mfma_kernel(const NarrowParams p, const T* __restrict__ wg) {
    __shared__ __attribute__((aligned(16))) char halo[NR_H * NR_H * NR_PIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;      // A: pixel (tile column) / 8-channel group; B: output channel / 8-channel group; D: column = output channel, rows 4 kq + r
    const int tile = blockIdx.x;
    const int oy0 = (tile / p.tiles_x) * NR_T, ox0 = (tile % p.tiles_x) * NR_T;
    const T* xb = (const T*)p.x + (long)blockIdx.y * p.H * p.W * p.ldx;
    floatx4 acc[4];                                // tile rows 4 wave + rb
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = floatx4{0.f, 0.f, 0.f, 0.f};
    const bool wrow = m < p.cout;

    for (int c0 = 0; c0 < p.cin; c0 += NR_CH) {
        const int nch = min(NR_CH, p.cin - c0);      // multiple of 32
        const int ngr = nch >> 3;
        // weight fragments of this chunk (requested before the halo is staged: both latencies overlap)
        u32x4 bf[9][2];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const bool ok = wrow && sl * 32 < nch;
                bf[tap][sl] = ok ? *reinterpret_cast<const u32x4*>(wg + (long)m * p.ldw + (long)tap * p.cin + c0 + sl * 32 + kq * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        // staging in two passes (round 6): first ALL of this thread's 16-byte units of the chunk are requested (eleven independent loads in flight — the one-pass loop
        // waited out one HBM latency per unit: 11 x 2 chunks x ~1.5 us was the kernel's 0.8 ms), then they are normalised and stored
        constexpr int NIT = (NR_H * NR_H * 8 + 255) / 256;      // 11 (the unit's channel group g = tid & 7 is the same in every iteration: 256 % 8 == 0)
        u32x4 raw[NIT];
        const int g = tid & 7;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 3;
            const int hy = pix / NR_H, hx = pix - hy * NR_H;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            const bool inb = pix < NR_H * NR_H && g < ngr && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            raw[it] = u32x4{0u, 0u, 0u, 0u};
            if (inb) raw[it] = *reinterpret_cast<const u32x4*>(xb + ((long)iy * p.W + ix) * p.ldx + c0 + g * 8);
        }
        // fused GroupNorm(+SiLU) of the input: a thread stages the same 8-channel group (tid & 7) of every pixel it handles, so its 24 coefficients
        // are loaded once per chunk; the staged value is what e2eft_groupnorm_fwd would have written (padding stays zero)
        float na[8], nm[8], nb[8];
        const bool nrm = p.nrm_ad != nullptr && (tid & 7) < ngr;
        if (nrm) {
            const int c = c0 + (tid & 7) * 8;
            // the sixteen (a, mean) floats of channels c .. c + 7 are consecutive, beta's eight values one 16-byte unit: five vector loads in flight instead of 24 scalar ones
            const floatx4* ad4 = reinterpret_cast<const floatx4*>(p.nrm_ad + ((long)blockIdx.y * p.cin + c) * 2);
            const floatx4 q0 = ad4[0], q1 = ad4[1], q2 = ad4[2], q3 = ad4[3];
            Vec16<T> bq;
            bq.raw = u32x4{0u, 0u, 0u, 0u};
            if (p.nrm_beta) bq = ld16((const T*)p.nrm_beta + c);
            const float av[8] = {q0[0], q0[2], q1[0], q1[2], q2[0], q2[2], q3[0], q3[2]};
            const float mv[8] = {q0[1], q0[3], q1[1], q1[3], q2[1], q2[3], q3[1], q3[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                nm[e] = mv[e];
                gn_fold(av[e], mv[e], p.nrm_beta ? to_f(bq.e[e]) : 0.f, na[e], nb[e]);      // (a2, d2): gn_apply_kernel's exp2-domain arithmetic (common.h)
            }
        }
        __syncthreads();                              // the previous chunk is consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 3;
            if (pix >= NR_H * NR_H || g >= ngr) continue;
            const int hy = pix / NR_H, hx = pix - hy * NR_H;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            Vec16<T> v;
            v.raw = raw[it];
            if (nrm && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = gn_act_u(__builtin_fmaf(to_f(v.e[e]), na[e], nb[e]), p.nrm_silu != 0);
                    asm("" : "+v"(t));      // the fp32 product exists before it is rounded to T, as in gn_apply_kernel (mul, then cvt_pk): without this the
                    v.e[e] = from_f<T>(t);  // compiler emits v_fma_mixlo_f16 (u * r rounded ONCE to fp16) and the two routes differ in the last bit
                }
            }
            *reinterpret_cast<u32x4*>(halo + pix * NR_PIX + g * 16) = v.raw;
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int ty = 4 * wave + rb;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const char* px = halo + ((ty + ky) * NR_H + m + kx) * NR_PIX + kq * 16;
                    acc[rb] = Mma16<T>::run(*reinterpret_cast<const u32x4*>(px), bf[ky * 3 + kx][0], acc[rb]);
                    if (nch > 32) acc[rb] = Mma16<T>::run(*reinterpret_cast<const u32x4*>(px + 64), bf[ky * 3 + kx][1], acc[rb]);
                }
        }
    }
    if (wrow) {
        const T* bias = (const T*)p.bias;
        const float bv = bias ? to_f(bias[m]) : 0.f;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int oy = oy0 + 4 * wave + rb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + 4 * kq + r;
                if (oy < p.H && ox < p.W)
                    ((T*)p.out)[(((long)blockIdx.y * p.H + oy) * p.W + ox) * p.ldo + m] = from_f<T>(p.alpha * (acc[rb][r] + bv));
            }
        }
    }
}

template <typename T> static void launch_narrow_t(const NarrowParams& p, dim3 grid, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (option(E2EFT_OPT_NARROW_MFMA) && p.cin % 32 == 0) {
            hipLaunchKernelGGL((conv3x3_narrow_mfma_kernel<T>), grid, dim3(256), 0, s, p, (const T*)p.w);
            return;
        }
    }
    const T* w = (const T*)p.w;
    switch (p.cout) {
        case 1: hipLaunchKernelGGL((conv3x3_narrow_kernel<T, 1>), grid, dim3(256), 0, s, p, w); break;
        case 2: hipLaunchKernelGGL((conv3x3_narrow_kernel<T, 2>), grid, dim3(256), 0, s, p, w); break;
        case 3: hipLaunchKernelGGL((conv3x3_narrow_kernel<T, 3>), grid, dim3(256), 0, s, p, w); break;
        default: hipLaunchKernelGGL((conv3x3_narrow_kernel<T, 4>), grid, dim3(256), 0, s, p, w); break;
    }
}

// returns -1 when the problem is not this kernel's (the caller then runs the implicit-GEMM path), else the launch status
bool conv3x3_narrow_eligible(const E2eftConvDesc* d, bool normed) {   // the descriptor's part of the test (pointers: 16-byte aligned)
    if (!option(E2EFT_OPT_NARROW_CONV)) return false;
    if (d->dtype != E2EFT_F16 && d->dtype != E2EFT_BF16 && d->dtype != E2EFT_F32) return false;
    const int epc = 16 / (int)dtype_size(d->dtype);
    if (d->dtype == E2EFT_F32 && normed) return false;   // (the fused GroupNorm lives in the 16-bit MFMA form)
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->c2 != 0) return false;
    if (d->cout < 1 || d->cout > NR_CO || d->hl != d->hin || d->wl != d->win || d->hout != d->hin || d->wout != d->win) return false;
    if (d->c1 % epc != 0 || d->ldx1 % epc != 0 || d->ldw % epc != 0) return false;
    if ((long)d->batch * d->hout * d->wout < 16384) return false;   // tiny problems: launch-bound either way, keep one code path
    if (d->c1 > 128) return false;   // measured: 320 -> 4 at 8 x 96^2 is 0.098 ms here vs 0.087 ms on the MFMA tile (five halo chunks per tile)
    if (normed && !(option(E2EFT_OPT_NARROW_MFMA) && d->c1 % 32 == 0 && option(E2EFT_OPT_FUSED_NORM))) return false;   // the fused norm lives in the MFMA form
    return true;
}

int launch_conv3x3_narrow(const E2eftConvDesc* d, const void* x1, const void* w, const void* bias, void* out, void* stream, const float* nrm_ad,
                          const void* nrm_beta, int nrm_silu) {
    if (!conv3x3_narrow_eligible(d, nrm_ad != nullptr) || ((uintptr_t)x1 & 15) || ((uintptr_t)w & 15)) return -1;
    NarrowParams p;
    p.nrm_ad = nrm_ad; p.nrm_beta = nrm_beta; p.nrm_silu = nrm_silu;
    p.x = x1; p.w = w; p.bias = bias; p.out = out;
    p.H = d->hin; p.W = d->win; p.cin = d->c1; p.ldx = d->ldx1; p.ldw = d->ldw; p.cout = d->cout; p.ldo = d->ldo;
    p.tiles_x = cdiv(d->win, NR_T); p.tiles_y = cdiv(d->hin, NR_T);
    p.alpha = d->alpha;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)d->batch);
    if (d->dtype == E2EFT_F16) launch_narrow_t<f16>(p, grid, s);
    else if (d->dtype == E2EFT_BF16) launch_narrow_t<bf16>(p, grid, s);
    else launch_narrow_t<float>(p, grid, s);
    tag_kernel("conv3x3_narrow%s_kernel", (d->dtype != E2EFT_F32 && option(E2EFT_OPT_NARROW_MFMA) && p.cin % 32 == 0) ? "_mfma" : "");
    return check_launch("conv3x3_narrow");
}

}  // namespace e2eft
