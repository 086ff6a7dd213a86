// igemm.h — parameter block shared by the implicit-GEMM kernel variants.
#pragma once
#include "common.h"
#include <type_traits>

namespace e2eft {

template <int V> using IConst = std::integral_constant<int, V>;

// -DE2EFT_STAMPS (build.py build_stamps(): lib/libe2eft_stamps.so, scripts/stamp_bench.py): thread 0 of every workgroup records the
// shader clock at the phase boundaries of igemm2_kernel — start, k-loop entry, k-loop exit, accumulators staged, end.
#ifdef E2EFT_STAMPS
static __device__ long long g_stamps[65536 * 8];
static __device__ long long g_stamps_rt[65536 * 2];   // the constant 100 MHz counter (s_memrealtime) at stamps 0 and 4: shader clock = cycles / ticks * 100 MHz
#define E2EFT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) { g_stamps[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0 || (i) == 4) g_stamps_rt[blockIdx.x * 2 + ((i) >> 2)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define E2EFT_STAMP(i) do { } while (0)
#endif

struct IgemmParams {
    const void* x1;
    const void* x2;
    const void* w;
    const void* bias;
    const void* rowadd;
    const void* residual;
    void* out;
    int M, N, K;
    int ldx1, ldx2, c1, cin;
    int hin, win, hl, wl, kh, kw, stride, pad_t, pad_l, hout, wout;
    float up_sh, up_sw;
    int zins;            // > 1: the source is read through a zero-insertion grid (dgrad of a strided conv): logical (y, x) is
                         // source (y / zins, x / zins) when both are multiples of zins and zero otherwise
    int ldw, ldr, ldo;
    int bias_along_m;
    int rows_per_img;
    float alpha;
    int nzi;
    long sa_o, sa_i, sw_o, sw_i, so_o, so_i, sr_o, sr_i;
    int mtiles, ntiles;
    float* gn_partial;   // optional: [img][gn_nslabs][N][3] (n, mean, M2) GroupNorm partials of the rounded output
    int gn_nslabs;       // row slabs (one per M-tile) per image
    int ksplit_taps;     // conv split-K: batch index zi covers filter taps [zi * ksplit_taps, (zi + 1) * ksplit_taps) (0 = whole K)
    const float* nrm_ad; // optional (igemm6 only): the A operand is read through GroupNorm(+SiLU): [image][cin][2] fp32 (a, mean) pairs of e2eft_groupnorm_fwd_stats ...
    const void* nrm_beta; // ... and the norm's beta [cin] (or null): value = (x - mean) * a + beta, then SiLU if nrm_silu
    int nrm_silu;
    int out_seg;         // > 0 (igemm5 only, e2eft_upconv2x_fwd): the output rows are laid out in SEGMENTS of out_seg consecutive GEMM rows, segment g starting
                         // 2 * out_seg pixel rows (of ldo elements) after segment g - 1: output row of GEMM row m = m + (m / out_seg) * out_seg.  That is one
                         // parity phase of a 2x-upsampled image: GEMM rows = low-resolution pixels (b, Y, X), out_seg = W, ldo = 2 * (pixel stride of the full
                         // image), so consecutive X land on every other pixel and consecutive Y on every other row.  Multiple of 16.
    int gn_islabs;       // > 0: statistics slabs per image in gn_partial (the image stride), when it differs from gn_nslabs (four phases share one buffer)
    // round 6 (igemm6 only, e2eft_conv2d_fwd_f32split): an fp32 convolution on the f16 matrix pipe.  The A operand is the two-term f16 split of an fp32 tensor,
    // planes [x0 | x1] side by side in one pixel (x * s = x0 + x1 to 22 bits, s a power of two); K runs over THREE blocks of split_c channels per tap — (x0, w0), (x0, w1),
    // (x1, w0) — so chunk channel ch reads plane channel ch - split_c once ch >= split_c; the weights arrive laid out that way.  Bias / residual / output are fp32:
    // out = acc * (alpha * *alpha_dev) + bias + residual   (alpha_dev: device scalar 1 / s of the activation split; alpha carries 1 / s_w and the caller's factor)
    int split_c;         // > 0: channels of one plane (cin = 3 * split_c)
    const float* alpha_dev;
    const float* alpha_dev2;   // a second optional device factor (1 / s_w of weights that were split on the device)
    int debug_flags;     // instrumented build (-DE2EFT_STAMPS) only: bit 0 / 1 / 2 = the persistent kernel issues its A-operand LDS-DMA never / on the
                         // first tap of a filter row only / on the first tap of a 64-channel chunk only (WRONG results; the price of operand
                         // delivery: profiles/r03d_a_operand_delivery_probe.txt, "what an A-reuse scheme could buy at most")
};

// ---------------------------------------------------------------------------------------------------------------
// Shared epilogue: the 32x32 MFMA accumulator layout (lane = column, registers = rows) would give 2-byte scattered
// stores and one dependent residual load per element.  Instead the fp32 tile goes through LDS once (the k-loop is
// finished, its buffers are free) and is written back row-wise: every thread owns 8 consecutive columns of a row,
// residual / output move as 16-byte vectors, a wave covers 4 full 256-byte rows per instruction.
//   out = alpha * (acc + bias + rowadd[img(m)]) + residual
// Optionally (p.gn_partial) the epilogue also emits the GroupNorm statistics of the tile it just wrote — per column, Welford
// over each thread's rows, butterfly + LDS merge over the workgroup — so that the consuming GroupNorm skips its statistics
// pass over HBM (one of its three passes).
template <typename T, int BM_, int BN_, int NT_>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, char* smem, int m0, int n0, int zo, int zi);

template <typename T, int BM_, int BN_, int NT_>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, char* smem, floatx16 (&acc)[2][2], int wm, int wn, int l31,
                                               int h, int m0, int n0, int zo, int zi) {
    constexpr int LDT = BN_ + 4;            // fp32 row stride of the staged tile (528 B for BN = 128)
    float* tile = reinterpret_cast<float*>(smem);
    __syncthreads();                        // every wave is done reading the last k-tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDT + wn * 64 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    E2EFT_STAMP(3);
    igemm_epilogue_rows<T, BM_, BN_, NT_>(p, smem, m0, n0, zo, zi);
}

// the row passes over a tile that is already staged in LDS as fp32 [BM_][BN_ + 4] (all threads past a barrier): bias / rowadd / alpha /
// residual, rounding, 16-byte stores, optional GroupNorm statistics.  
template <typename T, int BM_, int BN_, int NT_>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, char* smem, int m0, int n0, int zo, int zi) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int LDT = BN_ + 4;            // fp32 row stride of the staged tile (528 B for BN = 128)
    constexpr int CPR = BN_ / 8;            // 8-column chunks per row
    constexpr int RPP = NT_ / CPR;          // rows per pass
    constexpr int NPASS = BM_ / RPP;        // rows per thread
    constexpr int NWV = NT_ / 64;
    float* tile = reinterpret_cast<float*>(smem);
    float* gst = tile + BM_ * LDT;          // [NWV][16 chunks][8 cols][2] wave partials of the GroupNorm statistics

    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ rowadd = (const T*)p.rowadd;
    const T* __restrict__ res = p.residual ? (const T*)p.residual + zo * p.sr_o + zi * p.sr_i : nullptr;
    T* __restrict__ out = (T*)p.out + zo * p.so_o + zi * p.so_i;
    const int chunk = threadIdx.x % CPR, rbase = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    const bool col_ok = n < p.N;
    const bool vec = (p.N % 8 == 0) && (p.ldo % EPC == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     (!res || (p.ldr % EPC == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    const bool stats = p.gn_partial != nullptr;   // host guarantees: vec, whole tile inside one image, all rows valid
    const int nv = min(8, p.N - n);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (bias && !p.bias_along_m && col_ok && e < nv) ? to_f(bias[n + e]) : 0.f;
    float s_mean[8], s_m2[8];   // Welford state over this thread's NPASS rows (per column)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_mean[e] = s_m2[e] = 0.f;

    // ---- fast path: vector epilogue on a tile that lies completely inside the problem; every condition is workgroup-uniform and
    // hoisted, so the pass loop is straight-line code: x = fma(acc, alpha, bias*alpha) [+ rowadd*alpha] [+ residual], one
    // v_cvt_pk per column pair, one 16-byte store per row chunk.  (The generic loop below costs ~9k cycles per 256x128 tile in
    // exec-mask branches and scalar fallbacks even when none is taken — scripts/stamp_bench.py.)
    const bool full = vec && m0 + BM_ <= p.M && !(bias && p.bias_along_m);   // a ragged last N-tile only idles the threads of its missing chunks
    if (full) {
        const float al = p.alpha;
        float bva[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bva[e] = bv[e] * al;
        const bool one_img = p.rows_per_img % BM_ == 0 || p.rows_per_img >= p.M;
        const int img0 = rowadd ? m0 / p.rows_per_img : 0;
        T* __restrict__ orow = out + (long)(m0 + rbase) * p.ldo + n;
        const T* __restrict__ rrow = res ? res + (long)(m0 + rbase) * p.ldr + n : nullptr;
        const long ostep = (long)RPP * p.ldo, rstep = (long)RPP * p.ldr;
        // GroupNorm statistics: every thread of a column accumulates sum(x - pv), sum((x - pv)^2) about the same pivot pv (the tile's
        // first row before rowadd / residual: any value near the data does), so partial results merge by plain addition.  They are
        // taken from the fp32 value; rounding to T adds a variance of ~2^-22 x^2 (fp16) / 2^-16 x^2 (bf16) — below the kernels' noise.
        float pv[8];
        if (stats) {
            const floatx4 q0 = *reinterpret_cast<const floatx4*>(tile + chunk * 8);
            const floatx4 q1 = *reinterpret_cast<const floatx4*>(tile + chunk * 8 + 4);
            const float pr[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = fmaf(pr[e], al, bva[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = 0.f;
        }
        // The row passes work on float2 values: explicit packed math (v_pk_fma_f32 / v_pk_add_f32 on NATURAL register pairs — the pairs
        // are the dword pairs of the ds_read_b128 results, so no operand swizzle is needed; the swizzled forms the SLP vectorizer used
        // to generate here are the ones that misread on gfx950, DESIGN.md §3.6, and build.py keeps that vectorizer off).  The
        // statistics cost 4.6k cycles per 256x128 tile as scalar code (3 VALU per element) against 3.6k packed.
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 al2 = {al, al};
        f2 bva2[4], pv2[4], sm2[4], sq2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bva2[q] = f2{bva[2 * q], bva[2 * q + 1]};
            pv2[q] = f2{pv[2 * q], pv[2 * q + 1]};
            sm2[q] = f2{0.f, 0.f};
            sq2[q] = f2{0.f, 0.f};
        }
        // p.out_seg > 0 (e2eft_upconv2x_fwd: one parity phase of a 2x-upsampled image): GEMM row m lands on output row m + (m / out_seg) * out_seg
        auto seg_off = [&](const int row) -> long { return p.out_seg > 0 ? (long)((m0 + row) / p.out_seg) * p.out_seg * p.ldo : 0L; };
        auto run = [&](auto has_res, auto has_ra) {
            constexpr bool HAS_RES = decltype(has_res)::value, HAS_RA = decltype(has_ra)::value;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int row = rbase + pass * RPP;
                const floatx4 t0 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8);
                const floatx4 t1 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8 + 4);
                f2 x2[4] = {f2{t0[0], t0[1]}, f2{t0[2], t0[3]}, f2{t1[0], t1[1]}, f2{t1[2], t1[3]}};
#pragma unroll
                for (int q = 0; q < 4; ++q) x2[q] = __builtin_elementwise_fma(x2[q], al2, bva2[q]);
                if constexpr (HAS_RA) {
                    const int img = one_img ? img0 : (m0 + row) / p.rows_per_img;
                    const T* ra = rowadd + (long)img * p.N + n;
#pragma unroll
                    for (int q = 0; q < 8 / EPC; ++q) {
                        const Vec16<T> t = ld16(ra + q * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; e += 2) {
                            const int j = (q * EPC + e) >> 1;
                            x2[j] = __builtin_elementwise_fma(f2{to_f(t.e[e]), to_f(t.e[e + 1])}, al2, x2[j]);
                        }
                    }
                }
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int q = 0; q < 8 / EPC; ++q) {
                        const Vec16<T> t = ld16(rrow + pass * rstep + q * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; e += 2) {
                            const int j = (q * EPC + e) >> 1;
                            x2[j] += f2{to_f(t.e[e]), to_f(t.e[e + 1])};
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8 / EPC; ++q) {
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(x2[(q * EPC + e) >> 1][(q * EPC + e) & 1]);
                    st16(orow + pass * ostep + seg_off(row) + q * EPC, o);
                }
                if (stats) {   // uniform: shifted sums about a per-column pivot shared by the whole tile (sm2 = sum, sq2 = sum of squares)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f2 d = x2[q] - pv2[q];
                        sm2[q] += d;
                        sq2[q] = __builtin_elementwise_fma(d, d, sq2[q]);
                    }
                }
            }
        };
        if (col_ok) {
            if (res) {
                if (rowadd) run(std::true_type{}, std::true_type{});
                else run(std::true_type{}, std::false_type{});
            } else {
                if (rowadd) run(std::false_type{}, std::true_type{});
                else run(std::false_type{}, std::false_type{});
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_mean[2 * q] = sm2[q][0]; s_mean[2 * q + 1] = sm2[q][1];
            s_m2[2 * q] = sq2[q][0]; s_m2[2 * q + 1] = sq2[q][1];
        }
        if (stats) {   // uniform.  lanes l, l^16, l^32, l^48 own the same 8 columns: add; then the waves through LDS
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_mean[e] += __shfl_xor(s_mean[e], off, 64);
                    s_m2[e] += __shfl_xor(s_m2[e], off, 64);
                }
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            float* piv = gst + NWV * 16 * 8 * 2;
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gst[((wv * 16 + lane) * 8 + e) * 2] = s_mean[e];
                    gst[((wv * 16 + lane) * 8 + e) * 2 + 1] = s_m2[e];
                    if (wv == 0) piv[lane * 8 + e] = pv[e];
                }
            }
            __syncthreads();
            if (threadIdx.x < BN_ && n0 + (int)threadIdx.x < p.N) {
                const int c = threadIdx.x;
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) {
                    su += gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2];
                    sq += gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2 + 1];
                }
                const float nrow = (float)BM_;
                const int img = m0 / p.rows_per_img;
                const int slab = (m0 - img * p.rows_per_img) / BM_;
                float* o = p.gn_partial + (((long)img * p.gn_nslabs + slab) * p.N + n0 + c) * 3;
                o[0] = nrow;
                o[1] = piv[c] + su / nrow;
                o[2] = fmaxf(sq - su * su / nrow, 0.f);
            }
        }
        return;
    }
    {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int row = rbase + pass * RPP;
        const int m = m0 + row;
        if (m >= p.M || !col_ok) continue;
        const floatx4 t0 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8);
        const floatx4 t1 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8 + 4);
        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        const float bm = (bias && p.bias_along_m) ? to_f(bias[m]) : 0.f;
        const T* ra = rowadd ? rowadd + (long)(m / p.rows_per_img) * p.N + n : nullptr;
        if (vec) {
            float rv[8];
            if (res) {
#pragma unroll
                for (int q = 0; q < 8 / EPC; ++q) {
                    Vec16<T> t = ld16(res + (long)m * p.ldr + n + q * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) rv[q * EPC + e] = to_f(t.e[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e] + bv[e] + bm;
                if (ra) x += to_f(ra[e]);
                x *= p.alpha;
                if (res) x += rv[e];
                v[e] = x;
            }
#pragma unroll
            for (int q = 0; q < 8 / EPC; ++q) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(v[q * EPC + e]);
                st16(out + ((long)m + (p.out_seg > 0 ? (long)(m / p.out_seg) * p.out_seg : 0L)) * p.ldo + n + q * EPC, o);
                if (stats) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {   // statistics of what GroupNorm will read back: the rounded value
                        const float xr = to_f(o.e[e]);
                        const float d = xr - s_mean[q * EPC + e];
                        s_mean[q * EPC + e] += d * (1.0f / (float)(pass + 1));
                        s_m2[q * EPC + e] += d * (xr - s_mean[q * EPC + e]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < nv) {
                    float x = v[e] + bv[e] + bm;
                    if (ra) x += to_f(ra[e]);
                    x *= p.alpha;
                    if (res) x += to_f(res[(long)m * p.ldr + n + e]);
                    out[((long)m + (p.out_seg > 0 ? (long)(m / p.out_seg) * p.out_seg : 0L)) * p.ldo + n + e] = from_f<T>(x);
                }
            }
        }
    }
    }
    if (stats) {   // uniform branch
        // lanes l, l^16, l^32, l^48 own the same 8 columns (rbase differs): butterfly-merge equal-count triples
        float cnt = (float)NPASS;
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float om = __shfl_xor(s_mean[e], off, 64), o2 = __shfl_xor(s_m2[e], off, 64);
                const float dlt = om - s_mean[e];
                s_m2[e] = s_m2[e] + o2 + dlt * dlt * cnt * 0.5f;
                s_mean[e] = 0.5f * (s_mean[e] + om);
            }
            cnt *= 2.0f;
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gst[((wv * 16 + lane) * 8 + e) * 2] = s_mean[e];
                gst[((wv * 16 + lane) * 8 + e) * 2 + 1] = s_m2[e];
            }
        }
        __syncthreads();
        if (threadIdx.x < BN_ && n0 + (int)threadIdx.x < p.N) {   // one thread per column: merge the waves (4*NPASS rows each)
            const int c = threadIdx.x;
            float mean = gst[((0 * 16 + c / 8) * 8 + (c & 7)) * 2], m2 = gst[((0 * 16 + c / 8) * 8 + (c & 7)) * 2 + 1];
            float na = 4.0f * NPASS;
            const float nb = 4.0f * NPASS;
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const float om = gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2], o2 = gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2 + 1];
                const float dlt = om - mean, nt = na + nb;
                mean += dlt * (nb / nt);
                m2 += o2 + dlt * dlt * na * (nb / nt);
                na = nt;
            }
            const int img = m0 / p.rows_per_img;
            const int slab = (m0 - img * p.rows_per_img) / BM_;
            float* o = p.gn_partial + (((long)img * p.gn_nslabs + slab) * p.N + n0 + c) * 3;
            o[0] = na; o[1] = mean; o[2] = m2;
        }
    }
}

// v2: 256x128 tile, 8 waves, LDS-DMA (global_load_lds) 3-stage ring — igemm2.hip
int launch_igemm_v2(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);
// igemm5.hip: persistent workgroups walking a tile sequence (16-bit FAST path, M % 256 == 0, >= 2 tiles per CU); -1 when not eligible
int launch_igemm_persistent(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);
// igemm6.hip: persistent, the A operand of a 3x3 / stride-1 / pad-1 convolution as a 2-D halo patch in LDS; -1 when not eligible
int launch_igemm_patch(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);
bool igemm_patch_eligible(int dtype, int mode, IgemmParams& p, int nz);   // the host-side test of launch_igemm_patch alone (fills p.mtiles / ntiles / gn_nslabs)
// convin.hip: 3x3 / stride-1 / pad-1 convolutions with eight input channels (operands straight from global memory, persistent); -1 when not eligible
int launch_conv_thin_in(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);

}  // namespace e2eft
