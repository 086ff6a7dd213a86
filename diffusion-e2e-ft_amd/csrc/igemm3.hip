// igemm3.hip — implicit-GEMM variant built so that TWO 8-wave workgroups are resident per CU (fp16 / bf16, FAST-path problems).
//
// Why: cycle stamps of igemm2 (scripts/stamp_bench.py) show a fixed ~20k cycles per 256x128 workgroup that is not MFMA time —
// prologue DMA latency 3.8k, accumulator staging 2.7k, and above all the epilogue's global stores 9k (a CU drains only ~7 B/clk
// of stores, 64 KB per tile) + 4.3k of GroupNorm statistics — against 1.4-1.9k cycles per 64-wide k-tile.  With one workgroup
// per CU (igemm2 needs 147 KB of LDS) nothing overlaps it: 37 % of a conv 128->128 (18 k-tiles), 13 % at K = 4608.
//
// Here the k-tile is 32 wide: a stage is 24 KB (A 256 rows x 64 B + B 128 rows x 64 B), the 3-stage ring 72 KB, the fp32
// epilogue staging is done in two 128-row halves (67.6 KB) — 75.8 KB per workgroup, two per CU, <= 128 VGPRs per wave.  While one
// workgroup stores its tile the other one's MFMAs keep the matrix pipes busy.  Tile shape (256x128), wave tile (64x64 = 2x2
// 32x32x16 MFMAs), DMA bytes and fragment reads per MFMA are those of igemm2 (RESULT: no gain, see launch_igemm_v3); the lane-linear LDS image uses 64-byte rows with
// slot = chunk ^ ((row >> 2) & 3) (conflict-free ds_read_b128 for 16 consecutive rows).
#include "igemm.h"
#include <stdlib.h>
#include <type_traits>

namespace e2eft {

namespace g3 {
constexpr int BM = 256, BN = 128;
constexpr int A_STAGE = BM * 64, B_STAGE = BN * 64, STAGE = A_STAGE + B_STAGE, NSTAGE = 3;
constexpr int LDT = BN + 4;
constexpr int EPI_HALF = 128 * LDT * 4;            // fp32 staging of 128 rows
constexpr int LDS_BYTES = EPI_HALF + 8 * 1024;     // + GroupNorm-statistics scratch (>= NSTAGE * STAGE = 73728)
static_assert(LDS_BYTES >= NSTAGE * STAGE, "ring must fit");
constexpr unsigned int OOB = 0xF0000000u;          // byte offset beyond the descriptor's range: the load returns zeros
constexpr unsigned int RECORDS = 0xE0000000u;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int V> using IC = std::integral_constant<int, V>;

template <typename T> struct Mma;
template <> struct Mma<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};
}  // namespace g3

// grid (mtiles * ntiles, nz), 512 threads.  Host guarantees the FAST conditions (launch_igemm_v3).
template <typename T, int MODE>
__global__ __launch_bounds__(512, 4) void igemm3_kernel(const IgemmParams p) {
    using namespace g3;
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int EPC = 8;      // elements per 16-byte chunk
    constexpr int BK = 32;      // elements per k-tile
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int nblk = p.mtiles * p.ntiles;
    int lid;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = lid / p.ntiles, nt = lid - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int zo = z / p.nzi, zi = z - zo * p.nzi;

    const T* __restrict__ X1 = (const T*)p.x1 + zo * p.sa_o + zi * p.sa_i;
    const T* __restrict__ X2 = (const T*)p.x2;
    const T* __restrict__ W = (const T*)p.w + zo * p.sw_o + zi * p.sw_i;

    // ---- loader mapping: one wave-instruction = 1 KiB = 16 rows x 4 chunks, lane -> (row lane>>2, slot lane&3); the lane's
    // LOGICAL k-chunk is slot ^ ((row >> 2) & 3).  A pieces of wave w: rows 16w + .. and 128 + 16w + ..; B piece: rows 16w + ..
    const int pr = lane >> 2;
    const int jc = (lane & 3) ^ ((pr >> 2) & 3);
    const int lrow = 16 * wave + pr;
    long a_base[2];
    int a_iy0[2], a_ix0[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + lrow + 128 * i;
        a_ok[i] = m < p.M;
        if (MODE == 0) {
            a_base[i] = 0;
            a_iy0[i] = a_ix0[i] = 0;
        } else {
            const int hw = p.hout * p.wout;
            const int mm = a_ok[i] ? m : 0;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / p.wout, ox = rem - oy * p.wout;
            a_base[i] = b;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
        }
    }
    const bool w_ok = n0 + lrow < p.N;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets (k-step ks: logical chunk 2 ks + h), second 32-row sub-tile = + 32 * 64
    const int sw = (l31 >> 2) & 3;
    int aofs[2], bofs[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aofs[ks] = (wm * 64 + l31) * 64 + (((2 * ks + h) ^ sw) * 16);
        bofs[ks] = A_STAGE + (wn * 64 + l31) * 64 + (((2 * ks + h) ^ sw) * 16);
    }

    const int nk = (p.K + BK - 1) / BK;

    unsigned int off1[2], off2[2], cur_a[2], cur_b;
    int brel[2] = {0, 0};
    int tile_c = 0, tap = 0;
    const T* b1;
    const T* b2 = X2;
    if (MODE == 0) {
        b1 = X1 + (long)m0 * p.ldx1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            cur_a[i] = (a_ok[i] ? (unsigned)((lrow + 128 * i) * p.ldx1 + jc * EPC) * 2u : OOB) - 64u;
    } else {
        const int hw = p.hout * p.wout;
        const int b0 = m0 / hw;
        b1 = X1 + (long)b0 * p.hin * p.win * p.ldx1;
        if (X2) b2 = X2 + (long)b0 * p.hin * p.win * p.ldx2;
#pragma unroll
        for (int i = 0; i < 2; ++i) { brel[i] = (int)a_base[i] - b0; cur_a[i] = 0; }
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(b2 ? b2 : b1), 0, RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, RECORDS, 0x00020000);
    __amdgpu_buffer_rsrc_t rsa = rs1;
    cur_b = (w_ok ? (unsigned)(lrow * p.ldw + jc * EPC) * 2u : OOB) - 64u;

    auto retap = [&]() {
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
            int sy = iy, sx = ix;
            if (p.zins > 1) {
                ok = ok && (iy % p.zins == 0) && (ix % p.zins == 0);
                sy = iy / p.zins; sx = ix / p.zins;
            } else {
                if (p.hl != p.hin) sy = min((int)floorf(iy * p.up_sh), p.hin - 1);
                if (p.wl != p.win) sx = min((int)floorf(ix * p.up_sw), p.win - 1);
            }
            const unsigned pix = (unsigned)((brel[i] * p.hin + sy) * p.win + sx);
            off1[i] = ok ? (pix * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * 2u : OOB;
            off2[i] = ok ? (pix * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * 2u : OOB;
        }
    };
    auto advance = [&]() {
        if (MODE == 0) {
            cur_a[0] += 64u; cur_a[1] += 64u;
        } else {
            if (tile_c == 0) {
                retap();
                cur_a[0] = off1[0]; cur_a[1] = off1[1];
                rsa = rs1;
            } else if (tile_c == p.c1) {
                cur_a[0] = off2[0]; cur_a[1] = off2[1];
                rsa = rs2;
            } else {
                cur_a[0] += 64u; cur_a[1] += 64u;
            }
            tile_c += BK;
            if (tile_c >= p.cin) { tile_c = 0; ++tap; }
        }
        cur_b += 64u;
    };
    auto fire = [&](auto stage_c, auto piece_c) {
        constexpr int S = decltype(stage_c)::value, Q = decltype(piece_c)::value;
        char* sa = smem + S * STAGE + wave * 1024;
        if constexpr (Q < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr_t)(sa + Q * 8192), 16, cur_a[Q], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(sa + A_STAGE), 16, cur_b, 0, 0, 0);
    };
    auto mma = [&](const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        acc[0][0] = Mma<T>::run(a0, b0, acc[0][0]);
        acc[0][1] = Mma<T>::run(a0, b1, acc[0][1]);
        acc[1][0] = Mma<T>::run(a1, b0, acc[1][0]);
        acc[1][1] = Mma<T>::run(a1, b1, acc[1][1]);
    };
    // one k-tile: counted wait + barrier publish it; both k-steps' fragments are read up front, the DMA of tile kt+2 is spread
    // between the MFMA groups
    auto tile = [&](auto sc, auto dc, bool younger, bool more) {
        constexpr int S = decltype(sc)::value;
        if (younger) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) advance();
        const char* sb = smem + S * STAGE;
        u32x4 a0 = *reinterpret_cast<const u32x4*>(sb + aofs[0]);
        u32x4 a1 = *reinterpret_cast<const u32x4*>(sb + aofs[0] + 2048);
        u32x4 b0 = *reinterpret_cast<const u32x4*>(sb + bofs[0]);
        u32x4 b1 = *reinterpret_cast<const u32x4*>(sb + bofs[0] + 2048);
        if (more) fire(dc, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, a1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        // second k-step: its fragments reuse the first step's registers (128-VGPR budget for two workgroups per CU); the other
        // workgroup's MFMAs cover this read latency
        a0 = *reinterpret_cast<const u32x4*>(sb + aofs[1]);
        a1 = *reinterpret_cast<const u32x4*>(sb + aofs[1] + 2048);
        b0 = *reinterpret_cast<const u32x4*>(sb + bofs[1]);
        b1 = *reinterpret_cast<const u32x4*>(sb + bofs[1] + 2048);
        if (more) { fire(dc, IC<1>{}); fire(dc, IC<2>{}); }
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, a1, b0, b1);
        asm volatile("" ::: "memory");
    };

    // prologue: tiles 0 and 1 in flight
    advance();
    fire(IC<0>{}, IC<0>{}); fire(IC<0>{}, IC<1>{}); fire(IC<0>{}, IC<2>{});
    if (nk > 1) {
        advance();
        fire(IC<1>{}, IC<0>{}); fire(IC<1>{}, IC<1>{}); fire(IC<1>{}, IC<2>{});
    }
    int kt = 0;
    for (; kt + 5 <= nk; kt += 3) {   // steady state: every prefetch exists, every stage index is static
        tile(IC<0>{}, IC<2>{}, true, true);
        tile(IC<1>{}, IC<0>{}, true, true);
        tile(IC<2>{}, IC<1>{}, true, true);
    }
    if (kt < nk) { tile(IC<0>{}, IC<2>{}, kt + 1 < nk, kt + 2 < nk); ++kt; }
    if (kt < nk) { tile(IC<1>{}, IC<0>{}, kt + 1 < nk, kt + 2 < nk); ++kt; }
    if (kt < nk) { tile(IC<2>{}, IC<1>{}, kt + 1 < nk, kt + 2 < nk); ++kt; }
    if (kt < nk) { tile(IC<0>{}, IC<2>{}, kt + 1 < nk, kt + 2 < nk); ++kt; }

    // ---- epilogue: the fp32 tile goes through LDS in two 128-row halves, rows are written back with 16-byte vectors ----
    //   out = alpha * (acc + bias + rowadd[img(m)]) + residual   (+ GroupNorm partial statistics, see igemm.h)
    {
        constexpr int CPR = BN / 8;              // 16 chunks of 8 columns per row
        constexpr int RPP = 512 / CPR;           // 32 rows per pass
        constexpr int NPASS = 8;                 // rows per thread over both halves
        float* stg = reinterpret_cast<float*>(smem);
        float* gst = reinterpret_cast<float*>(smem + EPI_HALF);
        const T* __restrict__ bias = (const T*)p.bias;
        const T* __restrict__ rowadd = (const T*)p.rowadd;
        const T* __restrict__ res = p.residual ? (const T*)p.residual + zo * p.sr_o + zi * p.sr_i : nullptr;
        T* __restrict__ out = (T*)p.out + zo * p.so_o + zi * p.so_i;
        const int chunk = tid % CPR, rbase = tid / CPR;
        const int n = n0 + chunk * 8;
        const bool col_ok = n < p.N;
        const bool vec = (p.N % 8 == 0) && (p.ldo % EPC == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                         (!res || (p.ldr % EPC == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
        const bool stats = p.gn_partial != nullptr;   // host guarantees: vec, whole tile inside one image, all rows valid
        const int nv = min(8, p.N - n);
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && !p.bias_along_m && col_ok && e < nv) ? to_f(bias[n + e]) : 0.f;
        float s_mean[8], s_m2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s_mean[e] = s_m2[e] = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();   // half 0: every wave is done with the last k-tile; half 1: every thread has read half 0
            if ((wm >> 1) == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stg[((wm & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDT + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int pass = half * 4 + ps;
                const int row = rbase + ps * RPP;
                const int m = m0 + half * 128 + row;
                if (m >= p.M || !col_ok) continue;
                const floatx4 t0 = *reinterpret_cast<const floatx4*>(stg + row * LDT + chunk * 8);
                const floatx4 t1 = *reinterpret_cast<const floatx4*>(stg + row * LDT + chunk * 8 + 4);
                float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                const float bm = (bias && p.bias_along_m) ? to_f(bias[m]) : 0.f;
                const T* ra = rowadd ? rowadd + (long)(m / p.rows_per_img) * p.N + n : nullptr;
                if (vec) {
                    float rv[8];
                    if (res) {
                        Vec16<T> t = ld16(res + (long)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = to_f(t.e[e]);
                    }
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = v[e] + bv[e] + bm;
                        if (ra) x += to_f(ra[e]);
                        x *= p.alpha;
                        if (res) x += rv[e];
                        o.e[e] = from_f<T>(x);
                    }
                    st16(out + (long)m * p.ldo + n, o);
                    if (stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {   // statistics of what GroupNorm will read back: the rounded value
                            const float xr = to_f(o.e[e]);
                            const float d = xr - s_mean[e];
                            s_mean[e] += d * (1.0f / (float)(pass + 1));
                            s_m2[e] += d * (xr - s_mean[e]);
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
                            out[(long)m * p.ldo + n + e] = from_f<T>(x);
                        }
                    }
                }
            }
        }
        if (stats) {   // uniform branch; identical to igemm_epilogue's merge (NPASS rows per thread, 8 waves)
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
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gst[((wave * 16 + lane) * 8 + e) * 2] = s_mean[e];
                    gst[((wave * 16 + lane) * 8 + e) * 2 + 1] = s_m2[e];
                }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                const int c = tid;
                float mean = gst[((c / 8) * 8 + (c & 7)) * 2], m2 = gst[((c / 8) * 8 + (c & 7)) * 2 + 1];
                float na = 4.0f * NPASS;
                const float nb = 4.0f * NPASS;
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const float om = gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2], o2 = gst[((w * 16 + c / 8) * 8 + (c & 7)) * 2 + 1];
                    const float dlt = om - mean, ntot = na + nb;
                    mean += dlt * (nb / ntot);
                    m2 += o2 + dlt * dlt * na * (nb / ntot);
                    na = ntot;
                }
                const int img = m0 / p.rows_per_img;
                const int slab = (m0 - img * p.rows_per_img) / BM;
                float* o = p.gn_partial + (((long)img * p.gn_nslabs + slab) * p.N + n0 + c) * 3;
                o[0] = na; o[1] = mean; o[2] = m2;
            }
        }
    }
}

// returns -1 when the problem does not meet this variant's conditions (the caller falls back to igemm2)
int launch_igemm_v3(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    using namespace g3;
    if (dtype != E2EFT_F16 && dtype != E2EFT_BF16) return -1;
    // MEASURED (conv3x3, B = 8, fp16; igemm2 -> this kernel): 128->128 @768^2 731 -> 750 TF/s, 128->128 @384^2 636 -> 633,
    // 256->256 @384^2 886 -> 834, 512->512 @192^2 978 -> 886.  Two resident workgroups do NOT hide the store phase: a CU's
    // vector-memory pipe is what the epilogue saturates (64 store instructions at ~140 cycles each), and the other workgroup's
    // LDS-DMA loads queue behind it; staggering the second workgroup's start changes nothing.  Kept as an opt-in second
    // implementation (E2EFT_IGEMM3=1), off by default.
    static const int forced = [] { const char* e = getenv("E2EFT_IGEMM3"); return e ? atoi(e) : 0; }();
    if (forced != 1) return -1;
    bool fast;
    if (mode == 0) {
        fast = p.K % 32 == 0 && (long)256 * p.ldx1 * 2 < 0x40000000L && (long)128 * p.ldw * 2 < 0x40000000L;
    } else {
        const long img_bytes = (long)p.hin * p.win * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * 2;
        const long span_imgs = 256 / ((long)p.hout * p.wout) + 2;
        fast = p.cin % 32 == 0 && p.c1 % 32 == 0 && img_bytes * span_imgs < 0xD0000000L && (long)128 * p.ldw * 2 < 0x40000000L;
    }
    if (!fast) return -1;
    p.mtiles = cdiv(p.M, BM);
    p.ntiles = cdiv(p.N, BN);
    if (p.gn_partial) {
        const bool ok = nz == 1 && p.rows_per_img % BM == 0 && p.M % p.rows_per_img == 0 && p.N % 8 == 0 && p.ldo % 8 == 0 &&
                        (((uintptr_t)p.out) & 15) == 0 && (!p.residual || (p.ldr % 8 == 0 && (((uintptr_t)p.residual) & 15) == 0));
        if (ok) p.gn_nslabs = p.rows_per_img / BM;
        else p.gn_partial = nullptr;
    }
    const dim3 grid(p.mtiles * p.ntiles, nz, 1);
    if (dtype == E2EFT_F16) {
        if (mode) hipLaunchKernelGGL((igemm3_kernel<f16, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm3_kernel<f16, 0>), grid, dim3(512), 0, s, p);
    } else {
        if (mode) hipLaunchKernelGGL((igemm3_kernel<bf16, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm3_kernel<bf16, 0>), grid, dim3(512), 0, s, p);
    }
    return check_launch("igemm3");
}

}  // namespace e2eft
