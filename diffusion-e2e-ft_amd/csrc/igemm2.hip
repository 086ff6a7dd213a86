// igemm2.hip — implicit-GEMM conv / NT-GEMM, large-problem variant: LDS-DMA ring, 64x64 wave tiles, gfx950.
//
// Same contract as igemm.hip (out = alpha*(A W^T + bias + rowadd) + residual; im2col gather with zero padding, fused
// nearest upsample, two-source channel concat).  What changes is how operands reach the MFMAs:
//   * global -> LDS by LDS-DMA (1 KiB per wave-instruction, no staging VGPRs); the im2col gather is the per-lane SOURCE
//     offset; zero padding / ragged edges are out-of-range buffer offsets (the buffer bounds check writes zeros);
//   * the LDS image of a piece is lane-linear (8 rows x 128 B), bank conflicts are removed by an XOR swizzle applied to
//     the source chunk index and to the fragment reads alike: slot = chunk ^ ((row >> 1) & 7) — 16 consecutive rows of one
//     chunk column then cover all 16 sixteen-byte slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * ring of NSTAGE stages, prefetch distance NSTAGE-1 k-tiles, counted `s_waitcnt vmcnt(N)` (never 0 in the steady
//     state of the 3-stage ring) and ONE raw s_barrier per k-tile;
//   * FAST path (every k-tile inside one filter tap and one source: cin % BK == 0, c1 % BK == 0; GEMM: K % BK == 0):
//     tap / channel / source are wave-uniform scalars, a row's pixel offset is recomputed only when the tap changes, the
//     per-piece address work is ONE v_add (offset += 128 B), the buffer descriptor in use changes only at tap / source
//     boundaries; the k-loop is unrolled over the ring so that every LDS address is a register + immediate;
//   * DMA pieces are issued between the MFMA groups, fragments are fetched two groups ahead of their MFMAs.
// Two geometries (NW = waves per workgroup, wave tile always 64x64 = 2x2 MFMA 32x32, BN = 128):
//   NW = 8: 256x128 tile, 3-stage ring, one workgroup per CU;   NW = 4: 128x128 tile, 2 stages, two workgroups per CU.
#include "igemm.h"
#include <type_traits>

namespace e2eft {

constexpr int BN2 = 128;
constexpr int B_STAGE = BN2 * 128;
template <int NW> struct Geo {
    static constexpr int BM = NW * 32;
    static constexpr int A_STAGE = BM * 128;
    static constexpr int STAGE = A_STAGE + B_STAGE;
    static constexpr int NSTAGE = NW == 8 ? 3 : 2;
    static constexpr int BPIECES = 16 / NW;          // B pieces (8 rows each) per wave
    static constexpr int NPIECES = 4 + BPIECES;      // DMA instructions per wave per k-tile
    static constexpr int EPI_BYTES = BM * (BN2 + 4) * 4 + NW * 1024 + 512;   // staged fp32 tile + GroupNorm-statistics scratch + pivots
    static constexpr int LDS_BYTES = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;
};

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

template <typename T> struct Mma2;
template <> struct Mma2<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma2<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr unsigned int OOB_SENTINEL = 0xF0000000u;   // byte offset beyond SRD_RECORDS: the load returns zeros
constexpr unsigned int SRD_RECORDS = 0xE0000000u;

template <int V> using IC = std::integral_constant<int, V>;

// n / d for 0 <= n < 2^31, d >= 1 with a quotient below 2^22 (rows / image size, pixels / row length): float estimate (relative
// error ~2^-22, so off by at most one) + one correction — 8 instructions instead of the ~25 of the generic unsigned division,
// eight of which sit in front of the first DMA of every workgroup.
__device__ __forceinline__ int fast_div(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// Tried and measured, not kept (conv 512->512 @192^2, B = 8, fp16; this kernel: 940-970 TF/s): a rotated loop that reads the next
// tile's first fragments before the current tile's last MFMAs (820), a ping-pong schedule with the two waves of a SIMD in
// opposite L/M phases (750), 4 dedicated LDS-DMA loader waves + 8 pure consumer waves (917), static s_setprio for either
// half (no change), NW = 4 with two workgroups per CU (840).  See DESIGN.md §3 for the cycle-stamp breakdown.
template <typename T, int MODE, bool FAST, int NW>
__global__ __launch_bounds__(NW * 64) void igemm2_kernel(const IgemmParams p) {
    E2EFT_STAMP(0);
    using G = Geo<NW>;
    static constexpr int BM2 = G::BM, A_STAGE = G::A_STAGE, STAGE2 = G::STAGE, NSTAGE = G::NSTAGE, BPIECES = G::BPIECES, NPIECES = G::NPIECES;
    static constexpr int RSTEP = 8 * NW;   // row distance between a wave's consecutive pieces
    static constexpr int D = NSTAGE - 1;   // prefetch distance in k-tiles
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    static constexpr int EPC = 16 / (int)sizeof(T);
    static constexpr int BK = 128 / (int)sizeof(T);

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
    const int m0 = mt * BM2, n0 = nt * BN2;
    const int z = blockIdx.y;
    const int zo = z / p.nzi, zi = z - zo * p.nzi;

    const T* __restrict__ X1 = (const T*)p.x1 + zo * p.sa_o + zi * p.sa_i;
    const T* __restrict__ X2 = (const T*)p.x2;
    const int tap0 = (MODE == 1 && p.ksplit_taps > 0) ? zi * p.ksplit_taps : 0;   // split-K over filter-tap rows (FAST path only)
    const T* __restrict__ W = (const T*)p.w + zo * p.sw_o + zi * p.sw_i + (long)tap0 * p.cin;
    const T* zsrc = (const T*)g_zero16;

    // ---- loader mapping: a wave-instruction fills one 1-KiB piece = 8 rows x 8 chunks; lane -> (row lane>>3, slot lane&7).
    // A pieces: rows 8*wave + RSTEP*i (i < 4); B pieces: rows 8*wave + RSTEP*i (i < BPIECES).  The lane's LOGICAL k-chunk
    // is slot ^ ((row >> 1) & 7), identical for all of its pieces.
    const int lrow = 8 * wave + (lane >> 3);
    const int jc = (lane & 7) ^ ((lrow >> 1) & 7);
    long a_base[4];
    int a_iy0[4], a_ix0[4];
    bool a_ok[4];
    long w_base[4];
    bool w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + RSTEP * i;
        a_ok[i] = m < p.M;
        if (MODE == 0) {
            a_base[i] = (long)m * p.ldx1;
            a_iy0[i] = a_ix0[i] = 0;
        } else {
            const int hw = p.hout * p.wout;
            const int mm = a_ok[i] ? m : 0;
            const int b = fast_div(mm, hw);
            const int rem = mm - b * hw;
            const int oy = fast_div(rem, p.wout), ox = rem - oy * p.wout;
            a_base[i] = b;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
        }
    }
#pragma unroll
    for (int i = 0; i < BPIECES; ++i) {
        const int n = n0 + lrow + RSTEP * i;
        w_ok[i] = n < p.N;
        w_base[i] = (long)n * p.ldw;
    }

    floatx16 acc[2][2];
    auto zero_acc = [&]() {   // called after the first DMA pieces are in flight (FAST path): 64 v_mov under the cold-start latency
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    if constexpr (!FAST) zero_acc();

    // fragment byte offsets inside the LDS ring, per stage (static stage index in the unrolled loop -> register + immediate);
    // read-side swizzle: rows wm*64 + i*32 + l31 -> ((row >> 1) & 7) == (l31 >> 1) & 7; second row-subtile = +32*128.
    const int sw = (l31 >> 1) & 7;
    int aofs[3][4], bofs[3][4];
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = sizeof(T) == 2 ? (g * 2 + h) : (h * 4 + g);
            aofs[s][g] = s * STAGE2 + (wm * 64 + l31) * 128 + ((chunk ^ sw) * 16);
            bofs[s][g] = s * STAGE2 + A_STAGE + (wn * 64 + l31) * 128 + ((chunk ^ sw) * 16);
        }

    auto mma_group = [&](const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        if constexpr (sizeof(T) == 2) {
            acc[0][0] = Mma2<T>::run(a0, b0, acc[0][0]);
            acc[0][1] = Mma2<T>::run(a0, b1, acc[0][1]);
            acc[1][0] = Mma2<T>::run(a1, b0, acc[1][0]);
            acc[1][1] = Mma2<T>::run(a1, b1, acc[1][1]);
        } else {
            // fp32: MFMA 32x32x2 step s pairs k-slot s of the lower half (lanes 0-31, chunks 0-3) with k-slot s of the upper
            // half (lanes 32-63, chunks 4-7); the same pairing is used for A and W, so the sum is the plain dot product.
            const floatx4 fa0 = __builtin_bit_cast(floatx4, a0), fa1 = __builtin_bit_cast(floatx4, a1);
            const floatx4 fb0 = __builtin_bit_cast(floatx4, b0), fb1 = __builtin_bit_cast(floatx4, b1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb1[s], acc[1][1], 0, 0, 0);
            }
        }
    };
    auto wait_dma = [&](bool younger_in_flight) {   // this wave's pieces of the tile about to be published have landed
        if (NSTAGE == 3 && younger_in_flight) {
            if constexpr (NPIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    const int nk = ((MODE == 1 && p.ksplit_taps > 0 ? p.ksplit_taps * p.cin : p.K) + BK - 1) / BK;

    if constexpr (FAST) {
        // ============================ FAST path =====================================================================
        unsigned int off1[4], off2[4];          // per-row byte offsets of the current tap in x1 / x2 (OOB_SENTINEL if invalid)
        unsigned int cur_a[4], cur_b[4];        // byte offsets of the NEXT tile to issue (advanced by 128 B per tile)
        int brel[4] = {0, 0, 0, 0};
        int tile_c = 0;
        const T* b1;
        const T* b2 = X2;
        if (MODE == 0) {
            b1 = X1 + (long)m0 * p.ldx1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cur_a[i] = (a_ok[i] ? (unsigned)((lrow + RSTEP * i) * p.ldx1 + jc * EPC) * (unsigned)sizeof(T) : OOB_SENTINEL) - 128u;
        } else {
            const int hw = p.hout * p.wout;
            const int b0 = m0 / hw;             // first image touched by this tile (uniform)
            b1 = X1 + (long)b0 * p.hin * p.win * p.ldx1;
            if (X2) b2 = X2 + (long)b0 * p.hin * p.win * p.ldx2;
#pragma unroll
            for (int i = 0; i < 4; ++i) { brel[i] = (int)a_base[i] - b0; cur_a[i] = 0; }
        }
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, SRD_RECORDS, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(b2 ? b2 : b1), 0, SRD_RECORDS, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, SRD_RECORDS, 0x00020000);
        __amdgpu_buffer_rsrc_t rsa = rs1;       // descriptor of the A source in use (changes at tap / source boundaries)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            cur_b[i] = ((i < BPIECES && w_ok[i]) ? (unsigned)((lrow + RSTEP * i) * p.ldw + jc * EPC) * (unsigned)sizeof(T) : OOB_SENTINEL) - 128u;

        // plain convolutions (no fused upsample, no zero insertion): the source pixel of tap (ky, kx) is the tap-(0,0) pixel plus the
        // scalar (ky * win + kx), so a tap change costs two compares, one add and one select per row instead of the full index
        // arithmetic (measured: ~800 cycles per tap change with the general code, every 2nd k-tile at Cin = 128)
        const bool plain_taps = MODE == 1 && p.zins <= 1 && p.hl == p.hin && p.wl == p.win;
        unsigned int base1[4], base2[4];
        if (plain_taps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned pix0 = (unsigned)((brel[i] * p.hin + a_iy0[i]) * p.win + a_ix0[i]);   // wraps for padded taps; only used when valid
                base1[i] = (pix0 * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T);
                base2[i] = (pix0 * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T);
            }
        }
        int ky = MODE == 1 ? tap0 / p.kw : 0, kx = MODE == 1 ? tap0 - ky * p.kw : 0;   // the current tap, advanced incrementally
        auto retap = [&]() {   // per-row pixel offsets of the current filter tap (uniform branch, once per tap)
            if (plain_taps) {
                const unsigned d1 = (unsigned)((ky * p.win + kx) * p.ldx1) * (unsigned)sizeof(T);
                const unsigned d2 = (unsigned)((ky * p.win + kx) * p.ldx2) * (unsigned)sizeof(T);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = a_ok[i] && (unsigned)(a_iy0[i] + ky) < (unsigned)p.hl && (unsigned)(a_ix0[i] + kx) < (unsigned)p.wl;
                    off1[i] = ok ? base1[i] + d1 : OOB_SENTINEL;
                    off2[i] = ok ? base2[i] + d2 : OOB_SENTINEL;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
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
                off1[i] = ok ? (pix * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB_SENTINEL;
                off2[i] = ok ? (pix * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB_SENTINEL;
            }
        };
        auto advance = [&]() {   // offsets / descriptor of the next tile to issue (tiles are issued in order 0,1,2,...)
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur_a[i] += 128u;
            } else {
                if (tile_c == 0) {
                    retap();
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur_a[i] = off1[i];
                    rsa = rs1;
                } else if (tile_c == p.c1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur_a[i] = off2[i];
                    rsa = rs2;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur_a[i] += 128u;
                }
                tile_c += BK;
                if (tile_c >= p.cin) {
                    tile_c = 0;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
#pragma unroll
            for (int i = 0; i < BPIECES; ++i) cur_b[i] += 128u;
        };
        auto fire = [&](auto stage_c, auto piece_c) {
            constexpr int S = decltype(stage_c)::value, Q = decltype(piece_c)::value;
            char* sa = smem + S * STAGE2 + wave * 1024;
            if constexpr (Q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr_t)(sa + Q * (RSTEP * 128)), 16, cur_a[Q], 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(sa + A_STAGE + (Q - 4) * (RSTEP * 128)), 16, cur_b[Q - 4], 0, 0, 0);
        };
        auto fire_all = [&](auto stage_c) {
            fire(stage_c, IC<0>{}); fire(stage_c, IC<1>{}); fire(stage_c, IC<2>{}); fire(stage_c, IC<3>{});
            fire(stage_c, IC<4>{}); fire(stage_c, IC<5>{});
            if constexpr (NPIECES == 8) { fire(stage_c, IC<6>{}); fire(stage_c, IC<7>{}); }
        };
        // one k-tile: publish it (counted wait + barrier), then 4 MFMA groups with fragments fetched two groups ahead and
        // the DMA of tile kt+D (stage DS) spread between the groups
        auto tile = [&](auto sc, auto dc, bool younger, bool more) {
            constexpr int S = decltype(sc)::value;
            wait_dma(younger);
            __builtin_amdgcn_s_barrier();   // every wave's pieces of this tile are in LDS; every wave is done with the previous tile
            asm volatile("" ::: "memory");
            if (more) advance();
            u32x4 a0[3], a1[3], b0[3], b1[3];
            auto rd = [&](auto gc, auto slotc) {
                constexpr int g = decltype(gc)::value, slot = decltype(slotc)::value;
                a0[slot] = *reinterpret_cast<const u32x4*>(smem + aofs[S][g]);
                a1[slot] = *reinterpret_cast<const u32x4*>(smem + aofs[S][g] + 32 * 128);
                b0[slot] = *reinterpret_cast<const u32x4*>(smem + bofs[S][g]);
                b1[slot] = *reinterpret_cast<const u32x4*>(smem + bofs[S][g] + 32 * 128);
            };
            rd(IC<0>{}, IC<0>{});
            rd(IC<1>{}, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            rd(IC<2>{}, IC<2>{});
            if (more) { fire(dc, IC<0>{}); fire(dc, IC<1>{}); }
            __builtin_amdgcn_sched_barrier(0);
            mma_group(a0[0], a1[0], b0[0], b1[0]);
            __builtin_amdgcn_sched_barrier(0);
            rd(IC<3>{}, IC<0>{});
            if (more) { fire(dc, IC<2>{}); fire(dc, IC<3>{}); }
            __builtin_amdgcn_sched_barrier(0);
            mma_group(a0[1], a1[1], b0[1], b1[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                fire(dc, IC<4>{}); fire(dc, IC<5>{});
                if constexpr (NPIECES == 8) { fire(dc, IC<6>{}); fire(dc, IC<7>{}); }
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_group(a0[2], a1[2], b0[2], b1[2]);
            mma_group(a0[0], a1[0], b0[0], b1[0]);
            asm volatile("" ::: "memory");
        };

        // prologue: tiles 0 .. D-1 in flight
        advance();
        fire_all(IC<0>{});
        if constexpr (D == 2) {
            if (nk > 1) { advance(); fire_all(IC<1>{}); }
        }
        zero_acc();
        E2EFT_STAMP(1);
        int kt = 0;
        if constexpr (NSTAGE == 3) {
            for (; kt + 3 + D <= nk; kt += 3) {   // steady state: every prefetch exists, every stage index is static
                tile(IC<0>{}, IC<2>{}, true, true);
#ifdef E2EFT_STAMPS
                if (kt == 0) E2EFT_STAMP(5);
#endif
                tile(IC<1>{}, IC<0>{}, true, true);
                tile(IC<2>{}, IC<1>{}, true, true);
#ifdef E2EFT_STAMPS
                if (kt == 0) E2EFT_STAMP(6);
                if (kt == 3) E2EFT_STAMP(7);
#endif
            }
            // tail: at most 4 tiles left (kt is a multiple of 3)
            if (kt < nk) { tile(IC<0>{}, IC<2>{}, kt + 1 < nk, kt + D < nk); ++kt; }
            if (kt < nk) { tile(IC<1>{}, IC<0>{}, kt + 1 < nk, kt + D < nk); ++kt; }
            if (kt < nk) { tile(IC<2>{}, IC<1>{}, kt + 1 < nk, kt + D < nk); ++kt; }
            if (kt < nk) { tile(IC<0>{}, IC<2>{}, kt + 1 < nk, kt + D < nk); ++kt; }
        } else {
            for (; kt + 2 + D <= nk; kt += 2) {
                tile(IC<0>{}, IC<1>{}, false, true);
                tile(IC<1>{}, IC<0>{}, false, true);
            }
            if (kt < nk) { tile(IC<0>{}, IC<1>{}, false, kt + D < nk); ++kt; }
            if (kt < nk) { tile(IC<1>{}, IC<0>{}, false, kt + D < nk); ++kt; }
        }
    } else {
        // ============================ general path: per-lane pointers, zero block for padding ========================
        // incremental k decomposition of this thread's chunk: k = kt*BK + jc*EPC = (ky*kw + kx)*cin + c
        int k_cur = jc * EPC;
        int c_cur = 0, kx_cur = 0, ky_cur = 0;
        if (MODE == 1) {
            const int kpos = k_cur / p.cin;
            c_cur = k_cur - kpos * p.cin;
            ky_cur = kpos / p.kw;
            kx_cur = kpos - ky_cur * p.kw;
        }
        auto issue = [&](int stage) {   // DMA of the NEXT k-tile (tiles are issued in order 0,1,2,...)
            char* sa = smem + stage * STAGE2 + wave * 1024;
            char* sb = sa + A_STAGE;
            const bool kok = k_cur < p.K;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const T* src = (kok && a_ok[i]) ? X1 + a_base[i] + k_cur : zsrc;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + i * (RSTEP * 128)), 16, 0, 0);
                }
            } else {
                const bool second = c_cur >= p.c1;
                const T* sbase = second ? X2 : X1;
                const int ld = second ? p.ldx2 : p.ldx1;
                const int cc = second ? c_cur - p.c1 : c_cur;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int iy = a_iy0[i] + ky_cur, ix = a_ix0[i] + kx_cur;
                    bool ok = kok && a_ok[i] && (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
                    int sy = iy, sx = ix;
                    if (p.zins > 1) {
                        ok = ok && (iy % p.zins == 0) && (ix % p.zins == 0);
                        sy = iy / p.zins; sx = ix / p.zins;
                    } else {
                        if (p.hl != p.hin) sy = min((int)floorf(iy * p.up_sh), p.hin - 1);
                        if (p.wl != p.win) sx = min((int)floorf(ix * p.up_sw), p.win - 1);
                    }
                    const long pix = (a_base[i] * p.hin + sy) * p.win + sx;
                    const T* src = ok ? sbase + pix * ld + cc : zsrc;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + i * (RSTEP * 128)), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < BPIECES; ++i) {
                const T* src = (kok && w_ok[i]) ? W + w_base[i] + k_cur : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + i * (RSTEP * 128)), 16, 0, 0);
            }
            k_cur += BK;
            if (MODE == 1) {
                c_cur += BK;
                while (c_cur >= p.cin) {
                    c_cur -= p.cin;
                    if (++kx_cur == p.kw) { kx_cur = 0; ++ky_cur; }
                }
            }
        };
        issue(0);
        if (D == 2 && nk > 1) issue(1);
        for (int kt = 0; kt < nk; ++kt) {
            wait_dma(kt + 1 < nk);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + D < nk) issue((kt + D) % NSTAGE);
            const int so = (kt % NSTAGE) * STAGE2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(smem + so + aofs[0][g]);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(smem + so + aofs[0][g] + 32 * 128);
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(smem + so + bofs[0][g]);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(smem + so + bofs[0][g] + 32 * 128);
                mma_group(a0, a1, b0, b1);
            }
            asm volatile("" ::: "memory");
        }
    }

    // ---- epilogue: LDS-staged, vectorised (igemm.h) ----
    E2EFT_STAMP(2);
    igemm_epilogue<T, BM2, BN2, NW * 64>(p, smem, acc, wm, wn, l31, h, m0, n0, zo, zi);
    E2EFT_STAMP(4);
}

template <typename T, int MODE, int NW> static int launch2(IgemmParams& p, int nz, hipStream_t s) {
    p.mtiles = cdiv(p.M, Geo<NW>::BM);
    p.ntiles = cdiv(p.N, BN2);
    if (p.gn_partial) {   // statistics need whole tiles inside one image and the vector epilogue
        const bool ok = nz == 1 && p.rows_per_img % Geo<NW>::BM == 0 && p.M % p.rows_per_img == 0 && p.N % 8 == 0 &&
                        p.ldo % (16 / (int)sizeof(T)) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                        (!p.residual || (p.ldr % (16 / (int)sizeof(T)) == 0 && (((uintptr_t)p.residual) & 15) == 0));
        if (ok) p.gn_nslabs = p.rows_per_img / Geo<NW>::BM;
        else p.gn_partial = nullptr;
    }
    dim3 grid(p.mtiles * p.ntiles, nz, 1);
    constexpr int BK = 128 / (int)sizeof(T);
    bool fast;
    if (MODE == 0) {
        fast = p.K % BK == 0 && (long)256 * p.ldx1 * (long)sizeof(T) < 0x40000000L && (long)128 * p.ldw * (long)sizeof(T) < 0x40000000L;
    } else {
        const long img_bytes = (long)p.hin * p.win * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * (long)sizeof(T);
        const long span_imgs = 256 / ((long)p.hout * p.wout) + 2;
        fast = p.cin % BK == 0 && p.c1 % BK == 0 && img_bytes * span_imgs < 0xD0000000L && (long)128 * p.ldw * (long)sizeof(T) < 0x40000000L;
    }
    const bool nofast = option(E2EFT_OPT_IGEMM_GENERAL_OPERANDS) != 0;
    if (p.ksplit_taps > 0 && !(fast && !nofast && MODE == 1)) return fail(E2EFT_ERR_BAD_ARG, "igemm2: split-K needs the FAST conv path");
    if (fast && !nofast) hipLaunchKernelGGL((igemm2_kernel<T, MODE, true, NW>), grid, dim3(NW * 64), 0, s, p);
    else hipLaunchKernelGGL((igemm2_kernel<T, MODE, false, NW>), grid, dim3(NW * 64), 0, s, p);
    tag_kernel("igemm2_kernel<%s, %d, %s, %d>", sizeof(T) == 4 ? "float" : (std::is_same<T, f16>::value ? "_Float16" : "__bf16"), MODE, (fast && !nofast) ? "true" : "false", NW);
    return check_launch("igemm2");
}

template <typename T> static int launch2_t(int mode, IgemmParams& p, int nz, hipStream_t s) {
    const int forced_nw = option(E2EFT_OPT_IGEMM2_WAVES);
    // 256-row tiles (8 waves) halve the weight traffic and the per-tile fixed cost; 128-row tiles (4 waves, two workgroups per CU) only pay when the
    // 256-row grid would leave most of the machine idle.  16-bit, measured with the variant forced (profiles/r03k_variant_sweep.txt): at 150-180 tiles
    // of 256 rows (conv 1280->1280 @12^2 with split K, GEMM 4608 x 1280 x 1280 / x 5120, conv 1280->1280 / 2560->1280 @24^2) the 8-wave kernel is
    // 9-23 % faster than twice as many 128-row workgroups, at 360 tiles the two are equal: the switch sits at half a round.
    const long blocks256 = (long)cdiv(p.M, 256) * cdiv(p.N, BN2) * nz;
    const long switch_at = sizeof(T) == 4 ? 256 : 128;
    const int nw = forced_nw ? forced_nw : (blocks256 < switch_at ? 4 : 8);
    if (nw == 4) return mode ? launch2<T, 1, 4>(p, nz, s) : launch2<T, 0, 4>(p, nz, s);
    return mode ? launch2<T, 1, 8>(p, nz, s) : launch2<T, 0, 8>(p, nz, s);
}

int launch_igemm_v2(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    if (dtype == E2EFT_F32) return launch2_t<float>(mode, p, nz, s);
    if (dtype == E2EFT_F16) return launch2_t<f16>(mode, p, nz, s);
    if (dtype == E2EFT_BF16) return launch2_t<bf16>(mode, p, nz, s);
    return fail(E2EFT_ERR_BAD_ARG, "igemm2: bad dtype %d", dtype);
}

}  // namespace e2eft

#ifdef E2EFT_STAMPS
extern "C" int e2eft_debug_read_stamps(long long* host, int nworkgroups) {   // debug builds only; not part of include/e2eft.h
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps), (size_t)nworkgroups * 8 * sizeof(long long));
}
extern "C" int e2eft_debug_read_stamps_rt(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps_rt), (size_t)nworkgroups * 2 * sizeof(long long));
}
#endif
