// igemm4.hip — 3x3 / stride-1 / pad-1 convolution as implicit GEMM with ROW-STRIP REUSE of the A operand (gfx950, fp16 / bf16).
//
// Why: igemm2's k-loop is not MFMA-bound.  Per k-tile a 256x128 workgroup tile pulls 48 KB through LDS-DMA (A 32 KB + W 16 KB) for
// 4.2 MFLOP, and the k-tile takes the same ~0.87 us at 1.5, 1.8 and 2.0 GHz shader clock (scripts/stamp_bench.py, profiles/r02_*):
// it is bound by the L2 -> LDS delivery of those bytes (~31 B/clk/CU), 1.6-1.8x the 1024-cycle MFMA floor.  An implicit-GEMM A tile
// of filter tap (ky, kx) is the SAME 256 input pixels as tap (ky, kx-1) shifted by one pixel: with NHWC rows of one image stored
// back to back, output pixel m reads input pixel  m + (ky-1)*W + (kx-1)  for every stride-1 / same-size convolution.  So one
// STRIP of 258 consecutive input pixels (x 64 channels) serves the three taps kx = 0,1,2 of a filter row; the fragment reads of tap
// kx start kx rows further down.  A traffic per k-tile drops from 32 KB to 11 KB (33 pieces per three k-tiles instead of 96), the
// DMA instruction count from 144 to 81 per three k-tiles.
//
// What a strip cannot express is the zero padding: input pixel (y, W) is, in linear order, the real pixel (y+1, 0).  Validity belongs to
// the (output pixel, tap) pair, so it is applied to the A FRAGMENTS: every lane owns two output rows, knows their nine tap-validity
// bits, and zeroes its fragment registers for an invalid pair — only in k-tiles whose wave holds such a pair (wave-uniform test; at
// 768 px wide one 256-pixel tile in three contains a row end, and only for kx != 1).  Strip rows that fall outside the tensor are
// out-of-range buffer offsets (the bounds check writes zeros).
//
// OUTCOME (round 2, MI355X): parity green, speed equal to igemm2 (1079 vs 1104, 890 vs 871, 1005 vs 1017, 995 vs 1019 TFLOP/s on conv
// 512@192^2, 128@768^2, 256@384^2, 512@96^2).  The kernel runs at a HIGHER shader clock than igemm2 (2.0-2.1 vs 1.75-1.95 GHz, both
// measured from inside the kernel with s_memtime / s_memrealtime) but needs more cycles per k-tile (1900 vs 1600-1800): the chip settles
// at the same power either way.  Phase clocks inside a k-tile (-DE2EFT_STAMPS): DMA wait 80-180 cycles, barrier skew 350-430, barrier ->
// first MFMA 600-860, MFMA phase 975-1150 for the two waves of a SIMD together — the loop is bound by the barrier / fragment-read /
// MFMA serialisation of eight lockstep waves, not by operand delivery.  Variants measured on the way: 4 waves of 128x64 (one per SIMD,
// 0.75 KB of LDS reads per MFMA): 2600-2900 cycles per k-tile; requesting the next tap's first A fragments before the barrier: -5 %.
// Kept as an opt-in (E2EFT_STRIP=1) measurement instrument with its own parity test.
//
// K order: filter row ky -> 64-channel chunk -> kx.  LDS: two strip buffers (264 rows x 128 B, the igemm2 XOR swizzle) + a 3-stage ring
// of 128x64 weight tiles; strip s+1 and weight tile u+2 are in flight while tile u computes; counted vmcnt waits, one barrier per
// k-tile, everything else (wave tiles 64x64 = 2x2 MFMA 32x32x16, epilogue with fused GroupNorm statistics) as igemm2.
#include "igemm.h"
#include <stdlib.h>
#include <type_traits>

namespace e2eft {

namespace strip {
constexpr int BM = 256, BN = 128, NW = 8;
constexpr int ROWS = 264;                        // 256 + 2, rounded up to whole 8-row DMA pieces (33)
constexpr int STRIP = ROWS * 128;
constexpr int BTILE = BN * 128;
constexpr int NB = 3;
constexpr int B0 = 2 * STRIP;                    // byte offset of the weight ring
constexpr int RING = B0 + NB * BTILE;
constexpr int EPI = BM * (BN + 4) * 4 + NW * 1024 + 512;
constexpr int LDS = RING > EPI ? RING : EPI;
constexpr unsigned int OOB = 0xF0000000u;
constexpr unsigned int RECORDS = 0xE0000000u;
}  // namespace strip

template <typename T> struct Mma4;
template <> struct Mma4<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma4<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void* lptr4_t;

#ifdef E2EFT_STAMPS
// phase clocks INSIDE one steady-state k-tile (the 10th), per wave: before the DMA wait, after it, after the barrier, after the first two
// fragment groups are requested, at the end of the tile
static __device__ long long g_tile4[4096 * 8 * 8];
#define TSTAMP(i) do { if (u == 9 && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_tile4[(blockIdx.x * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
template <int V> using IC4 = std::integral_constant<int, V>;

__device__ __forceinline__ int fast_div4(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// 512 threads = 8 waves of 64x64 (2x2 MFMA 32x32x16), as igemm2.  Measured alternatives (profiles/r02_strip_*.txt): the same loop with 4 waves
// of 128x64 (one per SIMD, 0.75 KB of fragment reads per MFMA instead of 1 KB) runs 2600-2900 cycles per k-tile against 1900 here — a lone
// wave exposes every LDS round trip — so the two waves per SIMD stay.
template <typename T>
__global__ __launch_bounds__(512) void igemm4_strip_kernel(const IgemmParams p) {
    using namespace strip;
    E2EFT_STAMP(0);
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    const int tid = threadIdx.x, lane = tid & 63;
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
    const int Wd = p.win, hw = p.hin * p.win;
    const int nchunk = p.cin >> 6;
    const int nstrips = 3 * nchunk;
    const int ntiles_k = 3 * nstrips;

    const T* __restrict__ X1 = (const T*)p.x1;
    const T* __restrict__ X2 = (const T*)p.x2;
    const T* __restrict__ Wt = (const T*)p.w;

    // ---- loader mapping (LDS-DMA: one wave-instruction = 8 rows x 128 B, lane -> row lane>>3, 16-byte slot lane&7) --------------------
    const int r8 = lane >> 3, slot = lane & 7;
    // A strip: 33 pieces; wave w takes pi = w + 8 q (q < 4), wave 0 also piece 32.  Strip row j = 8 pi + r8, logical chunk = slot ^ ((j >> 1) & 7):
    // (j >> 1) & 7 = (4 pi + (r8 >> 1)) & 7 depends on pi's parity only, which is the wave's
    const int jcA = slot ^ (((r8 >> 1) + 4 * (wave & 1)) & 7);
    const unsigned jc16 = (unsigned)jcA * 16u;
    const int pbase = m0 - Wd - 1;                       // linear input pixel of strip row 0 at ky = 0 (may be negative)
    const int totpix = p.M;
    const int jrow0 = 8 * wave + r8;                     // strip row of piece q: jrow0 + 64 q
    const bool edge = pbase < 0 || pbase + 2 * Wd + ROWS > totpix;   // uniform: some strip row of this tile may fall outside the tensor
    // weights: 16 pieces per tile, wave w takes rows 8 w + r8 and + 64; the swizzle term (row >> 1) & 7 is the same for both
    const int lrowB = 8 * wave + r8;
    const int jcB = slot ^ ((lrowB >> 1) & 7);
    unsigned int wo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + lrowB + 64 * i;
        wo[i] = n < p.N ? (unsigned)((lrowB + 64 * i) * p.ldw + jcB * 8) * 2u : OOB;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(X1 + (long)pbase * p.ldx1), 0, RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)((X2 ? X2 : X1) + (long)pbase * (X2 ? p.ldx2 : p.ldx1)), 0, RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (long)n0 * p.ldw), 0, RECORDS, 0x00020000);

    // ---- validity of (output row, tap) for this lane's two fragment rows i0 = wm*64 + l31 and i0 + 32 -----------------------------------
    unsigned int vbits[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int m = m0 + wm * 64 + ii * 32 + l31;
        unsigned int vb = 0;
        if (m < p.M) {
            const int b = fast_div4(m, hw);
            const int rem = m - b * hw;
            const int oy = fast_div4(rem, Wd), ox = rem - oy * Wd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    if ((unsigned)(oy + ky - 1) < (unsigned)p.hin && (unsigned)(ox + kx - 1) < (unsigned)Wd) vb |= 1u << (ky * 3 + kx);
        }
        vbits[ii] = vb;
    }
    unsigned int anyinv = 0;                             // wave-uniform: taps for which some lane of this wave must zero a fragment row
#pragma unroll
    for (int t = 0; t < 9; ++t)
        if (__builtin_amdgcn_ballot_w64((((vbits[0] & vbits[1]) >> t) & 1u) == 0u) != 0) anyinv |= 1u << t;
    anyinv = __builtin_amdgcn_readfirstlane(anyinv);

    // ---- fragment byte offsets (read side of the swizzle), per kx ---------------------------------------------------------------------
    int aoff[3][4], boff[4];
    {
        const int i0 = wm * 64 + l31;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sw = ((i0 + kx) >> 1) & 7;
#pragma unroll
            for (int g = 0; g < 4; ++g) aoff[kx][g] = (i0 + kx) * 128 + (((g * 2 + h) ^ sw) << 4);
        }
        const int swb = (l31 >> 1) & 7;
#pragma unroll
        for (int g = 0; g < 4; ++g) boff[g] = B0 + (wn * 64 + l31) * 128 + (((g * 2 + h) ^ swb) << 4);
    }

    floatx16 acc[2][2];
    auto mma_group = [&](const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        acc[0][0] = Mma4<T>::run(a0, b0, acc[0][0]);
        acc[0][1] = Mma4<T>::run(a0, b1, acc[0][1]);
        acc[1][0] = Mma4<T>::run(a1, b0, acc[1][0]);
        acc[1][1] = Mma4<T>::run(a1, b1, acc[1][1]);
    };

    int u = 0;                         // k-tile counter
    // ---- issue state: plain per-strip values, recomputed in next_strip() (selecting between captured variables made the compiler park
    // them in scratch and reload them with flat loads that wait on vmcnt, i.e. on the DMA stream) ------------------------------------------
    int is_ky = 0, is_ch = 0;              // (ky, chunk) of the NEXT strip to issue
    int a_ld2 = p.ldx1 * 2, a_so = 0, a_prow = pbase;
    bool a_second = false;
    int b_so = 0, b_kx = 0, b_c0 = 0, b_tap0 = 0;   // NEXT weight tile to issue: byte offset of its k position, kx, channel chunk, 3 * ky
    auto fire_piece = [&](char* dst, int jrow) {
        unsigned vo = __umul24((unsigned)jrow, (unsigned)a_ld2) + jc16;
        if (edge) vo = (unsigned)(a_prow + jrow) < (unsigned)totpix ? vo : OOB;
        if (a_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr4_t)dst, 16, vo, a_so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr4_t)dst, 16, vo, a_so, 0, 0);
    };
    auto fire_a = [&](auto sbc, auto qc) {   // piece q of the next strip -> strip buffer SB
        constexpr int SB = decltype(sbc)::value, Q = decltype(qc)::value;
        fire_piece(smem + SB * STRIP + (wave + 8 * Q) * 1024, jrow0 + 64 * Q);
    };
    auto fire_a_extra = [&](auto sbc) {      // piece 32 (strip rows 256..263), wave 0 only
        constexpr int SB = decltype(sbc)::value;
        fire_piece(smem + SB * STRIP + 32 * 1024, 256 + r8);
    };
    auto next_strip = [&]() {
        if (++is_ch == nchunk) { is_ch = 0; ++is_ky; }
        const int c0 = is_ch << 6;
        a_second = c0 >= p.c1;
        const int ld = a_second ? p.ldx2 : p.ldx1;
        a_ld2 = ld * 2;
        a_so = (is_ky * Wd * ld + (a_second ? c0 - p.c1 : c0)) * 2;
        a_prow = pbase + is_ky * Wd;
    };
    auto fire_b = [&](auto stc) {            // both pieces of the next weight tile -> ring stage ST
        constexpr int ST = decltype(stc)::value;
        char* dst = smem + B0 + ST * BTILE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr4_t)dst, 16, wo[0], b_so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr4_t)(dst + 8192), 16, wo[1], b_so, 0, 0);
        if (++b_kx == 3) {                   // k position (tap * cin + c0): kx+1 -> + cin; next chunk -> back two taps, + 64; next ky -> next tap row
            b_kx = 0;
            b_c0 += 64;
            if (b_c0 == p.cin) { b_c0 = 0; b_tap0 += 3; }
            b_so = (b_tap0 * p.cin + b_c0) * 2;
        } else {
            b_so += p.cin * 2;
        }
    };
    auto wait_rt = [&](int n) {   // at most n of this wave's DMA instructions still in flight (tail of the k loop)
        switch (n) {
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };

    // A fragments live across tiles: for kx = 1, 2 the first two groups of a tile are requested BEFORE that tile's barrier, at the end of the
    // previous tile — they come from the strip that is already resident, only the weight tile is new — so that the post-barrier read burst
    // (all eight waves at once, nothing for the matrix pipe to do: 860 of 1900 cycles per k-tile, profiles/r02_strip_tile_phases.txt) is
    // 4 weight reads per wave instead of 12
    u32x4 a0[4], a1[4];
    // one k-tile: strip buffer SB, tap column KX (= weight ring stage).  STEADY: a next strip and weight tile u+2 exist, wait counts are
    // immediates (kx = 0: strip + weight tile u landed, weight tile u+1 (2 instructions) may fly; kx = 1, 2: also the 2 A pieces of the
    // previous phase); otherwise runtime flags for the last strip
    auto tile = [&](auto sbc, auto kxc, auto steadyc, int tap, bool moreA, bool moreB, int inflight) {
        constexpr int SB = decltype(sbc)::value, KX = decltype(kxc)::value;
        constexpr bool STEADY = decltype(steadyc)::value != 0;
        TSTAMP(0);
        if constexpr (STEADY) {
            if constexpr (KX == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            wait_rt(inflight);
        }
        TSTAMP(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TSTAMP(2);
        const char* sa = smem + SB * STRIP;
        const char* sb = smem + KX * BTILE;
        const bool mask = (anyinv >> tap) & 1u;
        const bool v0 = (vbits[0] >> tap) & 1u, v1 = (vbits[1] >> tap) & 1u;
        u32x4 b0[3], b1[3];
        auto rd_a = [&](const char* base, auto kxn, auto gc) {
            constexpr int K = decltype(kxn)::value, g = decltype(gc)::value;
            a0[g] = *reinterpret_cast<const u32x4*>(base + aoff[K][g]);
            a1[g] = *reinterpret_cast<const u32x4*>(base + aoff[K][g] + 32 * 128);
        };
        auto rd_b = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            b0[sl] = *reinterpret_cast<const u32x4*>(sb + boff[g]);
            b1[sl] = *reinterpret_cast<const u32x4*>(sb + boff[g] + 32 * 128);
        };
        auto mm = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            if (mask) {   // wave-uniform branch: zero padding of this tap for the rows that need it
                const u32x4 z = {0u, 0u, 0u, 0u};
                a0[g] = v0 ? a0[g] : z;
                a1[g] = v1 ? a1[g] : z;
            }
            mma_group(a0[g], a1[g], b0[sl], b1[sl]);
        };
        if constexpr (KX == 0) { rd_a(sa, IC4<KX>{}, IC4<0>{}); rd_a(sa, IC4<KX>{}, IC4<1>{}); }   // a new strip: readable only now
        rd_b(IC4<0>{}, IC4<0>{});
        rd_b(IC4<1>{}, IC4<1>{});
        __builtin_amdgcn_sched_barrier(0);
        rd_a(sa, IC4<KX>{}, IC4<2>{});
        rd_b(IC4<2>{}, IC4<2>{});
        if (STEADY || moreA) {   // the next strip goes to the OTHER buffer: free since the barrier above (its last readers finished the previous strip)
            if constexpr (KX == 0) { fire_a(IC4<SB ^ 1>{}, IC4<0>{}); fire_a(IC4<SB ^ 1>{}, IC4<1>{}); }
            if constexpr (KX == 1) { fire_a(IC4<SB ^ 1>{}, IC4<2>{}); fire_a(IC4<SB ^ 1>{}, IC4<3>{}); }
            if constexpr (KX == 2) { if (wave == 0) fire_a_extra(IC4<SB ^ 1>{}); next_strip(); }
        }
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(3);
        mm(IC4<0>{}, IC4<0>{});
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(4);
        rd_a(sa, IC4<KX>{}, IC4<3>{});
        rd_b(IC4<3>{}, IC4<0>{});
        if (STEADY || moreB) fire_b(IC4<(KX + 2) % 3>{});   // stage of tile u+2; its last readers finished tile u-1 (barrier above)
        __builtin_amdgcn_sched_barrier(0);
        mm(IC4<1>{}, IC4<1>{});
        __builtin_amdgcn_sched_barrier(0);
        mm(IC4<2>{}, IC4<2>{});
        if constexpr (KX < 2) {   // the next tap's first two A groups, from this same strip, ahead of the next barrier
            rd_a(sa, IC4<(KX + 1) % 3>{}, IC4<0>{});
            rd_a(sa, IC4<(KX + 1) % 3>{}, IC4<1>{});
        }
        mm(IC4<3>{}, IC4<0>{});
        asm volatile("" ::: "memory");
        TSTAMP(5);
    };

    // ---- prologue: strip 0 and weight tiles 0, 1 in flight ---------------------------------------------------------------------------
    fire_a(IC4<0>{}, IC4<0>{}); fire_a(IC4<0>{}, IC4<1>{}); fire_a(IC4<0>{}, IC4<2>{}); fire_a(IC4<0>{}, IC4<3>{});
    if (wave == 0) fire_a_extra(IC4<0>{});
    next_strip();
    fire_b(IC4<0>{});
    fire_b(IC4<1>{});
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    E2EFT_STAMP(1);

    auto strip_body = [&](auto sbc, int s, int ky) {
        if (s + 1 < nstrips) {       // a next strip exists, hence weight tiles u+1, u+2 for all three tiles: immediates everywhere
            tile(sbc, IC4<0>{}, IC4<1>{}, ky * 3 + 0, true, true, 0); ++u;
            tile(sbc, IC4<1>{}, IC4<1>{}, ky * 3 + 1, true, true, 0); ++u;
            tile(sbc, IC4<2>{}, IC4<1>{}, ky * 3 + 2, true, true, 0); ++u;
        } else {                     // last strip: nothing more to issue for A, weight tiles run out
            tile(sbc, IC4<0>{}, IC4<0>{}, ky * 3 + 0, false, u + 2 < ntiles_k, 2); ++u;
            tile(sbc, IC4<1>{}, IC4<0>{}, ky * 3 + 1, false, u + 2 < ntiles_k, u + 1 < ntiles_k ? 2 : 0); ++u;
            tile(sbc, IC4<2>{}, IC4<0>{}, ky * 3 + 2, false, false, 0); ++u;
        }
    };
    {
        int s = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int ch = 0; ch < nchunk; ++ch, ++s) {
                if (s & 1) strip_body(IC4<1>{}, s, ky);
                else strip_body(IC4<0>{}, s, ky);
            }
    }

    // ---- epilogue: LDS-staged, vectorised, fused GroupNorm statistics (igemm.h) --------------------------------------------------------
    E2EFT_STAMP(2);
    igemm_epilogue<T, BM, BN, NW * 64>(p, smem, acc, wm, wn, l31, h, m0, n0, 0, 0);
    E2EFT_STAMP(4);
}

// >= 0: launched (0 or an error code); -1: not eligible, use igemm2
int launch_igemm_strip(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    // OFF by default: measured EQUAL to igemm2 (-2 % .. +2 % on the four dominant shapes, profiles/r02_strip_v2_ab.txt) although it moves
    // 44 % fewer bytes through LDS-DMA — the experiment that shows the k-loop is not bound by L2 -> LDS delivery (see the header and DESIGN.md
    // §3).  E2EFT_STRIP=1 selects it for eligible problems, 2 also for small ones (tests/test_strip_conv_gpu.py keeps it correct).
    static const int enabled = [] { const char* e = getenv("E2EFT_STRIP"); return e ? atoi(e) : 0; }();
    if (!enabled || mode != 1 || nz != 1 || p.ksplit_taps > 0) return -1;
    if (dtype != E2EFT_F16 && dtype != E2EFT_BF16) return -1;
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.zins > 1) return -1;
    if (p.hl != p.hin || p.wl != p.win || p.hout != p.hin || p.wout != p.win) return -1;
    if (p.cin % 64 != 0 || p.c1 % 64 != 0 || p.K != 9 * p.cin) return -1;
    if (p.M != (p.M / (p.hin * p.win)) * p.hin * p.win) return -1;
    const long blocks256 = (long)cdiv(p.M, strip::BM) * cdiv(p.N, strip::BN);
    if (blocks256 < 256 && enabled != 2) return -1;         // small problems: igemm2's 128-row tiles fill the machine better (E2EFT_STRIP=2: tests force it)
    if ((long)(2 * p.win + strip::ROWS) * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * 2 >= 0x40000000L) return -1;
    if ((long)128 * p.ldw * 2 >= 0x40000000L) return -1;
    p.mtiles = cdiv(p.M, strip::BM);
    p.ntiles = cdiv(p.N, strip::BN);
    if (p.gn_partial) {   // statistics need whole tiles inside one image and the vector epilogue (as igemm2)
        const bool ok = p.rows_per_img % strip::BM == 0 && p.M % p.rows_per_img == 0 && p.N % 8 == 0 && p.ldo % 8 == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                        (!p.residual || (p.ldr % 8 == 0 && (((uintptr_t)p.residual) & 15) == 0));
        if (ok) p.gn_nslabs = p.rows_per_img / strip::BM;
        else p.gn_partial = nullptr;
    }
    dim3 grid(p.mtiles * p.ntiles, 1, 1);
    if (dtype == E2EFT_F16) hipLaunchKernelGGL((igemm4_strip_kernel<f16>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((igemm4_strip_kernel<bf16>), grid, dim3(512), 0, s, p);
    return check_launch("igemm4");
}

}  // namespace e2eft

#ifdef E2EFT_STAMPS
// the stamp arrays are per translation unit (static __device__ in igemm.h): this file's copies, for scripts/stamp_bench.py
extern "C" int e2eft_debug_read_stamps4(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps), (size_t)nworkgroups * 8 * sizeof(long long));
}
extern "C" int e2eft_debug_read_tile4(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_tile4), (size_t)nworkgroups * 64 * sizeof(long long));
}
extern "C" int e2eft_debug_read_stamps4_rt(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps_rt), (size_t)nworkgroups * 2 * sizeof(long long));
}
#endif
