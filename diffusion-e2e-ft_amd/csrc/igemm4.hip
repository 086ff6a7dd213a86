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
template <int V> using IC4 = std::integral_constant<int, V>;

__device__ __forceinline__ int fast_div4(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

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

    const T* __restrict__ X1 = (const T*)p.x1;
    const T* __restrict__ X2 = (const T*)p.x2;
    const T* __restrict__ Wt = (const T*)p.w;

    // ---- loader mapping (LDS-DMA: one wave-instruction = 8 rows x 128 B, lane -> row lane>>3, 16-byte slot lane&7) --------------------
    const int r8 = lane >> 3, slot = lane & 7;
    // A strip: piece pi = wave + 8 q (q < 4), plus piece 32 for wave 0; strip row j = 8 pi + r8; logical chunk = slot ^ ((j >> 1) & 7),
    // the same for all of a lane's pieces because pi keeps its parity
    const int jcA = slot ^ (((r8 >> 1) + 4 * (wave & 1)) & 7);
    const int pbase = m0 - Wd - 1;                       // linear input pixel of strip row 0 at ky = 0 (may be negative)
    unsigned int ro1[5], ro2[5];                         // per-piece byte offsets inside source 1 / source 2 (relative to pixel pbase)
    int jrow[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int pi = q < 4 ? wave + 8 * q : 32;
        jrow[q] = 8 * pi + r8;
        ro1[q] = (unsigned)(jrow[q] * p.ldx1 + jcA * 8) * 2u;
        ro2[q] = (unsigned)(jrow[q] * p.ldx2 + jcA * 8) * 2u;
    }
    const int totpix = p.M;
    // weights: rows n0 + 8 wave + r8 (+ 64), chunk swizzled by the row
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
        if (__builtin_amdgcn_ballot_w64(((vbits[0] & vbits[1]) >> t & 1u) == 0u) != 0) anyinv |= 1u << t;
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

    // ---- issue state ----------------------------------------------------------------------------------------------------------------
    int is_ky = 0, is_ch = 0;          // (ky, chunk) of the NEXT strip to issue
    int ib_ky = 0, ib_ch = 0, ib_kx = 0;   // (ky, chunk, kx) of the NEXT weight tile to issue
    auto fire_a = [&](auto sbc, auto qc) {   // piece q of the next strip -> strip buffer SB
        constexpr int SB = decltype(sbc)::value, Q = decltype(qc)::value;
        const int c0 = is_ch << 6;
        const bool second = c0 >= p.c1;
        const int P = pbase + is_ky * Wd + jrow[Q];
        const bool ok = (unsigned)P < (unsigned)totpix;
        const unsigned vo = ok ? (second ? ro2[Q] : ro1[Q]) : OOB;
        const int so = second ? (is_ky * Wd * p.ldx2 + (c0 - p.c1)) * 2 : (is_ky * Wd * p.ldx1 + c0) * 2;
        const int pi = Q < 4 ? wave + 8 * Q : 32;
        char* dst = smem + SB * STRIP + pi * 1024;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr4_t)dst, 16, vo, so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr4_t)dst, 16, vo, so, 0, 0);
    };
    auto next_strip = [&]() { if (++is_ch == nchunk) { is_ch = 0; ++is_ky; } };
    auto fire_b = [&](auto stc) {            // both pieces of the next weight tile -> ring stage ST
        constexpr int ST = decltype(stc)::value;
        const int so = ((ib_ky * 3 + ib_kx) * p.cin + (ib_ch << 6)) * 2;
        char* dst = smem + B0 + ST * BTILE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr4_t)dst, 16, wo[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr4_t)(dst + 8192), 16, wo[1], so, 0, 0);
        if (++ib_kx == 3) { ib_kx = 0; if (++ib_ch == nchunk) { ib_ch = 0; ++ib_ky; } }
    };
    auto wait_n = [&](int n) {   // at most n of this wave's DMA instructions still in flight
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };

    // one k-tile: strip buffer SB, tap column KX (= weight ring stage), tap index `tap`;  moreA: a next strip exists (its pieces are spread
    // over the three tiles of this strip), moreB: weight tile u+2 exists;  inflight: DMA instructions issued during the PREVIOUS tile
    // that are younger than everything this tile needs
    auto tile = [&](auto sbc, auto kxc, int tap, bool moreA, bool moreB, int inflight) {
        constexpr int SB = decltype(sbc)::value, KX = decltype(kxc)::value;
        wait_n(inflight);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sa = smem + SB * STRIP;
        const char* sb = smem + KX * BTILE;
        const bool mask = (anyinv >> tap) & 1u;
        const bool v0 = (vbits[0] >> tap) & 1u, v1 = (vbits[1] >> tap) & 1u;
        u32x4 a0[3], a1[3], b0[3], b1[3];
        auto rd = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            a0[sl] = *reinterpret_cast<const u32x4*>(sa + aoff[KX][g]);
            a1[sl] = *reinterpret_cast<const u32x4*>(sa + aoff[KX][g] + 32 * 128);
            b0[sl] = *reinterpret_cast<const u32x4*>(sb + boff[g]);
            b1[sl] = *reinterpret_cast<const u32x4*>(sb + boff[g] + 32 * 128);
            if (mask) {   // wave-uniform branch: zero padding of this tap for the rows that need it
                const u32x4 z = {0u, 0u, 0u, 0u};
                a0[sl] = v0 ? a0[sl] : z;
                a1[sl] = v1 ? a1[sl] : z;
            }
        };
        rd(IC4<0>{}, IC4<0>{});
        rd(IC4<1>{}, IC4<1>{});
        __builtin_amdgcn_sched_barrier(0);
        rd(IC4<2>{}, IC4<2>{});
        if (moreA) {   // the next strip goes to the OTHER buffer: free since the barrier above (its last readers finished the previous strip)
            if constexpr (KX == 0) { fire_a(IC4<SB ^ 1>{}, IC4<0>{}); fire_a(IC4<SB ^ 1>{}, IC4<1>{}); }
            if constexpr (KX == 1) { fire_a(IC4<SB ^ 1>{}, IC4<2>{}); fire_a(IC4<SB ^ 1>{}, IC4<3>{}); }
            if constexpr (KX == 2) { if (wave == 0) fire_a(IC4<SB ^ 1>{}, IC4<4>{}); next_strip(); }
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[0], a1[0], b0[0], b1[0]);
        __builtin_amdgcn_sched_barrier(0);
        rd(IC4<3>{}, IC4<0>{});
        if (moreB) fire_b(IC4<(KX + 2) % 3>{});   // stage of tile u+2; its last readers finished tile u-1 (barrier above)
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[1], a1[1], b0[1], b1[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[2], a1[2], b0[2], b1[2]);
        mma_group(a0[0], a1[0], b0[0], b1[0]);
        asm volatile("" ::: "memory");
    };

    // ---- prologue: strip 0 and weight tiles 0, 1 in flight ---------------------------------------------------------------------------
    fire_a(IC4<0>{}, IC4<0>{}); fire_a(IC4<0>{}, IC4<1>{}); fire_a(IC4<0>{}, IC4<2>{}); fire_a(IC4<0>{}, IC4<3>{});
    if (wave == 0) fire_a(IC4<0>{}, IC4<4>{});
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

    // A-piece instructions a wave issues in phase kx (the extra piece of wave 0 is always older than what the next tile may leave in flight)
    const int ntiles_k = 3 * nstrips;
    int u = 0;
    auto strip_body = [&](auto sbc, int s) {
        const int ky = s / nchunk;
        const bool moreA = s + 1 < nstrips;
        // inflight at the start of a tile = instructions issued in the previous tile after everything this tile needs:
        //   kx = 0: the strip AND weight tile u must have landed -> only weight tile u+1 (2) may be in flight
        //   kx = 1, 2: weight tile u -> the A pieces of the previous phase (2 if a next strip exists) + weight tile u+1 (2)
        const int nb1 = u + 1 < ntiles_k ? 2 : 0;
        tile(sbc, IC4<0>{}, ky * 3 + 0, moreA, u + 2 < ntiles_k, u == 0 ? 2 : nb1);
        ++u;
        tile(sbc, IC4<1>{}, ky * 3 + 1, moreA, u + 2 < ntiles_k, (moreA ? 2 : 0) + (u + 1 < ntiles_k ? 2 : 0));
        ++u;
        tile(sbc, IC4<2>{}, ky * 3 + 2, moreA, u + 2 < ntiles_k, (moreA ? 2 : 0) + (u + 1 < ntiles_k ? 2 : 0));
        ++u;
    };
    for (int s = 0; s < nstrips; s += 2) {
        strip_body(IC4<0>{}, s);
        if (s + 1 < nstrips) strip_body(IC4<1>{}, s + 1);
    }

    E2EFT_STAMP(2);
    igemm_epilogue<T, BM, BN, NW * 64>(p, smem, acc, wm, wn, l31, h, m0, n0, 0, 0);
    E2EFT_STAMP(4);
}

// >= 0: launched (0 or an error code); -1: not eligible, use igemm2
int launch_igemm_strip(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("E2EFT_STRIP"); return e ? atoi(e) : 1; }();
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
extern "C" int e2eft_debug_read_stamps4_rt(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps_rt), (size_t)nworkgroups * 2 * sizeof(long long));
}
#endif
