// igemm2.hip — implicit-GEMM conv / NT-GEMM, large-problem variant: 256x128 tile, 8 waves, LDS-DMA 3-stage ring.
//
// Same contract as igemm.hip (out = alpha*(A W^T + bias + rowadd) + residual; im2col gather with zero padding, fused
// nearest upsample, two-source channel concat).  What changes is how operands reach the MFMAs:
//   * global -> LDS by `global_load_lds_dwordx4` (LDS-DMA, 1 KiB per wave-instruction, no staging VGPRs); the im2col
//     gather is the per-lane SOURCE address, zero padding / ragged edges read a 16-byte zero block in device memory;
//   * the LDS image of a piece is lane-linear (8 rows x 128 B), so bank conflicts are removed by an XOR swizzle applied
//     to the source chunk index and to the fragment reads alike: slot = chunk ^ ((row >> 1) & 7) — 16 consecutive rows
//     of one chunk column then cover all 16 sixteen-byte slots of the 256-byte bank row (conflict-free ds_read_b128);
//   * 3-stage ring, prefetch distance 2 k-tiles: ~96 KiB of loads in flight per CU, counted `s_waitcnt vmcnt(6)`
//     (never 0 in the main loop) and ONE raw s_barrier per k-tile;
//   * 8 waves (4 x 2), each 64x64 = 2x2 MFMA 32x32 tiles: one workgroup per CU, two waves per SIMD.
#include "igemm.h"
#include <stdlib.h>

namespace e2eft {

// Two geometries of the same kernel (NW = waves per workgroup, wave tile always 64x64, BN = 128):
//   NW = 8: 256x128 tile, 3-stage ring (prefetch distance 2), one workgroup per CU;
//   NW = 4: 128x128 tile, 2 stages (prefetch distance 1), 66 KiB of LDS -> TWO workgroups per CU whose barriers drift
//           apart, so one workgroup's wait/barrier bubbles are filled by the other's MFMAs.
constexpr int BN2 = 128;
constexpr int B_STAGE = BN2 * 128;
template <int NW> struct Geo {
    static constexpr int BM = NW * 32;
    static constexpr int A_STAGE = BM * 128;
    static constexpr int STAGE = A_STAGE + B_STAGE;
    static constexpr int NSTAGE = NW == 8 ? 3 : 2;
    static constexpr int BPIECES = 16 / NW;          // B pieces (8 rows each) per wave
    static constexpr int NPIECES = 4 + BPIECES;      // DMA instructions per wave per k-tile
    static constexpr int EPI_BYTES = BM * (BN2 + 4) * 4;
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

// FAST: every k-tile lies inside one filter tap and one source tensor (cin % BK == 0, c1 % BK == 0; GEMM: K % BK == 0).
// Then tap / channel / source are wave-uniform scalars, a row's pixel offset is recomputed only when the tap changes,
// and the per-DMA address work is ONE v_add: loads go through `buffer_load ... lds` with a 32-bit byte offset relative
// to a per-workgroup base, invalid rows / padding taps carry an out-of-range offset and the buffer bounds check writes
// the zeros.  !FAST keeps the fully general per-lane address path (global_load_lds from a pointer or the zero block).
constexpr unsigned int OOB_SENTINEL = 0xF0000000u;
constexpr unsigned int SRD_RECORDS = 0xE0000000u;

template <typename T, int MODE, bool FAST, int NW>
__global__ __launch_bounds__(NW * 64) void igemm2_kernel(const IgemmParams p) {
    using G = Geo<NW>;
    static constexpr int BM2 = G::BM, A_STAGE = G::A_STAGE, STAGE2 = G::STAGE, NSTAGE = G::NSTAGE, BPIECES = G::BPIECES, NPIECES = G::NPIECES;
    static constexpr int RSTEP = 8 * NW;   // row distance between a wave's consecutive pieces
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
    const T* __restrict__ W = (const T*)p.w + zo * p.sw_o + zi * p.sw_i;
    const T* zsrc = (const T*)g_zero16;

    // ---- loader mapping: a wave-instruction fills one 1-KiB piece = 8 rows x 8 chunks; lane -> (row lane>>3, slot lane&7).
    // A pieces: rows 8*wave + 64*i (i < 4); B pieces: rows 8*wave + 64*i (i < 2).  The lane's LOGICAL k-chunk is
    // slot ^ ((row >> 1) & 7), identical for all of its pieces.
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
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / p.wout, ox = rem - oy * p.wout;
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

    // incremental k decomposition of this thread's chunk: k = kt*BK + jc*EPC = (ky*kw + kx)*cin + c
    int k_cur = jc * EPC;
    int c_cur = 0, kx_cur = 0, ky_cur = 0;
    if (MODE == 1) {
        const int kpos = k_cur / p.cin;
        c_cur = k_cur - kpos * p.cin;
        ky_cur = kpos / p.kw;
        kx_cur = kpos - ky_cur * p.kw;
    }

    auto issue = [&](int stage) {   // issue the DMA of the NEXT k-tile (tiles are issued in order 0,1,2,...)
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
                const bool ok = kok && a_ok[i] && (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
                int sy = iy, sx = ix;
                if (p.hl != p.hin) sy = min((int)floorf(iy * p.up_sh), p.hin - 1);
                if (p.wl != p.win) sx = min((int)floorf(ix * p.up_sw), p.win - 1);
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
        // advance to the next k-tile
        k_cur += BK;
        if (MODE == 1) {
            c_cur += BK;
            while (c_cur >= p.cin) {
                c_cur -= p.cin;
                if (++kx_cur == p.kw) { kx_cur = 0; ++ky_cur; }
            }
        }
    };

    // ---------------- FAST path state ----------------
    unsigned int off1[4] = {0, 0, 0, 0}, off2[4] = {0, 0, 0, 0}, woff[4];
    int brel[4] = {0, 0, 0, 0};
    int tile_c = 0, tap = 0;
    unsigned int kbytes = 0;
    __amdgpu_buffer_rsrc_t rs1, rs2, rsw;
    if constexpr (FAST) {
        const T* b1;
        const T* b2 = X2;
        if (MODE == 0) {
            b1 = X1 + (long)m0 * p.ldx1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                off1[i] = a_ok[i] ? (unsigned)((lrow + RSTEP * i) * p.ldx1 + jc * EPC) * (unsigned)sizeof(T) : OOB_SENTINEL;
        } else {
            const int hw = p.hout * p.wout;
            const int b0 = m0 / hw;                       // first image touched by this tile (uniform)
            b1 = X1 + (long)b0 * p.hin * p.win * p.ldx1;
            if (X2) b2 = X2 + (long)b0 * p.hin * p.win * p.ldx2;
#pragma unroll
            for (int i = 0; i < 4; ++i) brel[i] = (int)a_base[i] - b0;
        }
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, SRD_RECORDS, 0x00020000);
        rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(b2 ? b2 : b1), 0, SRD_RECORDS, 0x00020000);
        rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, SRD_RECORDS, 0x00020000);
#pragma unroll
        for (int i = 0; i < BPIECES; ++i)
            woff[i] = w_ok[i] ? (unsigned)((lrow + RSTEP * i) * p.ldw + jc * EPC) * (unsigned)sizeof(T) : OOB_SENTINEL;
    }
    auto retap = [&]() {   // per-row pixel offsets of the current filter tap (uniform branch, once per tap)
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
            int sy = iy, sx = ix;
            if (p.hl != p.hin) sy = min((int)floorf(iy * p.up_sh), p.hin - 1);
            if (p.wl != p.win) sx = min((int)floorf(ix * p.up_sw), p.win - 1);
            const unsigned pix = (unsigned)((brel[i] * p.hin + sy) * p.win + sx);
            off1[i] = ok ? (pix * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB_SENTINEL;
            off2[i] = ok ? (pix * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB_SENTINEL;
        }
    };
    // DMA of one k-tile = 6 pieces per wave (A0..A3, B0, B1).  prep_fast() computes the six 32-bit offsets (a v_add
    // each); fire_fast(stage, piece) issues one piece — the main loop spreads them between the MFMA groups so that the
    // ~60-180 cycle issue cost of an LDS-DMA instruction overlaps this wave's own MFMA execution.
    unsigned int voff[8];   // NPIECES <= 8 (literal size: a template-dependent bound breaks the host-side stub instantiation)
    bool use2 = false;
    auto prep_fast = [&]() {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) voff[i] = off1[i] + kbytes;
        } else {
            if (tile_c == 0) retap();
            use2 = tile_c >= p.c1;
            const unsigned cb = (unsigned)(use2 ? tile_c - p.c1 : tile_c) * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < 4; ++i) voff[i] = (use2 ? off2[i] : off1[i]) + cb;
            tile_c += BK;
            if (tile_c >= p.cin) { tile_c = 0; ++tap; }
        }
#pragma unroll
        for (int i = 0; i < BPIECES; ++i) voff[4 + i] = woff[i] + kbytes;
        kbytes += 128;
    };
    auto fire_fast = [&](int stage, int piece) {
        char* sa = smem + stage * STAGE2 + wave * 1024;
        if (piece < 4) {
            if (MODE == 1 && use2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr_t)(sa + piece * (RSTEP * 128)), 16, voff[piece], 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(sa + piece * (RSTEP * 128)), 16, voff[piece], 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(sa + A_STAGE + (piece - 4) * (RSTEP * 128)), 16, voff[piece], 0, 0, 0);
        }
    };
    auto issue_fast = [&](int stage) {
        prep_fast();
#pragma unroll
        for (int q = 0; q < NPIECES; ++q) fire_fast(stage, q);
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l31 >> 1) & 7;   // read-side swizzle: rows wm*64 + i*32 + l31 -> ((row >> 1) & 7) == (l31 >> 1) & 7
    // fragment byte offsets inside a stage (per lane, fixed for the whole kernel); second row-subtile = +32*128 immediate
    int aoff[4], boff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int chunk = sizeof(T) == 2 ? (g * 2 + h) : (h * 4 + g);
        aoff[g] = (wm * 64 + l31) * 128 + ((chunk ^ sw) * 16);
        boff[g] = A_STAGE + (wn * 64 + l31) * 128 + ((chunk ^ sw) * 16);
    }
    // One k-tile of MFMAs out of LDS stage `stage`; when `dma` is set the six DMA pieces of a later k-tile (offsets already
    // in voff[]) are issued between the MFMA groups: pieces (0,1) (2,3) (4,5) after the fragment reads of groups 0,1,2.
    auto compute = [&](int stage, bool dma, int dstage) {
        const char* sbase = smem + stage * STAGE2;
        if constexpr (sizeof(T) == 2) {
            // fragments are fetched TWO k-groups ahead of the MFMAs that consume them (3 register sets), so that the
            // ~130-cycle ds_read_b128 latency is always covered by a full group of 4 MFMAs (128 pipe cycles)
            u32x4 a0[3], a1[3], b0[3], b1[3];
            auto rd = [&](int g, int slot) {
                a0[slot] = *reinterpret_cast<const u32x4*>(sbase + aoff[g]);
                a1[slot] = *reinterpret_cast<const u32x4*>(sbase + aoff[g] + 32 * 128);
                b0[slot] = *reinterpret_cast<const u32x4*>(sbase + boff[g]);
                b1[slot] = *reinterpret_cast<const u32x4*>(sbase + boff[g] + 32 * 128);
            };
            rd(0, 0);
            rd(1, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks % 3;
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 4) rd(ks + 2, (ks + 2) % 3);
                if constexpr (FAST) {
                    if (dma) {
#pragma unroll
                        for (int q = ks * 2; q < (ks == 3 ? NPIECES : ks * 2 + 2); ++q) fire_fast(dstage, q);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (p.dbg & 1) {   // experiment: keep the fragment reads alive, skip the matrix pipe
                    asm volatile("" ::"v"(a0[c]), "v"(a1[c]), "v"(b0[c]), "v"(b1[c]));
                    continue;
                }
                acc[0][0] = Mma2<T>::run(a0[c], b0[c], acc[0][0]);
                acc[0][1] = Mma2<T>::run(a0[c], b1[c], acc[0][1]);
                acc[1][0] = Mma2<T>::run(a1[c], b0[c], acc[1][0]);
                acc[1][1] = Mma2<T>::run(a1[c], b1[c], acc[1][1]);
            }
        } else {
            // fp32: MFMA 32x32x2 step s pairs k-slot s of the lower half (lanes 0-31, chunks 0-3) with k-slot s of the
            // upper half (lanes 32-63, chunks 4-7); the same pairing is used for A and W.
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                floatx4 a0 = *reinterpret_cast<const floatx4*>(sbase + aoff[qd]);
                floatx4 a1 = *reinterpret_cast<const floatx4*>(sbase + aoff[qd] + 32 * 128);
                floatx4 b0 = *reinterpret_cast<const floatx4*>(sbase + boff[qd]);
                floatx4 b1 = *reinterpret_cast<const floatx4*>(sbase + boff[qd] + 32 * 128);
                if constexpr (FAST) {
                    if (dma) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = qd * 2; q < (qd == 3 ? NPIECES : qd * 2 + 2); ++q) fire_fast(dstage, q);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
                }
            }
        }
    };

    // ---- main loop: ring of NSTAGE stages, prefetch distance NSTAGE-1 k-tiles ----
    const int nk = (p.K + BK - 1) / BK;
    auto issue_any = [&](int stage) { if constexpr (FAST) issue_fast(stage); else issue(stage); };
    issue_any(0);
    if (NSTAGE == 3 && nk > 1) issue_any(1);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's DMA of tile kt has landed once only the younger tile's NPIECES loads are still outstanding
        if (NSTAGE == 3 && kt + 1 < nk) {
            if constexpr (NPIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt are in LDS; every wave is done reading tile kt-1
        asm volatile("" ::: "memory");
        const bool more = kt + (NSTAGE - 1) < nk && !(p.dbg & 2);
        const int dst = (kt + NSTAGE - 1) % NSTAGE;   // the stage tile kt-1 lived in
        if constexpr (FAST) {
            if (more) prep_fast();                    // offsets only; the DMA goes out inside compute()
            compute(kt % NSTAGE, more, dst);
        } else {
            if (more) issue(dst);
            compute(kt % NSTAGE, false, 0);
        }
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: LDS-staged, vectorised (igemm.h) ----
    igemm_epilogue<T, BM2, BN2, NW * 64>(p, smem, acc, wm, wn, l31, h, m0, n0, zo, zi);
}

template <typename T, int MODE, int NW> static int launch2(IgemmParams& p, int nz, hipStream_t s) {
    static const int dbg = [] { const char* e = getenv("E2EFT_IGEMM_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    p.mtiles = cdiv(p.M, Geo<NW>::BM);
    p.ntiles = cdiv(p.N, BN2);
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
    static const bool nofast = getenv("E2EFT_IGEMM_NOFAST") != nullptr;
    if (fast && !nofast) hipLaunchKernelGGL((igemm2_kernel<T, MODE, true, NW>), grid, dim3(NW * 64), 0, s, p);
    else hipLaunchKernelGGL((igemm2_kernel<T, MODE, false, NW>), grid, dim3(NW * 64), 0, s, p);
    return check_launch("igemm2");
}

template <typename T> static int launch2_t(int mode, IgemmParams& p, int nz, hipStream_t s) {
    static const int forced_nw = [] { const char* e = getenv("E2EFT_IGEMM2_NW"); return e ? atoi(e) : 0; }();
    // two 128-row workgroups per CU fill the machine better on mid-size problems; 256-row tiles halve the weight traffic on big ones
    const long blocks256 = (long)cdiv(p.M, 256) * cdiv(p.N, BN2) * nz;
    const int nw = forced_nw ? forced_nw : (blocks256 < 2048 ? 4 : 8);
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
