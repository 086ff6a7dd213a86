// igemm5.hip — PERSISTENT variant of the LDS-DMA implicit-GEMM kernel (igemm2.hip): one 8-wave workgroup per CU walks a sequence
// of 256x128 output tiles (gfx950, fp16 / bf16, the FAST operand path only).
//
// Why: igemm2 pays ~18-20k shader cycles (~10 us) of fixed cost per tile — workgroup launch, index arithmetic, a cold first k-tile
// (the first LDS-DMA of a fresh workgroup has nothing in flight in front of it), accumulators -> LDS (bound by ds_write_b32 at
// 64 B/clk), two workgroup barriers, row passes.  At K = 1152 (conv 128->128, the most expensive shape of the path) that is 37 % of
// the tile.  Here
//   * the k-tile stream never stops at a tile boundary: while the last two k-tiles of tile t run, the LDS-DMA of the first two
//     k-tiles of tile t+1 is already in flight (the ring keeps rotating; the address state of the loader is switched to the next
//     tile after the last piece of the current one has been issued);
//   * the epilogue has no workgroup-wide staging: every wave sends its own 64x64 accumulator block through a private 2 x 2 KB window
//     of the ring stage that the last k-tile just released — eight 8-row slices, slice s+1 written while slice s is read back
//     row-wise — so LDS writes, row math and 16-byte stores of different waves overlap instead of running in barrier-separated phases;
//   * the barrier that opens k-tile g+1 sits in front of the LAST TWO MFMA groups of k-tile g (every LDS read of k-tile g has been
//     requested and has returned by then): the skew of the eight waves runs under eight MFMAs — 1.45 k cycles per k-tile at K = 1152 and
//     1.3 k at K = 4608 instead of 1.8 k with the barrier at the top of a rolled loop (igemm2's unrolled loop gets this placement from the
//     compiler's scheduler, by accident of its freedom to move MFMAs across an asm barrier);
//   * the loader's per-row index arithmetic for the NEXT tile is done by one thread per output row at the top of the current tile (LDS
//     row table), the k-tile that switches the loader over only reads four table entries per lane;
//   * residual / bias / rowadd vectors of the first slices are requested at the top of the last-but-one k-tile, BEFORE the next tile's
//     first pieces (memory returns in order: their wait must not cover the pieces); every k-tile kind issues exactly six pieces, so all
//     waits in the tile stream are counted (`vmcnt(6)`, `vmcnt(6 + requests)`), the compiler's own wait for the operands is `vmcnt(12)`,
//     and the only drain per tile is one `vmcnt(0)` in the middle of the epilogue (as the builtin, so that the waitcnt pass knows);
//   * the first k-tile of a tile multiplies into the constant 0 (no accumulator clears) and carries the tile-entry arithmetic;
//   * launches WITHOUT a residual (template flag RES = false) use the packed epilogue: alpha / bias / row vector / statistics in the accumulator
//     layout (a lane owns two columns), values rounded there, two rows per dword through LDS — half the staging bytes, readers only store.
// Contract, operand layouts, ring, swizzle: as igemm2.hip.  Eligibility is decided on the host (launch_igemm_persistent returns -1 and the
// caller falls through to igemm2): 16-bit FAST path, M a multiple of 256, at least three k-tiles, the vector epilogue, at least two tiles
// per workgroup.  e2eft_set_option(E2EFT_OPT_PERSISTENT, 0) disables the variant (A/B runs), E2EFT_OPT_PERSISTENT_GRID shrinks the grid (tests).
//
// GroupNorm statistics (p.gn_partial): per column shifted sums about a per-wave pivot (the wave's first output row), a 14-exchange
// reduce-scatter over the eight row-lanes of a chunk, per-wave deposits in LDS merged over the four row-waves of a column with Chan's
// formula after the barrier that opens the next tile — the same (count, mean, M2) triples per 256-row slab that igemm2 emits.
//
// Measured (profiles/r02b_igemm5_phase_clocks_and_ab.txt, r02d_*): 34.5 k cycles per 256x128x1152 tile against 41.7 k in igemm2; at the
// 1400 W package power cap that is +6 % wall time on conv 128->128 @768^2, +14 % with a residual, +1-3 % at K >= 2304.  Tried and not
// kept: a 4th fragment slot requested across the barrier, row-table reads one MFMA group early, statistics merge under the next tile's
// MFMAs, a register-only (operand-swapped) epilogue for GEMM tiles, tap offsets after the fragment requests.
#include "igemm.h"
#include <atomic>
#include <type_traits>

namespace e2eft {

namespace pers {
constexpr int BM = 256, BN = 128, NW = 8;
constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128, STAGE = A_STAGE + B_STAGE;
constexpr int RING = 3 * STAGE;                  // 147,456 B
constexpr int DEP = NW * 64 * 3 * 4;             // per-wave GroupNorm deposits: (sum, sum of squares, pivot) per column
constexpr int ROWTAB = BM * 16;                  // (image, iy0, ix0, pixel) of the 256 output rows of the loader's NEXT tile, one thread per row
constexpr int LDS = RING + DEP + ROWTAB;
constexpr unsigned int OOB = 0xF0000000u;        // byte offset beyond RECORDS: the buffer load returns zeros
constexpr unsigned int RECORDS = 0xE0000000u;
}  // namespace pers

template <typename T> struct Mma5;
template <> struct Mma5<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma5<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

#ifdef E2EFT_STAMPS
// instrumented build: thread 0 of every workgroup records, for its first 32 tiles, the shader clock at: k-loop entry, after the steady
// k-tiles, after the last k-tile (= epilogue entry), epilogue exit (scripts/stamp5_bench.py)
static __device__ long long g_stamps5[512 * 32 * 8];
#define STAMP5(t, i) do { if (threadIdx.x == 0 && (t) < 32 && blockIdx.x < 512) g_stamps5[(blockIdx.x * 32 + (t)) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP5(t, i) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lptr5_t;
template <int V> using IC5 = std::integral_constant<int, V>;

__device__ __forceinline__ int fast_div5(int n, int d) {   // as igemm2.hip: float estimate + one correction (quotients below 2^22)
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// RES: the launch has a residual operand (fp32 sliced epilogue); otherwise the packed epilogue — one of the two per instantiation, for the register budget
// F32O (round 6, csrc/f32split.hip): an fp32 convolution / GEMM whose products are formed on the f16 matrix pipe.  A holds the two f16 planes [x0 | x1] of an fp32 operand
// side by side (split_c columns each), the weights arrive as [w0 | w1 | w0] per tap, K runs over the three blocks (x0, w0), (x0, w1), (x1, w0): the A offset walks
// split_c columns, starts over, then walks on into the second plane.  Bias / residual / output are fp32 (epilogue_f32 of the .inc).
template <typename T, int MODE, bool RES, bool F32O = false>
__global__ __launch_bounds__(512) void igemm5_kernel(const IgemmParams p, const int total_tiles) {
    using namespace pers;
    static_assert(!F32O || (RES && std::is_same<T, f16>::value), "fp32 output: the fp32-window epilogue, f16 split operands");
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    constexpr int EPC = 8;                 // elements per 16 bytes
    constexpr int BK = 64;
    constexpr int RSTEP = 8 * NW;          // row distance between a wave's consecutive DMA pieces
    typedef float f2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblk = p.mtiles * p.ntiles;
    const int nk = p.K / BK;               // host: K % BK == 0, nk >= 3

    // ---- this workgroup's tile sequence: the tile range is cut into eight contiguous chunks (one per XCD, so that neighbouring
    // tiles — same A rows, other N tile; same weights — share an L2), the workgroups of an XCD walk their chunk with stride nslots
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int q8 = total_tiles >> 3, r8 = total_tiles & 7;
    const int cbeg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cend = cbeg + (xcd < r8 ? q8 + 1 : q8);
    int u_dma = cbeg + slot;               // tile the LOADER works on (it runs up to two k-tiles ahead of the MFMAs)
    if (u_dma >= cend) return;

    // ---- loader mapping (igemm2.hip): one wave-instruction fills a 1-KiB piece = 8 rows x 8 sixteen-byte chunks
    const int lrow = 8 * wave + (lane >> 3);
    const int jc = (lane & 7) ^ ((lrow >> 1) & 7);
    const bool plain_taps = MODE == 1 && p.zins <= 1 && p.hl == p.hin && p.wl == p.win;
    const bool half_res = MODE == 1 && (p.zins == 2 || (p.zins <= 1 && p.hl == 2 * p.hin && p.wl == 2 * p.win));

    unsigned int off1[4], off2[4];          // per-row byte offsets of the current tap in x1 / x2 (OOB if padded)
    unsigned int cur_a[4], cur_b[2];        // byte offsets of the NEXT k-tile to issue (advanced by 128 B per k-tile)
    unsigned int base1[4], base2[4];
    int a_iy0[4], a_ix0[4], brel[4];
    int tile_c = 0, ky = 0, kx = 0;
    int d_m0 = 0, d_n0 = 0, d_zo = 0, d_zi = 0;   // coordinates of the loader's tile
    bool dma_done = false;
    __amdgpu_buffer_rsrc_t rs1, rs2, rsw, rsa;

    // Address state of the loader's next tile, in three steps so that the heavy part is not paid by every lane: (1) tile_coords —
    // uniform; (2) fill_rowtab — ONE thread per output row (waves 0-3) splits the row index into (image, y, x) and leaves
    // (image - b0, iy0, ix0, tap-(0,0) pixel) in LDS; (3) after a barrier finish_setup — every loader lane picks up its four rows
    // (one ds_read_b128 each) and derives byte offsets and descriptors.  (Doing the split per loader lane cost ~4k cycles per tile:
    // 8 float-reciprocal divisions x 8 waves beside 4 integer divisions of uniform values.)
    int d_b0 = 0;
    auto tile_coords = [&](const int u) {
        int z = 0, lid = u;
        if (nblk != total_tiles) { z = fast_div5(u, nblk); lid = u - z * nblk; }
        int mt = lid, nt = 0;
        if (p.ntiles > 1) { mt = fast_div5(lid, p.ntiles); nt = lid - mt * p.ntiles; }
        d_m0 = mt * BM; d_n0 = nt * BN;
        d_zo = 0; d_zi = z;
        if (p.nzi > 1 && z > 0) { d_zo = fast_div5(z, p.nzi); d_zi = z - d_zo * p.nzi; }
        else if (p.nzi == 1) { d_zo = z; d_zi = 0; }
        if (MODE == 1) d_b0 = fast_div5(d_m0, p.hout * p.wout);   // first image touched by this tile
    };
    // (the row table is written / read with inline-asm DS instructions: in front of LDS accesses it can see, the compiler drains
    // vmcnt — it has to assume that a pending LDS-DMA may alias them — and here pieces / stores are in flight by design)
    auto fill_rowtab = [&]() {
        if (MODE == 1 && tid < BM) {
            const int hw = p.hout * p.wout;
            const int m = d_m0 + tid;
            const int b = fast_div5(m, hw);
            const int rem = m - b * hw;
            const int oy = fast_div5(rem, p.wout), ox = rem - oy * p.wout;
            int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
            if (F32O && m >= p.M) iy0 = -0x4000;      // a row beyond the problem (ragged last tile, F32O only): every tap is padding
            const u32x4 v = {(unsigned)(b - d_b0), (unsigned)iy0, (unsigned)ix0, (unsigned)(((b - d_b0) * p.hin + iy0) * p.win + ix0)};
            const unsigned addr = (unsigned)(RING + DEP) + (unsigned)tid * 16u;
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
        }
    };
    auto finish_setup = [&]() {
        const T* X1 = (const T*)p.x1 + d_zo * p.sa_o + d_zi * p.sa_i;
        const T* W = (const T*)p.w + d_zo * p.sw_o + d_zi * p.sw_i;
        const T* b1;
        const T* b2 = (const T*)p.x2;
        if (MODE == 0) {
            b1 = X1 + (long)d_m0 * p.ldx1;
#pragma unroll
            for (int i = 0; i < 4; ++i) cur_a[i] = ((F32O && d_m0 + lrow + RSTEP * i >= p.M) ? OOB : (unsigned)((lrow + RSTEP * i) * p.ldx1 + jc * EPC) * (unsigned)sizeof(T)) - 128u;
        } else {
            b1 = X1 + (long)d_b0 * p.hin * p.win * p.ldx1;
            if (b2) b2 += (long)d_b0 * p.hin * p.win * p.ldx2;
            u32x4 rt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned addr = (unsigned)(RING + DEP) + (unsigned)(lrow + RSTEP * i) * 16u;
                asm volatile("ds_read_b128 %0, %1" : "=v"(rt[i]) : "v"(addr) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rt[0]), "+v"(rt[1]), "+v"(rt[2]), "+v"(rt[3]) :: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                brel[i] = (int)rt[i][0]; a_iy0[i] = (int)rt[i][1]; a_ix0[i] = (int)rt[i][2];
                cur_a[i] = 0;
                if (plain_taps) {   // pixel index wraps for padded taps; only used when valid
                    base1[i] = (rt[i][3] * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T);
                    base2[i] = (rt[i][3] * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T);
                }
            }
        }
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, RECORDS, 0x00020000);
        rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(b2 ? b2 : b1), 0, RECORDS, 0x00020000);
        rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)d_n0 * p.ldw), 0, RECORDS, 0x00020000);
        rsa = rs1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = d_n0 + lrow + RSTEP * i;
            cur_b[i] = (n < p.N ? (unsigned)((lrow + RSTEP * i) * p.ldw + jc * EPC) * (unsigned)sizeof(T) : OOB) - 128u;
        }
        tile_c = 0; ky = 0; kx = 0;
    };
    auto retap = [&]() {   // per-row pixel offsets of the current filter tap (uniform branch, once per tap)
        if (plain_taps) {
            const unsigned d1 = (unsigned)((ky * p.win + kx) * p.ldx1) * (unsigned)sizeof(T);
            const unsigned d2 = (unsigned)((ky * p.win + kx) * p.ldx2) * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(a_iy0[i] + ky) < (unsigned)p.hl && (unsigned)(a_ix0[i] + kx) < (unsigned)p.wl;
                off1[i] = ok ? base1[i] + d1 : OOB;
                off2[i] = ok ? base2[i] + d2 : OOB;
            }
            return;
        }
        if (half_res) {   // exact 2x nearest upsample (every fused upsample of the path) / stride-2 zero insertion: shifts instead of float / integer division
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                bool ok = (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
                if (p.zins == 2) ok = ok && (((iy | ix) & 1) == 0);
                const unsigned pix = (unsigned)((brel[i] * p.hin + (iy >> 1)) * p.win + (ix >> 1));
                off1[i] = ok ? (pix * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB;
                off2[i] = ok ? (pix * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            bool ok = (unsigned)iy < (unsigned)p.hl && (unsigned)ix < (unsigned)p.wl;
            int sy = iy, sx = ix;
            if (p.zins > 1) {
                ok = ok && (iy % p.zins == 0) && (ix % p.zins == 0);
                sy = iy / p.zins; sx = ix / p.zins;
            } else {
                if (p.hl != p.hin) sy = min((int)floorf(iy * p.up_sh), p.hin - 1);
                if (p.wl != p.win) sx = min((int)floorf(ix * p.up_sw), p.win - 1);
            }
            const unsigned pix = (unsigned)((brel[i] * p.hin + sy) * p.win + sx);
            off1[i] = ok ? (pix * (unsigned)p.ldx1 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB;
            off2[i] = ok ? (pix * (unsigned)p.ldx2 + (unsigned)(jc * EPC)) * (unsigned)sizeof(T) : OOB;
        }
    };
#ifdef E2EFT_STAMPS
    bool fire_a = true;
#endif
    auto advance = [&]() {   // offsets / descriptor of the next k-tile to issue (k-tiles of a tile are issued in order)
#ifdef E2EFT_STAMPS
        fire_a = !(p.debug_flags & 1) && !((p.debug_flags & 2) && kx != 0) && !((p.debug_flags & 4) && (kx != 0 || ky != 0));
#endif
        if (MODE == 0) {
            if constexpr (F32O) {   // block (x0, w1) reads the first plane again; (x1, w0) simply walks on
                const unsigned back = tile_c == p.split_c ? (unsigned)p.split_c * 2u : 0u;
                tile_c += BK;
#pragma unroll
                for (int i = 0; i < 4; ++i) cur_a[i] += 128u - back;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur_a[i] += 128u;
            }
        } else {
            if (F32O && tile_c == p.split_c) {   // block (x0, w1): the first plane again; (x1, w0) then walks on into the second plane
#pragma unroll
                for (int i = 0; i < 4; ++i) cur_a[i] = off1[i];
            } else if (tile_c == 0) {
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
        cur_b[0] += 128u;
        cur_b[1] += 128u;
    };
    // ring: three stage offsets that rotate once per k-tile (s_cur: consumed now, s_nxt: consumed next, s_dst: receives k-tile + 2)
    int s_cur = 0, s_nxt = STAGE, s_dst = 2 * STAGE;
    auto rotate = [&]() { const int t = s_cur; s_cur = s_nxt; s_nxt = s_dst; s_dst = t; };
    auto fire = [&](const int stage, auto piece_c) {
        constexpr int Q = decltype(piece_c)::value;
        char* sa = smem + stage + wave * 1024;
#ifdef E2EFT_STAMPS
        if (Q < 4 && !fire_a) return;
#endif
        if constexpr (Q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr5_t)(sa + Q * (RSTEP * 128)), 16, cur_a[Q], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr5_t)(sa + A_STAGE + (Q - 4) * (RSTEP * 128)), 16, cur_b[Q - 4], 0, 0, 0);
    };
    auto fire_all = [&](const int stage) {
        fire(stage, IC5<0>{}); fire(stage, IC5<1>{}); fire(stage, IC5<2>{}); fire(stage, IC5<3>{});
        fire(stage, IC5<4>{}); fire(stage, IC5<5>{});
    };

    int tseq = 0;
    (void)tseq;
    floatx16 acc[2][2];
    // fragment byte offsets inside a stage (read-side swizzle as igemm2.hip)
    const int sw = (l31 >> 1) & 7;
    int aofs[4], bofs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int chunk = g * 2 + h;
        aofs[g] = (wm * 64 + l31) * 128 + ((chunk ^ sw) * 16);
        bofs[g] = A_STAGE + (wn * 64 + l31) * 128 + ((chunk ^ sw) * 16);
    }
    auto mma_group = [&](const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        acc[0][0] = Mma5<T>::run(a0, b0, acc[0][0]);
        acc[0][1] = Mma5<T>::run(a0, b1, acc[0][1]);
        acc[1][0] = Mma5<T>::run(a1, b0, acc[1][0]);
        acc[1][1] = Mma5<T>::run(a1, b1, acc[1][1]);
    };

    // ---- epilogue operands requested ahead of their use
    const int er = lane >> 3, ec = lane & 7;          // row inside an 8-row slice, 8-column chunk inside the wave's 64 columns
    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ rowadd = (const T*)p.rowadd;
    constexpr bool has_res = RES && !F32O;            // (F32O requests its fp32 bias / residual inside epilogue_f32)
    const bool has_ra = rowadd != nullptr, stats = p.gn_partial != nullptr;
    Vec16<T> pre_res[4], pre_bias, pre_ra;            // residual rows of slices 0-3, bias and rowadd chunk of this lane
    T col_bias[2], col_ra[2];                          // packed epilogue (no residual): bias / rowadd of the two COLUMNS this lane owns in the accumulator layout
    long c_orow = 0, c_rrow = 0;                       // element offsets of this lane's first output / residual row in the MFMA-side tile
    int c_m0 = 0, c_n0 = 0, c_img = 0, c_ncl = 0;
    int c_mb = 0, c_seg0 = 0;                          // first GEMM row of this wave's 64-row block and (p.out_seg > 0) the output segment it starts in
    bool c_colok = false;

    // one k-tile.  KIND 0: steady state; 1: the loader has issued every k-tile of the current tile — at the top the epilogue's first
    // operands are requested (BEFORE the next tile's pieces: memory returns in order, and their wait must not cover the pieces) and the
    // loader's address state moves to the workgroup's next tile (row table filled at the top of the current tile), then the k-tile
    // runs like any other with the next tile's first k-tile as its interleaved issue; 2: last k-tile.  Every kind issues six pieces
    // (out-of-range offsets = zeros into a stage nobody reads when the workgroup has no further tile): with an unconditional VMEM
    // sequence the waits can count — the compiler's own s_waitcnt for the epilogue operands is vmcnt(12), not a drain.
    // The synchronisation that opens k-tile g+1 (counted DMA wait + barrier) is issued in front of the LAST TWO MFMA groups of k-tile g —
    // every LDS read of k-tile g has been requested by then, and the barrier skew of the eight waves disappears under 8 MFMAs (igemm2's
    // unrolled loop gets the same placement from the compiler's scheduler; a rolled loop has to spell it out).  nextwait: 0 none,
    // 2 vmcnt(6), 3 vmcnt(6 + NPRE) with NPRE = the operand requests issued at the top of a KIND-1 k-tile.
    const int npre = F32O ? 0 : has_res ? 4 + (bias ? 1 : 0) + (has_ra ? 1 : 0) : 2 * ((bias ? 1 : 0) + (has_ra ? 1 : 0));
    bool nxt = false;                                  // the workgroup has a tile after the current one (set at the top of a tile)
    long zoff_o = 0;
    auto enter_tile = [&]() {   // the MFMA side enters the tile the loader is (still) on; then the loader's coordinates move to the workgroup's next tile
        c_m0 = d_m0; c_n0 = d_n0;
        zoff_o = d_zo * p.so_o + d_zi * p.so_i;
        const long zoff_r = d_zo * p.sr_o + d_zi * p.sr_i;
        c_colok = c_n0 + wn * 64 + ec * 8 < p.N;
        c_ncl = c_colok ? c_n0 + wn * 64 + ec * 8 : c_n0;
        c_img = has_ra ? fast_div5(c_m0, p.rows_per_img) : 0;
        const int m = c_m0 + wm * 64 + er;          // slice s adds 8 * s rows
        c_mb = c_m0 + wm * 64;
        c_seg0 = p.out_seg > 0 ? fast_div5(c_mb, p.out_seg) : 0;      // (segmented output: er < 8 <= out_seg / 2 keeps row `er` in the block's first segment)
        c_orow = ((long)m + (long)c_seg0 * p.out_seg) * p.ldo + c_ncl;
        c_rrow = zoff_r + (long)m * p.ldr + c_ncl;
        nxt = u_dma + nslots < cend;
        if (nxt) tile_coords(u_dma + nslots);
    };
    auto ktile = [&](auto kind_c, const int nextwait) {
        constexpr int KIND = decltype(kind_c)::value;
        if (KIND == 0 || KIND == 3 || (KIND == 2 && !dma_done)) advance();
        const int sc = s_cur, sd = s_dst;
        u32x4 a0[3], a1[3], b0[3], b1[3];
        auto rd = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, slot = decltype(slotc)::value;
            a0[slot] = *reinterpret_cast<const u32x4*>(smem + sc + aofs[g]);
            a1[slot] = *reinterpret_cast<const u32x4*>(smem + sc + aofs[g] + 32 * 128);
            b0[slot] = *reinterpret_cast<const u32x4*>(smem + sc + bofs[g]);
            b1[slot] = *reinterpret_cast<const u32x4*>(smem + sc + bofs[g] + 32 * 128);
        };
        rd(IC5<0>{}, IC5<0>{});
        rd(IC5<1>{}, IC5<1>{});
        __builtin_amdgcn_sched_barrier(0);
        rd(IC5<2>{}, IC5<2>{});
        if constexpr (KIND != 1) { fire(sd, IC5<0>{}); fire(sd, IC5<1>{}); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == 3) {   // first k-tile of a tile: the accumulators start from the constant 0 (no 64 v_mov per tile)
            const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0][0] = Mma5<T>::run(a0[0], b0[0], z);
            acc[0][1] = Mma5<T>::run(a0[0], b1[0], z);
            acc[1][0] = Mma5<T>::run(a1[0], b0[0], z);
            acc[1][1] = Mma5<T>::run(a1[0], b1[0], z);
        } else {
            mma_group(a0[0], a1[0], b0[0], b1[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == 3) enter_tile();   // index arithmetic of this tile's epilogue and of the next tile: under the first MFMAs
        if constexpr (KIND == 1) {
            // every lane requests (columns beyond N read the tile's first chunk instead: the count of VMEM instructions must not depend on exec)
            if constexpr (F32O) {
            } else if (has_res) {
#pragma unroll
                for (int s = 0; s < 4; ++s) pre_res[s] = ld16((const T*)p.residual + c_rrow + (long)(s * 8) * p.ldr);
                if (bias) pre_bias = ld16(bias + c_ncl);
                if (has_ra) pre_ra = ld16(rowadd + (long)c_img * p.N + c_ncl);
            } else {   // packed epilogue: two column scalars per operand (columns beyond N read the tile's first column: unused, but the count stays)
                const int cA = c_n0 + wn * 64 + l31, cB = cA + 32;
                const int nA = cA < p.N ? cA : c_n0, nB = cB < p.N ? cB : c_n0;
                if (bias) { col_bias[0] = bias[nA]; col_bias[1] = bias[nB]; }
                if (has_ra) { col_ra[0] = rowadd[(long)c_img * p.N + nA]; col_ra[1] = rowadd[(long)c_img * p.N + nB]; }
            }
            u_dma += nslots;
            if (nxt) {
                finish_setup();
                advance();
            } else {
                dma_done = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) cur_a[i] = OOB;
                cur_b[0] = cur_b[1] = OOB;
            }
            __builtin_amdgcn_sched_barrier(0);
            fire(sd, IC5<0>{}); fire(sd, IC5<1>{});
        }
        rd(IC5<3>{}, IC5<0>{});
        fire(sd, IC5<2>{}); fire(sd, IC5<3>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[1], a1[1], b0[1], b1[1]);
        __builtin_amdgcn_sched_barrier(0);
        fire(sd, IC5<4>{}); fire(sd, IC5<5>{});
        if constexpr (KIND == 3) {
            if (nxt) fill_rowtab();   // consumed in this tile's last-but-one k-tile
        }
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0): this wave's reads of the current stage have RETURNED before the barrier lets others overwrite it
        if (nextwait == 2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if (nextwait == 3) {
            if (npre == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (npre == 1) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
            else if (npre == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (npre == 4) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
            else if (npre == 5) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[2], a1[2], b0[2], b1[2]);
        mma_group(a0[0], a1[0], b0[0], b1[0]);
        asm volatile("" ::: "memory");
        rotate();
    };

    // ---- epilogues and statistics merge: shared with igemm6.hip (rows of a wave's block are consecutive GEMM rows here)
    // p.out_seg > 0 (e2eft_upconv2x_fwd): row r (a multiple of 8; of 16 in the packed epilogue, out_seg is a multiple of 16) of the block may lie in a later output
    // segment than row 0 — every segment passed adds out_seg pixel rows (wave-uniform: one float-reciprocal division per slice)
    auto epi_rows_left = [&]() -> int { return F32O ? p.M - (c_m0 + wm * 64 + er) : 0x40000000; };   // (epilogue_f32: rows of this lane's column of slices that exist)
    auto epi_rofs = [&](const int r) -> long {
        if (p.out_seg <= 0) return (long)r;
        return (long)r + (long)(fast_div5(c_mb + r, p.out_seg) - c_seg0) * p.out_seg;
    };
    constexpr int EPI_DEP = RING;
#define EPI_STAMP(i) STAMP5(tseq, i)
#include "igemm_persistent_epilogue.inc"
#undef EPI_STAMP
    const float al_f32 = F32O ? p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f) * (p.alpha_dev2 ? *p.alpha_dev2 : 1.f) : 0.f;
    // ================================ main ==========================================================================
    tile_coords(u_dma);
    fill_rowtab();
    __syncthreads();
    finish_setup();
    advance(); fire_all(s_cur);
    advance(); fire_all(s_nxt);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k-tile 0 of the first tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bool first = true;
    for (;;) {
        STAMP5(tseq, 0);
        // embedded sync of k-tile kt opens k-tile kt + 1: after an epilogue the first two k-tiles are known to have landed
        ktile(IC5<3>{}, first ? 2 : 0);
        for (int kt = 1; kt < nk - 2; ++kt) ktile(IC5<0>{}, 2);
        STAMP5(tseq, 1);
        ktile(IC5<1>{}, 3);    // opens the last k-tile: younger = this k-tile's operand requests + the next tile's first pieces
        STAMP5(tseq, 4);
        ktile(IC5<2>{}, 0);    // opens the epilogue: barrier only
        STAMP5(tseq, 2);
        // after the rotation of the last k-tile its stage is s_dst (the next DMA destination): scratch until the next barrier
        if constexpr (F32O) epilogue_f32(s_dst, al_f32);
        else if constexpr (RES) epilogue(s_dst, zoff_o);
        else epilogue_packed(s_dst, zoff_o);
        STAMP5(tseq, 3);
        ++tseq;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // statistics deposits written; every slice window read
        __builtin_amdgcn_s_barrier();                        // opens k-tile 0 of the next tile (its pieces landed before slice 4 of every wave)
        asm volatile("" ::: "memory");
        if (stats) { const int img = c_m0 / p.rows_per_img; combine(img, (c_m0 - img * p.rows_per_img) / BM, c_n0); }
        STAMP5(tseq - 1, 7);
        if (dma_done) break;
        first = false;
    }
}

static std::atomic<long> g_pers_launches{0};      // debug counter (tests assert that the variant under test really ran)

int device_cus();   // api.hip: CU count of the current device
#ifdef E2EFT_STAMPS
extern int g_debug_flags5;
#endif

template <typename T, int MODE> static int launch5(IgemmParams& p, int nz, int total, int grid, hipStream_t s) {
    if constexpr (std::is_same<T, f16>::value) {
        if (p.split_c > 0) {   // fp32 operands as f16 split planes (launch_igemm_persistent checked the shape)
            hipLaunchKernelGGL((igemm5_kernel<f16, MODE, true, true>), dim3(grid), dim3(512), 0, s, p, total);
            tag_kernel("igemm5_kernel<_Float16, %d, true, f32split>", MODE);
            return check_launch("igemm5");
        }
    }
    if (p.residual) hipLaunchKernelGGL((igemm5_kernel<T, MODE, true>), dim3(grid), dim3(512), 0, s, p, total);
    else hipLaunchKernelGGL((igemm5_kernel<T, MODE, false>), dim3(grid), dim3(512), 0, s, p, total);
    tag_kernel("igemm5_kernel<%s, %d, %s>", std::is_same<T, f16>::value ? "_Float16" : "__bf16", MODE, p.residual ? "true" : "false");
    return check_launch("igemm5");
}

// the checks and the tile plan of launch_igemm_persistent (fills p.mtiles / p.ntiles / p.gn_nslabs): 0, or -1 = not this kernel's launch
static int persistent_plan(int dtype, int mode, IgemmParams& p, int nz, long& total_out, int& cus_out) {
    using namespace pers;
    if (!option(E2EFT_OPT_PERSISTENT)) return -1;
    if (dtype != E2EFT_F16 && dtype != E2EFT_BF16) return -1;
    if (p.ksplit_taps > 0 || p.bias_along_m) return -1;
    if (p.split_c > 0 && (dtype != E2EFT_F16 || nz != 1 || p.x2 || p.rowadd || p.split_c % 64 != 0 || p.cin != 3 * p.split_c || p.c1 != p.cin ||
                          (mode == 0 && p.K != 3 * p.split_c))) return -1;
    // whole 256-row tiles — except the fp32-split launches (F32O): their row table / A offsets zero the rows beyond M and epilogue_f32 masks them (the 18^2 UNet level
    // of the 576^2 recipe is 20.25 tiles per launch)
    if ((p.M % BM != 0 && !(p.split_c > 0 && p.out_seg == 0 && p.M > BM)) || p.K % 64 != 0 || p.K / 64 < 3) return -1;
    if (p.N % 8 != 0 || p.ldo % 8 != 0 || (((uintptr_t)p.out) & 15) != 0) return -1;
    if (p.residual && (p.ldr % 8 != 0 || (((uintptr_t)p.residual) & 15) != 0)) return -1;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return -1;
    if (p.rowadd && ((((uintptr_t)p.rowadd) & 15) != 0 || (p.rows_per_img % BM != 0 && p.rows_per_img < p.M))) return -1;
    if (nz > 1 && (p.so_o % 8 != 0 || p.so_i % 8 != 0 || p.sr_o % 8 != 0 || p.sr_i % 8 != 0)) return -1;
    if (p.out_seg != 0 && (p.out_seg < 16 || p.out_seg % 16 != 0 || nz != 1 || p.residual || p.M % p.out_seg != 0)) return -1;
    if (mode == 0) {
        if ((long)256 * p.ldx1 * 2 >= 0x40000000L || (long)128 * p.ldw * 2 >= 0x40000000L) return -1;
    } else {
        const long img_bytes = (long)p.hin * p.win * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * 2;
        const long span_imgs = 256 / ((long)p.hout * p.wout) + 2;
        if (p.cin % 64 != 0 || p.c1 % 64 != 0 || img_bytes * span_imgs >= 0xD0000000L || (long)128 * p.ldw * 2 >= 0x40000000L) return -1;
    }
    int g_pers_cus = device_cus();     // CUs of the CURRENT device (the one the caller's stream belongs to)
    if (g_pers_cus == 0) return -1;
    const int gopt = option(E2EFT_OPT_PERSISTENT_GRID);   // tests: a small grid sends small problems through the persistent kernel
    if (gopt >= 8 && gopt < g_pers_cus) g_pers_cus = gopt;
    const int mtiles = cdiv(p.M, BM), ntiles = cdiv(p.N, BN);
    const long total = (long)mtiles * ntiles * nz;
    if (4 * total < (long)option(E2EFT_OPT_PERSISTENT_MIN_QROUNDS) * g_pers_cus || total > 2000000000L || mtiles >= (1 << 22)) return -1;   // (tile / row splits use a float reciprocal: quotients below 2^22)
    p.mtiles = mtiles;
    p.ntiles = ntiles;
    if (p.gn_partial) {   // statistics need whole tiles inside one image; otherwise igemm2 may still be able to emit them (128-row slabs): fall through
        if (!(nz == 1 && p.rows_per_img % BM == 0 && p.M % p.rows_per_img == 0)) return -1;
        p.gn_nslabs = p.rows_per_img / BM;
    }
    total_out = total;
    cus_out = g_pers_cus;
    return 0;
}

bool igemm_persistent_eligible(int dtype, int mode, IgemmParams& p, int nz) {   // pure host arithmetic (csrc/f32split.hip asks before it splits anything)
    long total;
    int cus;
    return persistent_plan(dtype, mode, p, nz, total, cus) == 0;
}

int launch_igemm_persistent(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    long total;
    int g_pers_cus;
    if (persistent_plan(dtype, mode, p, nz, total, g_pers_cus) != 0) return -1;
    g_pers_launches.fetch_add(1, std::memory_order_relaxed);
#ifdef E2EFT_STAMPS
    p.debug_flags = g_debug_flags5;
#endif
    if (dtype == E2EFT_F16) return mode ? launch5<f16, 1>(p, nz, (int)total, g_pers_cus, s) : launch5<f16, 0>(p, nz, (int)total, g_pers_cus, s);
    return mode ? launch5<bf16, 1>(p, nz, (int)total, g_pers_cus, s) : launch5<bf16, 0>(p, nz, (int)total, g_pers_cus, s);
}

}  // namespace e2eft

#ifdef E2EFT_STAMPS
namespace e2eft { int g_debug_flags5 = 0; }
extern "C" void e2eft_debug_set_flags5(int f) { e2eft::g_debug_flags5 = f; }
extern "C" int e2eft_debug_read_stamps5(long long* host, int nworkgroups) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e2eft::g_stamps5), (size_t)nworkgroups * 32 * 8 * sizeof(long long));
}
#endif
extern "C" long e2eft_debug_persistent_launches(void) { return e2eft::g_pers_launches.load(); }   // not part of include/e2eft.h
