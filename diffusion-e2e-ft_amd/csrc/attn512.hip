// attn512.hip — fused attention forward for ONE head of width 512 (fp16 / bf16), gfx950: the mid-block attention of the SD VAE
// (reference: GeoWizard/geowizard/models/unet_2d_blocks.py:589-601 builds it — GroupNorm, q/k/v/out Linear with bias, a single 512-wide
// head over H*W tokens; at 768x768 input 9216 tokens per image, once in the encoder and once in the decoder).  Unfused it was
// QK^T -> [B, 9216, 9216] scores (1.36 GB at batch 8) -> row softmax in place -> PV: 5.4 GB of HBM traffic and 2 x 2.3 ms per inference
// step; here the scores never leave the registers.
//
// Decomposition.  What bounds a d = 512 head is the LDS: S^T = K Q^T and O^T += V^T P^T both take their A operand (a K / V^T fragment,
// 1 KiB per wave-instruction) from LDS, so LDS bytes per flop go with 1 / (queries per wave).  16 queries per wave (16x16x32 MFMAs,
// everything in 256 registers) needs 256 B/clk/CU — the whole LDS; 32 queries per wave (32x32x16 MFMAs) halves that, but then the O^T
// accumulator alone is 512 x 32 fp32 = 256 registers per lane.  So: ONE wave per SIMD with the full 512-register file — O^T in the 256
// accumulation registers, the wave's Q^T fragments in 128 VGPRs — four such waves (128 queries) per workgroup, one workgroup per CU.
//   * 32-key tiles.  K [32][512] and V [32][512] stay ROW-MAJOR in LDS and arrive by LDS-DMA (`buffer_load ... lds`, one 1-KiB row per
//     wave-instruction, 16 per wave and tile, no staging registers, keys beyond nk read zeros through the descriptor's bounds check);
//     the V^T fragments are read with ds_read_b64_tr_b16 (the hardware transpose read: a lane of a 16-lane group supplies the address
//     of 4 consecutive d of one key and receives 4 keys of one d), so nothing is transposed on the way in.  Row pitches 1040 B (K,
//     ds_read_b128: 4 dwords mod 64 per key) and 1088 B (V, transpose reads: 16 dwords mod 64 per key) keep both conflict-free.
//   * Both contractions are issued swapped (S^T = K Q^T, O^T += V^T P^T) as in attn.hip: a query is a lane column, softmax state is
//     lane-local, P feeds the second MFMA straight from the S^T registers; the key -> MFMA-k-slot permutation that implies is applied
//     to the V fragment addresses.
//   * The O^T update of tile t-1 is DELAYED into iteration t, where it is independent of that iteration's softmax: the wave is alone on
//     its SIMD, so the ~110 VALU instructions of softmax(t) only overlap matrix work if they are issued between MFMAs that do not
//     depend on them.  Buffers alive in iteration t: K(t), V(t-1) being read, K(t+1), V(t) landing — two of each.
//   * Running maximum with a threshold (flash-attention's deferred rescale): the reference maximum of a query only moves when a tile
//     exceeds it by more than 2^6 in the exponent domain, so probabilities are at most 64 (exact in fp16 / bf16 relative precision, fp32
//     accumulation) and the 256-register rescale `O^T *= alpha` — accumulation registers cannot be VALU operands: read, multiply, write
//     back, 768 instructions — runs a handful of times per 288 tiles instead of on every record of any of the wave's 32 queries.
//   * O^T lives in a[0:255] BY NAME: the 32 O^T MFMAs of a tile, the zero fill, the rescale and the final read-out are inline asm on
//     literal accumulation registers (attn512_regs.inc, generated).  Left to the register allocator the 16 blocks were rotated through
//     ~200 v_accvgpr_mov / read / write per tile at the loop's phi nodes (and spilled 445 registers with an unfenced rescale).
// Work per tile and wave: 32 + 32 MFMAs 32x32x16 (2048 cycles) against 16 exponentials per lane — MFMA-bound, unlike d = 64.
#include "common.h"
#include "attn512_regs.inc"
#include <type_traits>

namespace e2eft {

namespace a5 {
constexpr int D = 512, KT = 32;
constexpr int KP = D * 2 + 16;          // K row pitch 1040 B
constexpr int VP = D * 2 + 64;          // V row pitch 1088 B
constexpr int KTILE = KT * KP;          // 33280
constexpr int VTILE = KT * VP;          // 34816
constexpr int LDS = 2 * (KTILE + VTILE);   // 136192
constexpr float THR = 6.0f;             // deferred rescale: the reference maximum moves when a tile exceeds it by 2^THR
}  // namespace a5

struct Attn5Params {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int batch, nq, nk;
    int ldq, ldk, ldv, ldo;
    float c;  // scale * log2(e)
    // key-split tail (see the launcher): workgroups [0, n_main) own one 128-query block each over all keys and write the result;
    // workgroups n_main + S r + j own query block n_main + r over keys [j kchunk, (j + 1) kchunk) and write an unnormalised partial
    int nqb, n_main, nsplit, kchunk;
    float* part_o;    // [n_split blocks][128][512] fp32: O^T in the block's own reference frame
    float* part_ml;   // [n_split blocks][128][2]: (reference maximum in the exponent domain, row sum)
};

template <typename T> struct Mma512;
template <> struct Mma512<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
    }
};
template <> struct Mma512<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const f2 f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
    }
};

typedef __attribute__((address_space(3))) void* lptr512_t;
typedef short short4v512 __attribute__((ext_vector_type(4)));
template <int V> using IC512 = std::integral_constant<int, V>;
// ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 consecutive 16-bit elements (row i >> 2, columns 4 (i & 3) .. + 3
// of a 4 x 16 block) and receives column i of the block: rows 0 .. 3
__device__ __forceinline__ u32x2 tr_read512(const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v512*)p));
}

// O^T block DT (rows 32 DT .. + 31 of d, this wave's 32 queries) += A (V^T fragment) x B (P^T fragment), accumulator a[16 DT : 16 DT + 15].
// "s_nop 1": a VALU-written operand register needs two wait states in front of the MFMA that reads it, and inside an asm statement nobody
// pads.  "memory": keeps the fragment reads where the source puts them (two blocks ahead) instead of hoisting all 64 of a tile.
template <typename T, int DT> __device__ __forceinline__ void pv_block(const u32x4& a, const u32x4& b, float& chain) {
    // `chain` is not touched: as a read-write operand it pins the softmax slice that produces it BEFORE this MFMA and the one that consumes it
    // AFTER (pure VALU code is otherwise free to sink below the whole MFMA sequence, where nothing hides it)
    if constexpr (std::is_same<T, f16>::value)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c3:%c4], %1, %2, a[%c3:%c4]" : "+v"(chain) : "v"(a), "v"(b), "i"(16 * DT), "i"(16 * DT + 15) : "memory", E2EFT_A256_CLOBBERS);
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c3:%c4], %1, %2, a[%c3:%c4]" : "+v"(chain) : "v"(a), "v"(b), "i"(16 * DT), "i"(16 * DT + 15) : "memory", E2EFT_A256_CLOBBERS);
}
// read-out of block DT (after the last MFMA has retired: the caller issues the wait states once)
template <int DT> __device__ __forceinline__ floatx16 read_block(const int fence) {
    float x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
    asm volatile("v_accvgpr_read_b32 %0, a[%c16]\n\tv_accvgpr_read_b32 %1, a[%c16+1]\n\tv_accvgpr_read_b32 %2, a[%c16+2]\n\tv_accvgpr_read_b32 %3, a[%c16+3]\n\t"
                 "v_accvgpr_read_b32 %4, a[%c16+4]\n\tv_accvgpr_read_b32 %5, a[%c16+5]\n\tv_accvgpr_read_b32 %6, a[%c16+6]\n\tv_accvgpr_read_b32 %7, a[%c16+7]\n\t"
                 "v_accvgpr_read_b32 %8, a[%c16+8]\n\tv_accvgpr_read_b32 %9, a[%c16+9]\n\tv_accvgpr_read_b32 %10, a[%c16+10]\n\tv_accvgpr_read_b32 %11, a[%c16+11]\n\t"
                 "v_accvgpr_read_b32 %12, a[%c16+12]\n\tv_accvgpr_read_b32 %13, a[%c16+13]\n\tv_accvgpr_read_b32 %14, a[%c16+14]\n\tv_accvgpr_read_b32 %15, a[%c16+15]"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=v"(x8), "=v"(x9), "=v"(x10), "=v"(x11), "=v"(x12), "=v"(x13),
                   "=v"(x14), "=v"(x15)
                 : "i"(16 * DT), "v"(fence) : E2EFT_A256_CLOBBERS);
    return floatx16{x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
}

// S^T MFMAs as asm too, with the accumulator pinned to VGPRs ("+v"): a compiler-placed accumulator may be given a[..] registers between two
// asm statements (the clobber lists only protect a0-a255 ACROSS statements), i.e. on top of O^T.
template <typename T> __device__ __forceinline__ void s_mma_first(floatx16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (std::is_same<T, f16>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b) : "memory");
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b) : "memory");
}
template <typename T> __device__ __forceinline__ void s_mma(floatx16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (std::is_same<T, f16>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
// One row of a K / V tile by LDS-DMA, issued from asm: the compiler's waitcnt pass drains vmcnt in front of every LDS read it cannot prove
// disjoint from a pending LDS-DMA it knows of (here: all of them, once per phase) — what it does not see it does not wait for; the wave
// counts its own DMAs (one vmcnt(0) in front of the barrier that ends an iteration).  m0 = LDS byte address of the row (wave-uniform), the
// lanes land at m0 + 16 lane; "s_nop 0": m0 written by SALU -> LDS-DMA needs one wait state.  m0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void dma_row512(const __amdgpu_buffer_rsrc_t& rs, const unsigned voff, const unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}

// grid (ceil(nq / 128), batch), 256 threads = 4 waves x 32 queries, one wave per SIMD
template <typename T>
__global__ __launch_bounds__(256) void attn512_fwd_kernel(const Attn5Params p) {
    using namespace a5;
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int blk = blockIdx.x, part = -1, k_begin = 0, k_end = p.nk;
    if (blk >= p.n_main) {
        const int r = blk - p.n_main;
        const int rq = r / p.nsplit;
        part = r - rq * p.nsplit;
        blk = p.n_main + rq;
        k_begin = part * p.kchunk;
        k_end = min(p.nk, k_begin + p.kchunk);
    }
    const int b = blk / p.nqb;
    const int q0 = (blk - b * p.nqb) * 128 + wave * 32;

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (q = l31, hh) holds Q[q][16 ds + 8 hh .. + 7], ds = 0 .. 31 ----
    u32x4 qf[32];
    {
        const int qr = q0 + l31;
        const bool ok = qr < p.nq;
        const T* src = (const T*)p.q + ((long)b * p.nq + (ok ? qr : 0)) * p.ldq + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 32; ++ds) qf[ds] = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
    }

    // ---- LDS-DMA: one wave-instruction = one key row (64 lanes x 16 B); wave w moves rows w, w + 4, ... of a tile.  The descriptors span
    // exactly this image's nk rows and rows past the end are addressed out of range: they land as zeros (their scores are masked below; V rows
    // must be finite: 0 * p = 0).  Every iteration issues all 16 rows unconditionally (past the last tile: zeros into a buffer nobody reads).
    const unsigned kbytes = (unsigned)(((long)(p.nk - 1) * p.ldk + D) * sizeof(T));
    const unsigned vbytes = (unsigned)(((long)(p.nk - 1) * p.ldv + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.k + (long)b * p.nk * p.ldk), 0, kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.v + (long)b * p.nk * p.ldv), 0, vbytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)((lptr512_t)smem);
    const unsigned krow_b = (unsigned)(p.ldk * (int)sizeof(T)), vrow_b = (unsigned)(p.ldv * (int)sizeof(T));
    auto dma_k_row = [&](const int t, const int buf, const int i) {     // row wave + 4 i of K tile t
        const int r = wave + 4 * i;
        const unsigned key = (unsigned)(k_begin + t * KT + r);
        const unsigned off = key < (unsigned)p.nk ? key * krow_b + lane16 : 0xFFFFFFF0u;
        dma_row512(rsk, off, lds0 + (unsigned)(buf * KTILE + r * KP));
    };
    auto dma_v_row = [&](const int t, const int buf, const int i) {
        const int r = wave + 4 * i;
        const unsigned key = (unsigned)(k_begin + t * KT + r);
        const unsigned off = key < (unsigned)p.nk ? key * vrow_b + lane16 : 0xFFFFFFF0u;
        dma_row512(rsv, off, lds0 + (unsigned)(2 * KTILE + buf * VTILE + r * VP));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the Q loads: from here on the only VMEM operations in flight are the hand-counted DMA rows
    asm volatile(E2EFT_A256_ZERO "s_nop 3" ::: E2EFT_A256_CLOBBERS);
    float m_ref = -1e30f, l_run = 0.f;     // finite start: (m_ref - m_new) * c stays a number
    uint32_t pw[8];                         // packed probabilities of the previous tile (B operand of its delayed O^T update)
#pragma unroll
    for (int w = 0; w < 8; ++w) pw[w] = 0u;

    // per-lane address parts of the fragment reads
    const int kfrag = l31 * KP + hh * 16;                                        // + ds * 32
    // transpose read: 16-lane group g = lane >> 4 serves d columns 16 (g & 1) .. + 15 for the k-slot half hh = g >> 1; lane i of the group
    // supplies key (i >> 2) of the 4-key run and d offset 4 (i & 3)
    const int i16 = lane & 15, g4 = lane >> 4;
    const int vfrag = ((i16 >> 2) + 4 * hh) * VP + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;   // + (16 s2 [+ 8]) * VP + dt * 64

    const int nt = (k_end - k_begin + KT - 1) / KT;     // kchunk is a multiple of KT: only the sequence's last tile can be ragged
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_k_row(0, 0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- softmax of one tile, cut into 32 slices so that the caller can put one between every two MFMAs of the delayed O^T update.
    // State across slices: s (scores), mx, any (rescale wanted by some lane), alpha, mc, psum, np (the new packed probabilities).
    floatx16 s;
    float mx = 0.f, alpha = 1.f, mc = 0.f, psum = 0.f, e_lo = 0.f;
    bool any = false;
    uint32_t np[8];
    auto vslice = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n == 0) mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
        else if constexpr (n >= 1 && n <= 6) mx = fmaxf(fmaxf(mx, s[2 * n + 1]), s[2 * n + 2]);
        else if constexpr (n == 7) mx = fmaxf(mx, s[15]);
        else if constexpr (n == 8) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));        // the other 16 keys of this query
        else if constexpr (n == 9) {
            // the reference only moves when this tile exceeds it by more than 2^THR (per query; both lanes of a query agree: mx is shared)
            const bool need = (mx - m_ref) * p.c > THR;
            any = __builtin_amdgcn_ballot_w64(need) != 0;
            const float m_new = need ? fmaxf(m_ref, mx) : m_ref;
            alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * p.c);                // 1 when unchanged
            m_ref = m_new;
            mc = m_ref * p.c;
            l_run *= alpha;
            psum = 0.f;
        } else if constexpr (n >= 10 && n < 26) {
            constexpr int r = n - 10;
            const float e = __builtin_amdgcn_exp2f(fmaf(s[r], p.c, -mc));
            psum += e;
            if constexpr ((r & 1) == 0) e_lo = e;
            else np[r >> 1] = Mma512<T>::pack(e_lo, e);
        } else if constexpr (n == 26) l_run += psum;
    };
    auto softmax_plain = [&]() {
        vslice(IC512<0>{}); vslice(IC512<1>{}); vslice(IC512<2>{}); vslice(IC512<3>{}); vslice(IC512<4>{}); vslice(IC512<5>{}); vslice(IC512<6>{});
        vslice(IC512<7>{}); vslice(IC512<8>{}); vslice(IC512<9>{}); vslice(IC512<10>{}); vslice(IC512<11>{}); vslice(IC512<12>{}); vslice(IC512<13>{});
        vslice(IC512<14>{}); vslice(IC512<15>{}); vslice(IC512<16>{}); vslice(IC512<17>{}); vslice(IC512<18>{}); vslice(IC512<19>{}); vslice(IC512<20>{});
        vslice(IC512<21>{}); vslice(IC512<22>{}); vslice(IC512<23>{}); vslice(IC512<24>{}); vslice(IC512<25>{}); vslice(IC512<26>{});
    };

    // ---- S^T[key, q] = K(t) Q^T over d = 512: 32 MFMAs on two alternating accumulators (a dependent chain pays for every filler); the 16 DMA
    // rows of this iteration (K(t+1), V(t)) are issued between them — the wave is alone on its SIMD, whatever it issues outside an MFMA's
    // shadow is lost matrix time
    auto s_phase = [&](const int t) {
        const int buf = t & 1;
        const char* row = smem + buf * KTILE + kfrag;
        auto kf = [&](const int ds) { return *reinterpret_cast<const u32x4*>(row + ds * 32); };
        floatx16 sa, sb;
        u32x4 k0 = kf(0), k1 = kf(1), k2 = kf(2), k3 = kf(3);     // four fragments ahead of the MFMA that consumes them
        s_mma_first<T>(sa, k0, qf[0]); k0 = kf(4);
        s_mma_first<T>(sb, k1, qf[1]); k1 = kf(5);
        dma_k_row(t + 1, buf ^ 1, 0);
        s_mma<T>(sa, k2, qf[2]); k2 = kf(6);
        s_mma<T>(sb, k3, qf[3]); k3 = kf(7);
        dma_k_row(t + 1, buf ^ 1, 1);
#pragma unroll
        for (int i = 2; i < 16; i += 2) {       // MFMAs 2 i .. 2 i + 3
            s_mma<T>(sa, k0, qf[2 * i]);     if (2 * i + 4 < 32) k0 = kf(2 * i + 4);
            s_mma<T>(sb, k1, qf[2 * i + 1]); if (2 * i + 5 < 32) k1 = kf(2 * i + 5);
            if (i < 8) dma_k_row(t + 1, buf ^ 1, i); else dma_v_row(t, buf, i - 8);
            s_mma<T>(sa, k2, qf[2 * i + 2]); if (2 * i + 6 < 32) k2 = kf(2 * i + 6);
            s_mma<T>(sb, k3, qf[2 * i + 3]); if (2 * i + 7 < 32) k3 = kf(2 * i + 7);
            if (i + 1 < 8) dma_k_row(t + 1, buf ^ 1, i + 1); else dma_v_row(t, buf, i + 1 - 8);
        }
        // an MFMA's result registers must not be touched by a VALU for 18 wait states after issue
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sa), "+v"(sb));
        s = sa + sb;
        if (k_begin + t * KT + KT > p.nk) {   // last tile: keys beyond nk
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = k_begin + t * KT + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (j >= p.nk) s[r] = -INFINITY;
            }
        }
    };
    // O^T += V^T(tile in vbuf) P^T(pw), optionally with one softmax slice of the current tile behind every MFMA
    auto pv_phase = [&](const int vbuf, auto with_softmax) {
        constexpr bool SM = decltype(with_softmax)::value != 0;
        const char* sv = smem + 2 * KTILE + vbuf * VTILE + vfrag;
        const u32x4 pf0 = {pw[0], pw[1], pw[2], pw[3]}, pf1 = {pw[4], pw[5], pw[6], pw[7]};
        auto frag = [&](auto nc) -> u32x4 {                       // block n = 16 s2 + dt
            constexpr int n = decltype(nc)::value, s2 = n >> 4, dt = n & 15;
            const u32x2 v0 = tr_read512(sv + (16 * s2) * VP + dt * 64);
            const u32x2 v1 = tr_read512(sv + (16 * s2 + 8) * VP + dt * 64);
            return u32x4{v0[0], v0[1], v1[0], v1[1]};
        };
        u32x4 f0 = frag(IC512<0>{}), f1 = frag(IC512<1>{}), f2;
        float idle = 0.f;
        auto step = [&](auto nc, u32x4& cur, u32x4& nxt2) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n + 2 < 32) nxt2 = frag(IC512<(n + 2 < 32 ? n + 2 : 0)>{});
            // slices 0-8 carry the running tile maximum, slices 10-25 the probability sum: the value the slice behind this MFMA continues
            pv_block<T, (n & 15)>(cur, n < 16 ? pf0 : pf1, !SM ? idle : (n <= 9 ? mx : psum));
            if constexpr (SM) vslice(IC512<n>{});
        };
        // three fragment registers rotate: block n in `cur`, n + 1 already requested, n + 2 requested now
#define E2EFT_PV3(n) step(IC512<n>{}, f0, f2); step(IC512<n + 1>{}, f1, f0); step(IC512<n + 2>{}, f2, f1);
        E2EFT_PV3(0) E2EFT_PV3(3) E2EFT_PV3(6) E2EFT_PV3(9) E2EFT_PV3(12) E2EFT_PV3(15) E2EFT_PV3(18) E2EFT_PV3(21) E2EFT_PV3(24) E2EFT_PV3(27)
        step(IC512<30>{}, f0, f2); step(IC512<31>{}, f1, f0);
#undef E2EFT_PV3
    };
    auto end_iteration = [&]() {
        if (any) {   // uniform, rare: some query moved its reference — bring the 256 accumulators into the new frame (alpha = 1 for the other lanes)
            float t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile(E2EFT_A256_RESCALE : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(alpha) : E2EFT_A256_CLOBBERS);
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) pw[w] = np[w];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of K(t+1) and V(t) have landed
        __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is done reading K(t) and V(t-1)
        asm volatile("" ::: "memory");
    };

    // iteration 0: no O^T update yet
    s_phase(0);
    softmax_plain();
    end_iteration();
    // iterations 1 .. nt-1: S^T(t), then O^T update of tile t-1 with softmax(t) in its shadow.  Buffers: K(t) in kbuf[t & 1], V(t-1) in vbuf[(t-1) & 1]
    for (int t = 1; t < nt; ++t) {
        s_phase(t);
        pv_phase((t - 1) & 1, IC512<1>{});
        end_iteration();
    }
    pv_phase((nt - 1) & 1, IC512<0>{});

    // ---- epilogue: O / l, 8-byte stores of 4 consecutive d ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + l31;
    T* dst = (T*)p.out + ((long)b * p.nq + (qr < p.nq ? qr : 0)) * p.ldo;
    int fence = 0;
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, 0" : "=v"(fence) :: E2EFT_A256_CLOBBERS);   // the last MFMAs have retired before the first read-out
    float* po = nullptr;
    if (part >= 0) {   // key-split tail: unnormalised O^T (fp32), the frame it is in and the row sum go to the workspace; attn512_combine_kernel merges
        const long slot = (long)(blockIdx.x - p.n_main) * 128 + wave * 32 + l31;
        po = p.part_o + slot * D;
        if (hh == 0) {
            p.part_ml[slot * 2] = m_ref * p.c;
            p.part_ml[slot * 2 + 1] = l_tot;
        }
    }
    auto store_block = [&](auto dc) {
        constexpr int dt = decltype(dc)::value;
        const floatx16 o = read_block<dt>(fence);
        if (part >= 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<floatx4*>(po + dt * 32 + 8 * g + 4 * hh) = floatx4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
        } else if (qr < p.nq) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = Mma512<T>::pack(o[4 * g] * inv, o[4 * g + 1] * inv);
                w[1] = Mma512<T>::pack(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
            }
        }
    };
    store_block(IC512<0>{}); store_block(IC512<1>{}); store_block(IC512<2>{}); store_block(IC512<3>{}); store_block(IC512<4>{}); store_block(IC512<5>{});
    store_block(IC512<6>{}); store_block(IC512<7>{}); store_block(IC512<8>{}); store_block(IC512<9>{}); store_block(IC512<10>{}); store_block(IC512<11>{});
    store_block(IC512<12>{}); store_block(IC512<13>{}); store_block(IC512<14>{}); store_block(IC512<15>{});
}

// merge of the key-split tail: out[q] = sum_j O_j 2^(m_j - M) / sum_j l_j 2^(m_j - M), M = max_j m_j.  One thread per (query, 8 d values).
template <typename T>
__global__ __launch_bounds__(256) void attn512_combine_kernel(const Attn5Params p) {
    using namespace a5;
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    const int chunks = D / 8;
    const long row = it / chunks;            // (tail query block, row inside it)
    const int c8 = (int)(it - row * chunks) * 8;
    const int rq = (int)(row >> 7), ri = (int)(row & 127);
    const int nrem = (int)gridDim.x * 256 / chunks / 128;
    if (rq >= nrem) return;
    const int blk = p.n_main + rq;
    const int b = blk / p.nqb;
    const int qr = (blk - b * p.nqb) * 128 + ri;
    if (qr >= p.nq) return;
    float m[8], M = -INFINITY;
    for (int j = 0; j < p.nsplit; ++j) {
        m[j] = p.part_ml[(((long)rq * p.nsplit + j) * 128 + ri) * 2];
        M = fmaxf(M, m[j]);
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, L = 0.f;
    for (int j = 0; j < p.nsplit; ++j) {
        const long slot = ((long)rq * p.nsplit + j) * 128 + ri;
        const float w = __builtin_amdgcn_exp2f(m[j] - M);
        L += p.part_ml[slot * 2 + 1] * w;
        const floatx4 a0 = *reinterpret_cast<const floatx4*>(p.part_o + slot * D + c8);
        const floatx4 a1 = *reinterpret_cast<const floatx4*>(p.part_o + slot * D + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += a0[e] * w; acc[4 + e] += a1[e] * w; }
    }
    const float inv = 1.f / L;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(acc[e] * inv);
    st16((T*)p.out + ((long)b * p.nq + qr) * p.ldo + c8, o);
}

// Tail balancing.  A workgroup owns 128 queries and a CU holds one workgroup, so B * ceil(nq / 128) blocks run in ceil(blocks / CUs) rounds:
// 8 x 72 = 576 blocks on 256 CUs are 3 rounds for 2.25 rounds of work.  The blocks of the last, partial round (rem = blocks mod CUs) are cut S ways
// along the KEYS (S = CUs / rem, at most 8, at least 512 keys per part): the last round then lasts 1 / S of a round.  The parts leave
// unnormalised fp32 partials in the caller's workspace, a small kernel merges them.
static void attn512_plan(const E2eftAttnDesc* d, int cus, int& nqb, int& n_main, int& nsplit, int& kchunk) {
    nqb = cdiv(d->nq, 128);
    const long blocks = (long)d->batch * nqb;
    n_main = (int)blocks; nsplit = 1; kchunk = d->nk_seg;
    if (cus <= 0) return;
    const long rem = blocks % cus;
    if (rem == 0) return;
    long S = cus / rem;
    if (S > 8) S = 8;
    while (S > 1 && d->nk_seg / S < 512) --S;
    if (S < 2) return;
    kchunk = ((cdiv(d->nk_seg, (long)S) + a5::KT - 1) / a5::KT) * a5::KT;
    nsplit = cdiv(d->nk_seg, kchunk);
    if (nsplit < 2) { nsplit = 1; kchunk = d->nk_seg; return; }
    n_main = (int)(blocks - rem);
}

int device_cus();   // api.hip

}  // namespace e2eft

using namespace e2eft;

extern "C" size_t e2eft_attn512_workspace_bytes(const E2eftAttnDesc* d) {
    if (!d || d->batch <= 0 || d->nq <= 0 || d->nk_seg <= 0) return 0;
    int nqb, n_main, nsplit, kchunk;
    attn512_plan(d, device_cus(), nqb, n_main, nsplit, kchunk);
    if (nsplit < 2) return 0;
    const long tail = (long)d->batch * nqb - n_main;
    return (size_t)tail * nsplit * 128 * (512 + 2) * sizeof(float);
}

extern "C" int e2eft_attn512_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* workspace, size_t ws_bytes,
                                 void* stream) {
    E2EFT_REQUIRE(d && q && k && v && out, "attn512: null pointer");
    E2EFT_REQUIRE(d->dtype == E2EFT_F16 || d->dtype == E2EFT_BF16, "attn512: dtype %d unsupported (fp16 / bf16; fp32 uses the unfused path)", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->heads == 1 && d->nq > 0 && d->nk_seg > 0, "attn512: geometry (one head of width 512)");
    E2EFT_REQUIRE(d->kv_nseg == 1 && d->kv_bmod == d->batch, "attn512: plain self / cross attention only (kv_nseg = 1, kv_bmod = batch)");
    E2EFT_REQUIRE(d->ldq >= 512 && d->ldk >= 512 && d->ldv >= 512 && d->ldo >= 512, "attn512: row strides smaller than 512");
    E2EFT_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0, "attn512: row strides must be multiples of 8");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0, "attn512: alignment");
    E2EFT_REQUIRE(d->scale > 0.f, "attn512: scale must be positive");
    E2EFT_REQUIRE((long)d->nk_seg * (d->ldk > d->ldv ? d->ldk : d->ldv) * 2 < 0xFFFF0000L, "attn512: one image's keys / values must span less than 4 GB");
    Attn5Params p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.batch = d->batch; p.nq = d->nq; p.nk = d->nk_seg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.c = d->scale * 1.4426950408889634f;
    attn512_plan(d, device_cus(), p.nqb, p.n_main, p.nsplit, p.kchunk);
    const long blocks = (long)d->batch * p.nqb;
    const long tail = blocks - p.n_main;
    const size_t need = (size_t)tail * p.nsplit * 128 * (512 + 2) * sizeof(float);
    if (p.nsplit < 2 || !workspace || ws_bytes < need || ((uintptr_t)workspace & 15) != 0) {   // no (or too small a) workspace: every block runs all keys
        p.n_main = (int)blocks; p.nsplit = 1; p.kchunk = d->nk_seg;
    }
    p.part_o = (float*)workspace;
    p.part_ml = p.part_o ? p.part_o + (size_t)tail * p.nsplit * 128 * 512 : nullptr;
    const long nwg = p.n_main + (blocks - p.n_main) * p.nsplit;
    E2EFT_REQUIRE(nwg < 2147483647L, "attn512: grid");
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == E2EFT_F16) hipLaunchKernelGGL((attn512_fwd_kernel<f16>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn512_fwd_kernel<bf16>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    if (p.nsplit > 1) {
        const unsigned cb = (unsigned)((blocks - p.n_main) * 128 * (512 / 8) / 256);
        if (d->dtype == E2EFT_F16) hipLaunchKernelGGL((attn512_combine_kernel<f16>), dim3(cb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn512_combine_kernel<bf16>), dim3(cb), dim3(256), 0, s, p);
    }
    return check_launch("attn512_fwd");
}
