// attn.hip — fused attention forward for head dim 64 (fp16 / bf16), MFMA 32x32x16 + online softmax, gfx950.
//
// Work decomposition: one 256-thread workgroup = 128 query rows of one (image, head); each of its 4 waves owns 32 query rows.  K / V are consumed in 64-key tiles
// staged in LDS (both row-major; the V^T fragments come from the transpose read ds_read_b64_tr_b16), double-buffered with the next tile's global
// loads in flight under the current tile's MFMAs.
//
// Both contractions are issued "swapped" so that the query row lives on the lane axis of every MFMA result:
//   S^T[key, q] = K[key, :] . Q[q, :]      (A = K tile from LDS, B = Q^T fragments held in registers)
//   O^T[d,  q] += V^T[d, key] P^T[key, q]  (A = V^T tile from LDS, B = P^T straight from the S^T registers)
// so running max / sum / rescale are lane-local (one cross-half exchange per tile for the max), and the P
// operand of the second MFMA is exactly the register set the first MFMA produced (no LDS round trip):
// the key -> MFMA-k-slot permutation this implies is applied to the V fragment addresses instead.
#include "common.h"
#include <type_traits>

namespace e2eft {

constexpr int KROW = 144;  // K tile LDS row stride (128 data bytes + 16)  -> conflict-free ds_read_b128
constexpr int VROW = 192;  // V tile (row-major [key][64 d]) LDS row stride: the four key rows of a transpose read land on four different 64-byte windows
constexpr int KTILE = 64 * KROW;
constexpr int VTILE = 64 * VROW;
constexpr int KVBUF = KTILE + VTILE;

struct AttnParams {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int batch, heads, nq, nk_seg, kv_nseg, kv_bmod, nk_total;
    int nqb;   // query blocks per (image, head): ceil(nq / (128 QB))
    int ldq, ldk, ldv, ldo;
    float c;  // scale * log2(e)
    float* lse;  // optional [batch][heads][nq]: log2-sum-exp of the scaled scores (saved for e2eft_attn_bwd)
};

template <typename T> struct MmaA;
template <> struct MmaA<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct MmaA<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bhalf2v __attribute__((ext_vector_type(2)));

// pack two fp32 into one dword of T with a single v_cvt_pk_{f16,bf16}_f32
template <typename T> struct Pk;
template <> struct Pk<f16> {
    __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
        const float2v f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
    }
};
template <> struct Pk<bf16> {
    __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
        const float2v f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bhalf2v));
    }
};
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return Pk<T>::pack(lo, hi); }

typedef short short4va __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 consecutive 16-bit elements (row i >> 2, columns 4 (i & 3) .. + 3 of a
// 4 x 16 block) and receives column i of the block, rows 0 .. 3 — V stays row-major in LDS and is transposed on the way to the MFMA's A operand
__device__ __forceinline__ u32x2 tr_read(const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4va*)p));
}

// 1-D grid of nqb * heads * batch workgroups (nqb = ceil(nq / (128 QB))).  JOINT: keys come from kv_nseg = 2 batch-strided segments
// (GeoWizard), which costs an integer division per loaded row; the plain case indexes keys linearly.
// QB = query blocks of 32 rows per wave (1).  (Round 3 measured QB = 2 — 64 rows per wave, every K / V^T fragment read and every loader store
// serving two MFMAs: 742 against 726 TF/s at 9216 keys, +2 %: LDS / loader issue slots are not what binds; the per-score VALU work is — below.)
// Block -> (image, head, query block): all query blocks of one (image, head) run on ONE XCD (block id mod 8 = XCD, MI355X_MICROARCH.md), so the
// head's K / V (2.4 MB at 9216 keys) is fetched into one L2 instead of eight: PMC had 3.5x the algorithmic HBM bytes with the plain map.
// Measured dead ends: s_setprio(1) around the MFMA phases (771 -> 697 TF/s), v_dot2c row sums on the packed probabilities (717).
// launch_bounds(256, 2): two workgroups per CU caps the wave at 256 registers, which makes the compiler keep the MFMA
// accumulators in VGPRs — with the 512-register budget it parks O^T / S^T in AGPRs and pays a v_accvgpr_read + write per
// element per tile for the online-softmax rescale (measured: 255 of ~600 VALU instructions per tile).
// -DE2EFT_ATTN_PROBE=n (scripts/experiments/attn_probe.hip only; never defined in the library build): ceiling probes, WRONG results by construction.
// 1: K / V tile 0 pinned in LDS (no global loads, LDS stores or barrier in the loop); 2: 1 + no softmax VALU; 3: 1 + no LDS fragment reads in the loop;
// 4: production traffic without the softmax VALU.
#ifdef E2EFT_ATTN_PROBE
constexpr int PROBE = E2EFT_ATTN_PROBE;
#else
constexpr int PROBE = 0;
#endif
constexpr bool PROBE_PINNED = PROBE >= 1 && PROBE <= 3, PROBE_NOSM = PROBE == 2 || PROBE == 4, PROBE_NOLDS = PROBE == 3;

template <typename T, bool JOINT, int QB>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KVBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // ---- XCD-aware block map: pair = (image, head); pairs are dealt to the eight XCDs round-robin, each XCD walks its pairs' query blocks
    int b, head, qblk;
    {
        const int npair = p.batch * p.heads, nqb = p.nqb;
        const int L = blockIdx.x, full = (npair >> 3) << 3;          // pairs covered by complete rounds of eight
        if (L < full * nqb) {
            const int xcd = L & 7, idx = L >> 3;
            const int pr = (idx / nqb) * 8 + xcd;
            qblk = idx - (idx / nqb) * nqb;
            b = pr / p.heads; head = pr - b * p.heads;
        } else {                                                      // the remaining (< 8) pairs: plain order
            const int r = L - full * nqb;
            const int pr = full + r / nqb;
            qblk = r - (r / nqb) * nqb;
            b = pr / p.heads; head = pr - b * p.heads;
        }
    }
    const int q0 = qblk * (128 * QB) + wave * (32 * QB);

    const T* __restrict__ Q = (const T*)p.q;
    const T* __restrict__ K = (const T*)p.k;
    const T* __restrict__ V = (const T*)p.v;

    // ---- Q^T fragments (B operand): lane (q = l31, hh) holds Q[q][16 ds + 8 hh .. +7], ds = 0..3, for each of the wave's QB query blocks ----
    u32x4 qf[QB][4];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int qr = q0 + 32 * j + l31;
        const bool ok = qr < p.nq;
        const T* src = Q + ((long)b * p.nq + (ok ? qr : 0)) * p.ldq + head * 64 + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            Vec16<T> v;
            v.raw = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) v.e[e] = from_f<T>(to_f(v.e[e]) * p.c);     // Q' = Q * scale * log2(e): the MFMA delivers exponents
            qf[j][ds] = v.raw;
        }
    }

    // ---- loader: K and V tiles are both 64 keys x 8 sixteen-byte chunks, row-major: thread -> keys (tid / 8) and (tid / 8 + 32), chunk tid % 8.
    // Row pointers advance by a constant per tile; only a tile that reaches past the last key clamps its rows (scores masked below, p = 0).
    const int k_kc = tid & 7, k_r0 = tid >> 3;
    const int kvb0 = b % p.kv_bmod;
    auto key_row = [&](int j) -> long {  // global row index (in rows of the [kv_batch*nk_seg] matrix) of key j
        if (!JOINT) return (long)kvb0 * p.nk_seg + j;
        const int seg = j / p.nk_seg;
        return (long)(kvb0 + seg * p.kv_bmod) * p.nk_seg + (j - seg * p.nk_seg);
    };
    const T* kp[2];
    const T* vp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        kp[i] = K + key_row(min(k_r0 + 32 * i, p.nk_total - 1)) * p.ldk + head * 64 + k_kc * 8;
        vp[i] = V + key_row(min(k_r0 + 32 * i, p.nk_total - 1)) * p.ldv + head * 64 + k_kc * 8;
    }
    const long kstep = (long)64 * p.ldk, vstep = (long)64 * p.ldv;
    u32x4 rk[2], rv[2];
    auto load_tile = [&](int t) {
        const int base = t * 64;
        if (JOINT || base + 64 > p.nk_total) {      // (joint keys change segment somewhere in the sequence: index every row)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long r = key_row(min(base + k_r0 + 32 * i, p.nk_total - 1));
                rk[i] = *reinterpret_cast<const u32x4*>(K + r * p.ldk + head * 64 + k_kc * 8);
                rv[i] = *reinterpret_cast<const u32x4*>(V + r * p.ldv + head * 64 + k_kc * 8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rk[i] = *reinterpret_cast<const u32x4*>(kp[i] + t * kstep);
                rv[i] = *reinterpret_cast<const u32x4*>(vp[i] + t * vstep);
            }
        }
    };
    auto store_tile = [&](int buf) {
        char* sk = smem + buf * KVBUF;
        char* sv = sk + KTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(sk + (k_r0 + 32 * i) * KROW + k_kc * 16) = rk[i];
            *reinterpret_cast<u32x4*>(sv + (k_r0 + 32 * i) * VROW + k_kc * 16) = rv[i];
        }
    };
    // transpose-read address of this lane inside a V tile: 16-lane group g serves d columns 16 (g & 1) .. + 15 for the k-slot half hh = g >> 1;
    // lane i of the group supplies key (i >> 2) of a 4-key run and d offset 4 (i & 3)
    const int i16 = lane & 15;
    const int vfrag = ((i16 >> 2) + 4 * hh) * VROW + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;   // + first key of the run * VROW + dt * 64

    // ---- softmax state.  The kernel is bound by VALU issue (~135 VALU instructions against 16 MFMAs per 64-key tile), so two of
    // the four per-score VALU operations are moved onto the half-idle matrix pipe:
    //  * the scores come out of the MFMA already in the exponent domain AND relative to the reference maximum: Q is pre-multiplied by
    //    scale * log2(e) (once per workgroup) and the accumulator chain of S^T starts from a block holding -m_ref instead of 0, so a probability is
    //    ONE v_exp_f32 of the accumulator (no fma / subtract per score);
    //  * the row sums come from a fifth accumulator block: l^T += ones(32 x 16) P^T, 4 extra MFMAs per tile instead of 32 v_add_f32 (and the sum is
    //    taken over the ROUNDED probabilities, the ones that multiply V).
    // The reference maximum only moves when a tile exceeds it by more than 2^THR (deferred rescale, as attn512.hip): then — a uniform, rare branch
    // — the tile's scores are shifted, O^T and l^T rescaled and the -m_ref block rewritten.
    constexpr float THR = 6.0f;
    static_assert(QB == 1, "the VALU-light scheme keeps 2 x 16 extra accumulator registers per query block: one query block per wave");
    const u32x4 ones = std::is_same<T, f16>::value ? u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u} : u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    floatx16 o[2], lacc, cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; lacc[r] = 0.f; cinit[r] = 0.f; }
    float m_ref = 0.f;      // exponent domain; the first tile always re-references (first = true)
    bool first = true;
    asm volatile("" : "+v"(cinit));     // a register BLOCK of 16 equal values that stays put (not a splat re-materialised in front of every chain)
    u32x4 ones_v = ones;
    asm volatile("" : "+v"(ones_v));    // likewise the constant A operand of the row-sum MFMAs (else two v_mov_b64 from SGPRs in front of each)

    const int nt = (p.nk_total + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    u32x4 hk[2][4], hv[2][2][2];     // PROBE 3 only: the fragments of tile 0, read once
    if constexpr (PROBE_NOLDS) {
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) hk[kt2][ds] = *reinterpret_cast<const u32x4*>(smem + (kt2 * 32 + l31) * KROW + hh * 16 + ds * 32);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vb = smem + KTILE + vfrag + (kt2 * 32 + 16 * s2) * VROW;
                    const u32x2 v0 = tr_read(vb + dt * 64), v1 = tr_read(vb + 8 * VROW + dt * 64);
                    hv[kt2][s2][dt] = u32x4{v0[0], v0[1], v1[0], v1[1]};
                }
        }
    }

    for (int t = 0; t < nt; ++t) {
        const int buf = PROBE_PINNED ? 0 : (t & 1);
        const bool more = t + 1 < nt && !PROBE_PINNED;
        if (more) load_tile(t + 1);

        const char* sk = smem + buf * KVBUF;
        const char* sv = sk + KTILE;

        // ---- S'^T = K Q'^T - m_ref : two 32-key sub-tiles ----
        floatx16 s[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {     // (round 4: alternating the two blocks' MFMAs — no dependent neighbours — measured 781 against 795 TF/s this way round)
            const char* row = sk + (kt2 * 32 + l31) * KROW + hh * 16;
            if constexpr (PROBE_NOLDS) {
                s[kt2] = MmaA<T>::run(hk[kt2][0], qf[0][0], cinit);
#pragma unroll
                for (int ds = 1; ds < 4; ++ds) s[kt2] = MmaA<T>::run(hk[kt2][ds], qf[0][ds], s[kt2]);
            } else {
                s[kt2] = MmaA<T>::run(*reinterpret_cast<const u32x4*>(row), qf[0][0], cinit);
#pragma unroll
                for (int ds = 1; ds < 4; ++ds) s[kt2] = MmaA<T>::run(*reinterpret_cast<const u32x4*>(row + ds * 32), qf[0][ds], s[kt2]);
            }
        }
        // ---- mask keys beyond nk_total (last tile only) ----
        if (t * 64 + 64 > p.nk_total) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= p.nk_total) s[kt2][r] = -INFINITY;
                }
        }
        uint32_t pw[2][8];
        if constexpr (PROBE_NOSM) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int w = 0; w < 8; ++w) pw[kt2][w] = (__float_as_uint(s[kt2][2 * w]) & 0x3fff3fffu) ^ (__float_as_uint(s[kt2][2 * w + 1]) >> 18);
        } else {
        // ---- tile maximum relative to the reference (v_max3_f32: 16 instructions for 32 values) ----
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
        mx = fmaxf(fmaxf(mx, s[0][15]), s[1][0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
        mx = fmaxf(mx, s[1][15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (first || __builtin_amdgcn_ballot_w64(mx > THR) != 0) {   // uniform; after the first tiles: rare
            // per query: move the reference by delta >= 0 (first tile: to the tile maximum, whatever its sign), bring this tile's scores, O^T and
            // l^T into the new frame, rewrite the -m_ref block
            const float delta = first ? mx : (mx > THR ? mx : 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_ref += delta;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt2][r] -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; lacc[r] *= alpha; cinit[r] = -m_ref; }
            asm volatile("" : "+v"(cinit));
            first = false;
        }
        // ---- probabilities: one v_exp_f32 per score, packed straight into the B operand of the second MFMA ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int w = 0; w < 8; ++w)
                pw[kt2][w] = Pk<T>::pack(__builtin_amdgcn_exp2f(s[kt2][2 * w]), __builtin_amdgcn_exp2f(s[kt2][2 * w + 1]));
        }

        // ---- O^T += V^T P^T and l^T += 1 P^T ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pf = {pw[kt2][4 * s2], pw[kt2][4 * s2 + 1], pw[kt2][4 * s2 + 2], pw[kt2][4 * s2 + 3]};
                // k-slots 0-3 of half hh: keys kt2 * 32 + 16 s2 + 4 hh + 0..3, k-slots 4-7: the same + 8 (the S^T register order)
                const char* vb = sv + vfrag + (kt2 * 32 + 16 * s2) * VROW;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    if constexpr (PROBE_NOLDS) {
                        o[dt] = MmaA<T>::run(hv[kt2][s2][dt], pf, o[dt]);
                    } else {
                        const u32x2 v0 = tr_read(vb + dt * 64);
                        const u32x2 v1 = tr_read(vb + 8 * VROW + dt * 64);
                        const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                        o[dt] = MmaA<T>::run(vf, pf, o[dt]);
                    }
                }
                lacc = MmaA<T>::run(ones_v, pf, lacc);
            }
        }

        if (more) store_tile(buf ^ 1);
        if constexpr (!PROBE_PINNED) __syncthreads();
    }

    // ---- epilogue: O / l, 8-byte stores of 4 consecutive d (every accumulator register of lacc holds the query's full row sum) ----
    {
        const float l_tot = lacc[0];
        const float inv = 1.f / l_tot;
        const int qr = q0 + l31;
        if (p.lse && hh == 0 && qr < p.nq) p.lse[((long)b * p.heads + head) * p.nq + qr] = m_ref + __builtin_amdgcn_logf(l_tot);
        if (qr < p.nq) {
            T* dst = (T*)p.out + ((long)b * p.nq + qr) * p.ldo + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack2<T>(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
                    w[1] = pack2<T>(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// attn_fwd_dma_kernel — the same arithmetic per (query, key) as attn_fwd_kernel, instruction for instruction (swapped MFMAs, the same accumulation chains, scores
// relative to a reference maximum, row sums on the matrix pipe, deferred rescale: results are bit-identical), with another K / V delivery:
//   * the 64-key tiles arrive by LDS-DMA (`buffer_load ... lds`: one wave instruction = 8 key rows of 128 data bytes = 1 KiB, two K and two V pieces per wave and
//     tile) into a THREE-stage ring — no global -> VGPR -> ds_write staging (attn_fwd_kernel spends 4 global loads, 4 ds_write_b128 and 16 registers per thread and
//     tile on it, and stalls on vmcnt(0) in front of the stores when a load is late); the pieces of tile t + 2 are requested before tile t multiplies;
//   * LDS rows are the 128 data bytes (a DMA piece is lane-linear, there is no row padding), the bank spread comes from XOR swizzles on the SOURCE side, undone in
//     the fragment addresses: K chunk ^ ((row >> 1) & 7) (conflict-free ds_read_b128, as igemm2.hip), V chunk ^ (((row >> 1) & 1) << 2) (the four key rows of a
//     transpose read land on four different 64-byte windows, as wgrad.hip);
//   * one counted wait (this wave's pieces of tile t + 1; those of tile t + 2 may still be in flight) and one barrier per tile, as before.
// Still 4 waves and two workgroups per CU: the two waves of a SIMD belong to DIFFERENT workgroups, drift apart and overlap their MFMA and softmax phases — the
// 8-wave form of this delivery (one workgroup, both waves of a SIMD in lockstep behind one barrier) measured 765 against 795 TF/s (profiles/r04_attention_experiments.md).
// K / V are addressed through one 32-bit buffer descriptor per tensor: the launcher takes this kernel only below 3.5 GB.
namespace adma {
constexpr int KT = 64, HALF = KT * 128, STAGE = 2 * HALF;      // one 64-key tile: 8 KiB of K rows, 8 KiB of V rows
constexpr unsigned OOB = 0xF0000000u, RECORDS = 0xE0000000u;
}
typedef __attribute__((address_space(3))) void* lptr_a_t;
// one LDS-DMA piece (64 lanes x 16 B -> 1 KiB at m0) from asm: issued through the builtin the compiler would count it and drain vmcnt(0) in front of LDS reads it
// cannot prove disjoint; the kernel counts its own pieces.  m0 is saved and restored.
__device__ __forceinline__ void dma_piece_a(const __amdgpu_buffer_rsrc_t& rs, const unsigned voff, const unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}

// TPB = 64-key tiles per barrier.  1: three stages of one tile (48 KiB), tile t + 2 requested while tile t multiplies; 2: two stages of two tiles (64 KiB), the next
// pair requested while this pair multiplies — half the barriers (the measurement that asked for it: with TPB = 1 the DMA delivery runs exactly as fast as the
// register-staged kernel, 802 against 800 TF/s, while the probe with the tile pinned in LDS — no loads AND no barrier — runs at 957: what costs is the
// synchronisation of the four waves once per tile, not the instructions that move the data).
template <typename T, bool JOINT, int TPB>
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(const AttnParams p) {
    using namespace adma;
    constexpr int NSTAGE = 2, SSTAGE = TPB * STAGE, LDS = NSTAGE * SSTAGE;      // 32 KiB at TPB = 1: measured 772-776 against 749-752 TF/s (production) and 749-755 (three stages)
    __shared__ __attribute__((aligned(1024))) char smem[LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int b, head, qblk;
    {   // XCD-aware block map, as attn_fwd_kernel
        const int npair = p.batch * p.heads, nqb = p.nqb;
        const int L = blockIdx.x, full = (npair >> 3) << 3;
        if (L < full * nqb) {
            const int xcd = L & 7, idx = L >> 3;
            const int pr = (idx / nqb) * 8 + xcd;
            qblk = idx - (idx / nqb) * nqb;
            b = pr / p.heads; head = pr - b * p.heads;
        } else {
            const int r = L - full * nqb;
            const int pr = full + r / nqb;
            qblk = r - (r / nqb) * nqb;
            b = pr / p.heads; head = pr - b * p.heads;
        }
    }
    const int q0 = qblk * 128 + wave * 32;
    const T* __restrict__ Q = (const T*)p.q;

    // ---- loader: wave w moves key rows 8 w .. 8 w + 7 and 8 (w + 4) .. of every tile; lane -> row 8 w + (lane >> 3), LDS slot lane & 7 (the swizzle keys only see
    // (row >> 1) & 7 / & 1: unchanged by + 32)
    const int lrow = 8 * wave + (lane >> 3);
    const unsigned kch = (unsigned)(((lane & 7) ^ ((lrow >> 1) & 7)) * 16), vch = (unsigned)(((lane & 7) ^ (((lrow >> 1) & 1) << 2)) * 16);
    const int kvb0 = b % p.kv_bmod;
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.k + head * 64), 0, RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.v + head * 64), 0, RECORDS, 0x00020000);
    const unsigned ldkb = (unsigned)p.ldk * (unsigned)sizeof(T), ldvb = (unsigned)p.ldv * (unsigned)sizeof(T);
    const unsigned lds0 = (unsigned)(uintptr_t)((lptr_a_t)smem);
    auto fire = [&](const int G) {       // the 4 TPB pieces of tile group G this wave owns; keys beyond the last one fetch zeros (their scores are masked below)
#pragma unroll
        for (int sub = 0; sub < TPB; ++sub) {
            const unsigned dst0 = lds0 + (unsigned)((G % NSTAGE) * SSTAGE + sub * STAGE + wave * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = (G * TPB + sub) * KT + lrow + 32 * i;
                unsigned row;
                if (JOINT) {
                    const int seg = key / p.nk_seg;
                    row = (unsigned)((kvb0 + seg * p.kv_bmod) * p.nk_seg + (key - seg * p.nk_seg));
                } else {
                    row = (unsigned)(kvb0 * p.nk_seg + key);
                }
                const bool ok = key < p.nk_total;
                dma_piece_a(rsk, ok ? row * ldkb + kch : OOB, dst0 + (unsigned)(i * 4096));
                dma_piece_a(rsv, ok ? row * ldvb + vch : OOB, dst0 + (unsigned)(i * 4096 + HALF));
            }
        }
    };
    const int nt = (p.nk_total + KT - 1) / KT, ng = (nt + TPB - 1) / TPB;
    fire(0);

    // ---- Q'^T fragments (B operand), pre-multiplied by scale * log2(e)
    u32x4 qf[4];
    {
        const int qr = q0 + l31;
        const bool ok = qr < p.nq;
        const T* src = Q + ((long)b * p.nq + (ok ? qr : 0)) * p.ldq + head * 64 + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            Vec16<T> v;
            v.raw = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) v.e[e] = from_f<T>(to_f(v.e[e]) * p.c);
            qf[ds] = v.raw;
        }
    }
    // fragment addresses inside a stage.  K: row kt2 * 32 + l31, k-step ds: chunk (2 ds + hh) ^ ((l31 >> 1) & 7)
    const int kswz = (l31 >> 1) & 7;
    int kofs[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) kofs[ds] = l31 * 128 + (((2 * ds + hh) ^ kswz) << 4);
    // V^T (transpose read): 16-lane group g1 = (lane >> 4) & 1 serves d columns 16 g1 .. + 15 of the 32-column block dt, lane i16 supplies key row (i16 >> 2) + 4 hh of
    // a run and 4 d; the row's swizzle bit ((row >> 1) & 1 = (i16 >> 3) & 1) toggles the 64-byte half, i.e. dt
    const int i16 = lane & 15, vb1 = (i16 >> 3) & 1;
    const int vrow = ((i16 >> 2) + 4 * hh) * 128 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3);
    const int vofs[2] = {HALF + vrow + vb1 * 64, HALF + vrow + (1 - vb1) * 64};      // dt = 0, 1

    constexpr float THR = 6.0f;
    const u32x4 ones = std::is_same<T, f16>::value ? u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u} : u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    floatx16 o[2], lacc, cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; lacc[r] = 0.f; cinit[r] = 0.f; }
    float m_ref = 0.f;
    bool first = true;
    asm volatile("" : "+v"(cinit));
    u32x4 ones_v = ones;
    asm volatile("" : "+v"(ones_v));

    // group 0 (this wave's pieces; with TPB = 1 the four of tile 1 may still be in flight), then everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    auto tile = [&](const int t, const char* st) {
        // ---- S'^T = K Q'^T - m_ref : two 32-key sub-tiles ----
        floatx16 s[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            const char* row = st + kt2 * 32 * 128;
            s[kt2] = MmaA<T>::run(*reinterpret_cast<const u32x4*>(row + kofs[0]), qf[0], cinit);
#pragma unroll
            for (int ds = 1; ds < 4; ++ds) s[kt2] = MmaA<T>::run(*reinterpret_cast<const u32x4*>(row + kofs[ds]), qf[ds], s[kt2]);
        }
        if (t * 64 + 64 > p.nk_total) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= p.nk_total) s[kt2][r] = -INFINITY;
                }
        }
        uint32_t pw[2][8];
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
        mx = fmaxf(fmaxf(mx, s[0][15]), s[1][0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
        mx = fmaxf(mx, s[1][15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (first || __builtin_amdgcn_ballot_w64(mx > THR) != 0) {   // uniform; after the first tiles: rare
            const float delta = first ? mx : (mx > THR ? mx : 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_ref += delta;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt2][r] -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; lacc[r] *= alpha; cinit[r] = -m_ref; }
            asm volatile("" : "+v"(cinit));
            first = false;
        }
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int w = 0; w < 8; ++w)
                pw[kt2][w] = Pk<T>::pack(__builtin_amdgcn_exp2f(s[kt2][2 * w]), __builtin_amdgcn_exp2f(s[kt2][2 * w + 1]));

        // ---- O^T += V^T P^T and l^T += 1 P^T ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pf = {pw[kt2][4 * s2], pw[kt2][4 * s2 + 1], pw[kt2][4 * s2 + 2], pw[kt2][4 * s2 + 3]};
                const char* vb = st + (kt2 * 32 + 16 * s2) * 128;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const u32x2 v0 = tr_read(vb + vofs[dt]);
                    const u32x2 v1 = tr_read(vb + 8 * 128 + vofs[dt]);
                    const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                    o[dt] = MmaA<T>::run(vf, pf, o[dt]);
                }
                lacc = MmaA<T>::run(ones_v, pf, lacc);
            }
        }
    };

    for (int G = 0; G < ng; ++G) {
        // the stage being refilled held group G - 1 (TPB = 2) / G - 1 of three (TPB = 1): every wave finished reading it before the barrier that closed iteration G - 1
        constexpr int AHEAD = 1;
        if (G + AHEAD < ng) fire(G + AHEAD);
        const char* st = smem + (G % NSTAGE) * SSTAGE;
        tile(G * TPB, st);
        if constexpr (TPB == 2) {
            if (G * TPB + 1 < nt) tile(G * TPB + 1, st + STAGE);      // (uniform)
        }
        if (G + 1 < ng) {   // this wave's pieces of group G + 1 have landed (TPB = 1: those of G + 2 may fly); after the barrier everybody's have, and everybody is done with this stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    {   // ---- epilogue: as attn_fwd_kernel
        const float l_tot = lacc[0];
        const float inv = 1.f / l_tot;
        const int qr = q0 + l31;
        if (p.lse && hh == 0 && qr < p.nq) p.lse[((long)b * p.heads + head) * p.nq + qr] = m_ref + __builtin_amdgcn_logf(l_tot);
        if (qr < p.nq) {
            T* dst = (T*)p.out + ((long)b * p.nq + qr) * p.ldo + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack2<T>(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
                    w[1] = pack2<T>(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
                }
        }
    }
}

}  // namespace e2eft

namespace e2eft { int attn32_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream); }   // attn32.hip
using namespace e2eft;

extern "C" int e2eft_attn_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* stream) {
    return e2eft_attn_fwd_lse(d, q, k, v, out, nullptr, stream);
}

extern "C" int e2eft_attn_fwd_lse(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream) {
    E2EFT_REQUIRE(d && q && k && v && out, "attn: null pointer");
    E2EFT_REQUIRE(d->dtype == E2EFT_F16 || d->dtype == E2EFT_BF16 || d->dtype == E2EFT_F32, "attn: dtype %d unsupported", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->heads > 0 && d->nq > 0 && d->nk_seg > 0, "attn: geometry");
    E2EFT_REQUIRE(d->kv_nseg == 1 || d->kv_nseg == 2, "attn: kv_nseg must be 1 or 2");
    E2EFT_REQUIRE(d->kv_bmod > 0, "attn: kv_bmod");
    E2EFT_REQUIRE(d->scale > 0.f, "attn: scale must be positive");
    if (d->dtype == E2EFT_F32) return attn32_fwd(d, q, k, v, out, lse, stream);      // strict fp32 on v_mfma_f32_32x32x2_f32: attn32.hip
    const int w = d->heads * 64;
    E2EFT_REQUIRE(d->ldq >= w && d->ldk >= w && d->ldv >= w && d->ldo >= w, "attn: row strides smaller than heads*64");
    E2EFT_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0, "attn: row strides must be multiples of 8");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0, "attn: alignment");
    E2EFT_REQUIRE(d->scale > 0.f, "attn: scale must be positive");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk_seg = d->nk_seg;
    p.kv_nseg = d->kv_nseg; p.kv_bmod = d->kv_bmod; p.nk_total = d->nk_seg * d->kv_nseg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.c = d->scale * 1.4426950408889634f;
    p.lse = lse;
    hipStream_t s = (hipStream_t)stream;
    const bool joint = d->kv_nseg > 1;
    const long pairs = (long)d->heads * d->batch;
    p.nqb = cdiv(d->nq, 128);
    E2EFT_REQUIRE(pairs * p.nqb < 2147483647L, "attn: grid");
    dim3 grid((unsigned)(pairs * p.nqb));
    // LDS-DMA delivery (E2EFT_OPT_ATTN_DMA): K / V addressed through one 32-bit buffer descriptor each
    const long kv_rows = (long)d->kv_bmod * d->kv_nseg * d->nk_seg;
    const bool dma = option(E2EFT_OPT_ATTN_DMA) != 0 && kv_rows * (d->ldk > d->ldv ? d->ldk : d->ldv) * 2 < 0xE0000000L;
#define E2EFT_ATTN_LAUNCH(TT, JJ) do { if (dma) hipLaunchKernelGGL((attn_fwd_dma_kernel<TT, JJ, 1>), grid, dim3(256), 0, s, p); \
                                       else hipLaunchKernelGGL((attn_fwd_kernel<TT, JJ, 1>), grid, dim3(256), 0, s, p); } while (0)
    if (d->dtype == E2EFT_F16) {
        if (joint) E2EFT_ATTN_LAUNCH(f16, true); else E2EFT_ATTN_LAUNCH(f16, false);
    } else {
        if (joint) E2EFT_ATTN_LAUNCH(bf16, true); else E2EFT_ATTN_LAUNCH(bf16, false);
    }
#undef E2EFT_ATTN_LAUNCH
    return check_launch("attn_fwd");
}
