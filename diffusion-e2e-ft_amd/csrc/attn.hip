// attn.hip — fused attention forward for head dim 64 (fp16 / bf16), MFMA 32x32x16 + online softmax, gfx950.
//
// Work decomposition: one 256-thread workgroup = 128 or 256 query rows of one (image, head); each of its 4 waves owns
// 32 or 64 query rows (template parameter QB, chosen per launch).  K/V are consumed in 64-key tiles staged in LDS (K row-major, V transposed to [d][key] while it
// is written to LDS), double-buffered with the next tile's global loads in flight under the current tile's MFMAs.
//
// Both contractions are issued "swapped" so that the query row lives on the lane axis of every MFMA result:
//   S^T[key, q] = K[key, :] . Q[q, :]      (A = K tile from LDS, B = Q^T fragments held in registers)
//   O^T[d,  q] += V^T[d, key] P^T[key, q]  (A = V^T tile from LDS, B = P^T straight from the S^T registers)
// so running max / sum / rescale are lane-local (one cross-half exchange per tile for the max), and the P
// operand of the second MFMA is exactly the register set the first MFMA produced (no LDS round trip):
// the key -> MFMA-k-slot permutation this implies is applied to the V^T fragment addresses instead.
#include "common.h"

namespace e2eft {

constexpr int KROW = 144;  // K tile LDS row stride (128 data bytes + 16)  -> conflict-free ds_read_b128
constexpr int VROW = 136;  // V^T tile LDS row stride (128 + 8)            -> conflict-free ds_read_b64
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

// 1-D grid of nqb * heads * batch workgroups (nqb = ceil(nq / (128 QB))).  JOINT: keys come from kv_nseg = 2 batch-strided segments
// (GeoWizard), which costs an integer division per loaded row; the plain case indexes keys linearly.
// QB = query blocks of 32 rows per wave.  QB = 2 (64 rows per wave, 256 per workgroup) is the big-problem form: every K / V^T fragment read from
// LDS, every K / V row fetched and every LDS store of the loader then serves TWO MFMAs instead of one — the d = 64 head is bound by VALU + LDS
// issue slots, and 34 of the ~200 non-MFMA instructions of a 64-key tile are exactly those loads and stores.  QB = 1 keeps
// small problems (cross attention over a handful of keys, CLIP's 257 tokens, the 12^2 levels) on twice as many workgroups.
// Block -> (image, head, query block): all query blocks of one (image, head) run on ONE XCD (block id mod 8 = XCD, MI355X_MICROARCH.md), so the
// head's K / V (2.4 MB at 9216 keys) is fetched into one L2 instead of eight: PMC had 3.5x the algorithmic HBM bytes with the plain map.
// Measured dead ends: s_setprio(1) around the MFMA phases (771 -> 697 TF/s), v_dot2c row sums on the packed probabilities (717).
// launch_bounds(256, 2): two workgroups per CU caps the wave at 256 registers, which makes the compiler keep the MFMA
// accumulators in VGPRs — with the 512-register budget it parks O^T / S^T in AGPRs and pays a v_accvgpr_read + write per
// element per tile for the online-softmax rescale (measured: 255 of ~600 VALU instructions per tile).
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
            qf[j][ds] = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ---- loader mapping ----
    // K: 64 keys x 8 chunks(16 B): thread -> keys (tid/8) and (tid/8 + 32), chunk tid%8
    const int k_kc = tid & 7, k_r0 = tid >> 3;
    // V: thread -> key pair kp (keys 2kp, 2kp+1), d-chunk vc (8 d values)
    const int v_kp = l31, v_vc = 2 * wave + hh;
    const int kvb0 = b % p.kv_bmod;

    auto key_row = [&](int j) -> long {  // global row index (in rows of the [kv_batch*nk_seg] matrix) of key j
        if (!JOINT) return (long)kvb0 * p.nk_seg + j;
        const int seg = j / p.nk_seg;
        return (long)(kvb0 + seg * p.kv_bmod) * p.nk_seg + (j - seg * p.nk_seg);
    };

    u32x4 rk[2], rv[2];
    auto load_tile = [&](int t) {
        const int base = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = min(base + k_r0 + 32 * i, p.nk_total - 1);   // rows past the end: scores are masked below, p = 0
            rk[i] = *reinterpret_cast<const u32x4*>(K + key_row(j) * p.ldk + head * 64 + k_kc * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = min(base + 2 * v_kp + i, p.nk_total - 1);
            rv[i] = *reinterpret_cast<const u32x4*>(V + key_row(j) * p.ldv + head * 64 + v_vc * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* sk = smem + buf * KVBUF;
        char* sv = sk + KTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(sk + (k_r0 + 32 * i) * KROW + k_kc * 16) = rk[i];
        // transpose: V^T[8 vc + e][2 kp, 2 kp + 1]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // one v_perm_b32 per pair: {low halves} or {high halves} of the two keys' dwords
            const uint32_t w = __builtin_amdgcn_perm(rv[1][e >> 1], rv[0][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
            *reinterpret_cast<uint32_t*>(sv + (8 * v_vc + e) * VROW + v_kp * 4) = w;
        }
    };

    floatx16 o[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][dt][r] = 0.f;
        m_run[j] = -INFINITY;
        l_run[j] = 0.f;
    }

    const int nt = (p.nk_total + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);

        const char* sk = smem + buf * KVBUF;
        const char* sv = sk + KTILE;

        // ---- S^T = K Q^T : two 32-key sub-tiles x QB query blocks; one K fragment read feeds QB MFMAs ----
        floatx16 s[QB][2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            const char* row = sk + (kt2 * 32 + l31) * KROW + hh * 16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(row + ds * 32);
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    if (ds == 0) {
                        const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        s[j][kt2] = MmaA<T>::run(kf, qf[j][0], z);
                    } else {
                        s[j][kt2] = MmaA<T>::run(kf, qf[j][ds], s[j][kt2]);
                    }
                }
            }
        }
        // ---- mask keys beyond nk_total (last tile only) ----
        if (t * 64 + 64 > p.nk_total) {
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key >= p.nk_total) s[j][kt2][r] = -INFINITY;
                    }
        }
        // ---- online softmax (raw-score running max; exp2 with the scale folded into one fma) ----
        uint32_t pw[QB][2][8];
        bool resc = false;
        float alpha[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            // 32 values -> 1: v_max3_f32 halves the chain (the compiler fuses the nested fmaxf)
            float mx = fmaxf(fmaxf(s[j][0][0], s[j][0][1]), s[j][0][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[j][0][r]), s[j][0][r + 1]);
            mx = fmaxf(fmaxf(mx, s[j][0][15]), s[j][1][0]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[j][1][r]), s[j][1][r + 1]);
            mx = fmaxf(mx, s[j][1][15]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[j], mx);
            alpha[j] = __builtin_amdgcn_exp2f((m_run[j] - m_new) * p.c);
            const float mc = m_new * p.c;
            m_run[j] = m_new;
            // probabilities, packed straight into the B operand of the second MFMA
            float psum = 0.f;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float e0 = __builtin_amdgcn_exp2f(fmaf(s[j][kt2][2 * w], p.c, -mc));
                    const float e1 = __builtin_amdgcn_exp2f(fmaf(s[j][kt2][2 * w + 1], p.c, -mc));
                    pw[j][kt2][w] = Pk<T>::pack(e0, e1);
                    psum += e0 + e1;   // (v_dot2c on the packed word is one instruction per pair but measured slower beside the MFMAs: 717 vs 771 TF/s)
                }
            l_run[j] = l_run[j] * alpha[j] + psum;
            resc = resc || (alpha[j] != 1.0f);
        }
        if (__builtin_amdgcn_ballot_w64(resc) != 0) {   // the running max settles after a few tiles: skip the rescale then
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[j][dt][r] *= alpha[j];
        }

        // ---- O^T += V^T P^T : one V^T fragment read feeds QB MFMAs ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int kb = (kt2 * 32 + 16 * s2 + 4 * hh) * 2;  // byte offset of the first 4-key run
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vrow = sv + (dt * 32 + l31) * VROW + kb;
                    const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);
                    const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
                    const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        const u32x4 pf = {pw[j][kt2][4 * s2], pw[j][kt2][4 * s2 + 1], pw[j][kt2][4 * s2 + 2], pw[j][kt2][4 * s2 + 3]};
                        o[j][dt] = MmaA<T>::run(vf, pf, o[j][dt]);
                    }
                }
            }
        }

        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O / l, 8-byte stores of 4 consecutive d ----
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = 1.f / l_tot;
        const int qr = q0 + 32 * j + l31;
        if (p.lse && hh == 0 && qr < p.nq) p.lse[((long)b * p.heads + head) * p.nq + qr] = m_run[j] * p.c + __builtin_amdgcn_logf(l_tot);
        if (qr < p.nq) {
            T* dst = (T*)p.out + ((long)b * p.nq + qr) * p.ldo + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack2<T>(o[j][dt][4 * g] * inv, o[j][dt][4 * g + 1] * inv);
                    w[1] = pack2<T>(o[j][dt][4 * g + 2] * inv, o[j][dt][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
                }
        }
    }
}

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_attn_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* stream) {
    return e2eft_attn_fwd_lse(d, q, k, v, out, nullptr, stream);
}

extern "C" int e2eft_attn_fwd_lse(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream) {
    E2EFT_REQUIRE(d && q && k && v && out, "attn: null pointer");
    E2EFT_REQUIRE(d->dtype == E2EFT_F16 || d->dtype == E2EFT_BF16, "attn: dtype %d unsupported (fp16/bf16 only; fp32 uses the unfused path)", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->heads > 0 && d->nq > 0 && d->nk_seg > 0, "attn: geometry");
    E2EFT_REQUIRE(d->kv_nseg == 1 || d->kv_nseg == 2, "attn: kv_nseg must be 1 or 2");
    E2EFT_REQUIRE(d->kv_bmod > 0, "attn: kv_bmod");
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
    // 64 query rows per wave when that still leaves two 256-row workgroups per CU's worth of work and the key loop is long enough to matter
    const long pairs = (long)d->heads * d->batch;
    const bool wide = option(E2EFT_OPT_ATTN_WIDE) && pairs * cdiv(d->nq, 256) >= 512 && p.nk_total >= 256;
    p.nqb = cdiv(d->nq, wide ? 256 : 128);
    E2EFT_REQUIRE(pairs * p.nqb < 2147483647L, "attn: grid");
    dim3 grid((unsigned)(pairs * p.nqb));
#define E2EFT_ATTN_LAUNCH(TT, JJ)                                                                   \
    do {                                                                                            \
        if (wide) hipLaunchKernelGGL((attn_fwd_kernel<TT, JJ, 2>), grid, dim3(256), 0, s, p);       \
        else hipLaunchKernelGGL((attn_fwd_kernel<TT, JJ, 1>), grid, dim3(256), 0, s, p);            \
    } while (0)
    if (d->dtype == E2EFT_F16) {
        if (joint) E2EFT_ATTN_LAUNCH(f16, true); else E2EFT_ATTN_LAUNCH(f16, false);
    } else {
        if (joint) E2EFT_ATTN_LAUNCH(bf16, true); else E2EFT_ATTN_LAUNCH(bf16, false);
    }
#undef E2EFT_ATTN_LAUNCH
    return check_launch("attn_fwd");
}
