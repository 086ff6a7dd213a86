// attn.hip — fused attention forward for head dim 64 (fp16 / bf16), MFMA 32x32x16 + online softmax, gfx950.
//
// Work decomposition: one 256-thread workgroup = 128 query rows of one (image, head); each of its 4 waves owns
// 32 query rows.  K/V are consumed in 64-key tiles staged in LDS (K row-major, V transposed to [d][key] while it
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

// grid (ceil(nq/128), heads, batch).  JOINT: keys come from kv_nseg = 2 batch-strided segments (GeoWizard), which costs an
// integer division per loaded row; the plain case indexes keys linearly.
// Measured dead ends: s_setprio(1) around the MFMA phases (771 -> 697 TF/s), v_dot2c row sums on the packed probabilities (717).
// launch_bounds(256, 2): two workgroups per CU caps the wave at 256 registers, which makes the compiler keep the MFMA
// accumulators in VGPRs — with the 512-register budget it parks O^T / S^T in AGPRs and pays a v_accvgpr_read + write per
// element per tile for the online-softmax rescale (measured: 255 of ~600 VALU instructions per tile).
template <typename T, bool JOINT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KVBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const T* __restrict__ Q = (const T*)p.q;
    const T* __restrict__ K = (const T*)p.k;
    const T* __restrict__ V = (const T*)p.v;

    // ---- Q^T fragments (B operand): lane (q = l31, hh) holds Q[q][16 ds + 8 hh .. +7], ds = 0..3 ----
    u32x4 qf[4];
    {
        const int qr = q0 + l31;
        const bool ok = qr < p.nq;
        const T* src = Q + ((long)b * p.nq + (ok ? qr : 0)) * p.ldq + head * 64 + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            qf[ds] = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
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

    floatx16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

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

        // ---- S^T = K Q^T : two 32-key sub-tiles ----
        floatx16 s[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt2][r] = 0.f;
            const char* row = sk + (kt2 * 32 + l31) * KROW + hh * 16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                u32x4 kf = *reinterpret_cast<const u32x4*>(row + ds * 32);
                s[kt2] = MmaA<T>::run(kf, qf[ds], s[kt2]);
            }
        }
        // ---- mask keys beyond nk_total (last tile only) ----
        if (t * 64 + 64 > p.nk_total) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (j >= p.nk_total) s[kt2][r] = -INFINITY;
                }
        }
        // ---- online softmax (raw-score running max; exp2 with the scale folded into one fma) ----
        float mx = s[0][0];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
        const float mc = m_new * p.c;
        m_run = m_new;
        // probabilities, packed straight into the B operand of the second MFMA
        uint32_t pw[2][8];
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(s[kt2][2 * w], p.c, -mc));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(s[kt2][2 * w + 1], p.c, -mc));
                pw[kt2][w] = Pk<T>::pack(e0, e1);
                psum += e0 + e1;   // (v_dot2c on the packed word is one instruction per pair but measured slower beside the MFMAs: 717 vs 771 TF/s)
            }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // the running max settles after a few tiles: skip the rescale then
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pf = {pw[kt2][4 * s2], pw[kt2][4 * s2 + 1], pw[kt2][4 * s2 + 2], pw[kt2][4 * s2 + 3]};
                const int kb = (kt2 * 32 + 16 * s2 + 4 * hh) * 2;  // byte offset of the first 4-key run
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vrow = sv + (dt * 32 + l31) * VROW + kb;
                    const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);
                    const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
                    const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                    o[dt] = MmaA<T>::run(vf, pf, o[dt]);
                }
            }
        }

        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O / l, 8-byte stores of 4 consecutive d ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + l31;
    if (p.lse && hh == 0 && qr < p.nq) p.lse[((long)b * p.heads + head) * p.nq + qr] = m_run * p.c + __builtin_amdgcn_logf(l_tot);
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
    E2EFT_REQUIRE(d->heads <= 65535 && d->batch <= 65535, "attn: grid");
    E2EFT_REQUIRE(d->scale > 0.f, "attn: scale must be positive");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk_seg = d->nk_seg;
    p.kv_nseg = d->kv_nseg; p.kv_bmod = d->kv_bmod; p.nk_total = d->nk_seg * d->kv_nseg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.c = d->scale * 1.4426950408889634f;
    p.lse = lse;
    dim3 grid(cdiv(d->nq, 128), d->heads, d->batch);
    hipStream_t s = (hipStream_t)stream;
    const bool joint = d->kv_nseg > 1;
    if (d->dtype == E2EFT_F16) {
        if (joint) hipLaunchKernelGGL((attn_fwd_kernel<f16, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<f16, false>), grid, dim3(256), 0, s, p);
    } else {
        if (joint) hipLaunchKernelGGL((attn_fwd_kernel<bf16, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<bf16, false>), grid, dim3(256), 0, s, p);
    }
    return check_launch("attn_fwd");
}
