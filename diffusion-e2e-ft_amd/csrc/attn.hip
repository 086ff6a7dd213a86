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

template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    union { T t[2]; uint32_t u; } x;
    x.t[0] = from_f<T>(lo);
    x.t[1] = from_f<T>(hi);
    return x.u;
}

// grid (ceil(nq/128), heads, batch)
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
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
        const int seg = j / p.nk_seg;
        return (long)(kvb0 + seg * p.kv_bmod) * p.nk_seg + (j - seg * p.nk_seg);
    };

    u32x4 rk[2], rv[2];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto load_tile = [&](int t) {
        const int base = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = base + k_r0 + 32 * i;
            rk[i] = j < p.nk_total ? *reinterpret_cast<const u32x4*>(K + key_row(j) * p.ldk + head * 64 + k_kc * 8) : zero4;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = base + 2 * v_kp + i;
            rv[i] = j < p.nk_total ? *reinterpret_cast<const u32x4*>(V + key_row(j) * p.ldv + head * 64 + v_vc * 8) : zero4;
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
            const uint32_t lo = (rv[0][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            const uint32_t hi = (rv[1][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            *reinterpret_cast<uint32_t*>(sv + (8 * v_vc + e) * VROW + v_kp * 4) = lo | (hi << 16);
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
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kt2][r], p.c, -mc));
                s[kt2][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 pf;
#pragma unroll
                for (int w = 0; w < 4; ++w) pf[w] = pack2<T>(s[kt2][8 * s2 + 2 * w], s[kt2][8 * s2 + 2 * w + 1]);
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
    dim3 grid(cdiv(d->nq, 128), d->heads, d->batch);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == E2EFT_F16) hipLaunchKernelGGL((attn_fwd_kernel<f16>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16>), grid, dim3(256), 0, s, p);
    return check_launch("attn_fwd");
}
