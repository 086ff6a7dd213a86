// attn_bwd.hip — fused attention backward for head dim 64 (fp16 / bf16), gfx950.  Nothing of size Nq x Nk touches HBM:
// the probabilities are recomputed from q, k and the forward's log-sum-exp (e2eft_attn_fwd_lse), exactly as in the forward.
//
//   P  = 2^(q' k^T - lse2)               q' = round_T(c q), c = scale * log2(e): the forward (attn.hip) multiplies its Q fragments by c ONCE and rounds them,
//                                        so the recomputation does the same — with c applied to the fp32 score instead, P would differ from the
//                                        forward's by the rounding of q' (2^-9 relative per element in bf16) and its rows would not sum to one; lse2 from the forward
//   dV = P^T dO
//   dP = dO V^T,   dS = P o (dP - D),     D[q] = sum_d dO[q,d] O[q,d]
//   dQ = scale * dS K,   dK = scale * dS^T Q
//
// Two kernels, each the owner of its outputs (no atomics, deterministic):
//   attn_bwd_dkdv: a workgroup owns 128 keys (a wave 32, lane <-> key) and streams 64-query tiles through LDS.  S and dP come
//       out of the MFMAs with lanes = keys, registers = queries — which IS the B-operand layout of the two products that
//       contract over queries (dV^T = dO^T P, dK^T = Q^T dS), so P and dS feed them straight from registers; their A operands
//       are transposed LDS images of the dO / Q tile, written transposed by the loader (the forward's V^T recipe), with the
//       query -> MFMA-k-slot permutation folded into the fragment addresses.
//   attn_bwd_dq: the forward's structure (lane <-> query, 64-key tiles): S^T = K Q^T, dP^T = V dO^T, and dQ^T += K^T dS^T with
//       dS^T from registers and K^T from a transposed LDS image.
// MFMA operand convention (v_mfma_f32_32x32x16): lane l supplies A[i = l&31][k = 8(l>>5)..+7] and B[k = 8(l>>5)..+7][j = l&31];
// the result register r of lane l is D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31].
#include "common.h"

namespace e2eft {

constexpr int BROW = 144;           // row-major tile rows: 128 data bytes + 16 -> conflict-free ds_read_b128
constexpr int BTROW = 136;          // transposed tile rows: 128 + 8 -> conflict-free ds_read_b64
constexpr int BTILE = 64 * BROW;
constexpr int BTTILE = 64 * BTROW;

struct AttnBwdParams {
    const void* q;
    const void* k;
    const void* v;
    const void* dout;
    const float* lse;    // [B][H][nq]
    const float* dsum;   // [B][H][nq]
    void* dq;
    void* dk;
    void* dv;
    int batch, heads, nq, nk;
    int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
    float c, scale;
};

template <typename T> struct MmaB;
template <> struct MmaB<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct MmaB<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef float float2w __attribute__((ext_vector_type(2)));
typedef __bf16 bhalf2w __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ uint32_t packb(float lo, float hi);
template <> __device__ __forceinline__ uint32_t packb<f16>(float lo, float hi) {
    const float2w f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
}
template <> __device__ __forceinline__ uint32_t packb<bf16>(float lo, float hi) {
    const float2w f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bhalf2w));
}

// D[b][h][q] = sum_d dO[b,q,h*64+d] * O[b,q,h*64+d]; one thread per (row, head)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(long rows, int heads, int ldo, int lddo, int nq, const T* __restrict__ o,
                                                            const T* __restrict__ dout, float* __restrict__ dsum) {
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it >= rows * heads) return;
    const long row = it / heads;
    const int h = (int)(it - row * heads);
    const T* po = o + row * ldo + h * 64;
    const T* pd = dout + row * lddo + h * 64;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        Vec16<T> a = ld16(po + 8 * c), d = ld16(pd + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += to_f(a.e[e]) * to_f(d.e[e]);
    }
    const long b = row / nq, qi = row - b * nq;
    dsum[(b * heads + h) * nq + qi] = acc;
}

// shared loader pieces -----------------------------------------------------------------------------------------
// row-major copy of a 64-row x 64-col tile: thread -> rows tid/8 and tid/8 + 32, 16-byte chunk tid % 8
// transposed copy: thread -> row pair (2 rp, 2 rp + 1), 8-column chunk cc; LDS image [col][row] as row pairs per dword

// ---------------------------------------------------------------------------------------------------------------
// dK, dV.  grid (ceil(nk/128), heads, batch), 256 threads
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnBwdParams p) {
    // per stage: Q [64][BROW] | dO [64][BROW] | Q^T [64][BTROW] | dO^T [64][BTROW] | lse2[64] | D[64]
    constexpr int STAGE = 2 * BTILE + 2 * BTTILE + 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int key = blockIdx.x * 128 + wave * 32 + l31;
    const bool kvalid = key < p.nk;

    const T* __restrict__ Q = (const T*)p.q + (long)b * p.nq * p.ldq + head * 64;
    const T* __restrict__ DO = (const T*)p.dout + (long)b * p.nq * p.lddo + head * 64;
    const float* __restrict__ LSE = p.lse + ((long)b * p.heads + head) * p.nq;
    const float* __restrict__ DS = p.dsum + ((long)b * p.heads + head) * p.nq;

    // B operands held for the whole kernel: K[key][16 ds + 8 hh ..] and V[key][...]
    u32x4 kf[4], vf[4];
    {
        const long kr = (long)b * p.nk + min(key, p.nk - 1);
        const T* ks = (const T*)p.k + kr * p.ldk + head * 64 + 8 * hh;
        const T* vs = (const T*)p.v + kr * p.ldv + head * 64 + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            kf[ds] = *reinterpret_cast<const u32x4*>(ks + 16 * ds);
            vf[ds] = *reinterpret_cast<const u32x4*>(vs + 16 * ds);
        }
    }

    const int r_kc = tid & 7, r_r0 = tid >> 3;        // row-major loader
    const int t_rp = l31, t_cc = 2 * wave + hh;       // transposed loader

    u32x4 gq[2], gd[2], tq[2], td[2];
    float glse = 0.f, gds = 0.f;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto load_tile = [&](int t) {
        const int base = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qr = base + r_r0 + 32 * i;
            const bool ok = qr < p.nq;
            gq[i] = ok ? *reinterpret_cast<const u32x4*>(Q + (long)qr * p.ldq + r_kc * 8) : zero4;
            gd[i] = ok ? *reinterpret_cast<const u32x4*>(DO + (long)qr * p.lddo + r_kc * 8) : zero4;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qr = base + 2 * t_rp + i;
            const bool ok = qr < p.nq;
            tq[i] = ok ? *reinterpret_cast<const u32x4*>(Q + (long)qr * p.ldq + t_cc * 8) : zero4;
            td[i] = ok ? *reinterpret_cast<const u32x4*>(DO + (long)qr * p.lddo + t_cc * 8) : zero4;
        }
        if (tid < 64) {
            const int qr = base + tid;
            glse = qr < p.nq ? LSE[qr] : INFINITY;     // rows past the end: p = 2^(-inf) = 0
            gds = qr < p.nq ? DS[qr] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
        char* sq = smem + buf * STAGE;
        char* sd = sq + BTILE;
        char* sqt = sd + BTILE;
        char* sdt = sqt + BTTILE;
        float* sl = reinterpret_cast<float*>(sdt + BTTILE);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Vec16<T> qs;                       // the score operand is q' = round(c q) as in the forward; the transposed image below (dK's operand) stays q
            qs.raw = gq[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) qs.e[e] = from_f<T>(to_f(qs.e[e]) * p.c);
            *reinterpret_cast<u32x4*>(sq + (r_r0 + 32 * i) * BROW + r_kc * 16) = qs.raw;
            *reinterpret_cast<u32x4*>(sd + (r_r0 + 32 * i) * BROW + r_kc * 16) = gd[i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
            *reinterpret_cast<uint32_t*>(sqt + (8 * t_cc + e) * BTROW + t_rp * 4) = __builtin_amdgcn_perm(tq[1][e >> 1], tq[0][e >> 1], sel);
            *reinterpret_cast<uint32_t*>(sdt + (8 * t_cc + e) * BTROW + t_rp * 4) = __builtin_amdgcn_perm(td[1][e >> 1], td[0][e >> 1], sel);
        }
        if (tid < 64) { sl[tid] = glse; sl[64 + tid] = gds; }
    };

    floatx16 dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[dt][r] = dk[dt][r] = 0.f;

    const int nt = (p.nq + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);
        const char* sq = smem + buf * STAGE;
        const char* sd = sq + BTILE;
        const char* sqt = sd + BTILE;
        const char* sdt = sqt + BTTILE;
        const float* sl = reinterpret_cast<const float*>(sdt + BTTILE);

#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // S[q][key] and dP[q][key] for the 32 queries of this sub-tile
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const char* qrow = sq + (qt * 32 + l31) * BROW + hh * 16;
            const char* drow = sd + (qt * 32 + l31) * BROW + hh * 16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const u32x4 qa = *reinterpret_cast<const u32x4*>(qrow + ds * 32);
                s = MmaB<T>::run(qa, kf[ds], s);
            }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const u32x4 da = *reinterpret_cast<const u32x4*>(drow + ds * 32);
                dp = MmaB<T>::run(da, vf[ds], dp);
            }
            // P = 2^(c s - lse2[q]), dS = P (dP - D[q]); register r <-> query 32 qt + (r&3) + 8 (r>>2) + 4 hh
            uint32_t pw[8], dw[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 l4 = *reinterpret_cast<const floatx4*>(sl + qt * 32 + 8 * g + 4 * hh);
                const floatx4 d4 = *reinterpret_cast<const floatx4*>(sl + 64 + qt * 32 + 8 * g + 4 * hh);
                float pe[4], de[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[e] = __builtin_amdgcn_exp2f(s[4 * g + e] - l4[e]);
                    de[e] = pe[e] * (dp[4 * g + e] - d4[e]);
                }
                pw[2 * g] = packb<T>(pe[0], pe[1]); pw[2 * g + 1] = packb<T>(pe[2], pe[3]);
                dw[2 * g] = packb<T>(de[0], de[1]); dw[2 * g + 1] = packb<T>(de[2], de[3]);
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]  (two 16-query k-steps)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 pf = {pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
                const u32x4 df = {dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
                const int qb = (qt * 32 + 16 * s2 + 4 * hh) * 2;   // byte offset of this lane's first 4-query run
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* r1 = sdt + (dt * 32 + l31) * BTROW + qb;
                    const u32x2 a0 = *reinterpret_cast<const u32x2*>(r1);
                    const u32x2 a1 = *reinterpret_cast<const u32x2*>(r1 + 16);
                    dv[dt] = MmaB<T>::run(u32x4{a0[0], a0[1], a1[0], a1[1]}, pf, dv[dt]);
                    const char* r2 = sqt + (dt * 32 + l31) * BTROW + qb;
                    const u32x2 b0 = *reinterpret_cast<const u32x2*>(r2);
                    const u32x2 b1 = *reinterpret_cast<const u32x2*>(r2 + 16);
                    dk[dt] = MmaB<T>::run(u32x4{b0[0], b0[1], b1[0], b1[1]}, df, dk[dt]);
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // dK[key][d], dV[key][d]: register r of tile dt <-> d = 32 dt + (r&3) + 8 (r>>2) + 4 hh; 8-byte stores of 4 consecutive d
    if (kvalid) {
        T* dkp = (T*)p.dk + ((long)b * p.nk + key) * p.lddk + head * 64;
        T* dvp = (T*)p.dv + ((long)b * p.nk + key) * p.lddv + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = packb<T>(dk[dt][4 * g] * p.scale, dk[dt][4 * g + 1] * p.scale);
                w[1] = packb<T>(dk[dt][4 * g + 2] * p.scale, dk[dt][4 * g + 3] * p.scale);
                *reinterpret_cast<u32x2*>(dkp + dt * 32 + 8 * g + 4 * hh) = w;
                w[0] = packb<T>(dv[dt][4 * g], dv[dt][4 * g + 1]);
                w[1] = packb<T>(dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
                *reinterpret_cast<u32x2*>(dvp + dt * 32 + 8 * g + 4 * hh) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dQ.  grid (ceil(nq/128), heads, batch), 256 threads; lane <-> query
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    // per stage: K [64][BROW] | V [64][BROW] | K^T [64][BTROW]
    constexpr int STAGE = 2 * BTILE + BTTILE;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qr = blockIdx.x * 128 + wave * 32 + l31;
    const bool qvalid = qr < p.nq;

    const T* __restrict__ K = (const T*)p.k + (long)b * p.nk * p.ldk + head * 64;
    const T* __restrict__ V = (const T*)p.v + (long)b * p.nk * p.ldv + head * 64;

    u32x4 qf[4], dof[4];
    float lse2 = INFINITY, dsum = 0.f;
    {
        const long row = (long)b * p.nq + min(qr, p.nq - 1);
        const T* qs = (const T*)p.q + row * p.ldq + head * 64 + 8 * hh;
        const T* ds_ = (const T*)p.dout + row * p.lddo + head * 64 + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            Vec16<T> qv;                       // q' = round(c q): the forward's score operand
            qv.raw = *reinterpret_cast<const u32x4*>(qs + 16 * ds);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv.e[e] = from_f<T>(to_f(qv.e[e]) * p.c);
            qf[ds] = qv.raw;
            dof[ds] = *reinterpret_cast<const u32x4*>(ds_ + 16 * ds);
        }
        if (qvalid) {
            lse2 = p.lse[((long)b * p.heads + head) * p.nq + qr];
            dsum = p.dsum[((long)b * p.heads + head) * p.nq + qr];
        }
    }

    const int r_kc = tid & 7, r_r0 = tid >> 3;
    const int t_rp = l31, t_cc = 2 * wave + hh;
    u32x4 gk[2], gv[2], tk[2];
    auto load_tile = [&](int t) {
        const int base = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = min(base + r_r0 + 32 * i, p.nk - 1);   // rows past the end are masked through the scores
            gk[i] = *reinterpret_cast<const u32x4*>(K + (long)j * p.ldk + r_kc * 8);
            gv[i] = *reinterpret_cast<const u32x4*>(V + (long)j * p.ldv + r_kc * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = min(base + 2 * t_rp + i, p.nk - 1);
            tk[i] = *reinterpret_cast<const u32x4*>(K + (long)j * p.ldk + t_cc * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* sk = smem + buf * STAGE;
        char* sv = sk + BTILE;
        char* skt = sv + BTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(sk + (r_r0 + 32 * i) * BROW + r_kc * 16) = gk[i];
            *reinterpret_cast<u32x4*>(sv + (r_r0 + 32 * i) * BROW + r_kc * 16) = gv[i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
            *reinterpret_cast<uint32_t*>(skt + (8 * t_cc + e) * BTROW + t_rp * 4) = __builtin_amdgcn_perm(tk[1][e >> 1], tk[0][e >> 1], sel);
        }
    };

    floatx16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int nt = (p.nk + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);
        const char* sk = smem + buf * STAGE;
        const char* sv = sk + BTILE;
        const char* skt = sv + BTILE;

#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            // S^T[key][q] = K Q^T, dP^T[key][q] = V dO^T for the 32 keys of this sub-tile
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const char* krow = sk + (kt2 * 32 + l31) * BROW + hh * 16;
            const char* vrow = sv + (kt2 * 32 + l31) * BROW + hh * 16;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const u32x4 ka = *reinterpret_cast<const u32x4*>(krow + ds * 32);
                s = MmaB<T>::run(ka, qf[ds], s);
            }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const u32x4 va = *reinterpret_cast<const u32x4*>(vrow + ds * 32);
                dp = MmaB<T>::run(va, dof[ds], dp);
            }
            // dS^T = P^T (dP^T - D[q]); register r <-> key 64 t + 32 kt2 + (r&3) + 8 (r>>2) + 4 hh
            if (t * 64 + 64 > p.nk) {   // last tile only: keys past the end get p = 0
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= p.nk) s[r] = -INFINITY;
            }
            uint32_t dw[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float p0 = __builtin_amdgcn_exp2f(s[2 * w] - lse2);
                const float p1 = __builtin_amdgcn_exp2f(s[2 * w + 1] - lse2);
                dw[w] = packb<T>(p0 * (dp[2 * w] - dsum), p1 * (dp[2 * w + 1] - dsum));
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 df = {dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
                const int kb = (kt2 * 32 + 16 * s2 + 4 * hh) * 2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* r1 = skt + (dt * 32 + l31) * BTROW + kb;
                    const u32x2 a0 = *reinterpret_cast<const u32x2*>(r1);
                    const u32x2 a1 = *reinterpret_cast<const u32x2*>(r1 + 16);
                    dq[dt] = MmaB<T>::run(u32x4{a0[0], a0[1], a1[0], a1[1]}, df, dq[dt]);
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (qvalid) {
        T* dst = (T*)p.dq + ((long)b * p.nq + qr) * p.lddq + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = packb<T>(dq[dt][4 * g] * p.scale, dq[dt][4 * g + 1] * p.scale);
                w[1] = packb<T>(dq[dt][4 * g + 2] * p.scale, dq[dt][4 * g + 3] * p.scale);
                *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
            }
    }
}

}  // namespace e2eft

namespace e2eft {
int attn32_bwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, const void* out, const void* dout, int32_t lddo, const float* lse, void* dq,
               int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv, void* workspace, void* stream);   // attn32.hip
}
using namespace e2eft;

extern "C" size_t e2eft_attn_bwd_workspace_bytes(const E2eftAttnDesc* d) {
    if (!d || d->batch <= 0 || d->heads <= 0 || d->nq <= 0) return 0;
    return (size_t)d->batch * d->heads * d->nq * sizeof(float);
}

extern "C" int e2eft_attn_bwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, const void* out, const void* dout,
                              int32_t lddo, const float* lse, void* dq, int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv,
                              void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(d && q && k && v && out && dout && lse && dq && dk && dv && workspace, "attn_bwd: null pointer");
    E2EFT_REQUIRE(d->dtype == E2EFT_F16 || d->dtype == E2EFT_BF16 || d->dtype == E2EFT_F32, "attn_bwd: dtype %d unsupported (head dim 64)", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->heads > 0 && d->nq > 0 && d->nk_seg > 0, "attn_bwd: geometry");
    E2EFT_REQUIRE(d->kv_nseg == 1 && d->kv_bmod == d->batch, "attn_bwd: joint (segmented) keys are not supported; concatenate k / v");
    if (d->dtype == E2EFT_F32) {      // strict fp32: attn32.hip
        E2EFT_REQUIRE(d->heads <= 65535 && d->batch <= 65535, "attn_bwd: grid");
        const size_t need32 = e2eft_attn_bwd_workspace_bytes(d);
        if (ws_bytes < need32) return fail(E2EFT_ERR_WORKSPACE, "attn_bwd: workspace %zu < %zu", ws_bytes, need32);
        return attn32_bwd(d, q, k, v, out, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, workspace, stream);
    }
    const int w = d->heads * 64;
    E2EFT_REQUIRE(d->ldq >= w && d->ldk >= w && d->ldv >= w && d->ldo >= w && lddo >= w && lddq >= w && lddk >= w && lddv >= w, "attn_bwd: row strides");
    E2EFT_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                  "attn_bwd: row strides must keep 16-byte (inputs) / 8-byte (outputs) alignment");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout) & 15) == 0 &&
                      (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) == 0, "attn_bwd: alignment");
    E2EFT_REQUIRE(d->heads <= 65535 && d->batch <= 65535, "attn_bwd: grid");
    const size_t need = e2eft_attn_bwd_workspace_bytes(d);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "attn_bwd: workspace %zu < %zu", ws_bytes, need);
    AttnBwdParams p;
    p.q = q; p.k = k; p.v = v; p.dout = dout; p.lse = lse; p.dsum = (const float*)workspace;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk = d->nk_seg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.scale = d->scale;
    p.c = d->scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)d->batch * d->nq;
    const dim3 gprep(cdiv(rows * d->heads, 256));
    const dim3 gkv(cdiv(p.nk, 128), d->heads, d->batch), gq(cdiv(p.nq, 128), d->heads, d->batch);
    if (d->dtype == E2EFT_F16) {
        hipLaunchKernelGGL((attn_bwd_prep_kernel<f16>), gprep, dim3(256), 0, s, rows, d->heads, d->ldo, lddo, d->nq, (const f16*)out, (const f16*)dout, (float*)workspace);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<f16>), gkv, dim3(256), 0, s, p);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<f16>), gq, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((attn_bwd_prep_kernel<bf16>), gprep, dim3(256), 0, s, rows, d->heads, d->ldo, lddo, d->nq, (const bf16*)out, (const bf16*)dout, (float*)workspace);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<bf16>), gkv, dim3(256), 0, s, p);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16>), gq, dim3(256), 0, s, p);
    }
    return check_launch("attn_bwd");
}
