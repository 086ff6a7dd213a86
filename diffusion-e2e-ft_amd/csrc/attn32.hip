// attn32.hip — fused attention forward and backward for head dim 64 in STRICT fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation), gfx950.
//
// Why: the reference's training recipe runs in fp32 (`--mixed_precision "no"`, training/scripts/train_marigold_e2e_ft_depth.sh:15) with xformers' fused attention
// (training/train.py:308-318); until round 4 the fp32 path here ran attention UNFUSED — q k^T, softmax, P v as batched GEMMs with the N x N matrices in HBM, and
// five more GEMMs plus transposes in the backward: at configs[2] (16 images, 5 heads, 5184 tokens) 8.6 GB per score matrix and pass, GEMMs with N = 64 or K = 64 at
// 60-75 TF/s — about 450 ms of the 3.64 s step (profiles/r05a_bench_train_fp32_per_shape.tsv).  Here nothing of size Nq x Nk leaves the registers.
//
// The fp32 MFMA takes ONE float per lane and operand (A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31], 64 cycles per instruction and SIMD), so
//   * any operand can come from a row-major LDS tile with one ds_read_b32 (or a quarter of a ds_read_b128) per MFMA — no transposed images, no transpose reads:
//     the kernels are plain loops of (LDS read, MFMA), bound by the matrix pipe;
//   * the contraction index of a product may be visited in any order as long as A and B agree: k-slot (step s, half hh) is element 32 hh + s of a 64-wide row,
//     so a lane's 32 operand values are 32 CONSECUTIVE floats of its row (eight 16-byte loads), and the key / query slot of the second product is chosen to be
//     exactly the accumulator row the first product left in that register (P and dS feed their MFMAs from registers, in fp32).
// Structure as attn.hip / attn_bwd.hip: both forward contractions swapped (S^T = K Q'^T, O^T = V^T P^T) so that softmax state is lane-local; backward = two
// owner-computes kernels (dK / dV with lane = key, dQ with lane = query), probabilities recomputed from q, k and the forward's base-2 log-sum-exp.
// LDS tiles are [64 rows][68 floats]: 272-byte rows keep ds_read_b128 of 16 different rows conflict-free (row starts 4 dwords apart mod 64) and 16-byte alignment.
// MFMA work per 32 x 32 (query, key) block: forward 2 x 32 instructions, backward 7 x 32; softmax VALU (one v_exp_f32 per score) is ~5 % beside them.
#include "common.h"

namespace e2eft {

namespace a32 {
constexpr int PITCH = 68;                 // floats per LDS row
constexpr int TILEF = 64 * PITCH;         // floats per 64-row tile
constexpr float LN2 = 0.6931471805599453f;
}  // namespace a32

struct Attn32Params {
    const float* q;
    const float* k;
    const float* v;
    float* out;
    int batch, heads, nq, nk_seg, kv_nseg, kv_bmod, nk_total, nqb;
    int ldq, ldk, ldv, ldo;
    float c;        // scale * log2(e)
    float* lse;     // optional [batch][heads][nq]: base-2 log-sum-exp of the scaled scores
};

struct Attn32BwdParams {
    const float* q;
    const float* k;
    const float* v;
    const float* dout;
    const float* lse;
    const float* dsum;
    float* dq;
    float* dk;
    float* dv;
    int batch, heads, nq, nk;
    int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
    float c, scale;
};

__device__ __forceinline__ floatx16 mma32(float a, float b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx16 zero16() {
    floatx16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
// accumulator row (within its 32-row block) of register r in half hh: the MFMA result layout D[i = (r & 3) + 8 (r >> 2) + 4 hh][j = lane & 31]
__device__ __forceinline__ constexpr int arow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// D[32 x 32] += A[32 rows x 64] B^T where A rows live in LDS (row i of the block at `arows + i * PITCH`, this lane reads row l31) and the lane's B values are bf[0..31]
// (= element 32 hh + s of ITS column's 64-vector): 8 ds_read_b128 + 32 MFMAs
__device__ __forceinline__ floatx16 mma_rows(const float* arow_l31_hh, const float (&bf)[32], floatx16 acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(arow_l31_hh + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mma32(a[e], bf[4 * j + e], acc);
    }
    return acc;
}
// D[32 x 32] += A^T-from-rows: contraction over the 32 ROWS of an LDS block whose slot (s, hh) is row arow(s, hh); A[i = l31][slot] = tile[row][col0 + l31],
// B = the lane's registers bv[0..15] (an accumulator of a previous product, used as is): 16 ds_read_b32 + 16 MFMAs
__device__ __forceinline__ floatx16 mma_cols(const float* block_col_l31, const int hh, const floatx16& bv, floatx16 acc) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mma32(block_col_l31[arow(s, hh) * a32::PITCH], bv[s], acc);
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// forward: one 256-thread workgroup = 128 queries of one (image, head), a wave 32 queries; 64-key tiles of K and V double-buffered in LDS
template <bool JOINT>
__global__ __launch_bounds__(256, 2) void attn32_fwd_kernel(const Attn32Params p) {
    using namespace a32;
    __shared__ __attribute__((aligned(16))) float smem[4 * TILEF];      // [buffer][K | V][64][68]: 69,632 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    int b, head, qblk;
    {   // XCD-aware block map (attn.hip): all query blocks of one (image, head) on one XCD
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
    const int qr = qblk * 128 + wave * 32 + l31;
    const bool qok = qr < p.nq;

    float qf[32];       // Q'[q][32 hh + s] = c * Q: the B operand of S^T = K Q'^T
    {
        const float* src = p.q + ((long)b * p.nq + (qok ? qr : 0)) * p.ldq + head * 64 + 32 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            floatx4 v = *reinterpret_cast<const floatx4*>(src + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * j + e] = qok ? v[e] * p.c : 0.f;
        }
    }
    // loader: thread t moves columns 16 (t & 3) .. + 15 of key row t >> 2 of the tile, for K and for V
    const int lrow = tid >> 2, lcol = 16 * (tid & 3);
    const int kvb0 = b % p.kv_bmod;
    floatx4 kreg[4], vreg[4];
    auto load_tile = [&](const int t) {
        const int key = t * 64 + lrow;
        const bool ok = key < p.nk_total;
        long row;
        if (JOINT) {
            const int seg = key / p.nk_seg;
            row = (long)(kvb0 + seg * p.kv_bmod) * p.nk_seg + (key - seg * p.nk_seg);
        } else {
            row = (long)kvb0 * p.nk_seg + key;
        }
        const float* kp = p.k + (ok ? row : 0) * p.ldk + head * 64 + lcol;
        const float* vp = p.v + (ok ? row : 0) * p.ldv + head * 64 + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const floatx4 z = {0.f, 0.f, 0.f, 0.f};
            kreg[i] = ok ? *reinterpret_cast<const floatx4*>(kp + 4 * i) : z;
            vreg[i] = ok ? *reinterpret_cast<const floatx4*>(vp + 4 * i) : z;
        }
    };
    auto store_tile = [&](const int buf) {
        float* kt = smem + buf * 2 * TILEF + lrow * PITCH + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<floatx4*>(kt + 4 * i) = kreg[i];
            *reinterpret_cast<floatx4*>(kt + TILEF + 4 * i) = vreg[i];
        }
    };

    const int nt = (p.nk_total + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    floatx16 o[2] = {zero16(), zero16()};
    float m_run = -INFINITY, l_run = 0.f;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const float* kt = smem + buf * 2 * TILEF;
        const float* vt = kt + TILEF;
        floatx16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) s[kb] = mma_rows(kt + (kb * 32 + l31) * PITCH + 32 * hh, qf, zero16());
        if (t * 64 + 64 > p.nk_total) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + kb * 32 + arow(r, hh) >= p.nk_total) s[kb][r] = -INFINITY;
        }
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                  // finite: every tile holds at least one key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: 2^-inf = 0
        float lsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
                lsum += s[kb][r];
            }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // O^T[d, q] += V^T[d, key] P^T[key, q]: slot (s, hh) of the contraction is key row arow(s, hh) of the block — the row register s of S^T belongs to
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = mma_cols(vt + kb * 32 * PITCH + dt * 32 + l31, hh, s[kb], o[dt]);
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (p.lse && hh == 0 && qok) p.lse[((long)b * p.heads + head) * p.nq + qr] = m_run + __builtin_amdgcn_logf(l_tot);
    if (qok) {
        float* dst = p.out + ((long)b * p.nq + qr) * p.ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 w = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                *reinterpret_cast<floatx4*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// backward, D[q] = sum_d dO[q, d] O[q, d] per (image, head, query): one thread per row
__global__ __launch_bounds__(256) void attn32_bwd_prep_kernel(long rows, int heads, int ldo, int lddo, int nq, const float* __restrict__ out,
                                                              const float* __restrict__ dout, float* __restrict__ dsum) {
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it >= rows * heads) return;
    const long row = it / heads;
    const int head = (int)(it - row * heads);
    const float* o = out + row * ldo + head * 64;
    const float* g = dout + row * lddo + head * 64;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(o + 4 * j), c = *reinterpret_cast<const floatx4*>(g + 4 * j);
        acc += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
    }
    const long b = row / nq, qi = row - b * nq;
    dsum[(b * heads + head) * nq + qi] = acc;
}

// dK / dV: a workgroup owns 128 keys (a wave 32, lane <-> key); 64-query tiles of Q' = c Q and dO (plus their lse / D) stream through LDS.
//   S[q, key] = Q' K^T, dP[q, key] = dO V^T          A = tile rows (ds_read_b128), B = the lane's K / V row (registers, loaded once)
//   P = 2^(S - lse[q]), dS = P o (dP - D[q])          registers = queries: lse / D of the lane's 16 query rows come from LDS
//   dV^T[d, key] += dO^T P, dK^T[d, key] += Q'^T dS    A = tile columns (ds_read_b32), B = P / dS registers;   dK = ln 2 * dK' because Q' carries c = scale * log2 e
__global__ __launch_bounds__(256, 2) void attn32_bwd_dkdv_kernel(const Attn32BwdParams p) {
    using namespace a32;
    constexpr int BUF = 2 * TILEF + 128;      // Q' tile, dO tile, lse[64], D[64]
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];          // 70,656 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int key = blockIdx.x * 128 + wave * 32 + l31;
    const bool kok = key < p.nk;
    float kf[32], vf[32];
    {
        const float* kp = p.k + ((long)b * p.nk + (kok ? key : 0)) * p.ldk + head * 64 + 32 * hh;
        const float* vp = p.v + ((long)b * p.nk + (kok ? key : 0)) * p.ldv + head * 64 + 32 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const floatx4 a = *reinterpret_cast<const floatx4*>(kp + 4 * j), c = *reinterpret_cast<const floatx4*>(vp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) { kf[4 * j + e] = kok ? a[e] : 0.f; vf[4 * j + e] = kok ? c[e] : 0.f; }
        }
    }
    const int lrow = tid >> 2, lcol = 16 * (tid & 3);
    floatx4 qreg[4], greg[4];
    float lreg = 0.f, dreg = 0.f;
    auto load_tile = [&](const int t) {
        const int qi = t * 64 + lrow;
        const bool ok = qi < p.nq;
        const float* qp = p.q + ((long)b * p.nq + (ok ? qi : 0)) * p.ldq + head * 64 + lcol;
        const float* gp = p.dout + ((long)b * p.nq + (ok ? qi : 0)) * p.lddo + head * 64 + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const floatx4 z = {0.f, 0.f, 0.f, 0.f};
            floatx4 a = ok ? *reinterpret_cast<const floatx4*>(qp + 4 * i) : z;
            qreg[i] = a * p.c;
            greg[i] = ok ? *reinterpret_cast<const floatx4*>(gp + 4 * i) : z;
        }
        if (tid < 64) {     // lse / D of query row `tid` of the tile; rows beyond nq: lse = +inf makes every probability of the row 0
            const int q2 = t * 64 + tid;
            const bool ok2 = q2 < p.nq;
            const long li = ((long)b * p.heads + head) * p.nq + (ok2 ? q2 : 0);
            lreg = ok2 ? p.lse[li] : INFINITY;
            dreg = ok2 ? p.dsum[li] : 0.f;
        }
    };
    auto store_tile = [&](const int buf) {
        float* qt = smem + buf * BUF + lrow * PITCH + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<floatx4*>(qt + 4 * i) = qreg[i];
            *reinterpret_cast<floatx4*>(qt + TILEF + 4 * i) = greg[i];
        }
        if (tid < 64) {
            smem[buf * BUF + 2 * TILEF + tid] = lreg;
            smem[buf * BUF + 2 * TILEF + 64 + tid] = dreg;
        }
    };

    const int nt = (p.nq + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    floatx16 dv[2] = {zero16(), zero16()}, dk[2] = {zero16(), zero16()};
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        // the next tile goes global -> registers -> LDS in one go, BEFORE this tile's products (the other buffer is free since the barrier that closed iteration
        // t - 1): the 34 staging registers are dead while the 160 registers of operands and accumulators work — with the load at the top and the store at the
        // bottom of the iteration the kernel spilled 106 registers.  The load latency is covered by the second workgroup of the CU.
        if (t + 1 < nt) { load_tile(t + 1); store_tile(buf ^ 1); }
        const float* qt = smem + buf * BUF;
        const float* gt = qt + TILEF;
        const float* lt = qt + 2 * TILEF;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            floatx16 s = mma_rows(qt + (qb * 32 + l31) * PITCH + 32 * hh, kf, zero16());      // S[q, key]: registers = queries arow(r, hh) of the block
            floatx16 dp = mma_rows(gt + (qb * 32 + l31) * PITCH + 32 * hh, vf, zero16());
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 l4 = *reinterpret_cast<const floatx4*>(lt + qb * 32 + 8 * g + 4 * hh);
                const floatx4 d4 = *reinterpret_cast<const floatx4*>(lt + 64 + qb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pr = __builtin_amdgcn_exp2f(s[4 * g + e] - l4[e]);
                    s[4 * g + e] = pr;
                    dp[4 * g + e] = pr * (dp[4 * g + e] - d4[e]);
                }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = mma_cols(gt + qb * 32 * PITCH + dt * 32 + l31, hh, s, dv[dt]);
                dk[dt] = mma_cols(qt + qb * 32 * PITCH + dt * 32 + l31, hh, dp, dk[dt]);
            }
        }
        __syncthreads();
    }
    if (kok) {
        float* dkp = p.dk + ((long)b * p.nk + key) * p.lddk + head * 64;
        float* dvp = p.dv + ((long)b * p.nk + key) * p.lddv + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 a = {dk[dt][4 * g] * LN2, dk[dt][4 * g + 1] * LN2, dk[dt][4 * g + 2] * LN2, dk[dt][4 * g + 3] * LN2};
                const floatx4 c = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
                *reinterpret_cast<floatx4*>(dkp + dt * 32 + 8 * g + 4 * hh) = a;
                *reinterpret_cast<floatx4*>(dvp + dt * 32 + 8 * g + 4 * hh) = c;
            }
    }
}

// dQ: the forward's structure (lane <-> query, 64-key tiles of K and V in LDS).
//   S^T[key, q] = K Q'^T, dP^T[key, q] = V dO^T        A = tile rows, B = the lane's Q' / dO row (registers)
//   dS^T = P^T o (dP^T - D[q])                          lane-local lse / D
//   dQ^T[d, q] += K^T dS^T                              A = K tile columns, B = dS^T registers;   dQ = scale * dQ'
__global__ __launch_bounds__(256, 2) void attn32_bwd_dq_kernel(const Attn32BwdParams p) {
    using namespace a32;
    __shared__ __attribute__((aligned(16))) float smem[4 * TILEF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qr = blockIdx.x * 128 + wave * 32 + l31;
    const bool qok = qr < p.nq;
    float qf[32], gf[32];
    {
        const float* qp = p.q + ((long)b * p.nq + (qok ? qr : 0)) * p.ldq + head * 64 + 32 * hh;
        const float* gp = p.dout + ((long)b * p.nq + (qok ? qr : 0)) * p.lddo + head * 64 + 32 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const floatx4 a = *reinterpret_cast<const floatx4*>(qp + 4 * j), c = *reinterpret_cast<const floatx4*>(gp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qf[4 * j + e] = qok ? a[e] * p.c : 0.f; gf[4 * j + e] = qok ? c[e] : 0.f; }
        }
    }
    const long li = ((long)b * p.heads + head) * p.nq + (qok ? qr : 0);
    const float lse_q = qok ? p.lse[li] : INFINITY, d_q = qok ? p.dsum[li] : 0.f;
    const int lrow = tid >> 2, lcol = 16 * (tid & 3);
    floatx4 kreg[4], vreg[4];
    auto load_tile = [&](const int t) {
        const int key = t * 64 + lrow;
        const bool ok = key < p.nk;
        const float* kp = p.k + ((long)b * p.nk + (ok ? key : 0)) * p.ldk + head * 64 + lcol;
        const float* vp = p.v + ((long)b * p.nk + (ok ? key : 0)) * p.ldv + head * 64 + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const floatx4 z = {0.f, 0.f, 0.f, 0.f};
            kreg[i] = ok ? *reinterpret_cast<const floatx4*>(kp + 4 * i) : z;
            vreg[i] = ok ? *reinterpret_cast<const floatx4*>(vp + 4 * i) : z;
        }
    };
    auto store_tile = [&](const int buf) {
        float* kt = smem + buf * 2 * TILEF + lrow * PITCH + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<floatx4*>(kt + 4 * i) = kreg[i];
            *reinterpret_cast<floatx4*>(kt + TILEF + 4 * i) = vreg[i];
        }
    };
    const int nt = (p.nk + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    floatx16 dq[2] = {zero16(), zero16()};
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const float* kt = smem + buf * 2 * TILEF;
        const float* vt = kt + TILEF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            floatx16 s = mma_rows(kt + (kb * 32 + l31) * PITCH + 32 * hh, qf, zero16());
            floatx16 dp = mma_rows(vt + (kb * 32 + l31) * PITCH + 32 * hh, gf, zero16());
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool in = t * 64 + kb * 32 + arow(r, hh) < p.nk;
                const float pr = in ? __builtin_amdgcn_exp2f(s[r] - lse_q) : 0.f;
                dp[r] = pr * (dp[r] - d_q);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = mma_cols(kt + kb * 32 * PITCH + dt * 32 + l31, hh, dp, dq[dt]);
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (qok) {
        float* dst = p.dq + ((long)b * p.nq + qr) * p.lddq + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 w = {dq[dt][4 * g] * p.scale, dq[dt][4 * g + 1] * p.scale, dq[dt][4 * g + 2] * p.scale, dq[dt][4 * g + 3] * p.scale};
                *reinterpret_cast<floatx4*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
            }
    }
}

// ---- host side (called from e2eft_attn_fwd_lse / e2eft_attn_bwd for dtype E2EFT_F32; argument checks common to all dtypes are done there) ----------------
int attn32_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream) {
    const int w = d->heads * 64;
    E2EFT_REQUIRE(d->ldq >= w && d->ldk >= w && d->ldv >= w && d->ldo >= w && d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && d->ldo % 4 == 0,
                  "attn (fp32): row strides must cover heads * 64 floats and keep 16-byte alignment");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "attn (fp32): pointers must be 16-byte aligned");
    Attn32Params p;
    p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.out = (float*)out; p.lse = lse;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk_seg = d->nk_seg; p.kv_nseg = d->kv_nseg; p.kv_bmod = d->kv_bmod;
    p.nk_total = d->nk_seg * d->kv_nseg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.c = d->scale * 1.4426950408889634f;
    p.nqb = cdiv(d->nq, 128);
    const long blocks = (long)d->batch * d->heads * p.nqb;
    E2EFT_REQUIRE(blocks < 2147483647L, "attn (fp32): grid");
    hipStream_t s = (hipStream_t)stream;
    if (d->kv_nseg > 1) hipLaunchKernelGGL((attn32_fwd_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn32_fwd_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return check_launch("attn_fwd (fp32)");
}

int attn32_bwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, const void* out, const void* dout, int32_t lddo, const float* lse, void* dq,
               int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv, void* workspace, void* stream) {
    const int w = d->heads * 64;
    E2EFT_REQUIRE(d->ldq >= w && d->ldk >= w && d->ldv >= w && d->ldo >= w && lddo >= w && lddq >= w && lddk >= w && lddv >= w, "attn_bwd (fp32): row strides");
    E2EFT_REQUIRE(d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && d->ldo % 4 == 0 && lddo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                  "attn_bwd (fp32): row strides must keep 16-byte alignment");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                  "attn_bwd (fp32): pointers must be 16-byte aligned");
    Attn32BwdParams p;
    p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.dout = (const float*)dout; p.lse = lse; p.dsum = (const float*)workspace;
    p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk = d->nk_seg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.scale = d->scale;
    p.c = d->scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)d->batch * d->nq;
    hipLaunchKernelGGL(attn32_bwd_prep_kernel, dim3((unsigned)cdiv(rows * d->heads, 256)), dim3(256), 0, s, rows, d->heads, d->ldo, lddo, d->nq, (const float*)out,
                       (const float*)dout, (float*)workspace);
    hipLaunchKernelGGL(attn32_bwd_dkdv_kernel, dim3(cdiv(p.nk, 128), d->heads, d->batch), dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn32_bwd_dq_kernel, dim3(cdiv(p.nq, 128), d->heads, d->batch), dim3(256), 0, s, p);
    return check_launch("attn_bwd (fp32)");
}

}  // namespace e2eft
