// EXPERIMENT, NOT BUILT INTO libe2eft.so (round 2).  Compiles (hipcc --offload-arch=gfx950 -O3 -I diffusion-e2e-ft_amd/csrc -c), never run on
// the GPU: with 256 fp32 accumulator registers per lane (O^T of 32 queries x 512 d) the online-softmax RESCALE `o *= alpha` makes hipcc spill
// 445 VGPRs (876 B of scratch per lane); the same kernel without the rescale spills 6.  MFMA accumulators live in AGPRs, VALU cannot
// operate on AGPRs, and the allocator does not stream the 256 values through a few VGPRs (per-block sched_barriers change nothing).
// Ways on, with their price: 16x16x32 MFMA tiles (16 queries per wave, everything fits 256 registers, but 1 KB of LDS
// reads per 16-cycle MFMA = 250 B/clk/CU, the LDS ceiling); exact two-pass softmax (row maxima first: no rescale, 1.5x the MFMA work);
// pairs of waves splitting d (halves the accumulators, needs an LDS exchange of the partial S^T and 64-query workgroups: 2x the K/V
// traffic).  The VAE mid-block attention therefore still runs unfused (QK^T -> row softmax -> PV, 4.6 ms of a 118 ms step).
//
// attn512.hip — fused attention forward for head dim 512 (fp16 / bf16): the mid-block attention of the SD VAE (one 512-wide head over
// H*W tokens, /root/reference/GeoWizard/geowizard/models/unet_2d_blocks.py:589-601 builds it; at 768^2 input 9216 tokens per image,
// twice per path: encoder and decoder).  Unfused it was QK^T -> [B, 9216, 9216] scores (1.36 GB at batch 8) -> row softmax in place ->
// PV: 5.4 GB of HBM traffic and 4.6 ms per inference step; here the scores never leave the registers.
//
// Same scheme as attn.hip (swapped MFMAs: S^T = K Q^T, O^T += V^T P^T, so the softmax state of a query is lane-local and P feeds the
// second MFMA straight from the S^T registers), scaled to d = 512:
//   * one 256-thread workgroup = 128 queries, a wave owns 32; ONE wave per SIMD: the O^T accumulator alone is 16 MFMA blocks = 256
//     registers (AGPRs), the wave's Q^T fragments another 128 VGPRs;
//   * 32-key tiles: K [32][512] and V^T [512][32] double-buffered in LDS (2 x 69 KB), padded row pitches (1040 B / 72 B) keep the
//     ds_read_b128 / b64 fragment reads conflict-free;
//   * per tile and wave 64 MFMAs (32 for S^T over the 512-deep contraction, 32 for O^T over 16 d-blocks) against 16 exponentials per lane:
//     unlike d = 64 (512 MFMA cycles vs 528 exp cycles per tile) this kernel is MFMA-bound.
#include "common.h"

namespace e2eft {

namespace a5 {
constexpr int D = 512, KT = 32;
constexpr int KROW = D * 2 + 16;        // 1040 B per key row
constexpr int VROW = KT * 2 + 8;        // 72 B per d row of V^T
constexpr int KTILE = KT * KROW;        // 33280
constexpr int VTILE = D * VROW;         // 36864
constexpr int BUF = KTILE + VTILE;      // 70144
}  // namespace a5

struct Attn5Params {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int batch, heads, nq, nk;
    int ldq, ldk, ldv, ldo;
    float c;  // scale * log2(e)
};

template <typename T> struct Mma5;
template <> struct Mma5<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
    }
};
template <> struct Mma5<bf16> {
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

// grid (ceil(nq / 128), heads, batch), 256 threads
template <typename T>
__global__ __launch_bounds__(256) void attn512_fwd_kernel(const Attn5Params p) {
    using namespace a5;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const T* __restrict__ Q = (const T*)p.q + head * D;
    const T* __restrict__ K = (const T*)p.k + head * D;
    const T* __restrict__ V = (const T*)p.v + head * D;

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (q = l31, hh) holds Q[q][16 ds + 8 hh .. + 7], ds = 0 .. 31 ----
    u32x4 qf[32];
    {
        const int qr = q0 + l31;
        const bool ok = qr < p.nq;
        const T* src = Q + ((long)b * p.nq + (ok ? qr : 0)) * p.ldq + 8 * hh;
#pragma unroll
        for (int ds = 0; ds < 32; ++ds) qf[ds] = ok ? *reinterpret_cast<const u32x4*>(src + 16 * ds) : u32x4{0u, 0u, 0u, 0u};
    }

    // ---- loaders: K tile 32 keys x 64 chunks (16 B): thread -> chunk tid & 63, keys (tid >> 6) + 4 i, i < 8
    //               V tile: thread -> key pair kp = tid & 15 (keys 2 kp, 2 kp + 1), d chunks (tid >> 4) + 16 j, j < 4 (8 d values each)
    const int k_c = tid & 63, k_r = tid >> 6;
    const int v_kp = tid & 15, v_c = tid >> 4;
    // ONE 8 x 16-byte staging set per thread, used for the K rows of the next tile during the S^T phase and again for its V rows during the
    // O^T phase (the wave has 256 VGPRs next to the 256 accumulator registers; Q^T fragments take 128 of them)
    u32x4 st[8];
    auto load_k = [&](int t) {
        const int base = t * KT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = min(base + k_r + 4 * i, p.nk - 1);          // keys past the end are masked below (p = 0)
            st[i] = *reinterpret_cast<const u32x4*>(K + ((long)b * p.nk + j) * p.ldk + k_c * 8);
        }
    };
    auto store_k = [&](int buf) {
        char* sk = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(sk + (k_r + 4 * i) * KROW + k_c * 16) = st[i];
    };
    auto load_v = [&](int t) {
        const int base = t * KT;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = min(base + 2 * v_kp + i, p.nk - 1);
                st[jj * 2 + i] = *reinterpret_cast<const u32x4*>(V + ((long)b * p.nk + j) * p.ldv + (v_c + 16 * jj) * 8);
            }
    };
    auto store_v = [&](int buf) {   // transpose: V^T[8 c + e][2 kp, 2 kp + 1] — one v_perm_b32 per (d, key pair)
        char* sv = smem + buf * BUF + KTILE;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t w = __builtin_amdgcn_perm(st[jj * 2 + 1][e >> 1], st[jj * 2][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
                *reinterpret_cast<uint32_t*>(sv + (8 * (v_c + 16 * jj) + e) * VROW + v_kp * 4) = w;
            }
    };

    floatx16 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (p.nk + KT - 1) / KT;
    load_k(0);
    store_k(0);
    load_v(0);
    store_v(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_k(t + 1);
        const char* sk = smem + buf * BUF;
        const char* sv = sk + KTILE;

        // ---- S^T[key, q] = K Q^T over d = 512: 32 MFMAs ----
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            const char* row = sk + l31 * KROW + hh * 16;
#pragma unroll
            for (int ds = 0; ds < 32; ++ds) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(row + ds * 32);
                s = Mma5<T>::run(kf, qf[ds], s);
            }
        }
        if (t * KT + KT > p.nk) {   // last tile: keys beyond nk
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = t * KT + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (j >= p.nk) s[r] = -INFINITY;
            }
        }
        // ---- online softmax (lane-local: a lane's 16 values are 16 keys of ONE query; the other 16 keys sit in lane ^ 32) ----
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
        const float mc = m_new * p.c;
        m_run = m_new;
        uint32_t pw[8];
        float psum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float e0 = __builtin_amdgcn_exp2f(fmaf(s[2 * w], p.c, -mc));
            const float e1 = __builtin_amdgcn_exp2f(fmaf(s[2 * w + 1], p.c, -mc));
            pw[w] = Mma5<T>::pack(e0, e1);
            psum += e0 + e1;
        }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // the running max settles after a few tiles: skip the 256-register rescale then
#pragma unroll
            for (int dt = 0; dt < 16; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        if (more) { store_k(buf ^ 1); load_v(t + 1); }
        // ---- O^T[d, q] += V^T P^T: 16 d-blocks x 2 k-steps of 16 keys ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 pf = {pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
            const int kb = (16 * s2 + 4 * hh) * 2;   // byte offset of this lane's first 4-key run (the key -> MFMA-k-slot permutation of the S^T registers)
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const char* vrow = sv + (dt * 32 + l31) * VROW + kb;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow);
                const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
                const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                o[dt] = Mma5<T>::run(vf, pf, o[dt]);
            }
        }
        if (more) store_v(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O / l, 8-byte stores of 4 consecutive d ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int qr = q0 + l31;
    if (qr < p.nq) {
        T* dst = (T*)p.out + ((long)b * p.nq + qr) * p.ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = Mma5<T>::pack(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
                w[1] = Mma5<T>::pack(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * g + 4 * hh) = w;
            }
    }
}

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_attn512_fwd(const E2eftAttnDesc* d, const void* q, const void* k, const void* v, void* out, void* stream) {
    E2EFT_REQUIRE(d && q && k && v && out, "attn512: null pointer");
    E2EFT_REQUIRE(d->dtype == E2EFT_F16 || d->dtype == E2EFT_BF16, "attn512: dtype %d unsupported (fp16 / bf16; fp32 uses the unfused path)", d->dtype);
    E2EFT_REQUIRE(d->batch > 0 && d->heads > 0 && d->nq > 0 && d->nk_seg > 0, "attn512: geometry");
    E2EFT_REQUIRE(d->kv_nseg == 1 && d->kv_bmod == d->batch, "attn512: plain self / cross attention only (kv_nseg = 1, kv_bmod = batch)");
    const int w = d->heads * 512;
    E2EFT_REQUIRE(d->ldq >= w && d->ldk >= w && d->ldv >= w && d->ldo >= w, "attn512: row strides smaller than heads*512");
    E2EFT_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0, "attn512: row strides must be multiples of 8");
    E2EFT_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0, "attn512: alignment");
    E2EFT_REQUIRE(d->heads <= 65535 && d->batch <= 65535 && d->scale > 0.f, "attn512: grid / scale");
    Attn5Params p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.batch = d->batch; p.heads = d->heads; p.nq = d->nq; p.nk = d->nk_seg;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.c = d->scale * 1.4426950408889634f;
    dim3 grid(cdiv(d->nq, 128), d->heads, d->batch);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == E2EFT_F16) hipLaunchKernelGGL((attn512_fwd_kernel<f16>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn512_fwd_kernel<bf16>), grid, dim3(256), 0, s, p);
    return check_launch("attn512_fwd");
}
