// igemm.h — parameter block shared by the implicit-GEMM kernel variants.
#pragma once
#include "common.h"

namespace e2eft {

struct IgemmParams {
    const void* x1;
    const void* x2;
    const void* w;
    const void* bias;
    const void* rowadd;
    const void* residual;
    void* out;
    int M, N, K;
    int ldx1, ldx2, c1, cin;
    int hin, win, hl, wl, kh, kw, stride, pad_t, pad_l, hout, wout;
    float up_sh, up_sw;
    int ldw, ldr, ldo;
    int bias_along_m;
    int rows_per_img;
    float alpha;
    int nzi;
    long sa_o, sa_i, sw_o, sw_i, so_o, so_i, sr_o, sr_i;
    int mtiles, ntiles;
};

// ---------------------------------------------------------------------------------------------------------------
// Shared epilogue: the 32x32 MFMA accumulator layout (lane = column, registers = rows) would give 2-byte scattered
// stores and one dependent residual load per element.  Instead the fp32 tile goes through LDS once (the k-loop is
// finished, its buffers are free) and is written back row-wise: every thread owns 8 consecutive columns of a row,
// residual / output move as 16-byte vectors, a wave covers 4 full 256-byte rows per instruction.
//   out = alpha * (acc + bias + rowadd[img(m)]) + residual
template <typename T, int BM_, int BN_, int NT_>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, char* smem, floatx16 (&acc)[2][2], int wm, int wn, int l31,
                                               int h, int m0, int n0, int zo, int zi, bool is_consumer = true) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int LDT = BN_ + 4;            // fp32 row stride of the staged tile (528 B for BN = 128)
    constexpr int CPR = BN_ / 8;            // 8-column chunks per row
    constexpr int RPP = NT_ / CPR;          // rows per pass
    float* tile = reinterpret_cast<float*>(smem);
    __syncthreads();                        // every wave is done reading the last k-tile
    if (is_consumer) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDT + wn * 64 + j * 32 + l31] = acc[i][j][r];
    }
    __syncthreads();
    if (!is_consumer) return;               // dedicated loader waves (igemm2 NL > 0) only take part in the barriers

    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ rowadd = (const T*)p.rowadd;
    const T* __restrict__ res = p.residual ? (const T*)p.residual + zo * p.sr_o + zi * p.sr_i : nullptr;
    T* __restrict__ out = (T*)p.out + zo * p.so_o + zi * p.so_i;
    const int chunk = threadIdx.x % CPR, rbase = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    if (n >= p.N) return;
    const bool vec = (p.N % 8 == 0) && (p.ldo % EPC == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     (!res || (p.ldr % EPC == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    const int nv = min(8, p.N - n);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (bias && !p.bias_along_m && e < nv) ? to_f(bias[n + e]) : 0.f;
#pragma unroll 4
    for (int pass = 0; pass < BM_ / RPP; ++pass) {
        const int row = rbase + pass * RPP;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const floatx4 t0 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8);
        const floatx4 t1 = *reinterpret_cast<const floatx4*>(tile + row * LDT + chunk * 8 + 4);
        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        const float bm = (bias && p.bias_along_m) ? to_f(bias[m]) : 0.f;
        const T* ra = rowadd ? rowadd + (long)(m / p.rows_per_img) * p.N + n : nullptr;
        if (vec) {
            float rv[8];
            if (res) {
#pragma unroll
                for (int q = 0; q < 8 / EPC; ++q) {
                    Vec16<T> t = ld16(res + (long)m * p.ldr + n + q * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) rv[q * EPC + e] = to_f(t.e[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e] + bv[e] + bm;
                if (ra) x += to_f(ra[e]);
                x *= p.alpha;
                if (res) x += rv[e];
                v[e] = x;
            }
#pragma unroll
            for (int q = 0; q < 8 / EPC; ++q) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(v[q * EPC + e]);
                st16(out + (long)m * p.ldo + n + q * EPC, o);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < nv) {
                    float x = v[e] + bv[e] + bm;
                    if (ra) x += to_f(ra[e]);
                    x *= p.alpha;
                    if (res) x += to_f(res[(long)m * p.ldr + n + e]);
                    out[(long)m * p.ldo + n + e] = from_f<T>(x);
                }
            }
        }
    }
}

// v2: 256x128 tile, 8 waves, LDS-DMA (global_load_lds) 3-stage ring — igemm2.hip
int launch_igemm_v2(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);

}  // namespace e2eft
