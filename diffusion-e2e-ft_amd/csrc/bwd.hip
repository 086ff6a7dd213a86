// bwd.hip — HBM-bound pieces of the E2E-FT backward pass (training/train.py:470-568): the data movement that turns NHWC
// activations into the K-contiguous operands of the weight-gradient GEMMs (pixel index contiguous), column sums for bias /
// time-embedding gradients, nearest-upsample / SiLU / head / loss gradients, and the flat-buffer AdamW update.
//
// Weight gradients contract over PIXELS, the slow index of an NHWC tensor, while the MFMA GEMM kernel wants both operands
// K-contiguous.  So dW = dY^T · im2col(X) runs as   transpose(dY) [Cout, P]  x  im2col_t(X) [kh*kw*Cin, P]  through e2eft_gemm,
// whose output [Cout, kh*kw*Cin] is the OHWI weight layout itself.  Both producers are 64x64 LDS-tile transposes: 16-byte reads
// along channels, 16-byte writes along pixels.
#include "common.h"

namespace e2eft {

// ---------------------------------------------------------------------------------------------------------------
// 64 (rows) x 64 (cols) tile: src(r, c) yields the 16-byte chunk at row r0+r, columns c0+c..; written transposed as
// out[(c0 + c) * ld_out + r0 + r], rows >= rows_valid (up to the 64-row tile edge / rows_pad) are zero.
template <typename T> struct TileGeom {
    static constexpr int EPC = 16 / (int)sizeof(T);
    static constexpr int PITCH = 64 + 4 / (int)sizeof(T);   // elements; odd dword pitch -> conflict-free transposed reads
};

template <typename T, typename Src>
__device__ __forceinline__ void transpose_tile(T* tile, const Src& src, long r0, int c0, int cols, long rows_pad, T* __restrict__ out,
                                               long ld_out) {
    constexpr int EPC = TileGeom<T>::EPC, PITCH = TileGeom<T>::PITCH;
    constexpr int CPR = 64 / EPC;        // chunks per tile row
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < (64 * CPR) / 256; ++it) {
        const int idx = tid + it * 256;
        const int r = idx / CPR, cc = (idx % CPR) * EPC;
        Vec16<T> v;
        v.raw = u32x4{0u, 0u, 0u, 0u};
        if (c0 + cc < cols) v = src(r0 + r, c0 + cc);
        uint32_t* d = reinterpret_cast<uint32_t*>(tile + r * PITCH + cc);
        d[0] = v.raw[0]; d[1] = v.raw[1]; d[2] = v.raw[2]; d[3] = v.raw[3];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (64 * CPR) / 256; ++it) {
        const int idx = tid + it * 256;
        const int rc = (idx % CPR) * EPC, c = idx / CPR;
        if (c0 + c < cols && r0 + rc < rows_pad) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = tile[(rc + e) * PITCH + c];
            st16(out + (long)(c0 + c) * ld_out + r0 + rc, o);
        }
    }
}

// in[z][r][c] -> out[z][c][r]; grid (ceil(rows_pad/64), ceil(cols/64), batch)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(long rows, int cols, long ld_in, long bs_in, long rows_pad, long ld_out, long bs_out,
                                                        const T* __restrict__ in, T* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) T tile[64 * TileGeom<T>::PITCH];
    const T* src = in + (long)blockIdx.z * bs_in;
    auto ld = [&](long r, int c) {
        Vec16<T> v;
        v.raw = u32x4{0u, 0u, 0u, 0u};
        if (r < rows) v = ld16(src + r * ld_in + c);
        return v;
    };
    transpose_tile<T>(tile, ld, (long)blockIdx.x * 64, blockIdx.y * 64, cols, rows_pad, out + (long)blockIdx.z * bs_out, ld_out);
}

struct Im2colGeom {
    int batch, hin, win, hl, wl, c1, ldx1, c2, ldx2, kh, kw, stride, pad_t, pad_l, hout, wout;
    float up_sh, up_sw;
};

// col[(tap * Cin + c)][p], p = (b, oy, ox) over the conv OUTPUT pixels (zero beyond P up to Ppad); the gather is the forward's
// (padding, stride, fused nearest upsample, two-source concat).  grid (ceil(Ppad/64), ceil(Cin/64), kh*kw)
template <typename T>
__global__ __launch_bounds__(256) void im2col_t_kernel(Im2colGeom g, long P, long Ppad, long ldcol, const T* __restrict__ x1,
                                                       const T* __restrict__ x2, T* __restrict__ col) {
    __shared__ __attribute__((aligned(16))) T tile[64 * TileGeom<T>::PITCH];
    const int tap = blockIdx.z;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int cin = g.c1 + g.c2;
    const int hw = g.hout * g.wout;
    auto ld = [&](long p, int c) {
        Vec16<T> v;
        v.raw = u32x4{0u, 0u, 0u, 0u};
        if (p >= P) return v;
        const int b = (int)(p / hw);
        const int rem = (int)(p - (long)b * hw);
        const int oy = rem / g.wout, ox = rem - oy * g.wout;
        const int iy = oy * g.stride - g.pad_t + ky, ix = ox * g.stride - g.pad_l + kx;
        if ((unsigned)iy >= (unsigned)g.hl || (unsigned)ix >= (unsigned)g.wl) return v;
        int sy = iy, sx = ix;
        if (g.hl != g.hin) sy = min((int)floorf(iy * g.up_sh), g.hin - 1);
        if (g.wl != g.win) sx = min((int)floorf(ix * g.up_sw), g.win - 1);
        const long pix = ((long)b * g.hin + sy) * g.win + sx;
        return c < g.c1 ? ld16(x1 + pix * g.ldx1 + c) : ld16(x2 + pix * g.ldx2 + (c - g.c1));
    };
    transpose_tile<T>(tile, ld, (long)blockIdx.x * 64, blockIdx.y * 64, cin, Ppad, col + (long)tap * cin * ldcol, ldcol);
}

// ---------------------------------------------------------------------------------------------------------------
// column sums per row group: part[g][slab][c] (fp32); grid (ceil(nchunks/32), groups, nslabs), block = 32 chunk lanes x 8 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(long rows_per_group, int cols, long ld, long slab, const T* __restrict__ x,
                                                             float* __restrict__ part) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float sm[8][32 * EPC];
    const int chl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + chl) * EPC;
    const long r0 = (long)blockIdx.z * slab;
    const long r1 = min(r0 + slab, rows_per_group);
    const T* src = x + (long)blockIdx.y * rows_per_group * ld;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    if (c < cols) {
        for (long r = r0 + rl; r < r1; r += 8) {
            Vec16<T> v = ld16(src + r * ld + c);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += to_f(v.e[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) sm[rl][chl * EPC + e] = s[e];
    __syncthreads();
    for (int j = threadIdx.x; j < 32 * EPC; j += 256) {
        const int cc = blockIdx.x * 32 * EPC + j;
        if (cc < cols) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) a += sm[q][j];
            part[((long)blockIdx.y * gridDim.z + blockIdx.z) * cols + cc] = a;
        }
    }
}

// out[g][j] = alpha * sum_slab part[g][slab][j]; grid (ceil(n/256), groups)
__global__ __launch_bounds__(256) void colsum_final_kernel(int nslabs, int n, float alpha, const float* __restrict__ part, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* p = part + (long)blockIdx.y * nslabs * n;
    float a = 0.f;
    for (int r = 0; r < nslabs; ++r) a += p[(long)r * n + j];
    out[(long)blockIdx.y * n + j] = a * alpha;
}

// ---------------------------------------------------------------------------------------------------------------
// nearest-upsample backward: dx[b, sy, sx, :] = sum of dy over the logical pixels that read (sy, sx) in the forward gather
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(int batch, int hin, int win, int hl, int wl, int c, int lddy, int lddx, float up_sh,
                                                           float up_sw, const T* __restrict__ dy, T* __restrict__ dx) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cch = c / EPC;
    const long total = (long)batch * hin * win * cch;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pix = it / cch;
        const int ch = (int)(it - pix * cch) * EPC;
        const int b = (int)(pix / ((long)hin * win));
        const int rem = (int)(pix - (long)b * hin * win);
        const int sy = rem / win, sx = rem - sy * win;
        const int y0 = max(0, (int)floorf(sy / up_sh) - 1), y1 = min(hl - 1, (int)ceilf((sy + 1) / up_sh) + 1);
        const int x0 = max(0, (int)floorf(sx / up_sw) - 1), x1 = min(wl - 1, (int)ceilf((sx + 1) / up_sw) + 1);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int iy = y0; iy <= y1; ++iy) {
            if (min((int)floorf(iy * up_sh), hin - 1) != sy) continue;
            for (int ix = x0; ix <= x1; ++ix) {
                if (min((int)floorf(ix * up_sw), win - 1) != sx) continue;
                Vec16<T> v = ld16(dy + (((long)b * hl + iy) * wl + ix) * lddy + ch);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += to_f(v.e[e]);
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(acc[e]);
        st16(dx + pix * lddx + ch, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_bwd_kernel(long n, const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx) {
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) {
        const float z = to_f(x[it]);
        const float sg = 1.f / (1.f + __expf(-z));
        dx[it] = from_f<T>(to_f(dy[it]) * sg * (1.f + z * (1.f - sg)));
    }
}

// depth head backward (training/train.py:531-533): y = clamp(mean_c x, -1, 1) [* 0.5 + 0.5]; dx_c = dy/3 inside the clamp range
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void depth_head_bwd_kernel(long pixels, int ldx, int lddx, int cpad, int to_unit, const TI* __restrict__ x,
                                                             const TO* __restrict__ dy, TI* __restrict__ dx) {
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < pixels; it += (long)gridDim.x * 256) {
        const TI* s = x + it * ldx;
        const float v = (to_f(s[0]) + to_f(s[1]) + to_f(s[2])) / 3.0f;
        float g = (v >= -1.f && v <= 1.f) ? to_f(dy[it]) / 3.0f : 0.f;
        if (to_unit) g *= 0.5f;
        TI* d = dx + it * lddx;
        for (int ch = 0; ch < cpad; ++ch) d[ch] = from_f<TI>(ch < 3 ? g : 0.f);
    }
}

// normal head backward (train.py:535-538): n = x / (|x| + 1e-5), y = sign * clamp(n, -1, 1)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void normal_head_bwd_kernel(int batch, int hw, int ldx, int lddx, int cpad, int clampv, float sign,
                                                              const TI* __restrict__ x, const TO* __restrict__ dy, TI* __restrict__ dx) {
    const long total = (long)batch * hw;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long b = it / hw, pix = it - b * hw;
        const TI* s = x + it * ldx;
        const float n[3] = {to_f(s[0]), to_f(s[1]), to_f(s[2])};
        const float r = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const float inv = 1.0f / (r + 1e-5f);
        float dn[3], dot = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float o = n[ch] * inv;
            dn[ch] = to_f(dy[(b * 3 + ch) * hw + pix]) * sign;
            if (clampv && !(o >= -1.f && o <= 1.f)) dn[ch] = 0.f;
            dot += dn[ch] * n[ch];
        }
        const float k = r > 0.f ? dot * inv * inv / r : 0.f;
        TI* d = dx + it * lddx;
        for (int ch = 0; ch < cpad; ++ch) d[ch] = from_f<TI>(ch < 3 ? dn[ch] * inv - n[ch] * k : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// losses (training/util/loss.py), backward.  fp32 I/O, fp64 reductions like the forward.
__device__ __forceinline__ double wave_sum_d2(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// per image: Gs = sum m sign(r) p, Gh = sum m sign(r), r = s p + h - t
__global__ __launch_bounds__(256) void ssi_bwd_sums_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                           const uint8_t* __restrict__ mask, const float* __restrict__ ss,
                                                           double* __restrict__ gsum /* [B][2] */) {
    __shared__ double red[4][2];
    const int b = blockIdx.y;
    const float sc = ss[b * 2], sh = ss[b * 2 + 1];
    const float* p = pred + (long)b * hw;
    const float* t = tgt + (long)b * hw;
    const uint8_t* m = mask + (long)b * hw;
    double v0 = 0, v1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        if (m[i]) {
            const float r = sc * p[i] + sh - t[i];
            const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
            v0 += (double)(sg * p[i]);
            v1 += (double)sg;
        }
    }
    v0 = wave_sum_d2(v0); v1 = wave_sum_d2(v1);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v0; red[threadIdx.x >> 6][1] = v1; }
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(gsum + b * 2 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// grad_j = gout * m_j * (c0 sign(r_j) + c1 + c2 t_j + c3 p_j)  — the chain through the closed-form scale/shift included
__global__ __launch_bounds__(256) void ssi_bwd_apply_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ ss,
                                                            const double* __restrict__ sums /* [B][5] */, const double* __restrict__ acc /* [2] */,
                                                            const double* __restrict__ gsum, const float* __restrict__ gout,
                                                            float* __restrict__ dpred) {
    const int b = blockIdx.y;
    const double s = ss[b * 2], h = ss[b * 2 + 1];
    const double a00 = sums[b * 5], a01 = sums[b * 5 + 1], a11 = sums[b * 5 + 2], b0 = sums[b * 5 + 3], b1 = sums[b * 5 + 4];
    const double det = (double)((float)a00 * (float)a11 - (float)a01 * (float)a01);   // the forward's fp32 determinant decides validity
    const double nv = acc[1];
    const double Gs = gsum[b * 2], Gh = gsum[b * 2 + 1];
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const bool live = nv > 0.0 && !isnan(acc[0]);     // skipped loss term (no valid pixel / NaN loss, train.py:504,548): zero gradient
    if (det > 0.0 && live) {
        const double dd = a00 * a11 - a01 * a01;
        c0 = s / nv;
        c1 = (Gs * (-b1 + 2 * s * a01) + Gh * (-b0 + 2 * h * a01)) / (dd * nv);
        c2 = (Gs * a11 - Gh * a01) / (dd * nv);
        c3 = (-2 * s * a11 * Gs + Gh * (2 * b1 - 2 * h * a11)) / (dd * nv);
    }
    const float go = gout[0];
    const float f0 = (float)c0 * go, f1 = (float)c1 * go, f2 = (float)c2 * go, f3 = (float)c3 * go;
    const float sc = (float)s, sh = (float)h;
    const float* p = pred + (long)b * hw;
    const float* t = tgt + (long)b * hw;
    const uint8_t* m = mask + (long)b * hw;
    float* d = dpred + (long)b * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float g = 0.f;
        if (live && m[i]) {
            const float r = sc * p[i] + sh - t[i];
            const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
            g = f0 * sg + f1 + f2 * t[i] + f3 * p[i];
        }
        d[i] = g;
    }
}

// d/dp_c acos(clamp(p.t)) = -t_c / sqrt(1 - d^2) inside the clamp range (loss.py:61-66)
__global__ __launch_bounds__(256) void angular_bwd_kernel(int hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                          const uint8_t* __restrict__ mask, const double* __restrict__ acc,
                                                          const float* __restrict__ gout, float* __restrict__ dpred) {
    const int b = blockIdx.y;
    const float* p = pred + (long)b * 3 * hw;
    const float* t = tgt + (long)b * 3 * hw;
    const uint8_t* m = mask + (long)b * hw;
    float* d = dpred + (long)b * 3 * hw;
    const bool live = acc[1] > 0.0 && !isnan(acc[0]);   // skipped loss term (train.py:504,552): zero gradient
    const float k = live ? gout[0] / (float)acc[1] : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float g = 0.f;
        if (live && m[i]) {
            const float dt = p[i] * t[i] + p[hw + i] * t[hw + i] + p[2 * hw + i] * t[2 * hw + i];
            if (dt >= -1.f && dt <= 1.f) g = -k / sqrtf(1.f - dt * dt);
        }
        d[i] = live ? g * t[i] : 0.f;
        d[hw + i] = live ? g * t[hw + i] : 0.f;
        d[2 * hw + i] = live ? g * t[2 * hw + i] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// flat-buffer optimizer: all UNet parameters / gradients live in one fp32 buffer each (one RCCL all-reduce, one update launch)
__global__ __launch_bounds__(256) void sumsq_kernel(long n, const float* __restrict__ g, double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += (double)g[i] * (double)g[i];
    a = wave_sum_d2(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.AdamW semantics (decoupled decay), gradient pre-scaled by min(1, max_norm / (||g|| + 1e-6)) when sumsq is given
// (accelerator.clip_grad_norm_, train.py:562-563)
__global__ __launch_bounds__(256) void adamw_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, const double* __restrict__ sumsq, float gscale, float max_norm) {
    float clip = gscale;
    // a non-finite gradient norm (overflow in a 16-bit forward, NaN input) must not reach the master weights and moments: skip the
    // whole update, as the reference's step does nothing useful either once its loss term was dropped (train.py:548-553)
    if (sumsq && !isfinite(sumsq[0])) return;
    if (sumsq && max_norm > 0.f) {
        const float nrm = (float)sqrt(sumsq[0]) * gscale;
        clip *= fminf(1.f, max_norm / (nrm + 1e-6f));
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * clip;
        float pi = p[i];
        pi -= lr * wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// Guarded form with the step counter ON THE DEVICE (e2eft_adamw_step_guarded).  state[0] = number of APPLIED steps (the `step` of
// torch.optim.AdamW's bias correction), state[1] = number of SKIPPED steps.  adamw_prepare_kernel (one thread) looks at the gradient's
// sum of squares: finite -> state[0] += 1, coef = {clip factor, 1 - beta1^step, sqrt(1 - beta2^step), 1}; not finite (overflow in a 16-bit
// forward, NaN input) -> state[1] += 1, coef[3] = 0 and adamw_guarded_kernel leaves parameters AND moments alone.  So a skipped step does not
// advance the bias correction, the skip is visible to the caller (state[1]), and the host never waits for the device.
__global__ void adamw_prepare_kernel(long long* __restrict__ state, float* __restrict__ coef, const double* __restrict__ sumsq, float b1, float b2,
                                     float gscale, float max_norm) {
    const double ss = sumsq[0];
    if (!isfinite(ss)) {
        state[1] += 1;
        coef[0] = 0.f; coef[1] = 1.f; coef[2] = 1.f; coef[3] = 0.f;
        return;
    }
    const long long step = state[0] + 1;
    state[0] = step;
    float clip = gscale;
    if (max_norm > 0.f) clip *= fminf(1.f, max_norm / ((float)sqrt(ss) * gscale + 1e-6f));
    coef[0] = clip;
    coef[1] = (float)(1.0 - pow((double)b1, (double)step));
    coef[2] = (float)sqrt(1.0 - pow((double)b2, (double)step));
    coef[3] = 1.f;
}

__global__ __launch_bounds__(256) void adamw_guarded_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, float lr, float b1, float b2, float eps, float wd,
                                                            const float* __restrict__ coef) {
    if (coef[3] == 0.f) return;
    const float clip = coef[0], bc1 = coef[1], bc2_sqrt = coef[2];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * clip;
        float pi = p[i];
        pi -= lr * wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(long n, float mul, int accumulate, const TI* __restrict__ x, TO* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = to_f(x[i]) * mul;
        if (accumulate) v += to_f(y[i]);
        y[i] = from_f<TO>(v);
    }
}

static unsigned grid_for(long total) {
    long nb = (total + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 16384) nb = 16384;
    return (unsigned)nb;
}

#define E2EFT_DISPATCH2(dti, dto, TI, TO, ...) E2EFT_DISPATCH_DTYPE(dti, TI, { E2EFT_DISPATCH_DTYPE(dto, TO, __VA_ARGS__); })

}  // namespace e2eft

using namespace e2eft;

extern "C" int e2eft_transpose(int32_t dtype, int32_t batch, int64_t rows, int32_t cols, int64_t ld_in, int64_t batch_stride_in,
                               int64_t rows_pad, int64_t ld_out, int64_t batch_stride_out, const void* in, void* out, void* stream) {
    E2EFT_REQUIRE(in && out, "transpose: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "transpose: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && cols > 0 && cols % epc == 0 && ld_in >= cols && ld_in % epc == 0, "transpose: input shape");
    E2EFT_REQUIRE(rows_pad >= rows && rows_pad % epc == 0 && ld_out >= rows_pad && ld_out % epc == 0, "transpose: output shape");
    E2EFT_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0 && batch_stride_in % epc == 0 && batch_stride_out % epc == 0, "transpose: alignment");
    E2EFT_REQUIRE(cdiv(cols, 64) <= 65535, "transpose: too many columns");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(rows_pad, 64), cdiv(cols, 64), batch);
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, s, (long)rows, cols, (long)ld_in, (long)batch_stride_in,
                                                      (long)rows_pad, (long)ld_out, (long)batch_stride_out, (const T*)in, (T*)out));
    return check_launch("transpose");
}

extern "C" int e2eft_conv2d_im2col_t(const E2eftConvDesc* d, const void* x1, const void* x2, void* col, int64_t ldcol, void* stream) {
    E2EFT_REQUIRE(d && x1 && col, "im2col_t: null pointer");
    E2EFT_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "im2col_t: bad dtype");
    const int epc = 16 / (int)dtype_size(d->dtype);
    E2EFT_REQUIRE(d->c1 > 0 && d->c1 % epc == 0 && d->c2 >= 0 && d->c2 % epc == 0 && d->ldx1 >= d->c1 && d->ldx1 % epc == 0, "im2col_t: channels");
    E2EFT_REQUIRE(d->c2 == 0 || (x2 && d->ldx2 >= d->c2 && d->ldx2 % epc == 0), "im2col_t: second source");
    E2EFT_REQUIRE(d->batch > 0 && d->hout > 0 && d->wout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->hl >= d->hin && d->wl >= d->win, "im2col_t: geometry");
    const long P = (long)d->batch * d->hout * d->wout;
    E2EFT_REQUIRE(ldcol >= P && ldcol % epc == 0 && ((uintptr_t)col & 15) == 0, "im2col_t: ldcol");
    const long Pp = ldcol;   // every row is written over its whole length: zeros in [P, ldcol)
    Im2colGeom g;
    g.batch = d->batch; g.hin = d->hin; g.win = d->win; g.hl = d->hl; g.wl = d->wl; g.c1 = d->c1; g.ldx1 = d->ldx1; g.c2 = d->c2; g.ldx2 = d->ldx2;
    g.kh = d->kh; g.kw = d->kw; g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.hout = d->hout; g.wout = d->wout;
    g.up_sh = (float)d->hin / (float)d->hl; g.up_sw = (float)d->win / (float)d->wl;
    const int cin = d->c1 + d->c2;
    E2EFT_REQUIRE(cdiv(cin, 64) <= 65535 && d->kh * d->kw <= 65535, "im2col_t: grid");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(Pp, 64), cdiv(cin, 64), d->kh * d->kw);
    E2EFT_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((im2col_t_kernel<T>), grid, dim3(256), 0, s, g, P, Pp, (long)ldcol, (const T*)x1, (const T*)x2, (T*)col));
    return check_launch("im2col_t");
}

static int colsum_slabs(int64_t rows_per_group) {
    long ns = (rows_per_group + 511) / 512;
    return (int)(ns < 1 ? 1 : (ns > 64 ? 64 : ns));
}

extern "C" size_t e2eft_colsum_workspace_bytes(int32_t groups, int64_t rows_per_group, int32_t cols) {
    return groups > 0 && rows_per_group > 0 && cols > 0 ? (size_t)groups * colsum_slabs(rows_per_group) * (size_t)cols * sizeof(float) : 0;
}

extern "C" int e2eft_colsum(int32_t dtype, int32_t groups, int64_t rows_per_group, int32_t cols, int64_t ld, float alpha, const void* x, float* out,
                            void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(x && out && workspace, "colsum: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "colsum: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(groups > 0 && groups <= 65535 && rows_per_group > 0 && cols > 0 && cols % epc == 0 && ld >= cols && ld % epc == 0 && ((uintptr_t)x & 15) == 0,
                  "colsum: shape cols=%d ld=%ld", cols, (long)ld);
    const size_t need = e2eft_colsum_workspace_bytes(groups, rows_per_group, cols);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "colsum: workspace %zu < %zu", ws_bytes, need);
    const int ns = colsum_slabs(rows_per_group);
    const long slab = (rows_per_group + ns - 1) / ns;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(cols / epc, 32), groups, ns);
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(256), 0, s, (long)rows_per_group, cols, (long)ld, slab,
                                                      (const T*)x, (float*)workspace));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256), groups), dim3(256), 0, s, ns, cols, alpha, (const float*)workspace, out);
    return check_launch("colsum");
}

extern "C" int e2eft_upsample_nearest_bwd(int32_t dtype, int32_t batch, int32_t hin, int32_t win, int32_t hl, int32_t wl, int32_t c, int32_t lddy,
                                          int32_t lddx, const void* dy, void* dx, void* stream) {
    E2EFT_REQUIRE(dy && dx, "upsample_bwd: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "upsample_bwd: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(batch > 0 && hin > 0 && win > 0 && hl >= hin && wl >= win && c > 0 && c % epc == 0 && lddy >= c && lddy % epc == 0 && lddx >= c && lddx % epc == 0,
                  "upsample_bwd: shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((long)batch * hin * win * (c / epc));
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(g), dim3(256), 0, s, batch, hin, win, hl, wl, c, lddy, lddx,
                                                      (float)hin / (float)hl, (float)win / (float)wl, (const T*)dy, (T*)dx));
    return check_launch("upsample_nearest_bwd");
}

extern "C" int e2eft_silu_bwd(int32_t dtype, int64_t n, const void* x, const void* dy, void* dx, void* stream) {
    E2EFT_REQUIRE(x && dy && dx && n > 0, "silu_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((silu_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (long)n, (const T*)x, (const T*)dy, (T*)dx));
    return check_launch("silu_bwd");
}

extern "C" int e2eft_depth_head_bwd(int32_t dt_x, int32_t dt_y, int64_t pixels, int32_t ldx, int32_t lddx, int32_t cpad, int32_t to_unit,
                                    const void* x, const void* dy, void* dx, void* stream) {
    E2EFT_REQUIRE(x && dy && dx && pixels > 0 && ldx >= 3 && cpad >= 3 && lddx >= cpad, "depth_head_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH2(dt_x, dt_y, TI, TO, hipLaunchKernelGGL((depth_head_bwd_kernel<TI, TO>), dim3(grid_for(pixels)), dim3(256), 0, s, (long)pixels, ldx, lddx,
                                                          cpad, to_unit, (const TI*)x, (const TO*)dy, (TI*)dx));
    return check_launch("depth_head_bwd");
}

extern "C" int e2eft_normal_head_bwd(int32_t dt_x, int32_t dt_y, int32_t batch, int32_t hw, int32_t ldx, int32_t lddx, int32_t cpad, int32_t clampv,
                                     float sign, const void* x, const void* dy, void* dx, void* stream) {
    E2EFT_REQUIRE(x && dy && dx && batch > 0 && hw > 0 && ldx >= 3 && cpad >= 3 && lddx >= cpad, "normal_head_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH2(dt_x, dt_y, TI, TO, hipLaunchKernelGGL((normal_head_bwd_kernel<TI, TO>), dim3(grid_for((long)batch * hw)), dim3(256), 0, s, batch, hw, ldx,
                                                          lddx, cpad, clampv, sign, (const TI*)x, (const TO*)dy, (TI*)dx));
    return check_launch("normal_head_bwd");
}

// fwd_workspace: the buffer e2eft_ssi_loss_fwd filled (per-image sums, valid count); scale_shift: its [B][2] output
extern "C" int e2eft_ssi_loss_bwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                                  const float* scale_shift, const void* fwd_workspace, const float* grad_out, float* dpred, void* workspace,
                                  size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(pred && target && mask && scale_shift && fwd_workspace && grad_out && dpred && workspace, "ssi_loss_bwd: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && hw > 0, "ssi_loss_bwd: shape");
    if (ws_bytes < (size_t)batch * 2 * sizeof(double)) return fail(E2EFT_ERR_WORKSPACE, "ssi_loss_bwd: workspace too small");
    E2EFT_REQUIRE(((uintptr_t)workspace & 7) == 0, "ssi_loss_bwd: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const double* sums = (const double*)fwd_workspace;
    const double* acc = sums + (size_t)batch * 5;
    double* gsum = (double*)workspace;
    if (hipMemsetAsync(workspace, 0, (size_t)batch * 2 * sizeof(double), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "ssi_loss_bwd: memset failed");
    int nb = cdiv(hw, 256 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(ssi_bwd_sums_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, scale_shift, gsum);
    hipLaunchKernelGGL(ssi_bwd_apply_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, scale_shift, sums, acc, gsum, grad_out, dpred);
    return check_launch("ssi_loss_bwd");
}

extern "C" int e2eft_angular_loss_bwd(int32_t batch, int32_t hw, const float* pred, const float* target, const uint8_t* mask,
                                      const void* fwd_workspace, const float* grad_out, float* dpred, void* stream) {
    E2EFT_REQUIRE(pred && target && mask && fwd_workspace && grad_out && dpred, "angular_loss_bwd: null pointer");
    E2EFT_REQUIRE(batch > 0 && batch <= 65535 && hw > 0, "angular_loss_bwd: shape");
    hipStream_t s = (hipStream_t)stream;
    int nb = cdiv(hw, 256 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(angular_bwd_kernel, dim3(nb, batch), dim3(256), 0, s, hw, pred, target, mask, (const double*)fwd_workspace, grad_out, dpred);
    return check_launch("angular_loss_bwd");
}

// EMA of the parameters (diffusers EMAModel.step, used by GeoWizard/geowizard/training/train_depth_normal.py:352-353,785-786): shadow -= (1 - decay) * (shadow - param),
// torch's operation order (sub, mul, sub — no contraction), 16-byte vectors over the flat buffers: 12 B per element, HBM-bound
__global__ __launch_bounds__(256) void ema_kernel(long n, float* __restrict__ shadow, const float* __restrict__ param, float omd) {
#pragma clang fp contract(off)
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 s = reinterpret_cast<float4*>(shadow)[i];
        const float4 q = reinterpret_cast<const float4*>(param)[i];
        s.x = s.x - omd * (s.x - q.x); s.y = s.y - omd * (s.y - q.y); s.z = s.z - omd * (s.z - q.z); s.w = s.w - omd * (s.w - q.w);
        reinterpret_cast<float4*>(shadow)[i] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        shadow[i] = shadow[i] - omd * (shadow[i] - param[i]);
    }
}

extern "C" int e2eft_ema_step(int64_t n, float* shadow, const float* param, float one_minus_decay, void* stream) {
    E2EFT_REQUIRE(shadow && param && n > 0 && (((uintptr_t)shadow | (uintptr_t)param) & 15) == 0, "ema_step: bad args (16-byte aligned fp32 buffers)");
    unsigned nb = grid_for((n + 3) / 4);
    hipLaunchKernelGGL(ema_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (long)n, shadow, param, one_minus_decay);
    return check_launch("ema_step");
}

extern "C" int e2eft_sumsq(int64_t n, const float* g, double* out, void* stream) {
    E2EFT_REQUIRE(g && out && n > 0 && ((uintptr_t)out & 7) == 0, "sumsq: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(double), s) != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "sumsq: memset failed");
    unsigned nb = grid_for(n);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, s, (long)n, g, out);
    return check_launch("sumsq");
}

extern "C" int e2eft_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int32_t step, const double* grad_sumsq, float grad_scale, float max_norm, void* stream) {
    E2EFT_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "adamw: bad args");
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, s, (long)n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bc1,
                       bc2s, grad_sumsq, grad_scale, max_norm);
    return check_launch("adamw");
}

extern "C" int e2eft_adamw_step_guarded(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                                        float eps, float weight_decay, int64_t* state, float* coef, const double* grad_sumsq, float grad_scale,
                                        float max_norm, void* stream) {
    E2EFT_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && coef && grad_sumsq && n > 0, "adamw_guarded: bad args");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, s, (long long*)state, coef, grad_sumsq, beta1, beta2, grad_scale, max_norm);
    hipLaunchKernelGGL(adamw_guarded_kernel, dim3(grid_for(n)), dim3(256), 0, s, (long)n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
                       weight_decay, (const float*)coef);
    return check_launch("adamw_guarded");
}

extern "C" int e2eft_cast(int32_t dt_in, int32_t dt_out, int64_t n, float mul, int32_t accumulate, const void* x, void* y, void* stream) {
    E2EFT_REQUIRE(x && y && n > 0, "cast: bad args");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH2(dt_in, dt_out, TI, TO, hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(grid_for(n)), dim3(256), 0, s, (long)n, mul, accumulate,
                                                            (const TI*)x, (TO*)y));
    return check_launch("cast");
}
