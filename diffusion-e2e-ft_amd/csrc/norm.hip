// norm.hip — GroupNorm(+SiLU), LayerNorm, row softmax, GEGLU for NHWC / token tensors (HBM-bound streams).
//
// GroupNorm over NHWC: a group's channels are a short contiguous run at every pixel, so reading "one group at a
// time" would touch 8-160 bytes per 256-2560-byte pixel row.  Instead every workgroup streams whole pixel rows
// with 16-byte loads (thread <-> fixed channel chunk), keeps per-CHANNEL shifted sums in registers
// (pivot = first sample, so sum(x-p) / sum((x-p)^2) do not cancel), merges them to (n, mean, M2) triples with
// Chan's formula, and a tiny second kernel merges channels x slabs into per-(image, group) statistics and folds
// gamma/beta into a per-(image, channel) affine pair (a, d).  The apply kernel is then y = silu(a*x + d).
#include "common.h"

namespace e2eft {

struct Triple {
    float n, mean, m2;
};
__device__ __forceinline__ Triple merge(const Triple& A, const Triple& B) {
    if (B.n == 0.f) return A;
    if (A.n == 0.f) return B;
    Triple r;
    r.n = A.n + B.n;
    const float delta = B.mean - A.mean;
    const float f = B.n / r.n;
    r.mean = A.mean + delta * f;
    r.m2 = A.m2 + B.m2 + delta * delta * A.n * f;
    return r;
}

struct GnGeom {
    int batch, hw, c1, ldx1, c2, ldx2, C;
    int nchunks, cpb, nchb, pl;  // chunks per row, chunks per block, channel blocks, pixel lanes per block
    int nslabs, slab;            // pixel slabs per image and pixels per slab
};

// grid (nslabs, batch, nchb), block 256
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnGeom g, const T* __restrict__ x1, const T* __restrict__ x2,
                                                         float* __restrict__ partial /* [B][nslabs][C][3] */) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float sm[256 * 8 * 3];
    const int tid = threadIdx.x;
    const int chl = tid % g.cpb, pl = tid / g.cpb;
    const int ch = blockIdx.z * g.cpb + chl;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * g.slab;
    const int p1 = min(p0 + g.slab, g.hw);
    const bool active = pl < g.pl;
    const int c = ch * EPC;
    const T* src;
    int ld;
    if (c < g.c1) { src = x1 + c; ld = g.ldx1; } else { src = x2 + (c - g.c1); ld = g.ldx2; }
    float piv[EPC], s[EPC], ss[EPC];
    float n = 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) piv[e] = s[e] = ss[e] = 0.f;
    if (active) {
        for (int pix = p0 + pl; pix < p1; pix += g.pl) {
            Vec16<T> v = ld16(src + ((long)b * g.hw + pix) * ld);
            if (n == 0.f) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) piv[e] = to_f(v.e[e]);
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = to_f(v.e[e]) - piv[e];
                s[e] += d;
                ss[e] += d * d;
            }
            n += 1.f;
        }
    }
    // per-thread triples -> LDS [pl][chl][e]
    if (active) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float mean = 0.f, m2 = 0.f;
            if (n > 0.f) {
                const float ds = s[e] / n;
                mean = piv[e] + ds;
                m2 = fmaxf(ss[e] - s[e] * ds, 0.f);
            }
            float* o = sm + ((pl * g.cpb + chl) * EPC + e) * 3;
            o[0] = n; o[1] = mean; o[2] = m2;
        }
    }
    __syncthreads();
    // merge across pixel lanes: one thread per channel of this block
    const int cblk = g.cpb * EPC;
    for (int cc = tid; cc < cblk; cc += 256) {
        Triple acc = {0.f, 0.f, 0.f};
        for (int q = 0; q < g.pl; ++q) {
            const float* o = sm + (q * cblk + cc) * 3;
            Triple t = {o[0], o[1], o[2]};
            acc = merge(acc, t);
        }
        const int cg = blockIdx.z * cblk + cc;
        float* o = partial + (((long)b * g.nslabs + blockIdx.x) * g.C + cg) * 3;
        o[0] = acc.n; o[1] = acc.mean; o[2] = acc.m2;
    }
}

// grid (groups, batch), block 256: merge slabs x channels-of-group (the channels of a group may straddle the two sources,
// each source has its own slab count), write (rstd*gamma, mean) pairs
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(int C, int c1, int groups, float eps, const float* __restrict__ pa,
                                                          int nsa, const float* __restrict__ pb, int nsb,
                                                          const T* __restrict__ gamma, float* __restrict__ ad /* [B][C][2] */,
                                                          float* __restrict__ rstd_out /* [B][groups] */) {
    __shared__ float sm[256 * 3];
    const int grp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / groups;
    const int cbeg = grp * cpg, cend = cbeg + cpg;
    const int na = max(0, min(cend, c1) - cbeg);          // channels of this group that live in source 1
    const int items_a = na * nsa, items = items_a + (cpg - na) * nsb;
    const int c2 = C - c1;
    // Thread layout: consecutive threads -> consecutive channels of ONE slab (their triples are contiguous: 12 * cpg bytes), slab lanes above
    // them; four independent triples in flight per thread.  (The first version walked slab-major per channel: every thread of a wave in a
    // different 12 * C-byte row, one dependent round trip per item — 98 us for the 2304 slabs x 128 channels of a 768^2 layer, a third
    // of that now; the merge order differs, the result is the same Chan merge.)
    int P = 1;
    while (P < cpg && P < 256) P <<= 1;
    const int tx = tid & (P - 1), ty = tid / P, ny = 256 / P;
    Triple acc = {0.f, 0.f, 0.f};
    auto walk = [&](const float* __restrict__ base, const int ns, const int cstride, const int nch) {   // base: slab 0, first channel of the group in this source
        for (int cc = tx; cc < nch; cc += P) {
            const float* o = base + (long)cc * 3;
            const long step = (long)cstride * 3;
            int sl = ty;
            for (; sl + 3 * ny < ns; sl += 4 * ny) {
                const float* q0 = o + sl * step;
                const float* q1 = q0 + ny * step;
                const float* q2 = q1 + ny * step;
                const float* q3 = q2 + ny * step;
                const Triple t0 = {q0[0], q0[1], q0[2]}, t1 = {q1[0], q1[1], q1[2]}, t2 = {q2[0], q2[1], q2[2]}, t3 = {q3[0], q3[1], q3[2]};
                acc = merge(merge(merge(merge(acc, t0), t1), t2), t3);
            }
            for (; sl < ns; sl += ny) {
                const float* q0 = o + sl * step;
                const Triple t0 = {q0[0], q0[1], q0[2]};
                acc = merge(acc, t0);
            }
        }
    };
    if (na > 0) walk(pa + ((long)b * nsa * c1 + cbeg) * 3, nsa, c1, na);
    if (cpg - na > 0) walk(pb + ((long)b * nsb * c2 + (cbeg + na - c1)) * 3, nsb, c2, cpg - na);
    (void)items;
    sm[tid * 3] = acc.n; sm[tid * 3 + 1] = acc.mean; sm[tid * 3 + 2] = acc.m2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            Triple A = {sm[tid * 3], sm[tid * 3 + 1], sm[tid * 3 + 2]};
            Triple B = {sm[(tid + st) * 3], sm[(tid + st) * 3 + 1], sm[(tid + st) * 3 + 2]};
            Triple r = merge(A, B);
            sm[tid * 3] = r.n; sm[tid * 3 + 1] = r.mean; sm[tid * 3 + 2] = r.m2;
        }
        __syncthreads();
    }
    const float mean = sm[1];
    const float var = sm[2] / sm[0];
    const float rstd = rsqrtf(var + eps);
    if (tid == 0) rstd_out[(long)b * groups + grp] = rstd;   // kept for the backward pass
    for (int cc = tid; cc < cpg; cc += 256) {
        const int c = cbeg + cc;
        const float ga = gamma ? to_f(gamma[c]) : 1.f;
        ad[((long)b * C + c) * 2] = rstd * ga;   // y = (x - mean) * a + beta: no cancellation when |mean| >> std
        ad[((long)b * C + c) * 2 + 1] = mean;
    }
}

// grid (nslabs, batch, nchb)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnGeom g, int silu, int ldy, const T* __restrict__ x1,
                                                       const T* __restrict__ x2, const float* __restrict__ ad,
                                                       const T* __restrict__ beta, T* __restrict__ y) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int chl = tid % g.cpb, pl = tid / g.cpb;
    if (pl >= g.pl) return;
    const int ch = blockIdx.z * g.cpb + chl;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * g.slab;
    const int p1 = min(p0 + g.slab, g.hw);
    const int c = ch * EPC;
    const T* src;
    int ld;
    if (c < g.c1) { src = x1 + c; ld = g.ldx1; } else { src = x2 + (c - g.c1); ld = g.ldx2; }
    float a[EPC], mu[EPC], be[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        a[e] = ad[((long)b * g.C + c + e) * 2];
        mu[e] = ad[((long)b * g.C + c + e) * 2 + 1];
        be[e] = beta ? to_f(beta[c + e]) : 0.f;
        if constexpr (sizeof(T) == 2) gn_fold(a[e], mu[e], be[e], a[e], be[e]);      // 16-bit: (a2, d2) of the exp2-domain form (common.h); mu is unused then
    }
    auto one = [&](long row, const Vec16<T>& v) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float t;
            if constexpr (sizeof(T) == 2) {
                t = gn_act_u(__builtin_fmaf(to_f(v.e[e]), a[e], be[e]), silu != 0);
            } else {
                t = fmaf(to_f(v.e[e]) - mu[e], a[e], be[e]);
                if (silu) t = silu_f(t);
            }
            o.e[e] = from_f<T>(t);
        }
        st16(y + row * ldy + c, o);
    };
    int pix = p0 + pl;
    for (; pix + 3 * g.pl < p1; pix += 4 * g.pl) {   // four independent 16-byte loads in flight per thread
        const long row = (long)b * g.hw + pix;
        Vec16<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld16(src + (row + (long)u * g.pl) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(row + (long)u * g.pl, v[u]);
    }
    for (; pix < p1; pix += g.pl) {
        const long row = (long)b * g.hw + pix;
        one(row, ld16(src + row * ld));
    }
}

// fp32 GroupNorm(+SiLU) whose output leaves as the two-term f16 split of csrc/f32split.hip: planes [pixel][y0 (C) | y1 (C)] with y * s = y0 + y1, s a power of two
// derived (gn_split_scale_kernel below) from a bound of the output (|SiLU(t)| <= |t| <= max|gamma| * sqrt(elements per group) + max|beta|: the normalised value of one element of n is at most
// sqrt(n - 1)) — no maximum pass, and the consumer (e2eft_conv2d_fwd_f32split) never sees the fp32 tensor: 4 + 4 bytes per element like the plain apply pass.
// The conversions saturate (a bound that generous costs nothing: values 2^17 below it still carry 22 bits).  grid as gn_apply_kernel, one source.
// the scale of the planes from a bound of the output, on the device (one workgroup; no host read of the parameters — they may be training, and a captured graph
// cannot wait for the host): bound = max|gamma| * sqrt(n) + max|beta| -> s = 2^k with bound * s in [2^14, 2^15); scale[1] = s, scale[2] = 1 / s
__global__ __launch_bounds__(256) void gn_split_scale_kernel(int C, float sqrt_n, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale) {
    float mg = gamma ? 0.f : 1.f, mb = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        if (gamma) mg = fmaxf(mg, fabsf(gamma[c]));
        if (beta) mb = fmaxf(mb, fabsf(beta[c]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mg = fmaxf(mg, __shfl_xor(mg, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
    __shared__ float sg[4], sb[4];
    if ((threadIdx.x & 63) == 0) { sg[threadIdx.x >> 6] = mg; sb[threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mg = fmaxf(fmaxf(sg[0], sg[1]), fmaxf(sg[2], sg[3]));
        mb = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
        const float bound = mg * sqrt_n + mb;
        const unsigned bits = __float_as_uint(bound);
        const int ex = (int)((bits >> 23) & 0xffu);
        int k = 0;
        if (bound > 0.f && ex != 0xff) k = 14 - (ex == 0 ? -126 : ex - 127);      // bound in [2^(ex-127), 2^(ex-126)) -> bound * 2^k in [2^14, 2^15)
        k = k > 100 ? 100 : k < -100 ? -100 : k;
        scale[0] = bound;
        scale[1] = __uint_as_float((unsigned)(127 + k) << 23);
        scale[2] = __uint_as_float((unsigned)(127 - k) << 23);
    }
}

__global__ __launch_bounds__(256) void gn_apply_split_kernel(GnGeom g, int silu, int ldp, const float* __restrict__ scale, const float* __restrict__ x1, const float* __restrict__ ad,
                                                             const float* __restrict__ beta, f16* __restrict__ planes) {
    const float s = scale[1];
    const int tid = threadIdx.x;
    const int chl = tid % g.cpb, pl = tid / g.cpb;
    if (pl >= g.pl) return;
    const int ch = blockIdx.z * g.cpb + chl;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * g.slab;
    const int p1 = min(p0 + g.slab, g.hw);
    const int c = ch * 4;
    const float* src = x1 + c;
    float a[4], mu[4], be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = ad[((long)b * g.C + c + e) * 2];
        mu[e] = ad[((long)b * g.C + c + e) * 2 + 1];
        be[e] = beta ? beta[c + e] : 0.f;
    }
    auto one = [&](long row, const floatx4& v) {
        half4 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = fmaf(v[e] - mu[e], a[e], be[e]);      // gn_apply_kernel<float>'s arithmetic
            if (silu) t = silu_f(t);
            t = fminf(fmaxf(t * s, -65504.f), 65504.f);
            const f16 q0 = (f16)t;
            h0[e] = q0;
            h1[e] = (f16)(t - (float)q0);
        }
        *reinterpret_cast<half4*>(planes + row * ldp + c) = h0;
        *reinterpret_cast<half4*>(planes + row * ldp + g.C + c) = h1;
    };
    int pix = p0 + pl;
    for (; pix + 3 * g.pl < p1; pix += 4 * g.pl) {
        const long row = (long)b * g.hw + pix;
        floatx4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const floatx4*>(src + (row + (long)u * g.pl) * g.ldx1);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(row + (long)u * g.pl, v[u]);
    }
    for (; pix < p1; pix += g.pl) {
        const long row = (long)b * g.hw + pix;
        one(row, *reinterpret_cast<const floatx4*>(src + row * g.ldx1));
    }
}

static int gn_geom(const E2eftGroupNormDesc* d, GnGeom& g) {
    const int epc = 16 / (int)dtype_size(d->dtype);
    g.batch = d->batch; g.hw = d->hw; g.c1 = d->c1; g.ldx1 = d->ldx1; g.c2 = d->c2; g.ldx2 = d->ldx2;
    g.C = d->c1 + d->c2;
    g.nchunks = g.C / epc;
    int nb = 1;
    while (g.nchunks % nb != 0 || g.nchunks / nb > 256) ++nb;
    g.nchb = nb;
    g.cpb = g.nchunks / nb;
    g.pl = 256 / g.cpb;
    long ns = ((long)d->hw + (long)g.pl * 16 - 1) / ((long)g.pl * 16);
    if (ns < 1) ns = 1;
    if (ns > 1024) ns = 1024;
    g.slab = (int)((d->hw + ns - 1) / ns);
    g.nslabs = (d->hw + g.slab - 1) / g.slab;
    return 0;
}

static void gn_geom_one(int dtype, int batch, int hw, int c, int ld, GnGeom& g) {
    E2eftGroupNormDesc d1 = {};
    d1.dtype = dtype; d1.batch = batch; d1.hw = hw; d1.c1 = c; d1.ldx1 = ld; d1.c2 = 0; d1.ldx2 = 0;
    gn_geom(&d1, g);
}

template <typename T>
static int gn_run(const E2eftGroupNormDesc* d, const GnGeom& g, const void* x1, const void* x2, const void* gamma,
                  const void* beta, void* y, const float* pre1, int ns1, const float* pre2, int ns2, void* ws, hipStream_t s) {
    // workspace: [partial of source 1 (if computed here)] [partial of source 2 (if computed here)] [ad]
    float* wsp = (float*)ws;
    GnGeom g1, g2;
    gn_geom_one(d->dtype, d->batch, d->hw, d->c1, d->ldx1, g1);
    const float* pa = pre1;
    int nsa = ns1;
    if (!pa) {
        hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(g1.nslabs, g1.batch, g1.nchb), dim3(256), 0, s, g1, (const T*)x1, (const T*)nullptr, wsp);
        pa = wsp; nsa = g1.nslabs;
    }
    wsp += (size_t)d->batch * g1.nslabs * d->c1 * 3;
    const float* pb = pre2;
    int nsb = ns2;
    if (d->c2 > 0) {
        gn_geom_one(d->dtype, d->batch, d->hw, d->c2, d->ldx2, g2);
        if (!pb) {
            hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(g2.nslabs, g2.batch, g2.nchb), dim3(256), 0, s, g2, (const T*)x2, (const T*)nullptr, wsp);
            pb = wsp; nsb = g2.nslabs;
        }
        wsp += (size_t)d->batch * g2.nslabs * d->c2 * 3;
    } else {
        pb = pa; nsb = 1;
    }
    float* ad = wsp;
    float* rstd = ad + (size_t)d->batch * g.C * 2;
    hipLaunchKernelGGL((gn_finalize_kernel<T>), dim3(d->groups, g.batch), dim3(256), 0, s, g.C, d->c1, d->groups, d->eps, pa, nsa, pb, nsb,
                       (const T*)gamma, ad, rstd);
    // the apply pass has its own work split: short slabs (two unrolled iterations of four 16-byte loads per thread) in many workgroups —
    // 0.414 ms instead of 0.472 ms on 8 x 768^2 x 128 fp16, 5.8 of the ~5.8 TB/s this part copies at (scripts/stream_bench.hip,
    // profiles/r02_stream_bench.txt); the statistics kernels keep the coarser split, their partials are per slab
    if (!y) return check_launch("groupnorm_stats");   // statistics only: the consumer applies (x - mean) * a + beta itself (e2eft_conv2d_fwd_normed)
    GnGeom ga = g;
    // pixels per thread: 4 x n sixteen-byte loads.  n = 2 streams the big tensors fastest (r02 stream bench; re-measured in round 6: 768^2 C128 453 us against 476 at
    // n = 4), n = 4 the ones of <= 128 MB (96^2 C512 37.1 -> 34.4 us, 96^2 C320 26.2 -> 24.6: fewer, longer workgroups amortise the coefficient loads;
    // profiles/r06c_gn_apply_sweep.txt)
    const int it = option(E2EFT_OPT_GN_APPLY_ITERS);
    const long tensor_bytes = (long)d->batch * d->hw * g.C * (long)dtype_size(d->dtype);
    ga.slab = g.pl * 4 * (it > 0 ? it : (tensor_bytes <= (128L << 20) ? 4 : 2));
    ga.nslabs = (d->hw + ga.slab - 1) / ga.slab;
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(ga.nslabs, ga.batch, ga.nchb), dim3(256), 0, s, ga, d->silu, d->ldy, (const T*)x1, (const T*)x2, ad, (const T*)beta, (T*)y);
    return check_launch("groupnorm");
}

static size_t gn_ws_bytes(const E2eftGroupNormDesc* d) {
    GnGeom g1, g2;
    gn_geom_one(d->dtype, d->batch, d->hw, d->c1, d->ldx1, g1);
    size_t f = (size_t)d->batch * g1.nslabs * d->c1 * 3;
    if (d->c2 > 0) {
        gn_geom_one(d->dtype, d->batch, d->hw, d->c2, d->ldx2, g2);
        f += (size_t)d->batch * g2.nslabs * d->c2 * 3;
    }
    f += (size_t)d->batch * (d->c1 + d->c2) * 2 + (size_t)d->batch * d->groups;   // (a, mean) pairs, then rstd per (image, group)
    return f * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward.  With z = xhat*gamma + beta, dz = dy * act'(z):
//   S1[b,c] = sum_p dz, S2[b,c] = sum_p dz*xhat;  dgamma = sum_b S2, dbeta = sum_b S1;
//   m1[b,g] = sum_{c in g} gamma_c S1 / n, m2 likewise with S2;  dx = rstd * (dz*gamma - m1 - xhat*m2).
// Same streaming geometry as the forward (thread <-> 16-byte channel chunk, pixel lanes), the forward's (a, mean, rstd) are reused.
template <typename T, int EPC>
struct GnBwdCoef {
    float a[EPC], mu[EPC], rs[EPC], be[EPC];
};
template <typename T, int EPC>
__device__ __forceinline__ void gn_bwd_load(GnBwdCoef<T, EPC>& k, int b, int c, int C, int groups, const float* __restrict__ ad,
                                            const float* __restrict__ rstd, const T* __restrict__ beta) {
    const int cpg = C / groups;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        k.a[e] = ad[((long)b * C + c + e) * 2];
        k.mu[e] = ad[((long)b * C + c + e) * 2 + 1];
        k.rs[e] = rstd[(long)b * groups + (c + e) / cpg];
        k.be[e] = beta ? to_f(beta[c + e]) : 0.f;
    }
}
__device__ __forceinline__ float silu_grad_f(float z) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z));
    return sg * (1.f + z * (1.f - sg));
}

// grid (nslabs, batch, nchb): per-(image, slab, channel) sums S1, S2
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnGeom g, int groups, int silu, int lddy, const T* __restrict__ x1,
                                                             const T* __restrict__ x2, const T* __restrict__ dy,
                                                             const float* __restrict__ ad, const float* __restrict__ rstd,
                                                             const T* __restrict__ beta, float* __restrict__ part /* [B][nslabs][C][2] */) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float sm[256 * 8 * 2];
    const int tid = threadIdx.x;
    const int chl = tid % g.cpb, pl = tid / g.cpb;
    const int ch = blockIdx.z * g.cpb + chl;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * g.slab;
    const int p1 = min(p0 + g.slab, g.hw);
    const bool active = pl < g.pl;
    const int c = ch * EPC;
    const T* src;
    int ld;
    if (c < g.c1) { src = x1 + c; ld = g.ldx1; } else { src = x2 + (c - g.c1); ld = g.ldx2; }
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
        GnBwdCoef<T, EPC> k;
        gn_bwd_load<T, EPC>(k, b, c, g.C, groups, ad, rstd, beta);
        auto one = [&](const Vec16<T>& v, const Vec16<T>& d) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xc = to_f(v.e[e]) - k.mu[e];
                float dz = to_f(d.e[e]);
                if (silu) dz *= silu_grad_f(fmaf(xc, k.a[e], k.be[e]));
                s1[e] += dz;
                s2[e] += dz * xc * k.rs[e];
            }
        };
        int pix = p0 + pl;
        for (; pix + g.pl < p1; pix += 2 * g.pl) {
            const long row = (long)b * g.hw + pix;
            const Vec16<T> v0 = ld16(src + row * ld), d0 = ld16(dy + row * lddy + c);
            const Vec16<T> v1 = ld16(src + (row + g.pl) * ld), d1 = ld16(dy + (row + g.pl) * lddy + c);
            one(v0, d0);
            one(v1, d1);
        }
        for (; pix < p1; pix += g.pl) {
            const long row = (long)b * g.hw + pix;
            one(ld16(src + row * ld), ld16(dy + row * lddy + c));
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float* o = sm + ((pl * g.cpb + chl) * EPC + e) * 2;
            o[0] = s1[e]; o[1] = s2[e];
        }
    }
    __syncthreads();
    const int cblk = g.cpb * EPC;
    for (int cc = tid; cc < cblk; cc += 256) {
        float a1 = 0.f, a2 = 0.f;
        for (int q = 0; q < g.pl; ++q) { a1 += sm[(q * cblk + cc) * 2]; a2 += sm[(q * cblk + cc) * 2 + 1]; }
        float* o = part + (((long)b * g.nslabs + blockIdx.x) * g.C + blockIdx.z * cblk + cc) * 2;
        o[0] = a1; o[1] = a2;
    }
}

// grid (groups, batch): slab sums -> sc[b][c] = (S1, S2); mm[b][g] = (m1, m2).  Waves take channels, lanes take slabs.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(int C, int groups, int hw, int nslabs, const float* __restrict__ part,
                                                              const T* __restrict__ gamma, float* __restrict__ sc, float* __restrict__ mm) {
    __shared__ float red[2][4];
    const int grp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = C / groups;
    float m1 = 0.f, m2 = 0.f;
    for (int cc = wave; cc < cpg; cc += 4) {
        const int c = grp * cpg + cc;
        float a1 = 0.f, a2 = 0.f;
        for (int sl = lane; sl < nslabs; sl += 64) {
            const float* o = part + (((long)b * nslabs + sl) * C + c) * 2;
            a1 += o[0]; a2 += o[1];
        }
        a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            sc[((long)b * C + c) * 2] = a1;
            sc[((long)b * C + c) * 2 + 1] = a2;
        }
        const float ga = gamma ? to_f(gamma[c]) : 1.f;
        m1 += ga * a1; m2 += ga * a2;
    }
    if (lane == 0) { red[0][wave] = m1; red[1][wave] = m2; }
    __syncthreads();
    if (tid == 0) {
        const float n = (float)cpg * (float)hw;
        mm[((long)b * groups + grp) * 2] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
        mm[((long)b * groups + grp) * 2 + 1] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n;
    }
}

// grid (nslabs, batch, nchb)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnGeom g, int groups, int silu, int lddy, int lddx, const T* __restrict__ x1,
                                                           const T* __restrict__ x2, const T* __restrict__ dy, const float* __restrict__ ad,
                                                           const float* __restrict__ rstd, const float* __restrict__ mm,
                                                           const T* __restrict__ gamma, const T* __restrict__ beta, T* __restrict__ dx,
                                                           const T* __restrict__ addend, int ldadd) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int chl = tid % g.cpb, pl = tid / g.cpb;
    if (pl >= g.pl) return;
    const int ch = blockIdx.z * g.cpb + chl;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * g.slab;
    const int p1 = min(p0 + g.slab, g.hw);
    const int c = ch * EPC;
    const T* src;
    int ld;
    if (c < g.c1) { src = x1 + c; ld = g.ldx1; } else { src = x2 + (c - g.c1); ld = g.ldx2; }
    GnBwdCoef<T, EPC> k;
    gn_bwd_load<T, EPC>(k, b, c, g.C, groups, ad, rstd, beta);
    const int cpg = g.C / groups;
    float k1[EPC], k2[EPC], k3[EPC];   // dx = k1*dz - k2 - (x - mean)*k3
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int grp = (c + e) / cpg;
        const float ga = gamma ? to_f(gamma[c + e]) : 1.f;
        k1[e] = k.rs[e] * ga;
        k2[e] = k.rs[e] * mm[((long)b * groups + grp) * 2];
        k3[e] = k.rs[e] * k.rs[e] * mm[((long)b * groups + grp) * 2 + 1];
    }
    auto one = [&](long row, const Vec16<T>& v, const Vec16<T>& d) {
        Vec16<T> o, ad2;
        if (addend) ad2 = ld16(addend + row * ldadd + c);   // gradient arriving over the skip path of the same tensor (uniform branch)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float xc = to_f(v.e[e]) - k.mu[e];
            float dz = to_f(d.e[e]);
            if (silu) dz *= silu_grad_f(fmaf(xc, k.a[e], k.be[e]));
            float r = k1[e] * dz - k2[e] - xc * k3[e];
            if (addend) r += to_f(ad2.e[e]);
            o.e[e] = from_f<T>(r);
        }
        st16(dx + row * lddx + c, o);
    };
    int pix = p0 + pl;
    for (; pix + g.pl < p1; pix += 2 * g.pl) {   // two pixel rows (four 16-byte loads) in flight per thread
        const long row = (long)b * g.hw + pix;
        const Vec16<T> v0 = ld16(src + row * ld), d0 = ld16(dy + row * lddy + c);
        const Vec16<T> v1 = ld16(src + (row + g.pl) * ld), d1 = ld16(dy + (row + g.pl) * lddy + c);
        one(row, v0, d0);
        one(row + g.pl, v1, d1);
    }
    for (; pix < p1; pix += g.pl) {
        const long row = (long)b * g.hw + pix;
        one(row, ld16(src + row * ld), ld16(dy + row * lddy + c));
    }
}

// dgamma[c] = sum_b S2[b][c], dbeta[c] = sum_b S1[b][c]
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(int batch, int C, const float* __restrict__ sc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < batch; ++b) { a1 += sc[((long)b * C + c) * 2]; a2 += sc[((long)b * C + c) * 2 + 1]; }
    if (dbeta) dbeta[c] = a1;
    if (dgamma) dgamma[c] = a2;
}

static size_t gn_bwd_ws_bytes(const E2eftGroupNormDesc* d) {
    GnGeom g;
    gn_geom(d, g);
    const size_t C = d->c1 + d->c2;
    return ((size_t)d->batch * g.nslabs * C * 2 + (size_t)d->batch * C * 2 + (size_t)d->batch * d->groups * 2) * sizeof(float);
}

template <typename T>
static int gn_bwd_run(const E2eftGroupNormDesc* d, const GnGeom& g, const void* x1, const void* x2, const void* gamma, const void* beta,
                      const void* dy, int lddy, void* dx, int lddx, float* dgamma, float* dbeta, const float* fwd_ws, void* ws,
                      const void* addend, int ldadd, hipStream_t s) {
    const float* ad = fwd_ws + (gn_ws_bytes(d) / sizeof(float) - (size_t)d->batch * g.C * 2 - (size_t)d->batch * d->groups);
    const float* rstd = ad + (size_t)d->batch * g.C * 2;
    float* part = (float*)ws;
    float* sc = part + (size_t)d->batch * g.nslabs * g.C * 2;
    float* mm = sc + (size_t)d->batch * g.C * 2;
    const dim3 grid(g.nslabs, g.batch, g.nchb);
    hipLaunchKernelGGL((gn_bwd_partial_kernel<T>), grid, dim3(256), 0, s, g, d->groups, d->silu, lddy, (const T*)x1, (const T*)x2,
                       (const T*)dy, ad, rstd, (const T*)beta, part);
    hipLaunchKernelGGL((gn_bwd_finalize_kernel<T>), dim3(d->groups, g.batch), dim3(256), 0, s, g.C, d->groups, d->hw, g.nslabs, part,
                       (const T*)gamma, sc, mm);
    if (dx)
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), grid, dim3(256), 0, s, g, d->groups, d->silu, lddy, lddx, (const T*)x1, (const T*)x2,
                           (const T*)dy, ad, rstd, mm, (const T*)gamma, (const T*)beta, (T*)dx, (const T*)addend, ldadd);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(gn_bwd_params_kernel, dim3(cdiv(g.C, 256)), dim3(256), 0, s, g.batch, g.C, sc, dgamma, dbeta);
    return check_launch("groupnorm_bwd");
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers, exact two-pass statistics.
template <typename T, int MAXI>
__global__ __launch_bounds__(256) void layernorm_kernel(long rows, int c, int ldx, int ldy, float eps, const T* __restrict__ x,
                                                        const T* __restrict__ gamma, const T* __restrict__ beta, T* __restrict__ y) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = c / EPC;
    float v[MAXI][EPC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nchunks) {
            Vec16<T> t = ld16(x + row * ldx + ch * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { v[i][e] = to_f(t.e[e]); sum += v[i][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float dd = v[i][e] - mean; sq += dd * dd; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nchunks) {
            Vec16<T> ga = ld16(gamma + ch * EPC), be = ld16(beta + ch * EPC), o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>((v[i][e] - mean) * rstd * to_f(ga.e[e]) + to_f(be.e[e]));
            st16(y + row * ldy + ch * EPC, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Row softmax in place (block per row; the row is L2/L1 resident between the three sweeps).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(int ntot, long lds, float scale, T* __restrict__ s, int causal_nq) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float red[8];
    T* row = s + (long)blockIdx.x * lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // causal_nq > 0: row r belongs to query r % causal_nq and sees keys 0 .. that index (CLIP text tower); the rest of the row is zeroed
    const int n = causal_nq > 0 ? min(ntot, (int)(blockIdx.x % (unsigned)causal_nq) + 1) : ntot;
    const int nfull = n / EPC;
    const float sc = scale * 1.4426950408889634f;  // work in base 2
    float mx = -INFINITY;
    for (int ch = tid; ch < nfull; ch += 256) {
        Vec16<T> v = ld16(row + ch * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) mx = fmaxf(mx, to_f(v.e[e]));
    }
    for (int j = nfull * EPC + tid; j < n; j += 256) mx = fmaxf(mx, to_f(row[j]));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // scale may be negative in principle; we assume scale > 0 (checked on the host)
    const float mb = mx * sc;
    float sum = 0.f;
    for (int ch = tid; ch < nfull; ch += 256) {
        Vec16<T> v = ld16(row + ch * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) sum += exp2f(fmaf(to_f(v.e[e]), sc, -mb));
    }
    for (int j = nfull * EPC + tid; j < n; j += 256) sum += exp2f(fmaf(to_f(row[j]), sc, -mb));
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int ch = tid; ch < nfull; ch += 256) {
        Vec16<T> v = ld16(row + ch * EPC), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(exp2f(fmaf(to_f(v.e[e]), sc, -mb)) * inv);
        st16(row + ch * EPC, o);
    }
    const int npad = (ntot + EPC - 1) / EPC * EPC;
    for (int j = nfull * EPC + tid; j < npad; j += 256)
        row[j] = j < n ? from_f<T>(exp2f(fmaf(to_f(row[j]), sc, -mb)) * inv) : from_f<T>(0.f);
}

// ---------------------------------------------------------------------------------------------------
// GEGLU gate
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(long rows, int c, int ldh, int ldy, const T* __restrict__ hbuf, T* __restrict__ y) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cch = c / EPC;
    const long total = rows * cch;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long r = it / cch;
        const int ch = (int)(it - r * cch);
        Vec16<T> a = ld16(hbuf + r * ldh + ch * EPC), gt = ld16(hbuf + r * ldh + c + ch * EPC), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(to_f(a.e[e]) * gelu_erf_f(to_f(gt.e[e])));
        st16(y + r * ldy + ch * EPC, o);
    }
}


// LayerNorm backward: one wave per row (row + dy in registers); per-wave register accumulators of dgamma / dbeta over the rows
// the wave visits, written as one partial row per wave ([2][c] fp32) and reduced by the column-sum kernel afterwards.
//   dxhat = dy*gamma;  dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat))
template <typename T, int MAXI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(long rows, int c, int ldx, int lddy, int lddx, float eps, const T* __restrict__ x,
                                                            const T* __restrict__ gamma, const T* __restrict__ dy, T* __restrict__ dx,
                                                            float* __restrict__ part /* [gridDim.x*4][2][c] */) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = c / EPC;
    float dg[MAXI][EPC], db[MAXI][EPC];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) dg[i][e] = db[i][e] = 0.f;
    const float invc = 1.f / (float)c;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float v[MAXI][EPC], d[MAXI][EPC];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nchunks) {
                Vec16<T> t = ld16(x + row * ldx + ch * EPC), u = ld16(dy + row * lddy + ch * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { v[i][e] = to_f(t.e[e]); d[i][e] = to_f(u.e[e]); sum += v[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[i][e] = d[i][e] = 0.f;
            }
        }
        const float mean = wave_sum(sum) * invc;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (lane + 64 * i < nchunks) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float dd = v[i][e] - mean; sq += dd * dd; }
            }
        const float rstd = rsqrtf(wave_sum(sq) * invc + eps);
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nchunks) {
                Vec16<T> ga = ld16(gamma + ch * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float xh = (v[i][e] - mean) * rstd;
                    v[i][e] = xh;
                    dg[i][e] += d[i][e] * xh;
                    db[i][e] += d[i][e];
                    d[i][e] *= to_f(ga.e[e]);   // dxhat
                    a += d[i][e];
                    bsum += d[i][e] * xh;
                }
            }
        }
        a = wave_sum(a) * invc;
        bsum = wave_sum(bsum) * invc;
        if (dx) {
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nchunks) {
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o.e[e] = from_f<T>(rstd * (d[i][e] - a - v[i][e] * bsum));
                    st16(dx + row * lddx + ch * EPC, o);
                }
            }
        }
    }
    float* o = part + ((long)blockIdx.x * 4 + wave) * 2 * c;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { o[ch * EPC + e] = dg[i][e]; o[c + ch * EPC + e] = db[i][e]; }
        }
    }
}

// out[g][j] = sum_r part[g][r][j] (fp32); grid (ceil(n/256), groups)
__global__ __launch_bounds__(256) void rowsum_f32_kernel(int rows, int n, const float* __restrict__ part, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* p = part + (long)blockIdx.y * rows * n;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += p[(long)r * n + j];
    out[(long)blockIdx.y * n + j] = a;
}

// softmax backward in place on dp: ds = p * (dp - sum_j dp_j p_j) * scale   (rows of n valid columns, pad columns zeroed)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(int n, long lds, float scale, const T* __restrict__ pbuf, T* __restrict__ dpbuf) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float red[4];
    const T* prow = pbuf + (long)blockIdx.x * lds;
    T* drow = dpbuf + (long)blockIdx.x * lds;
    const int tid = threadIdx.x;
    const int npad = (n + EPC - 1) / EPC * EPC;   // forward zeroes p in the pad columns, so they contribute nothing
    float dot = 0.f;
    for (int ch = tid; ch < npad / EPC; ch += 256) {
        Vec16<T> pv = ld16(prow + ch * EPC), dv = ld16(drow + ch * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) dot += (ch * EPC + e < n) ? to_f(pv.e[e]) * to_f(dv.e[e]) : 0.f;
    }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    for (int ch = tid; ch < npad / EPC; ch += 256) {
        Vec16<T> pv = ld16(prow + ch * EPC), dv = ld16(drow + ch * EPC), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            o.e[e] = from_f<T>((ch * EPC + e < n) ? to_f(pv.e[e]) * (to_f(dv.e[e]) - dot) * scale : 0.f);
        st16(drow + ch * EPC, o);
    }
}

// GEGLU backward: dh[:, :c] = dy * gelu(gate), dh[:, c:] = dy * value * gelu'(gate)
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(long rows, int c, int ldh, int lddy, int lddh, const T* __restrict__ hbuf,
                                                        const T* __restrict__ dy, T* __restrict__ dh) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cch = c / EPC;
    const long total = rows * cch;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long r = it / cch;
        const int ch = (int)(it - r * cch);
        Vec16<T> a = ld16(hbuf + r * ldh + ch * EPC), gt = ld16(hbuf + r * ldh + c + ch * EPC), d = ld16(dy + r * lddy + ch * EPC), oa, og;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float g = to_f(gt.e[e]), dd = to_f(d.e[e]);
            const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
            oa.e[e] = from_f<T>(dd * g * cdf);
            og.e[e] = from_f<T>(dd * to_f(a.e[e]) * (cdf + g * pdf));
        }
        st16(dh + r * lddh + ch * EPC, oa);
        st16(dh + r * lddh + c + ch * EPC, og);
    }
}

}  // namespace e2eft

using namespace e2eft;

static int gn_validate(const E2eftGroupNormDesc* d) {
    E2EFT_REQUIRE(d, "groupnorm: null desc");
    E2EFT_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "groupnorm: bad dtype");
    const int epc = 16 / (int)dtype_size(d->dtype);
    const int C = d->c1 + d->c2;
    E2EFT_REQUIRE(d->batch > 0 && d->hw > 0 && d->groups > 0, "groupnorm: geometry");
    E2EFT_REQUIRE(d->c1 > 0 && d->c1 % epc == 0 && d->c2 >= 0 && d->c2 % epc == 0, "groupnorm: channels (%d,%d) must be multiples of %d", d->c1, d->c2, epc);
    E2EFT_REQUIRE(C % d->groups == 0, "groupnorm: %d channels not divisible by %d groups", C, d->groups);
    E2EFT_REQUIRE(d->ldx1 >= d->c1 && d->ldx1 % epc == 0 && d->ldy >= C && d->ldy % epc == 0, "groupnorm: strides");
    E2EFT_REQUIRE(d->c2 == 0 || (d->ldx2 >= d->c2 && d->ldx2 % epc == 0), "groupnorm: ldx2");
    return 0;
}

extern "C" size_t e2eft_groupnorm_workspace_bytes(const E2eftGroupNormDesc* d) {
    if (gn_validate(d)) return 0;
    return gn_ws_bytes(d);
}

extern "C" int e2eft_groupnorm_fwd_pre(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma,
                                       const void* beta, void* y, const float* partial1, int32_t nslabs1,
                                       const float* partial2, int32_t nslabs2, void* workspace, size_t ws_bytes, void* stream) {
    if (int e = gn_validate(d)) return e;
    E2EFT_REQUIRE(x1 && y && workspace, "groupnorm: null pointer");
    E2EFT_REQUIRE(d->c2 == 0 || x2, "groupnorm: x2 missing");
    E2EFT_REQUIRE((!partial1 || nslabs1 > 0) && (!partial2 || nslabs2 > 0), "groupnorm: precomputed statistics need a slab count");
    GnGeom g;
    gn_geom(d, g);
    const size_t need = gn_ws_bytes(d);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "groupnorm: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(g.batch <= 65535 && g.nchb <= 65535, "groupnorm: grid");
    E2EFT_DISPATCH_DTYPE(d->dtype, T, return gn_run<T>(d, g, x1, x2, gamma, beta, y, partial1, nslabs1, d->c2 > 0 ? partial2 : nullptr, nslabs2,
                                                      workspace, (hipStream_t)stream));
    return 0;
}

extern "C" size_t e2eft_groupnorm_coeff_offset(const E2eftGroupNormDesc* d) {
    if (gn_validate(d)) return 0;
    return gn_ws_bytes(d) - ((size_t)d->batch * (d->c1 + d->c2) * 2 + (size_t)d->batch * d->groups) * sizeof(float);
}

extern "C" int e2eft_groupnorm_fwd_stats(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const float* partial1,
                                         int32_t nslabs1, const float* partial2, int32_t nslabs2, void* workspace, size_t ws_bytes, void* stream) {
    if (int e = gn_validate(d)) return e;
    E2EFT_REQUIRE(x1 && workspace, "groupnorm_stats: null pointer");
    E2EFT_REQUIRE(d->c2 == 0 || x2, "groupnorm_stats: x2 missing");
    E2EFT_REQUIRE((!partial1 || nslabs1 > 0) && (!partial2 || nslabs2 > 0), "groupnorm_stats: precomputed statistics need a slab count");
    GnGeom g;
    gn_geom(d, g);
    const size_t need = gn_ws_bytes(d);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "groupnorm_stats: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(g.batch <= 65535 && g.nchb <= 65535, "groupnorm_stats: grid");
    E2EFT_DISPATCH_DTYPE(d->dtype, T, return gn_run<T>(d, g, x1, x2, gamma, nullptr, nullptr, partial1, nslabs1, d->c2 > 0 ? partial2 : nullptr, nslabs2,
                                                      workspace, (hipStream_t)stream));
    return 0;
}

// fp32 GroupNorm(+SiLU) -> f16 split planes (csrc/f32split.hip): the statistics exactly as e2eft_groupnorm_fwd_stats (the workspace afterwards is what
// e2eft_groupnorm_bwd expects), then gn_apply_split_kernel
extern "C" int e2eft_groupnorm_fwd_split(const E2eftGroupNormDesc* d, const float* x, const float* gamma, const float* beta, void* planes, int32_t ldp, float* scale,
                                         const float* partial1, int32_t nslabs1, void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(d && d->dtype == E2EFT_F32 && d->c2 == 0, "groupnorm_fwd_split: one fp32 source");
    E2EFT_REQUIRE(planes && ldp >= 2 * d->c1 && ldp % 4 == 0 && ((uintptr_t)planes & 7) == 0 && d->c1 % 4 == 0, "groupnorm_fwd_split: planes ldp=%d", (int)ldp);
    E2EFT_REQUIRE(scale && ((uintptr_t)scale & 3) == 0, "groupnorm_fwd_split: scale workspace");
    const int rc = e2eft_groupnorm_fwd_stats(d, x, nullptr, gamma, partial1, nslabs1, nullptr, 0, workspace, ws_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gn_split_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d->c1, sqrtf((float)d->hw * (float)(d->c1 / d->groups)), gamma, beta, scale);
    GnGeom g;
    gn_geom(d, g);
    const float* ad = (const float*)((const char*)workspace + e2eft_groupnorm_coeff_offset(d));
    const long tensor_bytes = (long)d->batch * d->hw * g.C * 4L;
    const int it = option(E2EFT_OPT_GN_APPLY_ITERS);
    g.slab = g.pl * 4 * (it > 0 ? it : (tensor_bytes <= (128L << 20) ? 4 : 2));
    g.nslabs = (d->hw + g.slab - 1) / g.slab;
    hipLaunchKernelGGL(gn_apply_split_kernel, dim3(g.nslabs, g.batch, g.nchb), dim3(256), 0, (hipStream_t)stream, g, d->silu, (int)ldp, (const float*)scale, x, ad, beta, (f16*)planes);
    return check_launch("groupnorm_fwd_split");
}

extern "C" int e2eft_groupnorm_fwd(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma,
                                   const void* beta, void* y, void* workspace, size_t ws_bytes, void* stream) {
    return e2eft_groupnorm_fwd_pre(d, x1, x2, gamma, beta, y, nullptr, 0, nullptr, 0, workspace, ws_bytes, stream);
}

extern "C" int e2eft_layernorm_fwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldx, int32_t ldy, float eps,
                                   const void* x, const void* gamma, const void* beta, void* y, void* stream) {
    E2EFT_REQUIRE(x && gamma && beta && y, "layernorm: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "layernorm: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && c > 0 && c % epc == 0 && ldx >= c && ldy >= c && ldx % epc == 0 && ldy % epc == 0, "layernorm: shape c=%d ldx=%d ldy=%d", c, ldx, ldy);
    const int nchunks = c / epc;
    E2EFT_REQUIRE(nchunks <= 64 * 8, "layernorm: c=%d too large", c);
    const long nblk = (rows + 3) / 4;
    E2EFT_REQUIRE(nblk < 2147483647L, "layernorm: too many rows");
    hipStream_t s = (hipStream_t)stream;
    const int iters = (nchunks + 63) / 64;
#define LN_LAUNCH(T, MAXI) hipLaunchKernelGGL((layernorm_kernel<T, MAXI>), dim3((unsigned)nblk), dim3(256), 0, s, (long)rows, c, ldx, ldy, eps, (const T*)x, (const T*)gamma, (const T*)beta, (T*)y)
    E2EFT_DISPATCH_DTYPE(dtype, T, {
        if (iters <= 1) LN_LAUNCH(T, 1);
        else if (iters <= 2) LN_LAUNCH(T, 2);
        else if (iters <= 4) LN_LAUNCH(T, 4);
        else LN_LAUNCH(T, 8);
    });
#undef LN_LAUNCH
    return check_launch("layernorm");
}

extern "C" int e2eft_softmax_rows(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, void* sbuf, void* stream) {
    E2EFT_REQUIRE(sbuf, "softmax: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "softmax: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && rows < 2147483647L && n > 0 && lds >= (n + epc - 1) / epc * epc && lds % epc == 0, "softmax: shape n=%d lds=%ld", n, (long)lds);
    E2EFT_REQUIRE(scale > 0.f, "softmax: scale must be positive");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)rows), dim3(256), 0, s, n, (long)lds, scale, (T*)sbuf, 0));
    return check_launch("softmax_rows");
}

extern "C" int e2eft_softmax_rows_causal(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, int32_t nq, void* sbuf, void* stream) {
    E2EFT_REQUIRE(sbuf, "softmax_causal: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "softmax_causal: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && rows < 2147483647L && n > 0 && lds >= (n + epc - 1) / epc * epc && lds % epc == 0, "softmax_causal: shape n=%d lds=%ld", n, (long)lds);
    E2EFT_REQUIRE(nq > 0 && nq <= n && rows % nq == 0, "softmax_causal: nq=%d must divide rows and be <= n (self-attention)", nq);
    E2EFT_REQUIRE(scale > 0.f, "softmax_causal: scale must be positive");
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)rows), dim3(256), 0, s, n, (long)lds, scale, (T*)sbuf, (int)nq));
    return check_launch("softmax_rows_causal");
}

extern "C" int e2eft_geglu_fwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldh, int32_t ldy, const void* h, void* y, void* stream) {
    E2EFT_REQUIRE(h && y, "geglu: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "geglu: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && c > 0 && c % epc == 0 && ldh >= 2 * c && ldh % epc == 0 && ldy >= c && ldy % epc == 0, "geglu: shape");
    const long total = rows * (c / epc);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((geglu_kernel<T>), dim3((unsigned)nb), dim3(256), 0, s, (long)rows, c, ldh, ldy, (const T*)h, (T*)y));
    return check_launch("geglu");
}

// ---------------------------------------------------------------------------------------------------------------
// backward entry points
extern "C" size_t e2eft_groupnorm_bwd_workspace_bytes(const E2eftGroupNormDesc* d) {
    if (gn_validate(d)) return 0;
    return gn_bwd_ws_bytes(d);
}

extern "C" int e2eft_groupnorm_bwd(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const void* beta,
                                   const void* dy, int32_t lddy, void* dx, int32_t lddx, float* dgamma, float* dbeta,
                                   const void* fwd_workspace, void* workspace, size_t ws_bytes, void* stream) {
    return e2eft_groupnorm_bwd_add(d, x1, x2, gamma, beta, dy, lddy, nullptr, 0, dx, lddx, dgamma, dbeta, fwd_workspace, workspace, ws_bytes, stream);
}

extern "C" int e2eft_groupnorm_bwd_add(const E2eftGroupNormDesc* d, const void* x1, const void* x2, const void* gamma, const void* beta,
                                       const void* dy, int32_t lddy, const void* dx_add, int32_t ldadd, void* dx, int32_t lddx, float* dgamma,
                                       float* dbeta, const void* fwd_workspace, void* workspace, size_t ws_bytes, void* stream) {
    if (int e = gn_validate(d)) return e;
    E2EFT_REQUIRE(x1 && dy && fwd_workspace && workspace, "groupnorm_bwd: null pointer");
    E2EFT_REQUIRE(d->c2 == 0 || x2, "groupnorm_bwd: x2 missing");
    const int epc = 16 / (int)dtype_size(d->dtype);
    const int C = d->c1 + d->c2;
    E2EFT_REQUIRE(lddy >= C && lddy % epc == 0 && (!dx || (lddx >= C && lddx % epc == 0)), "groupnorm_bwd: strides");
    E2EFT_REQUIRE(!dx_add || (dx && ldadd >= C && ldadd % epc == 0 && ((uintptr_t)dx_add & 15) == 0), "groupnorm_bwd: dx_add");
    GnGeom g;
    gn_geom(d, g);
    const size_t need = gn_bwd_ws_bytes(d);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "groupnorm_bwd: workspace %zu < %zu", ws_bytes, need);
    E2EFT_REQUIRE(g.batch <= 65535 && g.nchb <= 65535, "groupnorm_bwd: grid");
    E2EFT_DISPATCH_DTYPE(d->dtype, T, return gn_bwd_run<T>(d, g, x1, x2, gamma, beta, dy, lddy, dx, lddx, dgamma, dbeta,
                                                          (const float*)fwd_workspace, workspace, dx_add, ldadd, (hipStream_t)stream));
    return 0;
}

static const int LN_BWD_STAGE = 32;   // the per-wave partial rows are reduced in two deterministic stages (LN_BWD_STAGE groups, then one)

static int ln_bwd_blocks(int64_t rows) {
    long nb = (rows + 3) / 4;
    nb = nb > 512 ? 512 : nb;
    if (nb * 4 > LN_BWD_STAGE) nb = nb * 4 / LN_BWD_STAGE * LN_BWD_STAGE / 4;   // partial rows divisible by the stage count
    return (int)nb;
}

extern "C" size_t e2eft_layernorm_bwd_workspace_bytes(int64_t rows, int32_t c) {
    return rows > 0 && c > 0 ? ((size_t)ln_bwd_blocks(rows) * 4 + LN_BWD_STAGE) * 2 * (size_t)c * sizeof(float) : 0;
}

extern "C" int e2eft_layernorm_bwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldx, int32_t lddy, int32_t lddx, float eps,
                                   const void* x, const void* gamma, const void* dy, void* dx, float* dgamma_dbeta /* [2][c] */,
                                   void* workspace, size_t ws_bytes, void* stream) {
    E2EFT_REQUIRE(x && gamma && dy && dgamma_dbeta && workspace, "layernorm_bwd: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "layernorm_bwd: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && c > 0 && c % epc == 0 && ldx >= c && lddy >= c && ldx % epc == 0 && lddy % epc == 0 &&
                      (!dx || (lddx >= c && lddx % epc == 0)), "layernorm_bwd: shape c=%d", c);
    const int nchunks = c / epc;
    E2EFT_REQUIRE(nchunks <= 64 * 8, "layernorm_bwd: c=%d too large", c);
    const size_t need = e2eft_layernorm_bwd_workspace_bytes(rows, c);
    if (ws_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "layernorm_bwd: workspace %zu < %zu", ws_bytes, need);
    const int nblk = ln_bwd_blocks(rows);
    hipStream_t s = (hipStream_t)stream;
    const int iters = (nchunks + 63) / 64;
#define LNB_LAUNCH(T, MAXI) hipLaunchKernelGGL((layernorm_bwd_kernel<T, MAXI>), dim3((unsigned)nblk), dim3(256), 0, s, (long)rows, c, ldx, lddy, lddx, eps, (const T*)x, (const T*)gamma, (const T*)dy, (T*)dx, (float*)workspace)
    E2EFT_DISPATCH_DTYPE(dtype, T, {
        if (iters <= 1) LNB_LAUNCH(T, 1);
        else if (iters <= 2) LNB_LAUNCH(T, 2);
        else if (iters <= 4) LNB_LAUNCH(T, 4);
        else LNB_LAUNCH(T, 8);
    });
#undef LNB_LAUNCH
    const int prow = nblk * 4;
    if (prow > LN_BWD_STAGE && prow % LN_BWD_STAGE == 0) {
        float* stage = (float*)workspace + (size_t)prow * 2 * c;
        hipLaunchKernelGGL(rowsum_f32_kernel, dim3(cdiv(2 * c, 256), LN_BWD_STAGE), dim3(256), 0, s, prow / LN_BWD_STAGE, 2 * c, (const float*)workspace, stage);
        hipLaunchKernelGGL(rowsum_f32_kernel, dim3(cdiv(2 * c, 256), 1), dim3(256), 0, s, LN_BWD_STAGE, 2 * c, (const float*)stage, dgamma_dbeta);
    } else {
        hipLaunchKernelGGL(rowsum_f32_kernel, dim3(cdiv(2 * c, 256), 1), dim3(256), 0, s, prow, 2 * c, (const float*)workspace, dgamma_dbeta);
    }
    return check_launch("layernorm_bwd");
}

extern "C" int e2eft_softmax_bwd_rows(int32_t dtype, int64_t rows, int32_t n, int64_t lds, float scale, const void* p, void* dp, void* stream) {
    E2EFT_REQUIRE(p && dp, "softmax_bwd: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "softmax_bwd: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && rows < 2147483647L && n > 0 && lds >= (n + epc - 1) / epc * epc && lds % epc == 0, "softmax_bwd: shape n=%d lds=%ld", n, (long)lds);
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((softmax_bwd_rows_kernel<T>), dim3((unsigned)rows), dim3(256), 0, s, n, (long)lds, scale, (const T*)p, (T*)dp));
    return check_launch("softmax_bwd_rows");
}

extern "C" int e2eft_geglu_bwd(int32_t dtype, int64_t rows, int32_t c, int32_t ldh, int32_t lddy, int32_t lddh, const void* h, const void* dy,
                               void* dh, void* stream) {
    E2EFT_REQUIRE(h && dy && dh, "geglu_bwd: null pointer");
    E2EFT_REQUIRE(dtype >= 0 && dtype <= 2, "geglu_bwd: bad dtype");
    const int epc = 16 / (int)dtype_size(dtype);
    E2EFT_REQUIRE(rows > 0 && c > 0 && c % epc == 0 && ldh >= 2 * c && ldh % epc == 0 && lddy >= c && lddy % epc == 0 && lddh >= 2 * c && lddh % epc == 0,
                  "geglu_bwd: shape");
    const long total = rows * (c / epc);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipStream_t s = (hipStream_t)stream;
    E2EFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((geglu_bwd_kernel<T>), dim3((unsigned)nb), dim3(256), 0, s, (long)rows, c, ldh, lddy, lddh,
                                                      (const T*)h, (const T*)dy, (T*)dh));
    return check_launch("geglu_bwd");
}
