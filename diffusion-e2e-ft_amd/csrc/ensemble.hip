// ensemble.hip — test-time ensembling of the N predictions of one image (HBM-bound streams over an [N, P] fp32 stack).
//   ensemble_depths   — /root/reference/Marigold/marigold/util/ensemble.py:40-132 (twin: GeoWizard/geowizard/utils/depth_ensemble.py)
//   ensemble_normals  — /root/reference/Marigold/marigold/marigold_pipeline.py:58-71 (twin: GeoWizard/geowizard/utils/normal_ensemble.py)
//
// The reference evaluates its alignment objective with N(N-1)/2 full-resolution difference images per call.  Here the pairwise
// term comes from sufficient statistics gathered in ONE pass (Gram matrix G = D D^T, row sums): with a_i = s_i d_i + t_i,
//   sum_px (a_i - a_j)^2 = s_i^2 G_ii + s_j^2 G_jj - 2 s_i s_j G_ij + 2 (s_i S_i - s_j S_j)(t_i - t_j) + P (t_i - t_j)^2,
// so only the regulariser (min / max of the per-pixel median) touches the stack again.  All reductions are two-stage over a FIXED
// number of partials (no atomics): results are bit-reproducible.
#include "common.h"
#include <math.h>

namespace e2eft {

constexpr int ENS_MAX = 32;      // images per ensemble (the reference's default ensemble_size is 10)
constexpr int ENS_BLOCKS = 256;  // partials per reduction

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// block (256 threads) min / max -> out[0], out[1] by thread 0
__device__ __forceinline__ void block_minmax(float mn, float mx, float* out) {
    __shared__ float red[2][4];
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        out[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}

// ---- per-image min / max (ensemble.py:68-69): grid (ENS_BLOCKS, N) -> part [N][ENS_BLOCKS][2] ------------------------------
__global__ __launch_bounds__(256) void ens_minmax_partial(long npix, const float* __restrict__ x, float* __restrict__ part) {
    const float* p = x + (long)blockIdx.y * npix;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)ENS_BLOCKS * 256) {
        const float v = p[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, part + ((long)blockIdx.y * ENS_BLOCKS + blockIdx.x) * 2);
}
// grid (rows), block 256: out[row] = (min, max) over the row's ENS_BLOCKS partial pairs
__global__ __launch_bounds__(256) void ens_minmax_final(const float* __restrict__ part, float* __restrict__ out) {
    const float* p = part + (long)blockIdx.x * ENS_BLOCKS * 2;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < ENS_BLOCKS; i += 256) {
        mn = fminf(mn, p[i * 2]);
        mx = fmaxf(mx, p[i * 2 + 1]);
    }
    block_minmax(mn, mx, out + blockIdx.x * 2);
}

// ---- Gram matrix row + row sum: grid (ENS_BLOCKS, N), block 256 -> part [N][ENS_BLOCKS][N + 1] doubles -------------------
__global__ __launch_bounds__(256) void ens_gram_partial(int n, long npix, const float* __restrict__ x, double* __restrict__ part) {
    __shared__ double red[4][ENS_MAX + 1];
    const int i = blockIdx.y;
    const float* pi = x + (long)i * npix;
    double acc[ENS_MAX + 1];
#pragma unroll
    for (int j = 0; j <= ENS_MAX; ++j) acc[j] = 0.0;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)ENS_BLOCKS * 256) {
        const double di = pi[p];
#pragma unroll
        for (int j = 0; j < ENS_MAX; ++j)
            if (j < n) acc[j] += di * (double)x[(long)j * npix + p];
        acc[ENS_MAX] += di;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j <= ENS_MAX; ++j) {
        const double v = wave_sum_f64(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x <= n) {
        const int j = threadIdx.x == n ? ENS_MAX : threadIdx.x;
        part[((long)i * ENS_BLOCKS + blockIdx.x) * (n + 1) + threadIdx.x] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    }
}
// grid (N), block 64: sum the ENS_BLOCKS partials of row i in a fixed order
__global__ void ens_gram_final(int n, const double* __restrict__ part, double* __restrict__ gram, double* __restrict__ sums) {
    const int i = blockIdx.x, j = threadIdx.x;
    if (j > n) return;
    double s = 0.0;
    for (int b = 0; b < ENS_BLOCKS; ++b) s += part[((long)i * ENS_BLOCKS + b) * (n + 1) + j];
    if (j < n) gram[i * n + j] = s;
    else sums[i] = s;
}

// x * s + t as two roundings (torch: `input_images * s + t`, ensemble.py:119); HIP's __fmul_rn / __fadd_rn are plain operators that
// hipcc's default -ffp-contract=fast fuses, hence the pragma
__device__ __forceinline__ float mul_then_add(float x, float s, float t) {
#pragma clang fp contract(off)
    const float m = x * s;
    return m + t;
}

// ---- aligned stack -> per-pixel median (lower, as torch.median) + MAD, or mean + unbiased std; global min / max of the result ---
template <int NMAX> __device__ __forceinline__ void sort_asc(float (&a)[NMAX]) {   // odd-even transposition network, static indices
#pragma unroll
    for (int r = 0; r < NMAX; ++r) {
#pragma unroll
        for (int k = r & 1; k + 1 < NMAX; k += 2) {
            const float lo = fminf(a[k], a[k + 1]), hi = fmaxf(a[k], a[k + 1]);
            a[k] = lo;
            a[k + 1] = hi;
        }
    }
}

// grid (ENS_BLOCKS), block 256
template <int NMAX>
__global__ __launch_bounds__(256) void ens_depth_reduce(int n, long npix, const float* __restrict__ x, const float* __restrict__ sv,
                                                        const float* __restrict__ tv, int use_mean, float* __restrict__ pred,
                                                        float* __restrict__ unc, float* __restrict__ part /* [ENS_BLOCKS][2] */) {
    float s[NMAX], t[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        s[k] = k < n ? sv[k] : 0.f;
        t[k] = k < n ? tv[k] : 0.f;
    }
    float mn = INFINITY, mx = -INFINITY;
    const int mid = (n - 1) >> 1;   // torch.median returns the LOWER of the two middle values
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)ENS_BLOCKS * 256) {
        float a[NMAX];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) a[k] = k < n ? mul_then_add(x[(long)k * npix + p], s[k], t[k]) : INFINITY;
        float centre, spread;
        if (use_mean) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k < n) sum += a[k];
            centre = sum / (float)n;
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k < n) ss += (a[k] - centre) * (a[k] - centre);
            spread = sqrtf(ss / (float)(n - 1));                       // torch.std: Bessel's correction
        } else {
            float b[NMAX];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) b[k] = a[k];
            sort_asc<NMAX>(b);
            centre = 0.f;
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k == mid) centre = b[k];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) b[k] = k < n ? fabsf(a[k] - centre) : INFINITY;   // MAD (ensemble.py:124-127)
            sort_asc<NMAX>(b);
            spread = 0.f;
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k == mid) spread = b[k];
        }
        if (pred) pred[p] = centre;
        if (unc) unc[p] = spread;
        mn = fminf(mn, centre);
        mx = fmaxf(mx, centre);
    }
    block_minmax(mn, mx, part + blockIdx.x * 2);
}

// (x - min) / (max - min), uncertainty / (max - min)   (ensemble.py:129-133); the range stays on the device
__global__ __launch_bounds__(256) void ens_depth_finish(long npix, const float* __restrict__ minmax, float* __restrict__ pred,
                                                        float* __restrict__ unc) {
    const float mn = minmax[0], range = minmax[1] - minmax[0];
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        if (pred) pred[p] = (pred[p] - mn) / range;
        if (unc) unc[p] = unc[p] / range;
    }
}

// ---- normals: unit vectors, mean spherical angles, angular error of every prediction to the mean direction ------------------
// x, unit: [N][3][hw].  grid (ENS_BLOCKS), block 256 -> part [ENS_BLOCKS][N] doubles.  The loops over the members stay ROLLED (the
// unrolled version inlined atan2f / acosf 32 times: 256 VGPRs + 7 KB of scratch per lane, 5.8 ms for 10 x 768^2); the per-member
// error sums live in LDS, one private column per thread; the second loop re-reads the unit vectors this thread just wrote.
__global__ __launch_bounds__(256) void ens_normals_partial(int n, long hw, const float* __restrict__ x, float* __restrict__ unit,
                                                           double* __restrict__ part) {
    __shared__ double acc[ENS_MAX][256];
    __shared__ double red[4];
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < n; ++k) acc[k][tid] = 0.0;
    for (long p = (long)blockIdx.x * 256 + tid; p < hw; p += (long)ENS_BLOCKS * 256) {
        float phi = 0.f, theta = 0.f;
#pragma unroll 1
        for (int k = 0; k < n; ++k) {
            const float* v = x + (long)k * 3 * hw + p;
            const float a = v[0], b = v[hw], c = v[2 * hw];
            const float d = sqrtf(a * a + b * b + c * c) + 1e-5f;             // marigold_pipeline.py:61
            const float u0 = a / d, u1 = b / d, u2 = c / d;
            float* o = unit + (long)k * 3 * hw + p;
            o[0] = u0; o[hw] = u1; o[2 * hw] = u2;
            phi += atan2f(u1, u0);                                              // :62
            theta += atan2f(sqrtf(u0 * u0 + u1 * u1), u2);                      // :63
        }
        phi /= (float)n;
        theta /= (float)n;
        const float st = sinf(theta);
        const float r0 = st * cosf(phi), r1 = st * sinf(phi), r2 = cosf(theta);   // :64-67
        const float rn = fmaxf(sqrtf(r0 * r0 + r1 * r1 + r2 * r2), 1e-8f);
#pragma unroll 1
        for (int k = 0; k < n; ++k) {
            const float* o = unit + (long)k * 3 * hw + p;                       // written above by this very thread
            const float u0 = o[0], u1 = o[hw], u2 = o[2 * hw];
            const float un = fmaxf(sqrtf(u0 * u0 + u1 * u1 + u2 * u2), 1e-8f);   // F.cosine_similarity, eps = 1e-8
            float cs = (r0 * u0 + r1 * u1 + r2 * u2) / (rn * un);
            cs = fminf(fmaxf(cs, -0.999f), 0.999f);                              // :68
            acc[k][tid] += (double)acosf(cs);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        const double v = wave_sum_f64(acc[k][tid]);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) part[(long)blockIdx.x * n + k] = red[0] + red[1] + red[2] + red[3];
    }
}
__global__ void ens_normals_final(int n, const double* __restrict__ part, double* __restrict__ err) {
    const int k = threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
    for (int b = 0; b < ENS_BLOCKS; ++b) s += part[(long)b * n + k];
    err[k] = s;
}

static size_t ens_ws_bytes(int n) {
    const size_t a = (size_t)n * ENS_BLOCKS * 2 * sizeof(float);            // min / max partials
    const size_t b = (size_t)n * ENS_BLOCKS * (n + 1) * sizeof(double);     // Gram partials
    return a > b ? a : b;
}

}  // namespace e2eft

using namespace e2eft;

#define ENS_CHECK_N(n) E2EFT_REQUIRE((n) >= 1 && (n) <= ENS_MAX, "ensemble: n_img = %d outside [1, %d]", (int)(n), ENS_MAX)

extern "C" size_t e2eft_ensemble_workspace_bytes(int32_t n_img) {
    if (n_img < 1 || n_img > ENS_MAX) {
        fail(E2EFT_ERR_BAD_ARG, "ensemble: n_img = %d outside [1, %d]", (int)n_img, ENS_MAX);
        return 0;
    }
    return ens_ws_bytes(n_img);
}

extern "C" int e2eft_ensemble_minmax(int32_t n_img, int64_t npix, const float* x, float* out, void* workspace, size_t ws_bytes,
                                     void* stream) {
    ENS_CHECK_N(n_img);
    E2EFT_REQUIRE(npix >= 1 && x && out && workspace, "ensemble_minmax: null pointer or empty stack");
    E2EFT_REQUIRE(ws_bytes >= ens_ws_bytes(n_img), "ensemble_minmax: workspace %zu < %zu bytes", ws_bytes, ens_ws_bytes(n_img));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ens_minmax_partial, dim3(ENS_BLOCKS, n_img), dim3(256), 0, s, (long)npix, x, (float*)workspace);
    hipLaunchKernelGGL(ens_minmax_final, dim3(n_img), dim3(256), 0, s, (const float*)workspace, out);
    return check_launch("ensemble_minmax");
}

extern "C" int e2eft_ensemble_gram(int32_t n_img, int64_t npix, const float* x, double* gram, double* sums, void* workspace,
                                   size_t ws_bytes, void* stream) {
    ENS_CHECK_N(n_img);
    E2EFT_REQUIRE(npix >= 1 && x && gram && sums && workspace, "ensemble_gram: null pointer or empty stack");
    E2EFT_REQUIRE(ws_bytes >= ens_ws_bytes(n_img), "ensemble_gram: workspace %zu < %zu bytes", ws_bytes, ens_ws_bytes(n_img));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ens_gram_partial, dim3(ENS_BLOCKS, n_img), dim3(256), 0, s, (int)n_img, (long)npix, x, (double*)workspace);
    hipLaunchKernelGGL(ens_gram_final, dim3(n_img), dim3(64), 0, s, (int)n_img, (const double*)workspace, gram, sums);
    return check_launch("ensemble_gram");
}

extern "C" int e2eft_ensemble_depth_reduce(int32_t n_img, int64_t npix, const float* x, const float* scale, const float* shift,
                                           int32_t use_mean, float* pred, float* uncertainty, float* minmax, void* workspace,
                                           size_t ws_bytes, void* stream) {
    ENS_CHECK_N(n_img);
    E2EFT_REQUIRE(npix >= 1 && x && scale && shift && minmax && workspace, "ensemble_depth_reduce: null pointer or empty stack");
    E2EFT_REQUIRE(ws_bytes >= ens_ws_bytes(n_img), "ensemble_depth_reduce: workspace %zu < %zu bytes", ws_bytes, ens_ws_bytes(n_img));
    E2EFT_REQUIRE(!use_mean || n_img >= 2, "ensemble_depth_reduce: the mean / std reduction needs at least 2 images");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
#define ENS_LAUNCH(NM) hipLaunchKernelGGL((ens_depth_reduce<NM>), dim3(ENS_BLOCKS), dim3(256), 0, s, (int)n_img, (long)npix, x, scale, shift, (int)use_mean, pred, uncertainty, part)
    if (n_img <= 4) ENS_LAUNCH(4);
    else if (n_img <= 8) ENS_LAUNCH(8);
    else if (n_img <= 16) ENS_LAUNCH(16);
    else ENS_LAUNCH(32);
#undef ENS_LAUNCH
    hipLaunchKernelGGL(ens_minmax_final, dim3(1), dim3(256), 0, s, (const float*)part, minmax);
    return check_launch("ensemble_depth_reduce");
}

extern "C" int e2eft_ensemble_depth_finish(int64_t npix, const float* minmax, float* pred, float* uncertainty, void* stream) {
    E2EFT_REQUIRE(npix >= 1 && minmax && (pred || uncertainty), "ensemble_depth_finish: null pointer or empty image");
    hipLaunchKernelGGL(ens_depth_finish, dim3(cdiv(npix, 256 * 8) < 1024 ? cdiv(npix, 256 * 8) : 1024), dim3(256), 0, (hipStream_t)stream,
                       (long)npix, minmax, pred, uncertainty);
    return check_launch("ensemble_depth_finish");
}

extern "C" int e2eft_ensemble_normals(int32_t n_img, int64_t hw, const float* x, float* unit, double* err, void* workspace,
                                      size_t ws_bytes, void* stream) {
    ENS_CHECK_N(n_img);
    E2EFT_REQUIRE(hw >= 1 && x && unit && err && workspace, "ensemble_normals: null pointer or empty stack");
    E2EFT_REQUIRE(ws_bytes >= ens_ws_bytes(n_img), "ensemble_normals: workspace %zu < %zu bytes", ws_bytes, ens_ws_bytes(n_img));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ens_normals_partial, dim3(ENS_BLOCKS), dim3(256), 0, s, (int)n_img, (long)hw, x, unit, (double*)workspace);
    hipLaunchKernelGGL(ens_normals_final, dim3(1), dim3(64), 0, s, (int)n_img, (const double*)workspace, err);
    return check_launch("ensemble_normals");
}
