// convin.hip — 3x3 / stride-1 / pad-1 convolution whose INPUT has eight channels (one 16-byte unit per pixel): the VAE encoder's conv_in
// (3 -> 128 on the 8-channel padded image at full resolution), gfx950, fp16 / bf16.
//
// Why a kernel of its own: K = 72.  The implicit-GEMM kernels stage 64-deep k-tiles through LDS and pay their per-tile fixed cost every two k-tiles —
// igemm2 needs 0.75 ms for conv_in at 8 x 768^2 -> 128, three times the 0.22 ms it takes to WRITE the 1.2 GB output.  Here nothing is staged:
//   * an MFMA 32x32x16 k-step is two filter taps x eight channels, so the A fragment of lane (pixel, k-half h) at k-step j is ONE 16-byte global load:
//     the pixel's eight channels at tap 2 j + h (tap 9 = zeros); zero padding = an out-of-range buffer offset.  Ten loads per lane and tile, issued
//     for the NEXT tile before this tile's epilogue starts;
//   * the B fragments of a wave — two 32-column blocks x five k-steps — are ten 16-byte loads from the [Cout][9 * 8] weight matrix and stay in 40
//     registers for the whole launch (reloaded only when the N tile changes);
//   * 20 MFMAs per wave and tile, then igemm5 / igemm6's packed epilogue (alpha / bias in the accumulator layout, two rows per dword through a
//     wave-private LDS window, 16-byte stores) and their GroupNorm statistics — an output tile is an 8 x 32-pixel block of one image as in igemm6.
// Persistent workgroups (one per CU) walk contiguous per-XCD chunks of the tile range.  Bound: the output stream.
// Eligibility (host, launch_conv_thin_in returns -1): 16-bit, 3x3 / stride 1 / pad 1, exactly 8 input channels at pixel stride 8, one source, no
// residual / row vector, width % 32 == 0, height % 8 == 0, Cout % 8 == 0, at least two tiles per CU.  Summation order: taps in order, fp32.
#include "igemm.h"
#include <atomic>
#include <type_traits>

namespace e2eft {

namespace thin {
constexpr int BM = 256, BN = 128, NW = 8;
constexpr int TH = 8, TW = 32;
constexpr int WIN = NW * 4096;                 // the epilogue's per-wave windows
constexpr int DEP = NW * 64 * 3 * 4;
constexpr int LDS = WIN + DEP;
constexpr unsigned int OOB = 0xF0000000u;
constexpr unsigned int RECORDS = 0xE0000000u;
}  // namespace thin

template <typename T> struct MmaT;
template <> struct MmaT<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct MmaT<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ int fast_div_t(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

template <typename T>
__global__ __launch_bounds__(512) void conv_thin_in_kernel(const IgemmParams p, const int total_tiles) {
    using namespace thin;
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int er = lane >> 3, ec = lane & 7;
    const int H = p.hin, W = p.win;
    const int tw = W / TW, tpi = (H / TH) * tw;

    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int q8 = total_tiles >> 3, r8 = total_tiles & 7;
    const int cbeg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cend = cbeg + (xcd < r8 ? q8 + 1 : q8);
    int u = cbeg + slot;
    if (u >= cend) return;

    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ rowadd = nullptr;       // (names the shared epilogue expects; this kernel has neither a row vector nor a residual)
    constexpr bool has_res = false;
    const bool has_ra = false, stats = p.gn_partial != nullptr;
    Vec16<T> pre_res[4], pre_bias, pre_ra;
    T col_bias[2], col_ra[2];
    long c_orow = 0, c_rrow = 0;
    int c_n0 = -1, c_mt = 0, c_ncl = 0;
    bool c_colok = false;
    (void)pre_res; (void)pre_bias; (void)pre_ra; (void)rowadd; (void)c_rrow; (void)c_ncl;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, RECORDS, 0x00020000);
    // tile -> (M tile, N tile) with N fastest, M tile -> (image, 8 x 32 block)
    int t_mt = 0, t_n0 = 0, t_img = 0, t_oy0 = 0, t_ox0 = 0;
    auto coords = [&](const int v) {
        int mt = v, nt = 0;
        if (p.ntiles > 1) { mt = fast_div_t(v, p.ntiles); nt = v - mt * p.ntiles; }
        t_mt = mt; t_n0 = nt * BN;
        t_img = fast_div_t(mt, tpi);
        const int rem = mt - t_img * tpi;
        const int ty = fast_div_t(rem, tw);
        t_oy0 = ty * TH; t_ox0 = (rem - ty * tw) * TW;
    };
    // A fragments of tile (t_img, t_oy0, t_ox0): afr[i][j] = the eight channels of pixel (row 2 wm + i, column l31) shifted by tap 2 j + h
    u32x4 afr[2][5];
    auto load_a = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tap = 2 * j + h;
                const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), kx = tap - 3 * ky;
                const int iy = t_oy0 + 2 * wm + i + ky - 1, ix = t_ox0 + l31 + kx - 1;
                const bool ok = tap < 9 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned off = ok ? (unsigned)(((t_img * H + iy) * W + ix) * 16) : OOB;
                afr[i][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
            }
    };
    // B fragments of N tile n0: bfr[jb][j] = weights of output channel n0 + 64 wn + 32 jb + l31, taps 2 j + h
    u32x4 bfr[2][5];
    auto load_b = [&](const int n0) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int n = n0 + wn * 64 + jb * 32 + l31;
            const T* wr = (const T*)p.w + (long)(n < p.N ? n : 0) * p.ldw;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tap = 2 * j + h;
                bfr[jb][j] = (tap < 9 && n < p.N) ? *reinterpret_cast<const u32x4*>(wr + tap * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };

    floatx16 acc[2][2];
    auto epi_rofs = [&](const int r) -> long { return (long)(r & 31) + (long)(r >> 5) * W; };
    auto epi_rows_left = [&]() -> int { return 0x40000000; };
    constexpr int EPI_DEP = WIN;
#define EPI_STAMP(i) do { } while (0)
#include "igemm_persistent_epilogue.inc"
#undef EPI_STAMP
    (void)epilogue;

    coords(u);
    load_a();
    for (;;) {
        // ---- this tile: epilogue addressing, weights (if the N tile changed), the 20 MFMAs
        if (t_n0 != c_n0) {
            load_b(t_n0);
            const int cA = t_n0 + wn * 64 + l31, cB = cA + 32;
            if (bias) { col_bias[0] = bias[cA < p.N ? cA : t_n0]; col_bias[1] = bias[cB < p.N ? cB : t_n0]; }
        }
        c_n0 = t_n0; c_mt = t_mt;
        c_colok = c_n0 + wn * 64 + ec * 8 < p.N;
        c_ncl = c_colok ? c_n0 + wn * 64 + ec * 8 : c_n0;
        c_orow = (((long)t_img * H + t_oy0 + 2 * wm) * W + t_ox0 + er) * p.ldo + c_ncl;
        const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[0][0] = MmaT<T>::run(afr[0][0], bfr[0][0], z);
        acc[0][1] = MmaT<T>::run(afr[0][0], bfr[1][0], z);
        acc[1][0] = MmaT<T>::run(afr[1][0], bfr[0][0], z);
        acc[1][1] = MmaT<T>::run(afr[1][0], bfr[1][0], z);
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            acc[0][0] = MmaT<T>::run(afr[0][j], bfr[0][j], acc[0][0]);
            acc[0][1] = MmaT<T>::run(afr[0][j], bfr[1][j], acc[0][1]);
            acc[1][0] = MmaT<T>::run(afr[1][j], bfr[0][j], acc[1][0]);
            acc[1][1] = MmaT<T>::run(afr[1][j], bfr[1][j], acc[1][1]);
        }
        // ---- the next tile's operand loads fly under this tile's epilogue
        u += nslots;
        const bool more = u < cend;
        if (more) { coords(u); load_a(); }
        epilogue_packed(0, 0L);                    // windows at LDS offset 0 (+ wave * 4096 inside)
        if (stats) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // every wave's deposits are written
            asm volatile("" ::: "memory");
            { const int img = c_mt / tpi; combine(img, c_mt - img * tpi, c_n0); }
            __builtin_amdgcn_s_barrier();            // ... and merged, before the next tile's deposits overwrite them
            asm volatile("" ::: "memory");
        }
        if (!more) break;
    }
}

static std::atomic<long> g_thin_launches{0};
int device_cus();   // api.hip

int launch_conv_thin_in(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    using namespace thin;
    if (!option(E2EFT_OPT_PERSISTENT) || !option(E2EFT_OPT_THIN_INPUT_CONV)) return -1;
    if (mode != 1 || nz != 1 || (dtype != E2EFT_F16 && dtype != E2EFT_BF16)) return -1;
    if (p.ksplit_taps > 0 || p.bias_along_m || p.residual || p.rowadd || p.x2) return -1;
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.zins > 1) return -1;
    if (p.hl != p.hin || p.wl != p.win || p.hout != p.hin || p.wout != p.win) return -1;
    if (p.cin != 8 || p.c1 != 8 || p.ldx1 != 8 || p.K != 72 || p.ldw % 8 != 0) return -1;
    if (p.win % TW != 0 || p.hin % TH != 0) return -1;
    if (p.N % 8 != 0 || p.ldo % 8 != 0 || (((uintptr_t)p.out) & 15) != 0 || (((uintptr_t)p.x1) & 15) != 0 || (((uintptr_t)p.w) & 15) != 0) return -1;
    if ((long)p.M * 16 >= 0xD0000000L || p.M % (p.hin * p.win) != 0) return -1;
    int cus = device_cus();
    if (cus == 0) return -1;
    const int gopt = option(E2EFT_OPT_PERSISTENT_GRID);
    if (gopt >= 8 && gopt < cus) cus = gopt;
    const int mtiles = p.M / BM, ntiles = cdiv(p.N, BN);
    const long total = (long)mtiles * ntiles;
    if (total < 2L * cus || total > 2000000000L || mtiles >= (1 << 22)) return -1;
    if (p.gn_partial) {
        if (p.rows_per_img != p.hin * p.win) return -1;
        p.gn_nslabs = p.rows_per_img / BM;
    }
    p.mtiles = mtiles;
    p.ntiles = ntiles;
    g_thin_launches.fetch_add(1, std::memory_order_relaxed);
    if (dtype == E2EFT_F16) hipLaunchKernelGGL((conv_thin_in_kernel<f16>), dim3(cus), dim3(512), 0, s, p, (int)total);
    else hipLaunchKernelGGL((conv_thin_in_kernel<bf16>), dim3(cus), dim3(512), 0, s, p, (int)total);
    tag_kernel("conv_thin_in_kernel<%s>", dtype == E2EFT_F16 ? "_Float16" : "__bf16");
    return check_launch("conv_thin_in");
}

}  // namespace e2eft

extern "C" long e2eft_debug_thin_launches(void) { return e2eft::g_thin_launches.load(); }   // not part of include/e2eft.h
