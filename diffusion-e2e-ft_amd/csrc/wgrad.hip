// wgrad.hip — weight gradients of convolutions and Linears straight from the NHWC tensors (fp16 / bf16), gfx950.
//
//   dW[co][(ky, kx, ci)] = alpha * sum_p dY[p][co] * X[pixel(p) (+) (ky, kx)][ci]          (the reference: torch autograd of nn.Conv2d / nn.Linear in the
//   UNet, training/train.py:563 `accelerator.backward(loss)`; a Linear is the 1x1 case with p = token)
//
// Both operands are contracted over their SLOW index (pixels), while an MFMA fragment wants 8 consecutive k per lane.  Round 1 / 2 therefore
// materialised K-contiguous copies first — e2eft_transpose(dY), e2eft_conv2d_im2col_t(X) (9x the input for a 3x3 filter) — and ran the NT GEMM on
// them: 24 of the 544 ms of an E2E-FT step were those copies.  Here the tiles stay pixel-major in LDS, exactly as they lie in HBM, and the k-contiguous
// fragments are produced by the transpose read ds_read_b64_tr_b16 (a lane of a 16-lane group supplies the address of 4 consecutive channels of one
// pixel and receives 4 pixels of one channel): no transposed or im2col'ed tensor exists anywhere.
//
//   * workgroup = 128 output channels x 256 columns, or 256 x 128 where that grid multiplies less padding (64-channel chunks of (tap, ci) space: a chunk never straddles a tap or a concat source, so
//     its rows are one shifted window of ONE input tensor) x a range of pixels (split-K over the grid's z: a weight gradient has a handful of output
//     tiles and 10^4 - 10^5 pixels); 8 waves, 64 x 64 each, one workgroup per CU;
//   * k-tile = 64 pixels: six [64 pixels][64 channels] panels (two of dY, four of X) of 8 KB, filled by LDS-DMA in 1-KiB pieces of 8 pixel rows
//     (`buffer_load ... lds`; the im2col shift, the zero padding, the stride and the split's end are the per-lane source offset — out of range reads
//     zeros), three stages: the pieces of k-tile t + 2 fly while k-tile t multiplies; 16-byte chunk c of pixel row r lands in slot
//     c ^ (((r >> 1) & 1) << 2): the four rows of a transpose read then cover four different 64-byte bank windows (conflict-free), and the key is
//     constant per lane;
//   * partial sums leave as fp32 [split][Co][kh kw cin]; e2eft_colsum (the reduction the split-K NT path already used) adds the splits.
// Eligibility is decided here (returns E2EFT_ERR_UNSUPPORTED and the caller keeps the transpose + im2col_t + GEMM path): cin and c1 multiples
// of 64, no fused upsample, tensors below 4 GB.  fp32 (round 6): wgrad32_kernel below, the same decomposition on v_mfma_f32_32x32x2_f32.
#include "common.h"
#include <type_traits>

namespace e2eft {

namespace wg {
constexpr int BK = 64;                     // (8 waves per workgroup) tile: 64 NA output channels x 64 (6 - NA) columns, NA = 2 (128 x 256) or 4 (256 x 128)
constexpr int PANEL = 64 * 128;            // [64 pixels][64 channels] of 16-bit
constexpr int STAGE = 6 * PANEL;           // dY panels 0 .. NA - 1, then the X panels
constexpr int NSTAGE = 3;
constexpr int LDS = NSTAGE * STAGE;        // 147,456 B: one 8-wave workgroup per CU
}  // namespace wg

struct WgradParams {
    const void* dy;
    const void* x1;
    const void* x2;
    float* out;
    int ldy, ldx1, ldx2, c1, cin;
    int batch, hin, win, hout, wout, kh, kw, stride, pad_t, pad_l;
    int M, N, P;            // Co, kh * kw * cin, batch * hout * wout
    int kchunk, nsplit;     // pixels per split (multiple of 64)
    float alpha;
};

template <typename T> struct MmaW;
template <> struct MmaW<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct MmaW<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef short short4vw __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 tr_read_w(const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4vw*)p));
}
// one LDS-DMA piece (64 lanes x 16 B -> 1 KiB at m0) from asm: the compiler must not count it (it would drain vmcnt in front of LDS reads it cannot
// prove disjoint); the kernel counts its own pieces — six per wave and k-tile, always.  m0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void dma_piece_w(const __amdgpu_buffer_rsrc_t& rs, const unsigned voff, const unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ int fdiv_w(int n, int d) {   // float estimate + one correction (quotients below 2^22)
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// grid (ceil(N / 256), ceil(M / 128), nsplit), 512 threads.  Round 3, second version: the first one (128 x 128 tiles, 4 waves, two stages, two
// workgroups per CU) waited for k-tile t + 1 at the end of k-tile t — 16 MFMAs = 0.25 us of cover for a 1-2 us L2 round trip — and paid two float
// divisions + 64-bit multiplies per piece: 340-670 TF/s, latency-bound.  Now: three stages (the pieces of k-tile t + 2 are in flight while t multiplies,
// `vmcnt(6)` in front of the barrier is exact because every wave issues six pieces per k-tile, out-of-range ones fetch zeros), a lane's pixel is
// carried incrementally (one division per k-tile), one 8-wave workgroup per CU with a 128 x 256 tile (weights of a chunk's four (tap, channel)
// windows share the dY panels: 1.5 operand bytes per MFMA instead of 2).
// NA: dY panels per stage.  2: 128 output channels x 256 columns (two dY, four X panels); 4: 256 x 128 (four dY, two X) for Cout >> the column count
// of a 1x1 layer — the host picks the shape that multiplies less padding (wgrad_plan).
template <typename T, int NA>
__global__ __launch_bounds__(512) void wgrad_kernel(const WgradParams p) {
    using namespace wg;
    constexpr int NB = 6 - NA, BM = 64 * NA, BN = 64 * NB;
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NA == 2 ? wave >> 2 : wave >> 1, wn = NA == 2 ? wave & 3 : wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int k_begin = blockIdx.z * p.kchunk, k_end = min(p.P, k_begin + p.kchunk);
    const unsigned OOB = 0xFFFFFFF0u;

    // ---- the four 64-column chunks of this tile: (tap, first channel, source).  A chunk never straddles a tap or a concat source.
    const int cpt = p.cin >> 6;                        // chunks per tap
    int ci0[NB], ky[NB], kx[NB];
    bool cok[NB], src2[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = (n0 >> 6) + j;
        cok[j] = q * 64 < p.N;
        const int tap = cok[j] ? q / cpt : 0;
        ci0[j] = cok[j] ? (q - tap * cpt) * 64 : 0;
        ky[j] = tap / p.kw; kx[j] = tap - ky[j] * p.kw;
        src2[j] = ci0[j] >= p.c1;
    }
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)(((long)(p.P - 1) * p.ldy + p.M) * (long)sizeof(T)), 0x00020000);
    const long xpix = (long)p.batch * p.hin * p.win;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (unsigned)(((xpix - 1) * p.ldx1 + p.c1) * (long)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0,
                                                                         (unsigned)(((xpix - 1) * (p.x2 ? p.ldx2 : p.ldx1) + (p.x2 ? p.cin - p.c1 : p.c1)) * (long)sizeof(T)), 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)((lptr_t)smem);

    // ---- loader: wave w moves piece w (pixel rows 8 w .. 8 w + 7 of the k-tile) of all six panels.  Lane l: pixel row r = 8 w + (l >> 3), LDS slot
    // l & 7 of that row = source chunk (l & 7) ^ key(r).  The lane's output pixel advances by 64 per k-tile: (image, position inside the image) are
    // carried, only the row / column split is recomputed.
    const int r_kt = 8 * wave + (lane >> 3);
    const int sc8 = ((lane & 7) ^ (((r_kt >> 1) & 1) << 2)) * 8;      // first channel (of 64) of this lane's 16 bytes
    const int hw_out = p.hout * p.wout;
    int pix = k_begin + r_kt;                                          // this lane's output pixel in the NEXT k-tile to issue
    int bimg = fdiv_w(pix, hw_out);
    int rem = pix - bimg * hw_out;
    unsigned ycol[NA];                                                 // dY byte offsets of the lane's columns (OOB beyond M)
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int col = m0 + 64 * a + sc8;
        ycol[a] = col < p.M ? (unsigned)col * (unsigned)sizeof(T) : OOB;
    }
    const unsigned ldyb = (unsigned)p.ldy * (unsigned)sizeof(T);
    auto issue = [&](const int stage) {
        const bool pok = pix < k_end;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(stage * STAGE + wave * 1024)));
        const unsigned yrow = (unsigned)pix * ldyb;
#pragma unroll
        for (int a = 0; a < NA; ++a) dma_piece_w(rsy, (pok && ycol[a] != OOB) ? yrow + ycol[a] : OOB, dst + (unsigned)(a * PANEL));
        const int oy = fdiv_w(rem, p.wout), ox = rem - oy * p.wout;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        const int irow0 = bimg * p.hin;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int iy = iy0 + ky[c], ix = ix0 + kx[c];
            const bool ok = pok && cok[c] && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
            const unsigned ipix = (unsigned)((irow0 + iy) * p.win + ix);
            if (src2[c]) {      // (uniform branch: a descriptor select would leave the SGPRs)
                dma_piece_w(rs2, ok ? (ipix * (unsigned)p.ldx2 + (unsigned)(ci0[c] - p.c1 + sc8)) * (unsigned)sizeof(T) : OOB, dst + (unsigned)((NA + c) * PANEL));
            } else {
                dma_piece_w(rs1, ok ? (ipix * (unsigned)p.ldx1 + (unsigned)(ci0[c] + sc8)) * (unsigned)sizeof(T) : OOB, dst + (unsigned)((NA + c) * PANEL));
            }
        }
        pix += BK;
        rem += BK;
        while (rem >= hw_out) { rem -= hw_out; ++bimg; }
    };

    // ---- fragment addresses (transpose reads).  32x32x16 operand of lane (column = l31, k-slots 8 hh .. + 7): two reads of 4 pixel rows each.
    // 16-lane group g serves columns 16 (g & 1) .. + 15 of the 32-block; lane i of the group supplies pixel row (i >> 2) of the 4-row run and
    // channels 4 (i & 3) .. + 3.  row = 16 ks + 8 hh [+ 4] + (i >> 2): (row >> 1) & 1 = (i >> 3) & 1 — the swizzle key is a lane constant.
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int key = ((i16 >> 3) & 1) << 2;
    auto frag_off = [&](const int blk) {      // byte offset inside a panel of this lane's first read for 32-column block `blk` (0 / 1), k-step 0
        const int chunk = (4 * blk + 2 * g1 + ((i16 >> 1) & 1)) ^ key;
        return (8 * hh + (i16 >> 2)) * 128 + chunk * 16 + (i16 & 1) * 8;
    };
    const int fo[2] = {frag_off(0), frag_off(1)};
    auto frag = [&](const char* panel, const int blk, const int ks) -> u32x4 {
        const char* a = panel + fo[blk] + ks * (16 * 128);
        const u32x2 v0 = tr_read_w(a);
        const u32x2 v1 = tr_read_w(a + 4 * 128);
        return u32x4{v0[0], v0[1], v1[0], v1[1]};
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (k_end - k_begin + BK - 1) / BK;
    if (nkt > 0) {
        issue(0);
        issue(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // k-tile 0 (this wave's pieces; the barrier makes it everybody's)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int st = 0, st2 = 2;                                     // stage multiplied now / stage that receives k-tile kt + 2
        for (int kt = 0; kt < nkt; ++kt) {
            issue(st2);                                          // (beyond the split's end: zeros into a stage nobody reads — the count stays six)
            const char* pa = smem + st * STAGE + wm * PANEL;
            const char* pb = smem + st * STAGE + (NA + wn) * PANEL;
            // fragments one k-step ahead of their MFMAs; the synchronisation that opens k-tile kt + 1 sits in front of the LAST MFMA group (every
            // LDS read of this k-tile has returned by then): the barrier skew of the eight waves runs under four MFMAs (igemm5.hip's placement)
            u32x4 fa[2][2], fb[2][2];
            fa[0][0] = frag(pa, 0, 0); fa[0][1] = frag(pa, 1, 0); fb[0][0] = frag(pb, 0, 0); fb[0][1] = frag(pb, 1, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks & 1, n = c ^ 1;
                if (ks < 3) { fa[n][0] = frag(pa, 0, ks + 1); fa[n][1] = frag(pa, 1, ks + 1); fb[n][0] = frag(pb, 0, ks + 1); fb[n][1] = frag(pb, 1, ks + 1); }
                if (ks == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // this wave's pieces of k-tile kt + 1 have landed (kt + 2 may still fly); its reads of k-tile kt have returned
                    __builtin_amdgcn_s_barrier();                                  // ... everybody's
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[0][0] = MmaW<T>::run(fa[c][0], fb[c][0], acc[0][0]);
                acc[0][1] = MmaW<T>::run(fa[c][0], fb[c][1], acc[0][1]);
                acc[1][0] = MmaW<T>::run(fa[c][1], fb[c][0], acc[1][0]);
                acc[1][1] = MmaW<T>::run(fa[c][1], fb[c][1], acc[1][1]);
            }
            asm volatile("" ::: "memory");
            st2 = st;
            st = st == NSTAGE - 1 ? 0 : st + 1;
        }
    }
    // ---- partial tile out: fp32 [split][M][N]; a lane holds column n of 16 rows per block (rows (r & 3) + 8 (r >> 2) + 4 hh)
    float* out = p.out + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + 32 * j + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m < p.M) out[(long)m * p.N + n] = acc[i][j][r] * p.alpha;
            }
    }
}

// ---- strict fp32 (round 6) -----------------------------------------------------------------------------------------------------------------------------------
// The reference trains with `--mixed_precision "no"` (training/scripts/train_marigold_e2e_ft_depth.sh:15): its weight gradients are fp32 contractions over the
// pixels.  Rounds 3-5 kept them on e2eft_transpose + e2eft_conv2d_im2col_t + split-K batched GEMMs (two materialised K-contiguous copies, 9x the input for a 3x3
// filter, and 256-row GEMM tiles that a 320-channel layer fills to 62 %): 48 + 190 ms of the 2.95 s step at 50-100 TF/s.  The fp32 MFMA (v_mfma_f32_32x32x2_f32,
// exact fp32 products and sums) takes ONE float per lane and operand — A[i = lane & 31][k = lane >> 5] — so with the contraction index = pixel both operands are
// plain `ds_read_b32` of the tiles as they lie in HBM:   A = dY[pixel 2 ks + hh][co0 + l31],  B = X[pixel 2 ks + hh (+) tap][ci0 + l31]
// (32 consecutive floats per half-wave: conflict-free without any swizzle), no transposed image, no transpose read.  Same decomposition as the 16-bit kernel:
// 128 x 256 (or 256 x 128) tile, 8 waves of 64 x 64, panels [32 pixels][64 channels] of 8 KB filled by LDS-DMA (a wave's 1-KiB piece = 4 pixel rows of 256 B), three
// stages, six pieces per wave and k-tile (`vmcnt(6)` exact), split over pixels by whole rounds of the machine, fp32 partials reduced by e2eft_colsum.  A k-tile is
// 16 k-steps of 4 MFMAs x 64 cycles per wave: the matrix pipe is the only thing that is busy.
namespace wg32 {
constexpr int BK = 32;                     // pixels per k-tile
constexpr int PANEL = 32 * 256;            // [32 pixels][64 channels] of fp32
constexpr int STAGE = 6 * PANEL;
constexpr int NSTAGE = 3;
constexpr int LDS = NSTAGE * STAGE;        // 147,456 B
}  // namespace wg32

template <int NA>
__global__ __launch_bounds__(512) void wgrad32_kernel(const WgradParams p) {
    using namespace wg32;
    constexpr int NB = 6 - NA, BM = 64 * NA, BN = 64 * NB;
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NA == 2 ? wave >> 2 : wave >> 1, wn = NA == 2 ? wave & 3 : wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int k_begin = blockIdx.z * p.kchunk, k_end = min(p.P, k_begin + p.kchunk);
    const unsigned OOB = 0xFFFFFFF0u;

    const int cpt = p.cin >> 6;
    int ci0[NB], ky[NB], kx[NB];
    bool cok[NB], src2[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = (n0 >> 6) + j;
        cok[j] = q * 64 < p.N;
        const int tap = cok[j] ? q / cpt : 0;
        ci0[j] = cok[j] ? (q - tap * cpt) * 64 : 0;
        ky[j] = tap / p.kw; kx[j] = tap - ky[j] * p.kw;
        src2[j] = ci0[j] >= p.c1;
    }
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)(((long)(p.P - 1) * p.ldy + p.M) * 4L), 0x00020000);
    const long xpix = (long)p.batch * p.hin * p.win;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (unsigned)(((xpix - 1) * p.ldx1 + p.c1) * 4L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0,
                                                                         (unsigned)(((xpix - 1) * (p.x2 ? p.ldx2 : p.ldx1) + (p.x2 ? p.cin - p.c1 : p.c1)) * 4L), 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)((lptr_t)smem);

    // ---- loader: wave w moves piece w (pixel rows 4 w .. 4 w + 3 of the k-tile) of all six panels; lane l: row 4 w + (l >> 4), channels 4 (l & 15) .. + 3 of the chunk
    const int r_kt = 4 * wave + (lane >> 4);
    const int sc4 = (lane & 15) * 4;
    const int hw_out = p.hout * p.wout;
    int pix = k_begin + r_kt;
    int bimg = fdiv_w(pix, hw_out);
    int rem = pix - bimg * hw_out;
    unsigned ycol[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int col = m0 + 64 * a + sc4;
        ycol[a] = col < p.M ? (unsigned)col * 4u : OOB;
    }
    const unsigned ldyb = (unsigned)p.ldy * 4u;
    auto issue = [&](const int stage) {
        const bool pok = pix < k_end;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(stage * STAGE + wave * 1024)));
        const unsigned yrow = (unsigned)pix * ldyb;
#pragma unroll
        for (int a = 0; a < NA; ++a) dma_piece_w(rsy, (pok && ycol[a] != OOB) ? yrow + ycol[a] : OOB, dst + (unsigned)(a * PANEL));
        const int oy = fdiv_w(rem, p.wout), ox = rem - oy * p.wout;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        const int irow0 = bimg * p.hin;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int iy = iy0 + ky[c], ix = ix0 + kx[c];
            const bool ok = pok && cok[c] && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
            const unsigned ipix = (unsigned)((irow0 + iy) * p.win + ix);
            if (src2[c]) {      // (uniform branch)
                dma_piece_w(rs2, ok ? (ipix * (unsigned)p.ldx2 + (unsigned)(ci0[c] - p.c1 + sc4)) * 4u : OOB, dst + (unsigned)((NA + c) * PANEL));
            } else {
                dma_piece_w(rs1, ok ? (ipix * (unsigned)p.ldx1 + (unsigned)(ci0[c] + sc4)) * 4u : OOB, dst + (unsigned)((NA + c) * PANEL));
            }
        }
        pix += BK;
        rem += BK;
        while (rem >= hw_out) { rem -= hw_out; ++bimg; }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operand of k-step ks (pixels 2 ks, 2 ks + 1), 32-channel block blk: one float per lane
    const int fo = hh * 256 + l31 * 4;
    auto opnd = [&](const char* panel, const int blk, const int ks) -> float { return *reinterpret_cast<const float*>(panel + fo + ks * 512 + blk * 128); };

    const int nkt = (k_end - k_begin + BK - 1) / BK;
    if (nkt > 0) {
        issue(0);
        issue(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int st = 0, st2 = 2;
        for (int kt = 0; kt < nkt; ++kt) {
            issue(st2);
            const char* pa = smem + st * STAGE + wm * PANEL;
            const char* pb = smem + st * STAGE + (NA + wn) * PANEL;
            float fa[2][2], fb[2][2];
            fa[0][0] = opnd(pa, 0, 0); fa[0][1] = opnd(pa, 1, 0); fb[0][0] = opnd(pb, 0, 0); fb[0][1] = opnd(pb, 1, 0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int c = ks & 1, n = c ^ 1;
                if (ks < 15) { fa[n][0] = opnd(pa, 0, ks + 1); fa[n][1] = opnd(pa, 1, ks + 1); fb[n][0] = opnd(pb, 0, ks + 1); fb[n][1] = opnd(pb, 1, ks + 1); }
                if (ks == 15) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // this wave's pieces of k-tile kt + 1 have landed; its reads of k-tile kt have returned
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            st2 = st;
            st = st == NSTAGE - 1 ? 0 : st + 1;
        }
    }
    float* out = p.out + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + 32 * j + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m < p.M) out[(long)m * p.N + n] = acc[i][j][r] * p.alpha;
            }
    }
}

int device_cus();   // api.hip

// pixel split: one workgroup per CU and round.  Among the split counts that leave >= 8 k-tiles per workgroup the one with the best product of
// (filled fraction of the last round) x (k-tiles / (k-tiles + 3): the pipeline fill of a workgroup) — e.g. conv 320 -> 320 3x3 at 32 x 72^2:
// 36 tiles x 7 splits = 252 workgroups = 0.98 rounds, not 36 x 22 = 3.09.
static void wgrad_plan(long M, long N, long P, int& nsplit, int& kchunk, int& na, const int bk = wg::BK) {
    // tile shape: the one whose grid multiplies less padding (a 1x1 layer with 320 input and 2560 output channels: 10 x 3 tiles of 256 x 128 = 83 % useful
    // against 20 x 2 tiles of 128 x 256 = 62 %); ties keep 128 x 256
    const long area2 = cdiv(M, 128) * 128L * cdiv(N, 256) * 256L, area4 = cdiv(M, 256) * 256L * cdiv(N, 128) * 128L;
    na = area4 < area2 ? 4 : 2;
    const long tiles = na == 2 ? (long)cdiv(M, 128) * cdiv(N, 256) : (long)cdiv(M, 256) * cdiv(N, 128);
    int cus = device_cus();
    if (cus <= 0) cus = 256;
    const long ktiles = cdiv(P, bk);
    long cap = ktiles / 8;
    if (cap < 1) cap = 1;
    if (cap > 4L * cus) cap = 4L * cus;
    double best = -1.0;
    long best_ns = 1;
    for (long ns = 1; ns <= cap; ++ns) {
        const long kt = cdiv(ktiles, ns);                   // k-tiles per split
        const long real_ns = cdiv(ktiles, kt);
        const long w = tiles * real_ns, rounds = cdiv(w, (long)cus);
        const double score = (double)w / (double)(rounds * cus) * (double)kt / (double)(kt + 3);
        if (score > best * 1.0000001) { best = score; best_ns = real_ns; }
    }
    kchunk = (int)(cdiv(ktiles, best_ns) * bk);
    nsplit = (int)cdiv(P, kchunk);
}

}  // namespace e2eft

using namespace e2eft;

static int wgrad_check(const E2eftConvDesc* d, int lddy) {
    if (!d) return fail(E2EFT_ERR_BAD_ARG, "wgrad: null descriptor");
    if (d->dtype != E2EFT_F16 && d->dtype != E2EFT_BF16 && d->dtype != E2EFT_F32) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: dtype %d", d->dtype);
    const long es = (long)dtype_size(d->dtype);
    const int epc = (int)(16 / es);
    const int cin = d->c1 + d->c2;
    if (cin % 64 != 0 || d->c1 % 64 != 0) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: channel counts (%d, %d) must be multiples of 64", d->c1, d->c2);
    if (d->hl != d->hin || d->wl != d->win) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: fused upsample not supported");
    if (d->batch <= 0 || d->hin <= 0 || d->win <= 0 || d->hout <= 0 || d->wout <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0)
        return fail(E2EFT_ERR_BAD_ARG, "wgrad: geometry");
    const long P = (long)d->batch * d->hout * d->wout, xpix = (long)d->batch * d->hin * d->win;
    const long ldx = d->ldx1 > d->ldx2 ? d->ldx1 : d->ldx2;
    if (P * lddy * es >= 0xFFFF0000L || xpix * ldx * es >= 0xFFFF0000L) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: tensors of 4 GB and more");
    if (P >= (1L << 24)) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: more than 2^24 output pixels (the pixel split uses float reciprocals)");
    if ((long)d->hout * d->wout >= (1L << 22)) return fail(E2EFT_ERR_UNSUPPORTED, "wgrad: image too large for the index arithmetic");
    if (d->ldx1 % epc != 0 || (d->c2 > 0 && d->ldx2 % epc != 0) || lddy % epc != 0 || lddy < d->cout) return fail(E2EFT_ERR_BAD_ARG, "wgrad: row strides");
    return E2EFT_OK;
}

extern "C" size_t e2eft_conv2d_wgrad_workspace_bytes(const E2eftConvDesc* d, int32_t lddy) {
    if (wgrad_check(d, lddy) != E2EFT_OK) return 0;
    int nsplit, kchunk, na;
    const long N = (long)d->kh * d->kw * (d->c1 + d->c2);
    wgrad_plan(d->cout, N, (long)d->batch * d->hout * d->wout, nsplit, kchunk, na, d->dtype == E2EFT_F32 ? wg32::BK : wg::BK);
    return (size_t)nsplit * d->cout * N * sizeof(float);
}

extern "C" int e2eft_conv2d_wgrad(const E2eftConvDesc* d, const void* dy, int32_t lddy, const void* x1, const void* x2, float* partial, size_t partial_bytes,
                                  int32_t* nsplit_out, void* stream) {
    const int rc = wgrad_check(d, lddy);
    if (rc != E2EFT_OK) return rc;
    E2EFT_REQUIRE(dy && x1 && partial && nsplit_out && (d->c2 == 0 || x2), "wgrad: null pointer");
    E2EFT_REQUIRE((((uintptr_t)dy | (uintptr_t)x1 | (uintptr_t)x2) & 15) == 0, "wgrad: pointers must be 16-byte aligned");
    WgradParams p;
    p.dy = dy; p.x1 = x1; p.x2 = d->c2 > 0 ? x2 : nullptr; p.out = partial;
    p.ldy = lddy; p.ldx1 = d->ldx1; p.ldx2 = d->ldx2; p.c1 = d->c1; p.cin = d->c1 + d->c2;
    p.batch = d->batch; p.hin = d->hin; p.win = d->win; p.hout = d->hout; p.wout = d->wout;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->cout; p.N = d->kh * d->kw * p.cin; p.P = d->batch * d->hout * d->wout;
    int na;
    wgrad_plan(p.M, p.N, p.P, p.nsplit, p.kchunk, na, d->dtype == E2EFT_F32 ? wg32::BK : wg::BK);
    p.alpha = d->alpha;
    const size_t need = (size_t)p.nsplit * p.M * p.N * sizeof(float);
    if (partial_bytes < need) return fail(E2EFT_ERR_WORKSPACE, "wgrad: partial buffer %zu < %zu bytes", partial_bytes, need);
    *nsplit_out = p.nsplit;
    dim3 grid(cdiv(p.N, na == 2 ? 256 : 128), cdiv(p.M, na == 2 ? 128 : 256), p.nsplit);
    E2EFT_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "wgrad: grid");
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == E2EFT_F32) {
        if (na == 2) hipLaunchKernelGGL((wgrad32_kernel<2>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((wgrad32_kernel<4>), grid, dim3(512), 0, s, p);
    } else if (d->dtype == E2EFT_F16) {
        if (na == 2) hipLaunchKernelGGL((wgrad_kernel<f16, 2>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((wgrad_kernel<f16, 4>), grid, dim3(512), 0, s, p);
    } else {
        if (na == 2) hipLaunchKernelGGL((wgrad_kernel<bf16, 2>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((wgrad_kernel<bf16, 4>), grid, dim3(512), 0, s, p);
    }
    return check_launch("wgrad");
}
