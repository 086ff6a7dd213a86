// igemm6.hip — persistent implicit GEMM for 3x3 / stride-1 / pad-1 convolutions whose A operand is a 2-D HALO PATCH held in LDS
// (gfx950, fp16 / bf16): the variant of igemm5.hip that stops re-fetching every input pixel once per filter tap.
//
// Why: igemm5 walks K tap-major — k-tile (ky, kx, 64 channels) — and fetches a fresh 256-row x 128-byte A stage for every k-tile, so a
// 64-channel slice of an input pixel travels L2 -> LDS nine times per output tile.  Measured with the A pieces of eight taps in nine
// switched off (profiles/r03d_a_operand_delivery_probe.txt): +7.7 % at 128->128 @768^2, +9.2 % at 256->256 @384^2, +11.2 % at
// 512->512 @192^2 — operand delivery (LDS-DMA issue, LDS write bandwidth, L2 reads and their energy at the package power cap), not MFMA
// issue, is what the k-loop of the 3x3 layers still pays for.  Here
//   * an output tile is an 8-row x 32-column block of ONE image (256 GEMM rows), K runs chunk-major — k-tile (64 channels, tap) — and the
//     A operand of all nine taps of a chunk is one (8+2) x (32+2)-pixel patch: 340 rows of 128 bytes, 43 LDS-DMA pieces instead of
//     9 x 32 = 288;
//   * two patch buffers: while the nine k-tiles of chunk c multiply out of one, the patch of chunk c+1 (or chunk 0 of the workgroup's
//     next tile) arrives in the other — at most one A piece per wave and k-tile, beside the two B pieces that the 3-stage weight ring
//     needs: three pieces in k-tiles 0-5 of a chunk, two in 6-8 (the counts are compile-time constants of the unrolled k-tiles, so all
//     waits are counted; the five patch pieces beyond the 43rd fetch zeros into a dump kilobyte);
//   * the fragment of output pixel (y, x), tap (ky, kx) is patch row (y + ky) * 34 + x + kx: 32 consecutive patch rows per MFMA operand, the
//     same XOR chunk swizzle as igemm2 / igemm5 keyed by the PATCH row (conflict-free for any start row); the nine k-tiles of a chunk are
//     unrolled, the tap shift is an immediate;
//   * zero padding costs nothing: patch rows outside the image carry an out-of-range buffer offset;
//   * no row table, no per-tap offset arithmetic: a lane's six patch rows are fixed for the whole launch, a tile switch is six
//     multiply-adds;
//   * tile-entry arithmetic, epilogue operand prefetch, barrier placement, both epilogues (fp32 sliced with a residual, packed without),
//     GroupNorm statistics: as igemm5.hip, with rows of the tile mapped to the 8 x 32 block (a wave's 64 rows = two image rows).
// Eligibility (host, launch_igemm_patch returns -1 and the caller falls through to igemm5): conv mode, 3x3, stride 1, pad 1, plain or with
// the exact 2x nearest upsample fused into the read (the patch is built in the upsampled geometry), no zero insertion, width a multiple
// of 32, height of 8, channels (each concat part) multiples of 64 and at least 128, the vector epilogue, at least two tiles per
// workgroup.  e2eft_set_option(E2EFT_OPT_PATCH_CONV, 0) disables it.
// Measured (profiles/r03h_patch_conv_ab.txt, one box, igemm5 -> igemm6, TF/s): 128->128 @768^2 896 -> 1011, 256->256 @384^2 1007 -> 1115, 512->512 @192^2
// 1065 -> 1182, 256->128 @768^2 1011 -> 1138: +10.7 ... +13.8 %; the step of BASELINE configs[1] 107.0 -> 99.1 ms.
// NORM variant (inference, cout <= 128; template flag): the input is read through a GroupNorm(+SiLU) that was never applied — see the kernel's comment.
// LDS: 2 x 43 KiB patch buffers + 3 x 16 KiB weight ring + 6 KiB statistics deposits + 8 KiB dump = 148 KiB (+ 5 KiB coefficient table with NORM).
// Summation order: per output element k runs (chunk, tap) instead of (tap, chunk) — fp32 accumulation, results differ from igemm5 in the
// last bits (documented in include/e2eft.h; deterministic run to run).
#include "igemm.h"
#include <atomic>
#include <type_traits>

namespace e2eft {

namespace patchk {
constexpr int BM = 256, BN = 128, NW = 8;
constexpr int TH = 8, TW = 32;
constexpr int B_STAGE = BN * 128;
constexpr int DEP = NW * 64 * 3 * 4;
constexpr int NORM_CMAX = 640;                                          // NORM: input channels the (a, mean, beta) table of one image holds
constexpr unsigned int OOB = 0xF0000000u;
constexpr unsigned int RECORDS = 0xE0000000u;
// KT x KT taps (3: the 3x3 / pad-1 convolutions; 2: round 6, the 2x2 parity phases of the 2x-upsampler convolutions, e2eft_upconv2x_fwd)
template <int KT> struct Geo {
    static constexpr int NT = KT * KT;                                      // k-tiles per 64-channel chunk
    static constexpr int PW = TW + KT - 1, PH = TH + KT - 1, PROWS = PH * PW;   // 340 (3x3) / 297 (2x2) patch rows
    static constexpr int PPIECES = (PROWS + 7) / 8;                         // 43 / 38 one-KiB pieces
    static constexpr int NPI = (PPIECES + NW - 1) / NW;                     // pieces per wave: 6 / 5
    static constexpr int PATCH = PPIECES * 1024;                            // 44,032 / 38,912 B
    static constexpr int OFF_B = 2 * PATCH;
    static constexpr int OFF_DEP = OFF_B + 3 * B_STAGE;
    static constexpr int OFF_DUMP = OFF_DEP + DEP;
    static constexpr int LDS = OFF_DUMP + NW * 1024;                        // 151,552 / 141,312 B
    static constexpr int OFF_TAB = LDS, OFF_TABM = OFF_TAB + NORM_CMAX * 4;   // a2[C], d2[C] (fp32; common.h::gn_fold) of the loader's image
    static constexpr int LDS_NORM = OFF_TABM + NORM_CMAX * 4;               // 156,672 B
    // A pieces a wave issues in k-tile t of a chunk (the next chunk's patch): 3x3 — one in k-tiles 0-5; 2x2 — two in k-tiles 0 and 1, one in k-tile 2.  Never in the
    // chunk's LAST k-tile: the wait that closes it may leave only that k-tile's own two weight pieces in flight (the next chunk reads the patch right after it)
    static constexpr int a_count(int t) { return KT == 3 ? (t < 6 ? 1 : 0) : (t < 2 ? 2 : t == 2 ? 1 : 0); }
    static constexpr int a_first(int t) { return KT == 3 ? t : 2 * t; }
};
}  // namespace patchk

template <typename T> struct Mma6;
template <> struct Mma6<f16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma6<bf16> {
    __device__ static __forceinline__ floatx16 run(const u32x4& a, const u32x4& b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bhalf8, a), __builtin_bit_cast(bhalf8, b), c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void* lptr6_t;
template <int V> using IC6 = std::integral_constant<int, V>;

__device__ __forceinline__ int fast_div6(int n, int d) {   // float estimate + one correction (quotients below 2^22)
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// NORM: the input is read through GroupNorm(+SiLU) — the statistics exist (e2eft_groupnorm_fwd_stats), the apply pass does not: every lane normalises, in
// place in LDS, exactly the 16-byte units of the patch it fetched itself (k-tile t of a chunk: piece t - 2, landed since the previous k-tile's wait), with the
// arithmetic of gn_apply_kernel (norm.hip) — the convolution's result is bit-identical to GroupNorm followed by igemm6.  Padding rows stay zero.
// F32O (round 6, e2eft_conv2d_fwd_f32split): an fp32 convolution whose products are formed on the f16 matrix pipe — A is the two-plane f16 split of an fp32 tensor,
// the weights are split the same way, K runs over the three blocks (x0, w0), (x0, w1), (x1, w0) (IgemmParams::split_c), bias / residual / output are fp32 (epilogue_f32).
template <typename T, bool RES, bool NORM, int KT, bool F32O = false>
__global__ __launch_bounds__(512) void igemm6_kernel(const IgemmParams p, const int total_tiles) {
    using namespace patchk;
    using G = Geo<KT>;
    static_assert(KT == 3 || (KT == 2 && !NORM), "taps: 3x3, or 2x2 without the fused GroupNorm");
    static_assert(!F32O || (RES && !NORM && std::is_same<T, f16>::value), "fp32 output: the fp32-window epilogue, f16 split operands");
    constexpr int NT = G::NT, PW = G::PW, PROWS = G::PROWS, PPIECES = G::PPIECES, NPI = G::NPI, PATCH = G::PATCH, OFF_B = G::OFF_B, OFF_DEP = G::OFF_DEP, OFF_DUMP = G::OFF_DUMP;
    constexpr int OFF_TAB = G::OFF_TAB, OFF_TABM = G::OFF_TABM;
    __shared__ __attribute__((aligned(128))) char smem[NORM ? G::LDS_NORM : G::LDS];
    typedef float f2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int nch = p.cin >> 6;              // 64-channel chunks, host: >= 2
    const int H = p.hl, W = p.wl;            // the convolution's (logical) input = output size; with the fused 2x nearest upsample the source is hin x win = H/2 x W/2
    const int Hs = p.hin, Ws = p.win;
    const bool up2 = p.hl != p.hin;
    const int tw = W / TW, tpi = (H / TH) * tw;   // tiles per image row / per image
    // p.out_seg > 0 (= W; e2eft_upconv2x_fwd): this launch is ONE parity phase of a 2x-upsampled image — p.out points at the phase's first pixel, p.ldo is twice the pixel
    // pitch of the full-resolution tensor (GEMM rows x, x + 1 are two output pixels apart) and every image row y starts 2 W such strides after row y - 1
    const int Wo = (KT == 2 && p.out_seg > 0) ? 2 * W : W;
    const int pad_t = KT == 3 ? 1 : p.pad_t, pad_l = KT == 3 ? 1 : p.pad_l;   // (3x3: compile-time; a 2x2 phase: 1 - py, 1 - px)

    // ---- tile sequence of this workgroup: as igemm5 (eight contiguous chunks of the tile range, one per XCD)
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int q8 = total_tiles >> 3, r8 = total_tiles & 7;
    const int cbeg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cend = cbeg + (xcd < r8 ? q8 + 1 : q8);
    int u_dma = cbeg + slot;
    if (u_dma >= cend) return;

    // ---- loader state
    // A: piece j = wave + 8 i (i = 0..5) covers patch rows 8 j .. 8 j + 7; this lane's row and its swizzled 16-byte chunk are fixed
    const int jc16 = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * 16;   // ((patch row >> 1) & 7 = (4 j + (lane >> 4)) & 7, 32 i = 0 mod 8)
    int pyx[NPI];                             // (patch y << 16) | patch x of the lane's row in piece i; y = 0x4000 for rows beyond the patch
    int pix[NPI];                             // pixel index of that row inside the loader's image, -1 = padding / no tile
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int prow = 8 * (wave + 8 * i) + (lane >> 3);
        const int py = prow / PW;             // (once per launch)
        pyx[i] = prow < PROWS ? (py << 16) | (prow - PW * py) : (0x4000 << 16);
    }
    // B: as igemm5 (rows lrow, lrow + 64 of the 128-row weight stage)
    const int lrow = 8 * wave + (lane >> 3);
    const int jcb = (lane & 7) ^ ((lrow >> 1) & 7);
    unsigned int brow[2];
    int d_n0 = 0, d_img = 0, d_oy0 = 0, d_ox0 = 0, d_mt = 0;   // the loader's tile
    bool dma_done = false;
    __amdgpu_buffer_rsrc_t rs1, rs2, rsw, rsa;
    unsigned int a_ldb = 0, a_coff = 0;       // A pieces of the chunk being fetched: pixel pitch in bytes, channel + swizzle offset
    int n_c0 = 0, tab_img = -1;               // NORM: first of this lane's eight channels in the chunk being fetched; image whose coefficients are in LDS
    const unsigned lds_base = (unsigned)(uintptr_t)((lptr6_t)smem);

    auto tile_coords = [&](const int u) {
        int mt = u, nt = 0;
        if (p.ntiles > 1) { mt = fast_div6(u, p.ntiles); nt = u - mt * p.ntiles; }
        d_mt = mt; d_n0 = nt * BN;
        d_img = fast_div6(mt, tpi);
        const int rem = mt - d_img * tpi;
        const int ty = fast_div6(rem, tw);
        d_oy0 = ty * TH; d_ox0 = (rem - ty * tw) * TW;
    };
    auto set_a = [&](const bool valid) {      // A address state of the loader's tile (d_*): pixel indices of the six rows, image descriptors
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int iy = d_oy0 - pad_t + (pyx[i] >> 16), ix = d_ox0 - pad_l + (pyx[i] & 0xffff);
            const bool ok = valid && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            pix[i] = !ok ? -1 : up2 ? (iy >> 1) * Ws + (ix >> 1) : iy * W + ix;   // (fused upsample: four patch rows fetch the same source pixel — L2 hits)
        }
        const T* b1 = (const T*)p.x1 + (long)d_img * Hs * Ws * p.ldx1;
        const T* b2 = p.x2 ? (const T*)p.x2 + (long)d_img * Hs * Ws * p.ldx2 : b1;
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, RECORDS, 0x00020000);
        rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)b2, 0, RECORDS, 0x00020000);
        if constexpr (NORM) {
            if (valid && d_img != tab_img) {   // (uniform, once per image and workgroup) nobody reads the table between k-tile 8 of a chunk and k-tile 2 of the next
                tab_img = d_img;
                float* ta = reinterpret_cast<float*>(smem + OFF_TAB);
                float* tm = reinterpret_cast<float*>(smem + OFF_TABM);
                for (int c = tid; c < p.cin; c += 512) {      // (a2, d2) of the exp2-domain form (common.h::gn_fold — the operations gn_apply_kernel performs per thread)
                    float a2, d2;
                    gn_fold(p.nrm_ad[((long)d_img * p.cin + c) * 2], p.nrm_ad[((long)d_img * p.cin + c) * 2 + 1], p.nrm_beta ? to_f(((const T*)p.nrm_beta)[c]) : 0.f, a2, d2);
                    ta[c] = a2;
                    tm[c] = d2;
                }
            }
        }
    };
    auto set_b = [&](const bool valid) {      // B address state of the loader's tile
        rsw = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.w + (long)d_n0 * p.ldw), 0, RECORDS, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = d_n0 + lrow + 64 * i;
            brow[i] = (valid && n < p.N) ? (unsigned)((lrow + 64 * i) * p.ldw + jcb * 8) * (unsigned)sizeof(T) : OOB;
        }
    };
    auto set_chunk = [&](const int cn) {      // source of chunk cn's patch: x1 for channels below c1, x2 above (concat)
        int ch = cn * 64;
        if constexpr (F32O) { if (ch >= p.split_c) ch -= p.split_c; }   // blocks (x0, w0), (x0, w1), (x1, w0): plane channels 0 .. C-1, 0 .. C-1, C .. 2C-1
        if (ch < p.c1) { rsa = rs1; a_ldb = (unsigned)p.ldx1 * (unsigned)sizeof(T); a_coff = (unsigned)ch * (unsigned)sizeof(T) + (unsigned)jc16; }
        else { rsa = rs2; a_ldb = (unsigned)p.ldx2 * (unsigned)sizeof(T); a_coff = (unsigned)(ch - p.c1) * (unsigned)sizeof(T) + (unsigned)jc16; }
        n_c0 = ch + (jc16 >> 1);
    };
    int pcur = 0, pnext = PATCH;              // patch buffer being multiplied / being filled
    int bs_cur = OFF_B, bs_nxt = OFF_B + B_STAGE, bs_dst = OFF_B + 2 * B_STAGE;
    auto fire_a = [&](auto ic, const int pdst) {   // piece i (0 .. NPI-1) of the patch being fetched (pieces beyond the last: zeros into the dump kilobyte)
        constexpr int i = decltype(ic)::value;
        unsigned off = OOB;
        int dst = OFF_DUMP + wave * 1024;
        if constexpr (i < NPI) {
            off = pix[i] < 0 ? OOB : (unsigned)pix[i] * a_ldb + a_coff;
            const int j = wave + 8 * i;
            if (j < PPIECES) dst = pdst + j * 1024;   // (wave-uniform)
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr6_t)(smem + dst), 16, off, 0, 0, 0);
    };
    auto fire_b = [&](const int stage, const unsigned kofs) {
        char* sb = smem + stage + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr6_t)sb, 16, brow[0] + kofs, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr6_t)(sb + 64 * 128), 16, brow[1] + kofs, 0, 0, 0);
    };

    // NORM: piece i of the patch in buffer `pbuf` (fetched by this wave, landed): x -> act((x - mean) * a + beta) in place, in three steps that a
    // k-tile spreads over its second half: the LDS reads (patch unit, coefficients) are requested in front of the embedded barrier — its lgkmcnt(0)
    // covers them —, four values are computed behind each of the last two MFMA groups, then the unit is written back.  Inline-asm DS instructions
    // (in front of LDS accesses it can see the compiler may drain vmcnt: pieces are in flight by design).
    u32x4 n_x;
    floatx4 n_a, n_m;           // coefficients of four channels at a time: a2, d2 (natural register pairs for the packed fp32 instructions)
    u32x4 n_o;
    unsigned n_addr = 0, n_t = 0;
    bool n_on = false;
    const bool n_silu = p.nrm_silu != 0;
    auto norm_issue = [&](auto ic, const int pbuf) {   // the patch unit and the coefficients of its first four channels
        constexpr int i = decltype(ic)::value;
        const int j = wave + 8 * i;
        n_on = j < PPIECES;                                          // (wave-uniform) pieces beyond the 43rd: the arithmetic runs on the dump kilobyte and nothing is stored —
        // no branch: a basic-block boundary would keep the arithmetic out of the MFMA shadow
        n_addr = lds_base + (unsigned)(n_on ? pbuf + j * 1024 : OFF_DUMP + wave * 1024) + (unsigned)lane * 16u;
        n_t = lds_base + (unsigned)OFF_TAB + (unsigned)n_c0 * 4u;
        const unsigned ad = n_addr, ta = n_t;
        u32x4 x0;
        floatx4 t0, t1;         // (asm outputs into locals: clang does not capture variables that appear only as asm operands of a nested generic lambda)
        asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(ad) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(ta) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(t1) : "v"(ta), "n"(NORM_CMAX * 4) : "memory");
        n_x = x0; n_a = t0; n_m = t1;
    };
    // round 6: inside the k-loop a lane's sixteen coefficients (a2, d2 of its channel octet — fixed per launch, only the chunk moves) are loaded ONCE per chunk, in the
    // chunk's first norm k-tile, and stay in registers for its six pieces: four of the five ds_read_b128 of a norm k-tile were these (DESIGN.md "Measured (round 6)")
    floatx4 c_a0, c_a1, c_d0, c_d1;
    auto norm_issue_unit = [&](auto ic, const int pbuf) {
        constexpr int i = decltype(ic)::value;
        const int j = wave + 8 * i;
        n_on = j < PPIECES;
        n_addr = lds_base + (unsigned)(n_on ? pbuf + j * 1024 : OFF_DUMP + wave * 1024) + (unsigned)lane * 16u;
        const unsigned ad = n_addr;
        u32x4 x0;
        asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(ad) : "memory");
        n_x = x0;
    };
    auto norm_load_coef = [&]() {
        const unsigned ta = lds_base + (unsigned)OFF_TAB + (unsigned)n_c0 * 4u;
        floatx4 t0, t1, t2, t3;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(ta) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(t1) : "v"(ta) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(t2) : "v"(ta), "n"(NORM_CMAX * 4) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(t3) : "v"(ta), "n"(NORM_CMAX * 4 + 16) : "memory");
        c_a0 = t0; c_a1 = t1; c_d0 = t2; c_d1 = t3;
    };
    auto norm_ready_k = [&](const bool with_coef) {   // as norm_ready: the unit (and, in a chunk's first norm k-tile, the coefficients) are the oldest LDS reads in flight
        u32x4 x0 = n_x;
        if (with_coef) {
            floatx4 t0 = c_a0, t1 = c_a1, t2 = c_d0, t3 = c_d1;
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x0), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) :: "memory");
            c_a0 = t0; c_a1 = t1; c_d0 = t2; c_d1 = t3;
        } else {
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x0) :: "memory");
        }
        n_x = x0;
    };
    auto norm_pair_k = [&](auto hc, auto qc) {   // norm_pair on the persistent coefficients; the result replaces the unit's dword in place
        constexpr int hf = decltype(hc)::value, q2 = decltype(qc)::value;
        typedef T T2 __attribute__((ext_vector_type(2)));
        Vec16<T> v;
        v.raw = n_x;
        constexpr int e = 4 * hf + 2 * q2;
        const floatx4& ca = hf ? c_a1 : c_a0;
        const floatx4& cd = hf ? c_d1 : c_d0;
        const f2 xx = {to_f(v.e[e]), to_f(v.e[e + 1])};
        const f2 aa = {ca[2 * q2], ca[2 * q2 + 1]}, dd = {cd[2 * q2], cd[2 * q2 + 1]};
        const f2 u = __builtin_elementwise_fma(xx, aa, dd);
        f2 t;
        {
            float r0 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-u[0]), GN_L2E, GN_L2E));
            float r1 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-u[1]), GN_L2E, GN_L2E));
            if (!n_silu) { r0 = GN_LN2; r1 = GN_LN2; }
            t = u * f2{r0, r1};
        }
        n_x[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, T2));      // (in place: the pair's own dword — four registers fewer than a separate result)
    };
    auto norm_store_k = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const unsigned ad = (n_on && pix[i] >= 0) ? n_addr : lds_base + (unsigned)(OFF_DUMP + wave * 1024) + (unsigned)lane * 16u;
        const u32x4 ov = n_x;
        asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(ov) : "memory");
    };
    auto norm_issue2 = [&]() {   // the coefficients of the unit's last four channels (into the registers the first half is done with)
        const unsigned ta = n_t;
        floatx4 t0, t1;
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(t0) : "v"(ta) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(t1) : "v"(ta), "n"(NORM_CMAX * 4 + 16) : "memory");
        n_a = t0; n_m = t1;
    };
    auto norm_wait = [&]() {
        u32x4 x0 = n_x;
        floatx4 t0 = n_a, t1 = n_m;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(t0), "+v"(t1) :: "memory");
        n_x = x0; n_a = t0; n_m = t1;
    };
    // the three reads of norm_issue are the OLDEST of the (at most) eleven LDS reads in flight when this runs (eight fragment reads were requested behind them;
    // LDS returns in order, LDS-DMA counts on vmcnt, the loop has no scalar loads): at most eight outstanding = they are back
    auto norm_ready = [&]() {
        u32x4 x0 = n_x;
        floatx4 t0 = n_a, t1 = n_m;
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x0), "+v"(t0), "+v"(t1) :: "memory");
        n_x = x0; n_a = t0; n_m = t1;
    };
    auto norm_mark = [&]() {    // behind a wait that already covered norm_issue2's reads (the barrier's lgkmcnt(0)): only tells the compiler where the values exist
        floatx4 t0 = n_a, t1 = n_m;
        asm volatile("" : "+v"(t0), "+v"(t1) :: "memory");
        n_a = t0; n_m = t1;
    };
    auto norm_pair = [&](auto hc, auto qc) {   // values 4 hf + 2 q2, + 1 of the unit: gn_apply_kernel's arithmetic (norm.hip) on a natural fp32 pair — packed fma / mul, one v_exp + one v_fma + one v_rcp per value, one packed convert (13 VALU instructions; 16 before round 6)
        constexpr int hf = decltype(hc)::value, q2 = decltype(qc)::value;
        typedef T T2 __attribute__((ext_vector_type(2)));
        Vec16<T> v;
        v.raw = n_x;
        constexpr int e = 4 * hf + 2 * q2;
        const f2 xx = {to_f(v.e[e]), to_f(v.e[e + 1])};
        const f2 aa = {n_a[2 * q2], n_a[2 * q2 + 1]}, dd = {n_m[2 * q2], n_m[2 * q2 + 1]};
        const f2 u = __builtin_elementwise_fma(xx, aa, dd);      // log2(e) * ((x - mean) * a + beta): gn_apply_kernel's exp2-domain arithmetic (common.h)
        f2 t;
        {   // gn_act_u (common.h): u * rcp(fma(exp2(-u), log2 e, log2 e)), or u * ln 2 without SiLU (a select, not a branch: see norm_issue).  The constant operands
            // stay scalar instructions: a packed form would broadcast them with op_sel — the operand-swizzle family of DESIGN.md §3.6; fma / the last mul run on natural pairs
            float r0 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-u[0]), GN_L2E, GN_L2E));
            float r1 = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-u[1]), GN_L2E, GN_L2E));
            if (!n_silu) { r0 = GN_LN2; r1 = GN_LN2; }
            t = u * f2{r0, r1};
        }
        n_o[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, T2));
    };
    auto norm_half = [&](auto hc) { norm_pair(hc, IC6<0>{}); norm_pair(hc, IC6<1>{}); };   // (prologue only: nothing to hide behind)
    // one MFMA group with one value pair's arithmetic in its shadow: the matrix pipe takes 32 cycles per MFMA, the wave's VALU issues meanwhile — as long as
    // the instructions alternate.  (Round 3 placed each half behind a whole group: the pipe idled while 30 VALU instructions ran, 806-861 TF/s against 1160-1190.)
    auto interleave4 = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU, then three behind each further MFMA (13 per value pair)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
    };
    auto norm_store = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // padding rows (zeros from the out-of-range fetch) stay zero and pieces beyond the 43rd do not exist: their results go to the dump kilobyte.  An address
        // select, not a branch: the compiler sinks the whole arithmetic into a conditional block around the store, i.e. out of the MFMA shadow
        const unsigned ad = (n_on && pix[i] >= 0) ? n_addr : lds_base + (unsigned)(OFF_DUMP + wave * 1024) + (unsigned)lane * 16u;
        const u32x4 ov = n_o;
        asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(ov) : "memory");
    };

    // "all but the newest n vector-memory operations of this wave are complete, and every LDS read has returned": immediates only, n is a compile-time constant
    // everywhere except the one k-tile per tile that also requests the epilogue's operands (uniform npre)
    auto wait_vm = [&](const int n) {
        switch (n) {
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
    };
    floatx16 acc[2][2];
    // B fragment byte offsets inside a stage (read-side swizzle as igemm2.hip); A: computed per k-tile from the patch row
    int bofs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bofs[g] = (wn * 64 + l31) * 128 + (((g * 2 + h) ^ ((l31 >> 1) & 7)) * 16);
    const int rbase = 2 * wm * PW + l31;      // patch row of this lane's first output pixel at tap (0, 0)
    auto mma_group = [&](const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        acc[0][0] = Mma6<T>::run(a0, b0, acc[0][0]);
        acc[0][1] = Mma6<T>::run(a0, b1, acc[0][1]);
        acc[1][0] = Mma6<T>::run(a1, b0, acc[1][0]);
        acc[1][1] = Mma6<T>::run(a1, b1, acc[1][1]);
    };

    // ---- epilogue operands requested ahead of their use (as igemm5)
    const int er = lane >> 3, ec = lane & 7;
    const T* __restrict__ bias = (const T*)p.bias;
    const T* __restrict__ rowadd = (const T*)p.rowadd;
    constexpr bool has_res = RES && !F32O;      // (F32O requests its fp32 bias / residual inside epilogue_f32)
    const bool has_ra = rowadd != nullptr, stats = p.gn_partial != nullptr;
    Vec16<T> pre_res[4], pre_bias, pre_ra;
    T col_bias[2], col_ra[2];
    long c_orow = 0, c_rrow = 0;              // element offsets of output / residual pixel (image row 2 wm of the tile, column er) at this lane's chunk
    int c_n0 = 0, c_img = 0, c_mt = 0, c_ncl = 0;
    bool c_colok = false;
    const int npre = F32O ? 0 : has_res ? 4 + (bias ? 1 : 0) + (has_ra ? 1 : 0) : 2 * ((bias ? 1 : 0) + (has_ra ? 1 : 0));
    bool nxt = false;

    auto enter_tile = [&]() {   // the MFMA side enters the tile the loader is on; then the loader's coordinates move to the workgroup's next tile
        c_n0 = d_n0; c_img = d_img; c_mt = d_mt;
        c_colok = c_n0 + wn * 64 + ec * 8 < p.N;
        c_ncl = c_colok ? c_n0 + wn * 64 + ec * 8 : c_n0;
        const long py0 = (long)d_img * H + d_oy0 + 2 * wm;
        c_orow = (py0 * Wo + d_ox0 + er) * p.ldo + c_ncl;
        c_rrow = (py0 * W + d_ox0 + er) * p.ldr + c_ncl;
        nxt = u_dma + nslots < cend;
        if (nxt) tile_coords(u_dma + nslots);
    };

    // one k-tile = (chunk c, tap t).  CK: 1 first chunk of a tile (t = 0 multiplies into the constant 0 and carries the tile-entry arithmetic),
    // 2 last chunk (t = 0: the A state moves to the workgroup's next tile — this chunk's pieces are that tile's first patch; t = 7: epilogue
    // operands requested, then the B state moves on; t = 8: last k-tile), 0 otherwise.  Every k-tile issues the two B pieces of k-tile + 2, k-tiles
    // 0-5 one piece of the next patch as well.  The synchronisation that opens the next k-tile sits in front of the last two MFMA groups (igemm5.hip).
    auto ktile = [&](auto tc, auto ckc, const int c) {
        constexpr int t = decltype(tc)::value, CK = decltype(ckc)::value;
        constexpr int TS = (t / KT) * PW + (t % KT);
        constexpr int AN = G::a_count(t), AF = G::a_first(t);   // A pieces of the next patch this k-tile issues
        unsigned kofs;                         // byte offset in a weight row of k-tile + 2 (uniform)
        if constexpr (t + 2 < NT) kofs = (unsigned)((t + 2) * p.cin + c * 64) * (unsigned)sizeof(T);
        else if constexpr (CK == 2) kofs = (unsigned)((t + 2 - NT) * p.cin) * (unsigned)sizeof(T);
        else kofs = (unsigned)((t + 2 - NT) * p.cin + (c + 1) * 64) * (unsigned)sizeof(T);
        const int pr0 = rbase + TS, pr1 = pr0 + PW;
        const int A0 = pcur + pr0 * 128 + ((((pr0 >> 1) & 7) ^ h) << 4);
        const int A1 = pcur + pr1 * 128 + ((((pr1 >> 1) & 7) ^ h) << 4);
        const int sb = bs_cur;
        u32x4 a0[3], a1[3], b0[3], b1[3];
        auto rd = [&](auto gc, auto slotc) {
            constexpr int g = decltype(gc)::value, sl = decltype(slotc)::value;
            a0[sl] = *reinterpret_cast<const u32x4*>(smem + (A0 ^ (g << 5)));
            a1[sl] = *reinterpret_cast<const u32x4*>(smem + (A1 ^ (g << 5)));
            b0[sl] = *reinterpret_cast<const u32x4*>(smem + sb + bofs[g]);
            b1[sl] = *reinterpret_cast<const u32x4*>(smem + sb + bofs[g] + 32 * 128);
        };
        constexpr bool NRM = NORM && t >= 2 && t <= 7;      // this k-tile normalises piece t - 2 of the next patch (landed since the previous k-tile's wait)
        // the unit and the coefficients of its first four channels are requested FIRST: LDS returns in order, so the wait the compiler puts in front of group 0's
        // first MFMA (for fragments requested after these) covers them
        if constexpr (NRM) { norm_issue_unit(IC6<t - 2>{}, pnext); if constexpr (t == 2) norm_load_coef(); }
        rd(IC6<0>{}, IC6<0>{});
        rd(IC6<1>{}, IC6<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NRM) norm_ready_k(t == 2);
        rd(IC6<2>{}, IC6<2>{});
        if constexpr (!(CK == 2 && t == NT - 2)) fire_b(bs_dst, kofs);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CK == 1 && t == 0) {
            const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0][0] = Mma6<T>::run(a0[0], b0[0], z);
            acc[0][1] = Mma6<T>::run(a0[0], b1[0], z);
            acc[1][0] = Mma6<T>::run(a1[0], b0[0], z);
            acc[1][1] = Mma6<T>::run(a1[0], b1[0], z);
        } else {
            mma_group(a0[0], a1[0], b0[0], b1[0]);
        }
        if constexpr (NRM) { norm_pair_k(IC6<0>{}, IC6<0>{}); interleave4(); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CK == 1 && t == 0) enter_tile();
        if constexpr (t == 0) {
            if constexpr (CK == 2) { set_a(nxt); set_chunk(0); }
            else set_chunk(c + 1);
        }
        if constexpr (CK == 2 && t == NT - 2) {
            // every lane requests (columns beyond N read the tile's first chunk instead: the count of VMEM instructions must not depend on exec)
            if constexpr (F32O) {
            } else if (has_res) {
#pragma unroll
                for (int s = 0; s < 4; ++s) pre_res[s] = ld16((const T*)p.residual + c_rrow + (long)(s * 8) * p.ldr);
                if (bias) pre_bias = ld16(bias + c_ncl);
                if (has_ra) pre_ra = ld16(rowadd + (long)c_img * p.N + c_ncl);
            } else {
                const int cA = c_n0 + wn * 64 + l31, cB = cA + 32;
                const int nA = cA < p.N ? cA : c_n0, nB = cB < p.N ? cB : c_n0;
                if (bias) { col_bias[0] = bias[nA]; col_bias[1] = bias[nB]; }
                if (has_ra) { col_ra[0] = rowadd[(long)c_img * p.N + nA]; col_ra[1] = rowadd[(long)c_img * p.N + nB]; }
            }
            u_dma += nslots;
            set_b(nxt);
            if (!nxt) dma_done = true;
            __builtin_amdgcn_sched_barrier(0);
            fire_b(bs_dst, kofs);
        }
        rd(IC6<3>{}, IC6<0>{});
        if constexpr (AN >= 1) fire_a(IC6<AF>{}, pnext);   // (3x3: k-tiles 6-8 of a chunk issue the two weight pieces only)
        if constexpr (AN >= 2) fire_a(IC6<AF + 1>{}, pnext);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[1], a1[1], b0[1], b1[1]);
        if constexpr (NRM) { norm_pair_k(IC6<0>{}, IC6<1>{}); interleave4(); }
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0): this wave's reads of the current stage have RETURNED before the barrier lets others overwrite it
        if constexpr (CK == 2 && t == NT - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (CK == 2 && t == NT - 2) wait_vm(2 + AN + npre);   // younger than the previous k-tile's pieces: this k-tile's operand requests + its weight / patch pieces
        else wait_vm(2 + AN);                                               // all but this k-tile's own pieces have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0[2], a1[2], b0[2], b1[2]);
        if constexpr (NRM) { norm_pair_k(IC6<1>{}, IC6<0>{}); interleave4(); __builtin_amdgcn_sched_barrier(0); }
        mma_group(a0[0], a1[0], b0[0], b1[0]);
        if constexpr (NRM) { norm_pair_k(IC6<1>{}, IC6<1>{}); interleave4(); __builtin_amdgcn_sched_barrier(0); norm_store_k(IC6<t - 2>{}); }
        asm volatile("" ::: "memory");
        { const int x = bs_cur; bs_cur = bs_nxt; bs_nxt = bs_dst; bs_dst = x; }
        if constexpr (t == NT - 1) { const int x = pcur; pcur = pnext; pnext = x; }
    };
    auto chunk = [&](auto ckc, const int c) {
        ktile(IC6<0>{}, ckc, c); ktile(IC6<1>{}, ckc, c); ktile(IC6<2>{}, ckc, c); ktile(IC6<3>{}, ckc, c);
        if constexpr (KT == 3) {
            ktile(IC6<4>{}, ckc, c); ktile(IC6<5>{}, ckc, c); ktile(IC6<6>{}, ckc, c); ktile(IC6<7>{}, ckc, c); ktile(IC6<8>{}, ckc, c);
        }
    };

    // ---- epilogues and statistics merge: shared with igemm5.hip; rows of a wave's 64-row block are two image rows of 32 pixels here.
    // Packed (no residual): rounds of 16 rows = half an image row; fp32 (residual): slices 4-7 = the wave's second image row.
    auto epi_rofs = [&](const int r) -> long { return (long)(r & 31) + (long)(r >> 5) * Wo; };
    auto epi_rows_left = [&]() -> int { return 0x40000000; };      // (tiles are whole 8 x 32 blocks of one image)
    constexpr int EPI_DEP = OFF_DEP;
#define EPI_STAMP(i) do { } while (0)
#include "igemm_persistent_epilogue.inc"
#undef EPI_STAMP
    const float al_f32 = F32O ? p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f) * (p.alpha_dev2 ? *p.alpha_dev2 : 1.f) : 0.f;      // (epilogue_f32 of the .inc)
    // ================================ main ==========================================================================
    tile_coords(u_dma);
    set_a(true);
    set_b(true);
    set_chunk(0);
    fire_a(IC6<0>{}, pcur); fire_a(IC6<1>{}, pcur); fire_a(IC6<2>{}, pcur);
    fire_a(IC6<3>{}, pcur); fire_a(IC6<4>{}, pcur);
    if constexpr (NPI > 5) fire_a(IC6<5>{}, pcur);
    fire_b(bs_cur, 0u);
    fire_b(bs_nxt, (unsigned)p.cin * (unsigned)sizeof(T));
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the first patch and k-tile 0's weights
    if constexpr (NORM) {
        __syncthreads();                                      // the coefficient table (written in set_a) is visible
        auto whole = [&](auto ic) { norm_issue(ic, pcur); norm_wait(); norm_half(IC6<0>{}); norm_issue2(); norm_wait(); norm_half(IC6<1>{}); norm_store(ic); };
        whole(IC6<0>{}); whole(IC6<1>{}); whole(IC6<2>{}); whole(IC6<3>{}); whole(IC6<4>{}); whole(IC6<5>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (;;) {
        chunk(IC6<1>{}, 0);
        for (int c = 1; c < nch - 1; ++c) chunk(IC6<0>{}, c);
        chunk(IC6<2>{}, nch - 1);
        // the patch buffer of the last chunk (pnext after the swap) is scratch until the next tile's second patch is requested
        if constexpr (F32O) epilogue_f32(pnext, al_f32);
        else if constexpr (RES) epilogue(pnext, 0L);
        else epilogue_packed(pnext, 0L);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // statistics deposits written; every slice window read
        __builtin_amdgcn_s_barrier();                        // opens k-tile 0 of the next tile
        asm volatile("" ::: "memory");
        if (stats) { const int img = c_mt / tpi; combine(img, c_mt - img * tpi, c_n0); }
        if (dma_done) break;
    }
}

static std::atomic<long> g_patch_launches{0};

int device_cus();   // api.hip

template <typename T> static int launch6(IgemmParams& p, int total, int grid, hipStream_t s) {
    if constexpr (std::is_same<T, f16>::value) {
        if (p.split_c > 0) {   // an fp32 convolution from f16 split planes (patch_eligible): 3x3, or a 2x2 parity phase of an upsampler convolution
            if (p.kh == 2) hipLaunchKernelGGL((igemm6_kernel<f16, true, false, 2, true>), dim3(grid), dim3(512), 0, s, p, total);
            else hipLaunchKernelGGL((igemm6_kernel<f16, true, false, 3, true>), dim3(grid), dim3(512), 0, s, p, total);
            tag_kernel("igemm6_kernel<_Float16, true, false, %d, f32split>", p.kh);
            return check_launch("igemm6");
        }
    }
    if (p.kh == 2) {   // a 2x2 parity phase of an upsampler convolution: no residual, no fused GroupNorm (patch_eligible)
        hipLaunchKernelGGL((igemm6_kernel<T, false, false, 2>), dim3(grid), dim3(512), 0, s, p, total);
        tag_kernel("igemm6_kernel<%s, false, false, 2>", std::is_same<T, f16>::value ? "_Float16" : "__bf16");
        return check_launch("igemm6");
    }
    if (p.nrm_ad) {
        if (p.residual) hipLaunchKernelGGL((igemm6_kernel<T, true, true, 3>), dim3(grid), dim3(512), 0, s, p, total);
        else hipLaunchKernelGGL((igemm6_kernel<T, false, true, 3>), dim3(grid), dim3(512), 0, s, p, total);
    } else {
        if (p.residual) hipLaunchKernelGGL((igemm6_kernel<T, true, false, 3>), dim3(grid), dim3(512), 0, s, p, total);
        else hipLaunchKernelGGL((igemm6_kernel<T, false, false, 3>), dim3(grid), dim3(512), 0, s, p, total);
    }
    tag_kernel("igemm6_kernel<%s, %s, %s>", std::is_same<T, f16>::value ? "_Float16" : "__bf16", p.residual ? "true" : "false", p.nrm_ad ? "true" : "false");
    return check_launch("igemm6");
}

static bool patch_eligible(int dtype, int mode, IgemmParams& p, int nz, int& grid_out, long& total_out) {
    using namespace patchk;
    if (!option(E2EFT_OPT_PATCH_CONV) || !option(E2EFT_OPT_PERSISTENT)) return false;
    if (mode != 1 || nz != 1 || (dtype != E2EFT_F16 && dtype != E2EFT_BF16)) return false;
    if (p.ksplit_taps > 0 || p.bias_along_m) return false;
    if (p.stride != 1 || p.zins > 1) return false;
    const bool taps3 = p.kh == 3 && p.kw == 3 && p.pad_t == 1 && p.pad_l == 1;
    if (p.split_c > 0 && (dtype != E2EFT_F16 || p.x2 || p.nrm_ad || p.rowadd || p.cin != 3 * p.split_c || p.split_c % 64 != 0 || p.hl != p.hin)) return false;
    // round 6: a 2x2 convolution with top / left pads of 0 or 1 — one parity phase of a 2x-upsampler convolution (e2eft_upconv2x_fwd); its rows may be written as
    // segments of a full-resolution image (out_seg = the row length)
    const bool taps2 = p.kh == 2 && p.kw == 2 && (unsigned)p.pad_t <= 1u && (unsigned)p.pad_l <= 1u && !p.residual && !p.nrm_ad && !p.x2 && (p.out_seg == 0 || p.out_seg == p.wl);
    if (!taps3 && !(taps2 && option(E2EFT_OPT_PATCH_CONV_2X2))) return false;
    if (taps3 && p.out_seg != 0) return false;
    const bool same = p.hl == p.hin && p.wl == p.win, up2 = taps3 && p.hl == 2 * p.hin && p.wl == 2 * p.win;   // plain, or the exact 2x nearest upsample fused into the read
    if (!(same || up2) || p.hout != p.hl || p.wout != p.wl) return false;
    if (p.wl % TW != 0 || p.hl % TH != 0) return false;
    if (p.cin % 64 != 0 || p.c1 % 64 != 0 || p.cin < 128 || p.K != p.kh * p.kw * p.cin) return false;
    if (p.nrm_ad && (p.x2 || p.cin > NORM_CMAX || p.N > BN || !option(E2EFT_OPT_FUSED_NORM))) return false;   // one N tile: every N tile would redo the normalisation
    if (p.N % 8 != 0 || p.ldo % 8 != 0 || (((uintptr_t)p.out) & 15) != 0) return false;
    if (p.ldx1 % 8 != 0 || (((uintptr_t)p.x1) & 15) != 0 || (p.x2 && (p.ldx2 % 8 != 0 || (((uintptr_t)p.x2) & 15) != 0))) return false;
    if (p.ldw % 8 != 0 || (((uintptr_t)p.w) & 15) != 0) return false;
    if (p.residual && (p.ldr % 8 != 0 || (((uintptr_t)p.residual) & 15) != 0)) return false;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return false;
    if (p.rowadd && ((((uintptr_t)p.rowadd) & 15) != 0 || p.rows_per_img != p.hl * p.wl)) return false;
    const long img_bytes = (long)p.hin * p.win * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * 2;
    if (img_bytes >= 0xD0000000L || (long)128 * p.ldw * 2 >= 0x40000000L) return false;
    if (p.M % (p.hl * p.wl) != 0) return false;
    int cus = device_cus();
    if (cus == 0) return false;
    const int gopt = option(E2EFT_OPT_PERSISTENT_GRID);
    if (gopt >= 8 && gopt < cus) cus = gopt;
    const int mtiles = p.M / BM, ntiles = cdiv(p.N, BN);
    const long total = (long)mtiles * ntiles;
    if (4 * total < (long)option(E2EFT_OPT_PERSISTENT_MIN_QROUNDS) * cus || total > 2000000000L || mtiles >= (1 << 22)) return false;
    if (p.gn_partial) {
        if (p.rows_per_img != p.hl * p.wl) return false;
        p.gn_nslabs = p.rows_per_img / BM;
    }
    p.mtiles = mtiles;
    p.ntiles = ntiles;
    grid_out = cus;
    total_out = total;
    return true;
}

bool igemm_patch_eligible(int dtype, int mode, IgemmParams& p, int nz) {
    int grid;
    long total;
    return patch_eligible(dtype, mode, p, nz, grid, total);
}

int launch_igemm_patch(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s) {
    int grid;
    long total;
    if (!patch_eligible(dtype, mode, p, nz, grid, total)) return -1;
    g_patch_launches.fetch_add(1, std::memory_order_relaxed);
    return dtype == E2EFT_F16 ? launch6<f16>(p, (int)total, grid, s) : launch6<bf16>(p, (int)total, grid, s);
}

}  // namespace e2eft

extern "C" long e2eft_debug_patch_launches(void) { return e2eft::g_patch_launches.load(); }   // not part of include/e2eft.h
