// attn_probe.hip — stand-alone harness around csrc/attn.hip for ceiling probes of the d = 64 attention kernel (no torch, no library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DE2EFT_ATTN_PROBE=<n> scripts/experiments/attn_probe.hip -o /tmp/attn_probe_<n>
//   /tmp/attn_probe_<n> [B=8] [heads=5] [N=9216] [iters=20]
// PROBE 0: the production kernel.  1: K / V pinned in LDS (tile 0 re-used: no global loads, no LDS stores, no barrier in the loop) = the MFMA + softmax-VALU +
// LDS-fragment-read stream.  2: as 1 without the softmax VALU (probabilities = the raw accumulator bits).  3: as 1 without the LDS fragment reads in the loop
// (fragments of tile 0 kept in registers where they fit: K only).  4: production traffic (loads, stores, barrier) without the softmax VALU.
// Results of probes != 0 are WRONG by construction; they price the phases.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../diffusion-e2e-ft_amd/csrc/attn.hip"

namespace e2eft {
static char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : fail(3, "%s: %s", what, hipGetErrorString(e)); }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 5, N = argc > 3 ? atoi(argv[3]) : 9216, iters = argc > 4 ? atoi(argv[4]) : 20;
    const size_t n = (size_t)B * N * 3 * H * 64;
    std::vector<_Float16> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 8) % 2001 - 1000) / 500.0f); }
    _Float16 *qkv, *out;
    hipMalloc(&qkv, n * 2); hipMalloc(&out, (size_t)B * N * H * 64 * 2);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    E2eftAttnDesc d;
    d.dtype = E2EFT_F16; d.batch = B; d.heads = H; d.nq = N; d.nk_seg = N; d.kv_nseg = 1; d.kv_bmod = B;
    d.ldq = d.ldk = d.ldv = 3 * H * 64; d.ldo = H * 64; d.scale = 0.125f;
    for (int i = 0; i < 3; ++i) if (e2eft_attn_fwd(&d, qkv, qkv + H * 64, qkv + 2 * H * 64, out, nullptr)) { printf("error: %s\n", e2eft::err_buf()); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) e2eft_attn_fwd(&d, qkv, qkv + H * 64, qkv + 2 * H * 64, out, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    std::vector<_Float16> o((size_t)B * N * H * 64);
    hipMemcpy(o.data(), out, o.size() * 2, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < o.size(); i += 997) cs += (double)o[i];
    // spot check against fp64 on the host: a few (image, head, query) rows, first / middle / last query blocks included
    double worst = 0;
    const int ld = 3 * H * 64;
    const int qs[6] = {0, 31, 257, N / 2 + 5, N - 129, N - 1};
    for (int bi = 0; bi < B; bi += (B > 1 ? B - 1 : 1))
        for (int hi = 0; hi < H; hi += (H > 1 ? H - 1 : 1))
            for (int qi = 0; qi < 6; ++qi) {
                const int q = qs[qi] < 0 ? 0 : qs[qi];
                const _Float16* qp = &h[((size_t)bi * N + q) * ld + hi * 64];
                std::vector<double> sc(N);
                double mx = -1e300;
                for (int j = 0; j < N; ++j) {
                    const _Float16* kp = &h[((size_t)bi * N + j) * ld + H * 64 + hi * 64];
                    double a = 0;
                    for (int e = 0; e < 64; ++e) a += (double)qp[e] * (double)kp[e];
                    sc[j] = a * 0.125; if (sc[j] > mx) mx = sc[j];
                }
                double l = 0; std::vector<double> acc(64, 0.0);
                for (int j = 0; j < N; ++j) {
                    const double pj = exp(sc[j] - mx); l += pj;
                    const _Float16* vp = &h[((size_t)bi * N + j) * ld + 2 * H * 64 + hi * 64];
                    for (int e = 0; e < 64; ++e) acc[e] += pj * (double)vp[e];
                }
                for (int e = 0; e < 64; ++e) {
                    const double err = fabs(acc[e] / l - (double)o[((size_t)bi * N + q) * H * 64 + hi * 64 + e]);
                    if (err > worst) worst = err;
                }
            }
    printf("spot check vs fp64 (24 rows): max |err| %.3e\n", worst);
#ifndef E2EFT_ATTN_PROBE
#define E2EFT_ATTN_PROBE 0
#endif
    printf("probe %d: attn B%d h%d N%d fp16: %.3f ms  %.1f TFLOP/s  checksum %.4f\n", E2EFT_ATTN_PROBE, B, H, N, ms, 4.0 * B * H * (double)N * N * 64 / ms / 1e9, cs);
    return 0;
}
extern "C" const char* e2eft_last_error(void) { return e2eft::err_buf(); }
