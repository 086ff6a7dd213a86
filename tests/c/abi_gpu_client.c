/* A plain-C client of include/e2eft.h that LAUNCHES kernels: what a non-Python, non-torch host does with the drop-in boundary.  Compiled with gcc by
 * tests/test_abi_gpu_client.py, linked against nothing but libdl and libm; loads libamdhip64.so (device memory, copies, synchronisation) and libe2eft.so at
 * run time.  Known-answer tests (SURVEY.md §8c "analytic KATs"), fp32 so that no half conversion is needed on the host:
 *   1. e2eft_groupnorm_fwd(+SiLU) on an input that is constant inside every group: variance 0 -> the normalised value is 0 -> y = SiLU(beta[c]);
 *   2. e2eft_conv2d_fwd, 3x3 / pad 1, weights = identity on the centre tap: y = x + bias (borders included: the zero padding contributes nothing);
 *   3. the two chained the way ResnetBlock2D does (norm -> conv) on the device buffers, caller-owned workspace, caller's stream = 0.
 * usage: abi_gpu_client <libe2eft.so> [libamdhip64.so] */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "e2eft.h"

typedef int (*hipMalloc_fn)(void**, size_t);
typedef int (*hipFree_fn)(void*);
typedef int (*hipMemcpy_fn)(void*, const void*, size_t, int);
typedef int (*hipSync_fn)(void);
typedef int (*version_fn)(void);
typedef const char* (*err_fn)(void);
typedef size_t (*gn_ws_fn)(const E2eftGroupNormDesc*);
typedef int (*gn_fn)(const E2eftGroupNormDesc*, const void*, const void*, const void*, const void*, void*, void*, size_t, void*);
typedef int (*conv_fn)(const E2eftConvDesc*, const void*, const void*, const void*, const void*, const void*, const void*, void*, void*);

#define SYM(T, h, name) T name##_p = (T)dlsym(h, #name); if (!name##_p) { fprintf(stderr, "missing %s\n", #name); return 2; }
#define HIP(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> hip error %d\n", #call, rc_); return 10; } } while (0)
enum { H2D = 1, D2H = 2 };

static float silu(float v) { return v / (1.0f + expf(-v)); }

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* hip = dlopen(argc > 2 ? argv[2] : "libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hip) { fprintf(stderr, "dlopen hip: %s\n", dlerror()); return 1; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    SYM(hipMalloc_fn, hip, hipMalloc)
    SYM(hipFree_fn, hip, hipFree)
    SYM(hipMemcpy_fn, hip, hipMemcpy)
    hipSync_fn hipSync_p = (hipSync_fn)dlsym(hip, "hipDeviceSynchronize"); if (!hipSync_p) { fprintf(stderr, "missing hipDeviceSynchronize\n"); return 2; }
    SYM(version_fn, h, e2eft_version)
    SYM(err_fn, h, e2eft_last_error)
    SYM(gn_ws_fn, h, e2eft_groupnorm_workspace_bytes)
    SYM(gn_fn, h, e2eft_groupnorm_fwd)
    SYM(conv_fn, h, e2eft_conv2d_fwd)
    if (e2eft_version_p() != E2EFT_VERSION) { fprintf(stderr, "header %d != library %d\n", E2EFT_VERSION, e2eft_version_p()); return 3; }

    enum { B = 2, H = 24, W = 40, C = 64, G = 32 };
    const size_t n = (size_t)B * H * W * C;
    float* x = (float*)malloc(n * 4), *y = (float*)malloc(n * 4), *wt = (float*)calloc((size_t)C * 9 * C, 4);
    float gamma[C], beta[C], bias[C];
    for (int c = 0; c < C; ++c) { gamma[c] = 1.0f + 0.01f * c; beta[c] = 0.5f * (float)(c % 7 - 3); bias[c] = 0.25f * (float)(c % 5 - 2); }
    /* constant inside every (image, group): value depends on the image and the group only */
    for (size_t i = 0; i < n; ++i) { const int c = (int)(i % C), b = (int)(i / ((size_t)H * W * C)); x[i] = 1.5f * (float)(c / (C / G)) - 3.0f * b; }
    for (int o = 0; o < C; ++o) wt[(size_t)o * 9 * C + 4 * C + o] = 1.0f;            /* OHWI rows: W[n][(ky*3 + kx)*C + c], centre tap (1,1) = identity */

    void *dx, *dy, *dz, *dg, *db, *dbias, *dw, *dws;
    HIP(hipMalloc_p(&dx, n * 4)); HIP(hipMalloc_p(&dy, n * 4)); HIP(hipMalloc_p(&dz, n * 4));
    HIP(hipMalloc_p(&dg, C * 4)); HIP(hipMalloc_p(&db, C * 4)); HIP(hipMalloc_p(&dbias, C * 4)); HIP(hipMalloc_p(&dw, (size_t)C * 9 * C * 4));
    HIP(hipMemcpy_p(dx, x, n * 4, H2D)); HIP(hipMemcpy_p(dg, gamma, C * 4, H2D)); HIP(hipMemcpy_p(db, beta, C * 4, H2D));
    HIP(hipMemcpy_p(dbias, bias, C * 4, H2D)); HIP(hipMemcpy_p(dw, wt, (size_t)C * 9 * C * 4, H2D));

    E2eftGroupNormDesc g;
    memset(&g, 0, sizeof g);
    g.dtype = E2EFT_F32; g.batch = B; g.hw = H * W; g.c1 = C; g.ldx1 = C; g.groups = G; g.ldy = C; g.silu = 1; g.eps = 1e-5f;
    const size_t ws = e2eft_groupnorm_workspace_bytes_p(&g);
    if (ws == 0) { fprintf(stderr, "groupnorm ws: %s\n", e2eft_last_error_p()); return 4; }
    HIP(hipMalloc_p(&dws, ws));
    int rc = e2eft_groupnorm_fwd_p(&g, dx, NULL, dg, db, dy, dws, ws, NULL);
    if (rc != E2EFT_OK) { fprintf(stderr, "groupnorm_fwd rc %d: %s\n", rc, e2eft_last_error_p()); return 5; }

    E2eftConvDesc d;
    memset(&d, 0, sizeof d);
    d.dtype = E2EFT_F32; d.batch = B; d.hin = d.hl = d.hout = H; d.win = d.wl = d.wout = W; d.c1 = C; d.ldx1 = C; d.kh = d.kw = 3; d.stride = 1;
    d.pad_t = d.pad_l = 1; d.cout = C; d.ldo = C; d.ldw = 9 * C; d.alpha = 1.0f;
    rc = e2eft_conv2d_fwd_p(&d, dy, NULL, dw, dbias, NULL, NULL, dz, NULL);          /* norm -> conv, as ResnetBlock2D chains them */
    if (rc != E2EFT_OK) { fprintf(stderr, "conv2d_fwd rc %d: %s\n", rc, e2eft_last_error_p()); return 6; }
    HIP(hipSync_p());
    HIP(hipMemcpy_p(y, dy, n * 4, D2H));
    double e1 = 0.0, e2 = 0.0;
    for (size_t i = 0; i < n; ++i) { const double e = fabs((double)y[i] - (double)silu(beta[i % C])); if (e > e1) e1 = e; }
    HIP(hipMemcpy_p(y, dz, n * 4, D2H));
    for (size_t i = 0; i < n; ++i) { const double e = fabs((double)y[i] - ((double)silu(beta[i % C]) + bias[i % C])); if (e > e2) e2 = e; }
    /* and the convolution alone on the raw input (non-constant across groups / images): y = x + bias */
    rc = e2eft_conv2d_fwd_p(&d, dx, NULL, dw, dbias, NULL, NULL, dz, NULL);
    if (rc != E2EFT_OK) { fprintf(stderr, "conv2d_fwd rc %d: %s\n", rc, e2eft_last_error_p()); return 6; }
    HIP(hipSync_p());
    HIP(hipMemcpy_p(y, dz, n * 4, D2H));
    double e3 = 0.0;
    for (size_t i = 0; i < n; ++i) { const double e = fabs((double)y[i] - ((double)x[i] + bias[i % C])); if (e > e3) e3 = e; }
    printf("e2eft %d on the GPU from plain C: groupnorm+silu KAT max err %.3g, norm->conv KAT %.3g, identity conv KAT %.3g\n", e2eft_version_p(), e1, e2, e3);
    hipFree_p(dx); hipFree_p(dy); hipFree_p(dz); hipFree_p(dg); hipFree_p(db); hipFree_p(dbias); hipFree_p(dw); hipFree_p(dws);
    return (e1 <= 2e-5 && e2 <= 2e-5 && e3 <= 1e-5) ? 0 : 7;
}
