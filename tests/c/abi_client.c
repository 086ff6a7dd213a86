/* A plain-C client of include/e2eft.h: what a non-Python host (the boundary the task statement asks for) sees.  Compiled with gcc by
 * tests/test_abi.py::test_plain_c_client, linked against nothing but libdl; loads libe2eft.so at run time.  No GPU needed: it only
 * exercises entry points that validate arguments or size workspaces on the host. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "e2eft.h"

#define SYM(T, name) T name##_p = (T)dlsym(h, #name); if (!name##_p) { fprintf(stderr, "missing %s\n", #name); return 2; }

typedef int (*version_fn)(void);
typedef const char* (*err_fn)(void);
typedef size_t (*gn_ws_fn)(const E2eftGroupNormDesc*);
typedef int (*gemm_fn)(const E2eftGemmDesc*, const void*, const void*, const void*, const void*, void*, void*);
typedef size_t (*ens_ws_fn)(int32_t);
typedef int (*ens_minmax_fn)(int32_t, int64_t, const float*, float*, void*, size_t, void*);

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    SYM(version_fn, e2eft_version)
    SYM(err_fn, e2eft_last_error)
    SYM(gn_ws_fn, e2eft_groupnorm_workspace_bytes)
    SYM(gemm_fn, e2eft_gemm)
    SYM(ens_ws_fn, e2eft_ensemble_workspace_bytes)
    SYM(ens_minmax_fn, e2eft_ensemble_minmax)
    if (e2eft_version_p() != E2EFT_VERSION) { fprintf(stderr, "header %d != library %d\n", E2EFT_VERSION, e2eft_version_p()); return 3; }

    E2eftGroupNormDesc g;
    memset(&g, 0, sizeof g);
    g.dtype = E2EFT_F16; g.batch = 2; g.hw = 96 * 96; g.c1 = 320; g.ldx1 = 320; g.groups = 32; g.ldy = 320; g.eps = 1e-5f;
    const size_t ws = e2eft_groupnorm_workspace_bytes_p(&g);
    if (ws == 0) { fprintf(stderr, "groupnorm ws: %s\n", e2eft_last_error_p()); return 4; }

    E2eftGemmDesc d;
    memset(&d, 0, sizeof d);
    d.dtype = E2EFT_F16; d.m = 4; d.n = 4; d.k = 6;           /* K must be a multiple of 8 for 16-bit operands */
    char buf[64];
    const int rc = e2eft_gemm_p(&d, buf, buf, NULL, NULL, buf, NULL);
    if (rc != E2EFT_ERR_BAD_ARG || !strstr(e2eft_last_error_p(), "multiple of 8")) { fprintf(stderr, "gemm rc %d: %s\n", rc, e2eft_last_error_p()); return 5; }

    if (e2eft_ensemble_workspace_bytes_p(10) == 0 || e2eft_ensemble_workspace_bytes_p(33) != 0) return 6;   /* at most 32 members */
    if (e2eft_ensemble_minmax_p(33, 16, (const float*)buf, (float*)buf, buf, 64, NULL) != E2EFT_ERR_BAD_ARG) return 7;
    printf("e2eft %d ok: groupnorm workspace %zu bytes; error text: %s\n", e2eft_version_p(), ws, e2eft_last_error_p());
    return 0;
}
