// api.hip — error plumbing and version for libe2eft.
#include "common.h"
#include <atomic>

namespace e2eft {

static thread_local char g_err[512] = "";

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(E2EFT_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return E2EFT_OK;
}

// CU count of the CURRENT device, cached per device ordinal (one process may drive several GPUs); 0 on failure.  Rounded down to a multiple of 8 (XCDs).
static std::atomic<int> g_cus_of_device[64];
int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = g_cus_of_device[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 0;
        n &= ~7;
        g_cus_of_device[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static thread_local char g_tag[96] = "";
void tag_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_tag, sizeof(g_tag), fmt, ap);
    va_end(ap);
}
const char* last_tag() { return g_tag; }

static std::atomic<int> g_opt[E2EFT_OPT_COUNT] = {{1}, {0}, {1}, {1}, {0}, {0}, {1}, {1}, {1}, {1}, {1}, {1}, {2}, {0}, {1}};

int option(int key) { return g_opt[key].load(std::memory_order_relaxed); }

}  // namespace e2eft

extern "C" int e2eft_set_option(int32_t key, int32_t value) {
    using namespace e2eft;
    E2EFT_REQUIRE(key >= 0 && key < E2EFT_OPT_COUNT, "set_option: unknown key %d", key);
    bool ok = value == 0 || value == 1;
    if (key == E2EFT_OPT_PERSISTENT_GRID) ok = value == 0 || (value >= 8 && value % 8 == 0 && value <= 4096);
    if (key == E2EFT_OPT_IGEMM2_WAVES) ok = value == 0 || value == 4 || value == 8;
    if (key == E2EFT_OPT_PERSISTENT_MIN_QROUNDS) ok = value >= 1 && value <= 64;
    if (key == E2EFT_OPT_GN_APPLY_ITERS) ok = value >= 0 && value <= 16;
    E2EFT_REQUIRE(ok, "set_option: value %d out of range for key %d", value, key);
    g_opt[key].store(value, std::memory_order_relaxed);
    return E2EFT_OK;
}
extern "C" int e2eft_get_option(int32_t key) { return key >= 0 && key < E2EFT_OPT_COUNT ? e2eft::option(key) : -1; }

// debugging / measurement aid, not part of include/e2eft.h: the symbol (as rocprofv3 prints it, template arguments included) of the kernel the
// calling thread's last implicit-GEMM launch went to
extern "C" const char* e2eft_debug_last_kernel(void) { return e2eft::last_tag(); }

extern "C" int e2eft_version(void) { return E2EFT_VERSION; }
// identity of the SOURCES this binary was built from: sha256 over csrc/*, include/*.h and the compiler flags, computed by build.py and passed as
// -DE2EFT_BUILD_ID (hex string).  Measurements that are collected in one process and quoted by another (the PMC traffic figure in bench.py's
// roofline) carry it, so "same build" is an equality of ids, not of launch counts.
#ifndef E2EFT_BUILD_ID
#define E2EFT_BUILD_ID "unstamped"
#endif
static const char k_build_id[] = "E2EFT_BUILD_ID=" E2EFT_BUILD_ID;      // (the marker lets build.py read the id from the file without loading it)
extern "C" const char* e2eft_build_id(void) { return k_build_id + 15; }
extern "C" const char* e2eft_last_error(void) { return e2eft::err_buf(); }
