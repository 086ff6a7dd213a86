// api.hip — error plumbing and version for libe2eft.
#include "common.h"

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

}  // namespace e2eft

extern "C" int e2eft_version(void) { return E2EFT_VERSION; }
extern "C" const char* e2eft_last_error(void) { return e2eft::err_buf(); }
