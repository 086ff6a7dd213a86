/*
 * e2eft_debug.h — instrumentation exports of libe2eft.so.  NOT part of the drop-in contract of e2eft.h: nothing here is needed to run the path, nothing
 * here changes a result.  The tests use the launch counters to prove WHICH kernel served a call (tests/test_patch_conv_gpu.py, test_persistent_gpu.py,
 * test_thin_conv_gpu.py, test_fused_norm_conv_gpu.py); bench.py --detail uses the kernel tag to print the symbol next to every shape.
 *
 * State: three process-global, monotonically increasing atomic launch counters and one thread-local string.  They are the only mutable state of the
 * library besides what e2eft.h documents (thread-local error string, e2eft_set_option); reading them has no side effect.
 */
#ifndef E2EFT_DEBUG_H
#define E2EFT_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* name + template arguments of the kernel the calling thread's last launcher enqueued ("igemm6_kernel<_Float16, false, true>", ...), "" if none */
const char* e2eft_debug_last_kernel(void);
/* launches of igemm6_kernel (halo-patch convolution) / igemm5_kernel (persistent implicit GEMM) / conv_thin_in_kernel since the library was loaded */
long e2eft_debug_patch_launches(void);
long e2eft_debug_persistent_launches(void);
long e2eft_debug_thin_launches(void);

/* Instrumented build only (-DE2EFT_STAMPS, lib/libe2eft_stamps.so, scripts/stamp_bench.py; absent from libe2eft.so): per-workgroup phase clocks */
int e2eft_debug_read_stamps(long long* host, int nworkgroups);
int e2eft_debug_read_stamps_rt(long long* host, int nworkgroups);
int e2eft_debug_read_stamps5(long long* host, int nworkgroups);
void e2eft_debug_set_flags5(int flags);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* E2EFT_DEBUG_H */
