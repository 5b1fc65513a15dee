// Library-wide helpers: version and last-error bookkeeping (no device state).
#include "asr_common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

void asr_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

static thread_local const char* g_last_kernel = "";

void asr_set_last_kernel(const char* name) { g_last_kernel = name; }

extern "C" int asr_version(void) { return 102; }          // 102: asr_tap_gemm_nt_splitk, the larger asr_ctc_workspace (round 5)
extern "C" const char* asr_last_kernel(void) { return g_last_kernel; }
extern "C" const char* asr_last_error(void) { return g_err; }
