// Library-wide helpers: version and last-error bookkeeping (no device state).
#include "asr_common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

void asr_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

extern "C" int asr_version(void) { return 100; }
extern "C" const char* asr_last_error(void) { return g_err; }
