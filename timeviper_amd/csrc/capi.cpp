// Error plumbing and ABI version of libtimeviper_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/timeviper_hip.h"

static thread_local char g_err[512] = "";

void tv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tv_abi_version(void) { return 12; }
extern "C" const char* tv_last_error(void) { return g_err; }
#ifndef TV_BUILD_ID
#define TV_BUILD_ID "unknown"
#endif
// hash of the sources this library was built from (timeviper_amd/build.py source_id())
extern "C" const char* tv_build_id(void) { return TV_BUILD_ID; }
