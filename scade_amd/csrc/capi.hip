// C-ABI plumbing shared by every scade_hip entry point: thread-local last
// error string, launch checking, version.  No exceptions cross the boundary.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void scade_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

int scade_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    scade_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" const char* scade_last_error(void) { return g_last_error; }
extern "C" int scade_version(void) { return 1; }
