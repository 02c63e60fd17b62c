// Error channel and identity of libjoeys2t_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/joeys2t_hip.h"

static thread_local char g_err[512] = "";

void js2t_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* js2t_last_error(void) { return g_err; }
extern "C" int js2t_abi_version(void) { return 1; }

// Deterministic mode (the reference asks cuDNN for one: helpers.py:93-104, set_seed): every kernel on the Transformer S2T train
// path that sums through floating-point atomics takes an ordered form instead - see js2t_set_deterministic in the header.
int g_js2t_deterministic = 0;
extern "C" void js2t_set_deterministic(int on) { g_js2t_deterministic = on != 0; }
extern "C" int js2t_get_deterministic(void) { return g_js2t_deterministic; }
