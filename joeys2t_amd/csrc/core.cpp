// Error channel and identity of libjoeys2t_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/joeys2t_hip.h"

void js2t_set_error(const char* fmt, ...);
int js2t_ctx_value(int key);
void js2t_ctx_override(int key, int value);

static thread_local char g_err[512] = "";

void js2t_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* js2t_last_error(void) { return g_err; }
extern "C" int js2t_abi_version(void) { return 1; }

// Settings that belong to a CALLER, not to the process (a thin, stateless boundary): deterministic mode (the reference
// asks cuDNN for one: helpers.py:93-104, set_seed - every kernel on the train path that sums through floating-point atomics then takes
// an ordered form) and the kernel-selection rules of js2t_gemm.  Round 5 kept them in process globals: two train steps in one process,
// one of them deterministic, raced on the switch.  Now they live in a js2t_ctx the caller binds to its thread around its launches;
// nothing bound = the built-in defaults.  The old setters stay as process-wide TEST overrides (they win over any context).
struct js2t_ctx_s {
  int32_t v[JS2T_CTX_NKEYS];
};
static int32_t g_override[JS2T_CTX_NKEYS] = {0, -1, -1, -1, -1};  // deterministic: 0 = no override; modes: -1 = no override
static thread_local js2t_ctx_s* t_bound = nullptr;

int js2t_ctx_value(int key) {
  if (key < 0 || key >= JS2T_CTX_NKEYS) return -1;
  const int32_t bound = t_bound ? t_bound->v[key] : -1;
  if (key == JS2T_CTX_DETERMINISTIC) return g_override[key] > 0 || bound > 0;
  return g_override[key] >= 0 ? g_override[key] : bound;
}
void js2t_ctx_override(int key, int value) {
  if (key >= 0 && key < JS2T_CTX_NKEYS) g_override[key] = value;
}
extern "C" js2t_ctx js2t_ctx_create(void) {
  js2t_ctx_s* c = new js2t_ctx_s;
  for (int i = 0; i < JS2T_CTX_NKEYS; ++i) c->v[i] = -1;  // -1: the built-in default / automatic rule
  return c;
}
extern "C" void js2t_ctx_destroy(js2t_ctx ctx) {
  if (ctx && t_bound == ctx) t_bound = nullptr;
  delete ctx;
}
extern "C" int js2t_ctx_set(js2t_ctx ctx, int32_t key, int32_t value) {
  if (!ctx || key < 0 || key >= JS2T_CTX_NKEYS) {
    js2t_set_error("ctx_set: null context or unknown key %d", (int)key);
    return JS2T_ERR_INVALID;
  }
  ctx->v[key] = value < 0 ? -1 : value;
  return JS2T_OK;
}
extern "C" int js2t_ctx_get(js2t_ctx ctx, int32_t key) { return (ctx && key >= 0 && key < JS2T_CTX_NKEYS) ? ctx->v[key] : -1; }
extern "C" js2t_ctx js2t_ctx_bind(js2t_ctx ctx) {
  js2t_ctx prev = t_bound;
  t_bound = ctx;
  return prev;
}
extern "C" int js2t_ctx_effective(int32_t key) { return js2t_ctx_value(key); }

extern "C" void js2t_set_deterministic(int on) { js2t_ctx_override(JS2T_CTX_DETERMINISTIC, on != 0); }
extern "C" int js2t_get_deterministic(void) { return js2t_ctx_value(JS2T_CTX_DETERMINISTIC); }
