// libmavahip.so: error channel and version of the C ABI declared in include/mava_hip.h.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "ctx.h"

static thread_local char g_err[512] = "";

void mava_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mava_last_error(void) { return g_err; }

extern "C" int mava_abi_version(void) { return 3; }  // 3: round-3 signatures (mava_ctx handle instead of process-wide setters)

// ---- context handle (ctx.h) -------------------------------------------------------------------------------------------
enum { CTX_MATMUL_MODE = 0, CTX_CRITIC_AGGREGATION = 1, CTX_GAE_VARIANT = 2, CTX_POLICY_VARIANT = 3, CTX_H2_LAUNCHES = 4, CTX_TRAIN_VARIANT = 5, CTX_W8_LAUNCHES = 6, CTX_W1_SPLIT_FRESH = 7 };

extern "C" int mava_ctx_create(mava_ctx** out) {
  MAVA_ARG_CHECK(out != nullptr, 0, "mava_ctx_create: null output pointer");
  mava_ctx* c = new mava_ctx();
  c->matmul_mode = 0;
  c->critic_aggregation = 1;
  c->gae_variant = 0;
  c->policy_variant = 0;
  c->h2_launches = 0;
  c->train_variant = 0;
  c->w8_launches = 0;
  c->w1_split[0] = c->w1_split[1] = nullptr;
  c->w1_fresh[0] = c->w1_fresh[1] = 0;
  *out = c;
  return MAVA_OK;
}

extern "C" int mava_ctx_destroy(mava_ctx* c) {
  if (c == nullptr) return MAVA_OK;
  for (int i = 0; i < 2; ++i)
    if (c->w1_split[i] != nullptr) (void)hipFree(c->w1_split[i]);  // (synchronises with the launches that read it)
  delete c;
  return MAVA_OK;
}

extern "C" int mava_ctx_set(mava_ctx* c, int key, long value) {
  MAVA_ARG_CHECK(c != nullptr, 0, "mava_ctx_set: null context");
  switch (key) {
    case CTX_MATMUL_MODE:
      MAVA_ARG_CHECK(value == 0 || value == 1, 1, "mava_ctx_set: matmul mode %ld (0 = exact f32, 1 = f16x2)", value);
      c->matmul_mode = (int)value;
      return MAVA_OK;
    case CTX_CRITIC_AGGREGATION: c->critic_aggregation = value ? 1 : 0; return MAVA_OK;
    case CTX_GAE_VARIANT: c->gae_variant = (int)value; return MAVA_OK;
    case CTX_POLICY_VARIANT: c->policy_variant = (int)value; return MAVA_OK;
    case CTX_H2_LAUNCHES: c->h2_launches = value; return MAVA_OK;
    case CTX_TRAIN_VARIANT: c->train_variant = (int)value; return MAVA_OK;
    case CTX_W8_LAUNCHES: c->w8_launches = value; return MAVA_OK;
    case CTX_W1_SPLIT_FRESH:
      MAVA_ARG_CHECK(value == 0, 1, "mava_ctx_set: MAVA_CTX_W1_SPLIT_FRESH can only be cleared (0) by the caller");
      c->w1_fresh[0] = c->w1_fresh[1] = 0;
      return MAVA_OK;
    default: mava_set_error("mava_ctx_set: unknown key %d", key); return MAVA_EARG(2);
  }
}

extern "C" int mava_ctx_get(const mava_ctx* c, int key, long* value) {
  MAVA_ARG_CHECK(c != nullptr && value != nullptr, 0, "mava_ctx_get: null argument");
  switch (key) {
    case CTX_MATMUL_MODE: *value = c->matmul_mode; return MAVA_OK;
    case CTX_CRITIC_AGGREGATION: *value = c->critic_aggregation; return MAVA_OK;
    case CTX_GAE_VARIANT: *value = c->gae_variant; return MAVA_OK;
    case CTX_POLICY_VARIANT: *value = c->policy_variant; return MAVA_OK;
    case CTX_H2_LAUNCHES: *value = c->h2_launches; return MAVA_OK;
    case CTX_TRAIN_VARIANT: *value = c->train_variant; return MAVA_OK;
    case CTX_W8_LAUNCHES: *value = c->w8_launches; return MAVA_OK;
    case CTX_W1_SPLIT_FRESH: *value = c->w1_fresh[1]; return MAVA_OK;
    default: mava_set_error("mava_ctx_get: unknown key %d", key); return MAVA_EARG(2);
  }
}
