// Per-learner context of libmavahip.so (include/mava_hip.h "context"): every setting that used to be a process-wide
// knob (arithmetic of the matrix products, critic aggregation, kernel variants) and every workspace the library
// allocates on its own (the pre-split W1 copies of the wide f16x2 gradient kernels) lives in a handle the caller
// creates, passes to the calls that depend on it and destroys.  A NULL handle means the defaults (exact f32,
// aggregation on, default variants); nothing in the library is shared between two handles.
#pragma once
#include "common.h"

struct mava_ctx {
  int matmul_mode;         // 0: exact-f32 MFMA kernels; 1: split-f16 ("f16x2") kernels where instantiated
  int critic_aggregation;  // 1: a critic input row shared by the A agents of an index is evaluated once
  int gae_variant;         // 0: default chunk / lane mapping of mava_gae_f32 (others: tools/gae_sweep.py)
  int policy_variant;      // 0: default acting-step launch (others: tools/policy_bench.py)
  int train_variant;       // f16x2 gradient kernels: bit 0 = four-wave kernels only (default: eight-wave actor kernel where instantiated); bit 1 = never skip the x_lo products
  long h2_launches;        // diagnostic: gradient launches of this handle that ran on the f16x2 kernels
  long w8_launches;        // diagnostic: ... of which on the eight-wave kernel (ppo_train_w8.hip)
  void* w1_split[2];       // f16x2, inputs wider than 95: pre-split W1 in fragment order (actor, critic), lazily allocated
  int w1_fresh[2];         // one-shot: the copy already matches the parameters of the NEXT gradient launch (written by the Adam
                           // launch of mava_ppo_finish_f32); cleared by that launch, and by the caller whenever parameters may
                           // have changed in between (MAVA_CTX_W1_SPLIT_FRESH)
};

static inline int mava_ctx_matmul_mode(const mava_ctx* c) { return c ? c->matmul_mode : 0; }
static inline int mava_ctx_critic_aggregation(const mava_ctx* c) { return c ? c->critic_aggregation : 1; }
static inline int mava_ctx_gae_variant(const mava_ctx* c) { return c ? c->gae_variant : 0; }
static inline int mava_ctx_policy_variant(const mava_ctx* c) { return c ? c->policy_variant : 0; }
