// Arguments of one fused PPO minibatch launch, shared by the exact-f32 kernel (ppo_train.hip) and the
// split-f16 kernel (ppo_train_h2.hip).
#pragma once
#include <stdint.h>

#include "common.h"

struct TrainTask {
  const float* params;
  const float* x;          // (rows_x, din)
  int din, no, xshare, xv;
  int A;                   // agent rows per (t,e) index
  int agg;                 // critic only: 1, or A when the A agents of an index share one input row and are
                           // aggregated into it (one network pass per (t,e) row instead of A identical ones)
  const int32_t* idx;      // minibatch (t*E+e) indices, or null => idx_base + b
  long idx_base;
  int Rb;                  // (t,e) rows in the minibatch; agent rows R = Rb * A
  const uint8_t* mask;     // (TE*A, no) or null
  const int32_t* action;   // (TE*A)
  const float* action_f;   // continuous head: (TE*A, no) actions in (-1, 1); the raw scales follow the MLP in params
  uint32_t seed_lo, seed_hi, ent_step, row_offset;  // continuous head: Philox key / counters of the entropy sample
  float min_scale;         // continuous head: scale = softplus(log_std) + min_scale
  const float* old_logp;   // (TE*A)
  const float* adv;        // (TE*A)
  const double* stats;     // STATS_BLOCKS x {sum, sumsq} partials of the minibatch advantages
  const float* old_value;  // (TE*A)
  const float* targets;    // (TE*A)
  float clip_eps, ent_coef, vf_coef;
  float* slab;
  long slab_stride;
  unsigned long long* stamps;  // diagnostic builds only (-DMAVA_STAMPS): per-phase cycle sums of block 0
  int force_xlo;           // f16x2 kernels: always run the x_lo products (MAVA_CTX_TRAIN_VARIANT bit 1; tests: same bits either way)
};

// ppo_train_h2.hip: the same fused kernel on v_mfma_f32_32x32x16_f16 with every operand split into two f16 terms
// (hi + lo, f32 accumulation).  Returns MAVA_OK, or 1 when the shape is not instantiated (the caller then runs the
// exact-f32 kernel), or a negative error code.
struct mava_ctx;
// ppo_train_w8.hip: the discrete actor's kernel on eight waves (two per SIMD, 16 features per wave, 16x16x32 MFMAs); same
// return convention.  Tried first by mava_train_h2_launch unless the handle's MAVA_CTX_TRAIN_VARIANT is 1.
int mava_train_w8_launch(const TrainTask& tk, int n_slab, bool actor, hipStream_t s);
int mava_train_h2_launch(mava_ctx* ctx, const TrainTask& tk, int n_slab, bool actor, hipStream_t s);
