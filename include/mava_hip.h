/* libmavahip.so - C ABI of the MI355X-native PPO hot path (gfx950 only).
 *
 * Drop-in boundary: Mava has no FFI; the reusable seam is the Python
 * learner contract  learn(LearnerState) -> ExperimentOutput  (mava/types.py:146-154,
 * call site mava/systems/ppo/ff_mappo.py:497).  mava_amd/learner.py keeps that contract and
 * drives the device work through the entry points below with ctypes.  Every pointer is a
 * DEVICE pointer (torch.Tensor.data_ptr()) unless marked "host"; buffers are caller-owned,
 * contiguous, and never retained past the call.  Every function is asynchronous on the given
 * stream and returns 0 on success, -(hipError_t) for a runtime failure or -1000-k for a
 * rejected argument; the message is available from mava_last_error().
 *
 * Each entry cites the reference code it replaces (paths relative to the Mava repository).
 */
#ifndef MAVA_HIP_H
#define MAVA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mava_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------------------------- */
const char* mava_last_error(void); /* thread-local text of the last failure */
int mava_abi_version(void);

/* ---- context handle: the library keeps NO process-wide state ("no hidden globals except handles").
 * Everything that selects behaviour - the arithmetic of the matrix products, the critic aggregation, kernel variants for
 * bench sweeps - and every workspace the library allocates itself (the pre-split W1 copy of the wide f16x2 gradient
 * kernels) belongs to a handle the caller creates per learner and passes as the first argument of the entry points that
 * depend on it.  NULL is accepted everywhere and means the defaults (exact f32, aggregation on, default variants).  Two
 * handles share nothing; launches that share one handle must be issued on one stream (they share its W1 workspace).
 * In Mava the same choices are properties of the jitted learner function (mava/systems/ppo/ff_mappo.py:388-389). */
typedef struct mava_ctx mava_ctx;
enum {
  MAVA_CTX_MATMUL_MODE = 0,        /* 0 (default): exact-f32 MFMA; 1: "f16x2" - every matrix operand split into two f16 terms
                                      (hi + lo), three f16 MFMAs per product, f32 accumulation: 5.3x less matrix-pipe time, the
                                      same 1e-4 gradient parity tests; applies where a split-f16 kernel is instantiated
                                      (ppo_train_h2.hip, rec_dense_h2.hip, rec_gru_h2.hip), other shapes run exact f32 */
  MAVA_CTX_CRITIC_AGGREGATION = 1, /* 1 (default): a critic input row shared by the A agents of a (t,e) index (RWARE's global
                                      state, mava/wrappers/jumanji.py:53-59) is evaluated once and the sum of its agents' loss
                                      gradients back-propagated - the same gradient as A identical passes; 0: one pass per agent */
  MAVA_CTX_GAE_VARIANT = 2,        /* bench sweeps (tools/gae_sweep.py): chunk / lane mapping of mava_gae_f32, 0 = default */
  MAVA_CTX_POLICY_VARIANT = 3,     /* 0 = per-wave acting kernel / hybrid launch (default), 1 = per-wave only, 2 = block-cooperative */
  MAVA_CTX_H2_LAUNCHES = 4,        /* diagnostic counter: gradient launches of this handle that ran on the f16x2 kernels */
  MAVA_CTX_TRAIN_VARIANT = 5,      /* f16x2 gradient kernels, bit 0: 0 (default) = the eight-wave actor kernel (ppo_train_w8.hip) where it
                                      is instantiated, 1 = the four-wave kernels (ppo_train_h2.hip) only; bit 1: always run the x_lo
                                      products that inputs exact in f16 skip (same bits either way: tests); for A/B measurements */
  MAVA_CTX_W8_LAUNCHES = 6,        /* diagnostic counter: ... of which on the eight-wave kernel */
  MAVA_CTX_W1_SPLIT_FRESH = 7      /* read: 1 while the handle's pre-split W1 copy (wide f16x2 critic) already matches the parameters of
                                      the next gradient launch (mava_ppo_finish_f32 wrote it); write 0: the caller changed parameters
                                      by other means - the next launch re-splits them itself */
};
int mava_ctx_create(mava_ctx** out);
int mava_ctx_destroy(mava_ctx* ctx); /* frees the handle's workspaces; NULL is a no-op */
int mava_ctx_set(mava_ctx* ctx, int key, long value);
int mava_ctx_get(const mava_ctx* ctx, int key, long* value);

/* ---- GAE: mava/systems/ppo/ff_mappo.py:112-139 (_calculate_gae);
 *           mava/systems/ppo/rec_mappo.py:177-199 when last_done != NULL.
 * reward,value,adv,tgt: (T,N) f32 time-major; done: (T,N) u8; last_val: (N) f32;
 * last_done: (N) u8 or NULL.  Recurrent mode: done[t] is the flag ENTERING step t and the mask
 * of step t is done[t+1] (done[T] := last_done). */
int mava_gae_f32(const mava_ctx* ctx, const float* reward, const float* value, const uint8_t* done,
                 const float* last_val, const uint8_t* last_done, int T, int N, float gamma,
                 float lambda, float* adv, float* tgt, mava_stream_t s);

/* ---- epoch permutation: jax.random.permutation(key, batch_size) of ff_mappo.py:272-273 / rec_mappo.py:277-279.
 * out[0..n) = a bijection of [0, n) determined by (seed, counter): 16-round keyed Feistel network over [0, 2^ceil(log2 n))
 * with cycle walking (mava_amd/csrc/permutation.hip; bit-exact restatement oracle/permutation.py).  1 <= n < 2^31. */
int mava_permutation_i32(long n, uint64_t seed, uint64_t counter, int32_t* out, mava_stream_t s);

/* ---- optimiser: optax.chain(clip_by_global_norm, adam(eps=1e-5)) per network,
 *      mava/systems/ppo/ff_mappo.py:359-366 (definition) and :241-250 (application);
 *      LR schedule mava/utils/training.py:20-64; pmean scaling ff_mappo.py:224-238.
 * p,g,m,v: flat f32 over all segments (networks); seg_off: host int[n_seg+1]; seg_lr: host
 * float[n_seg]; count: device int32[n_seg] (incremented).  g holds the SUM over replicas and
 * ranks; grad_scale = 1/(update_batch_size * n_devices).  loss_sums (device, 3 floats:
 * actor_loss, entropy, value_loss - same summation) and metrics_out (device, 4 floats:
 * total_loss, value_loss, actor_loss, entropy; ff_mappo.py:255-265) may both be NULL. */
int mava_clip_adam(float* p, const float* g, float* m, float* v, int32_t* count,
                   const int* seg_off, const float* seg_lr, int n_seg, float grad_scale,
                   float max_norm, int decay, int steps_per_update, int num_updates, float b1,
                   float b2, float eps, const float* loss_sums, float vf_coef, float ent_coef,
                   float* metrics_out, mava_stream_t s);

/* Tail of one minibatch on ONE rank with ONE update-batch replica (ff_mappo.py:224-266 with trivial pmeans) in two launches
 * instead of six: (1) the fixed-order sums of BOTH networks' gradient slabs (slab_a rows [Pa grads | actor_loss, entropy],
 * slab_c rows [Pc grads | value_loss]) into g = [actor | critic | actor_loss, entropy, value_loss, pad] - bit-identical to
 * two mava_slab_reduce2_f32 calls - leaving each block's squared-norm partial in the workspace; (2) clip + Adam as
 * mava_clip_adam (norms from those partials), the step-count increment and, through an arrival ticket, the re-split of a
 * wide f16x2 critic's W1 (critic_din in [96, 287], handle in f16x2 mode; 0 = none) for the handle's next gradient launch by
 * the block that finishes last.  workspace: mava_ppo_finish_workspace_bytes(Pa, Pc) bytes, 8-byte aligned, zeroed once by
 * the caller and owned by this call sequence afterwards. */
size_t mava_ppo_finish_workspace_bytes(int Pa, int Pc);
/* (n_slab = 0: `g` already holds the summed gradient and loss sums - the multi-rank path, whose all-reduce sits between the slab
 * sums and Adam: only the Adam launch runs, norms from g, still carrying the count increment and the W1 re-split.) */
int mava_ppo_finish_f32(mava_ctx* ctx, const float* slab_a, long stride_a, const float* slab_c, long stride_c, int n_slab,
                        int Pa, int Pc, float* g, float* p, float* m, float* v, int32_t* count, float lr_a, float lr_c,
                        float grad_scale, float max_norm, int decay, int steps_per_update, int num_updates, float b1,
                        float b2, float eps, float vf_coef, float ent_coef, float* metrics_out, int critic_din,
                        void* workspace, size_t workspace_bytes, mava_stream_t s);

/* out[i] = (accumulate ? out[i] : 0) + sum_b slab[b*slab_stride + i], b ascending. */
int mava_slab_reduce_f32(const float* slab, int n_slab, long slab_stride, int n, int accumulate,
                         float* out, mava_stream_t s);

/* Same, with row columns [0,n_main) summed into out_main and [n_main, n_main+n_tail) into out_tail
 * (the loss sums that follow a gradient in a slab row). */
int mava_slab_reduce2_f32(const float* slab, int n_slab, long slab_stride, int n_main,
                          float* out_main, int n_tail, float* out_tail, int accumulate,
                          mava_stream_t s);

/* ---- networks: mava/networks.py:39-58 (MLPTorso [128,128] relu), :88-124
 *      (DiscreteActionHead), :172-207 (FeedForwardActor / FeedForwardValueNet).
 * Flat parameter layout [W1(din,128) | b1 | W2(128,128) | b2 | W3(128,n_out) | b3], kernels
 * row-major (in,out) as in flax.linen.Dense. */
int mava_mlp_param_count(int din, int n_out);

/* out (rows, n_out) = net(x[row / x_share]);  x: (ceil(rows/x_share), din). */
int mava_mlp_forward_f32(const mava_ctx* ctx, const float* params, int din, int n_out, const float* x, int x_share,
                         int rows, float* out, mava_stream_t s);

/* (Every *_continuous entry point and mava_rec_step_packed_f32 take ContinuousActionHead.min_scale - mava/networks.py:134,162,
 * default 1e-3: scale = softplus(log_std) + min_scale - right after the action dimension.)
 * One acting step, mava/systems/ppo/ff_mappo.py:80-85: actor forward + action mask
 * (networks.py:116-120) + Categorical sample / log_prob (distributions.py:146-165), and critic
 * forward, in one launch.  rows = envs*agents.  agents_view (rows, actor_din);
 * action_mask (rows, n_actions) u8 or NULL; critic_input (critic_rows/critic_share.., critic_din)
 * with critic row r reading input row r / critic_share; each critic output is written
 * value_broadcast times (value index r*value_broadcast + b).  Sampling: Gumbel-max on
 * Philox4x32-10 keyed by seed with counter (row_offset + row, step, word-group, "POLI").
 * forced_action (rows) i32 or NULL: score these actions instead of sampling.
 * logits (rows, n_actions) or NULL: raw (unmasked) logits out.
 * step_base (device, or NULL): a word added to `step` on the device, so that a captured HIP graph of a whole rollout
 * replays with the next counters (mava_synth_rware_step takes t_base likewise).
 * rows == 0 or critic_rows == 0 skips that half (its pointers may then be NULL): the learner runs the actor
 * half on the acting stream and the value half (mava_mlp_forward_f32) on a side stream. */
int mava_policy_step_f32(const mava_ctx* ctx, const float* actor_params, int actor_din, int n_actions,
                         const float* agents_view, const uint8_t* action_mask,
                         const float* critic_params, int critic_din, const float* critic_input,
                         int critic_share, int critic_rows, int value_broadcast, int rows,
                         uint64_t seed, uint32_t step, const uint32_t* step_base, uint32_t row_offset, int greedy,
                         const int32_t* forced_action, int32_t* action, float* log_prob,
                         float* value, float* logits, mava_stream_t s);

/* ---- PPO minibatch gradients: mava/systems/ppo/ff_mappo.py:150-218 (losses + value_and_grad)
 *      with the shuffle of :268-285 applied as an index vector (no shuffled copy).
 * The trajectory is the flat time-major batch of TE = T*E env rows with A agent rows each:
 * every per-agent array is (TE*A, ...).  A minibatch is Rb env rows: idx[b] (int32, a slice of the
 * epoch permutation) or idx_base + b when idx is NULL.  Each of the n_slab persistent blocks
 * writes one partial slab of slab_stride floats: [gradient in parameter layout | loss sums];
 * mava_slab_reduce_f32 sums the slabs in a fixed order.  All sums are already divided by the
 * minibatch element count Rb*A (the .mean() of the reference).  Row arithmetic is 32-bit: TE*A < 2^31. */

int mava_adv_stats_blocks(void); /* number of (sum, sumsq) f64 pairs mava_adv_stats_f64 writes */

/* advantage statistics for ff_mappo.py:164 (mean / population std over the whole minibatch) */
int mava_adv_stats_f64(const float* advantages, const int32_t* idx, long idx_base, int Rb, int A,
                       double* partials, mava_stream_t s);
/* the same for n_batch minibatches in one launch: minibatch j uses idx[j * idx_stride ...) and writes
 * partials[j][mava_adv_stats_blocks()][2] (all epochs x minibatches of an update, once GAE has run) */
int mava_adv_stats_batched_f64(const float* advantages, const int32_t* idx, long idx_stride, int Rb, int A,
                               int n_batch, double* partials, mava_stream_t s);

/* _actor_loss_fn, ff_mappo.py:150-180.  slab tail: [actor_loss, entropy]. */
int mava_ppo_actor_grad_f32(mava_ctx* ctx, const float* params, int din, int n_actions, const float* agents_view,
                            const uint8_t* action_mask, const int32_t* action,
                            const float* old_log_prob, const float* advantages,
                            const double* adv_stats, const int32_t* idx, long idx_base, int Rb,
                            int A, float clip_eps, float ent_coef, float* slab, long slab_stride,
                            int n_slab, mava_stream_t s);

/* Continuous action head: Independent(TanhTransformed(Normal(loc, softplus(log_std) + 1e-3))),
 * mava/networks.py:127-169 + mava/distributions.py:24-91 (log_prob clipped at +-0.999, sampled entropy).
 * actor_params = [MLP(actor_din -> 128 -> 128 -> action_dim) | log_std(action_dim)], action_dim <= 16.
 *
 * mava_policy_step_continuous_f32: the acting step of mava_policy_step_f32 with action (rows, action_dim) float in
 * (-1, 1) = tanh(loc + scale * noise) (greedy: the mode tanh(loc), distributions.py:75-77); forced_action (or NULL)
 * scores given actions instead; mean (or NULL) receives loc.  The critic half is unchanged.
 *
 * mava_ppo_actor_grad_continuous_f32: _actor_loss_fn ff_mappo.py:150-180 for this head; the entropy sample of
 * ff_mappo.py:176-177 is drawn from Philox(counter (row_offset + trajectory row, ent_step, dim/2), key seed).
 * slab row = [MLP gradient | d/d log_std | actor_loss, entropy]. */
int mava_policy_step_continuous_f32(const mava_ctx* ctx, const float* actor_params, int actor_din, int action_dim, float min_scale,
                                    const float* agents_view, const float* critic_params, int critic_din,
                                    const float* critic_input, int critic_share, int critic_rows,
                                    int value_broadcast, int rows, uint64_t seed, uint32_t step,
                                    const uint32_t* step_base, uint32_t row_offset, int greedy,
                                    const float* forced_action, float* action, float* log_prob, float* value,
                                    float* mean, mava_stream_t s);
int mava_ppo_actor_grad_continuous_f32(const float* params, int din, int action_dim, float min_scale, const float* agents_view,
                                       const float* action, const float* old_log_prob, const float* advantages,
                                       const double* adv_stats, const int32_t* idx, long idx_base, int Rb, int A,
                                       float clip_eps, float ent_coef, uint64_t seed, uint32_t ent_step,
                                       uint32_t row_offset, float* slab, long slab_stride, int n_slab,
                                       mava_stream_t s);

/* _critic_loss_fn, ff_mappo.py:182-201.  critic_input row = agent_row / x_share.
 * slab tail: [value_loss, unused]. */
int mava_ppo_critic_grad_f32(mava_ctx* ctx, const float* params, int din, const float* critic_input, int x_share,
                             const float* old_value, const float* targets, const int32_t* idx,
                             long idx_base, int Rb, int A, float clip_eps, float vf_coef,
                             float* slab, long slab_stride, int n_slab, mava_stream_t s);

/* Both gradient kernels read MAVA_CTX_MATMUL_MODE, the critic also MAVA_CTX_CRITIC_AGGREGATION, from their handle. */

/* ---- synthetic RWARE-shaped environment (measurement stand-in for the third-party Jumanji
 *      RobotWarehouse stepped at mava/systems/ppo/ff_mappo.py:88).  Wrapper semantics follow
 *      mava/wrappers/observation.py:41-53, jumanji.py:53-59,128-143, auto_reset_wrapper.py:88-101
 *      and episode_metrics.py:78-111.  One call = one vectorised env.step (or reset when is_reset).
 * state: step_count (E,A) i32, run_return/ep_return (E) f32, run_length/ep_length (E) i32.
 * outputs: agents_view (E,A,A+O), global_state (E,gs_tiles,W) with gs_tiles in {1,A} and W = A*O (concatenated
 * raw views, RWARE) when state_dim == 0 or W = state_dim (independent state vector, SMAX-shaped),
 * action_mask (E,A,n_actions) u8, obs_step_count (E,A) i32; transition: reward (E,A) f32,
 * done (E,A) u8, info_return (E) f32, info_length (E) i32, info_terminal (E) u8.
 * reward_mode 0: team reward Bernoulli(0.02), independent of the actions (the measurement workload);
 * reward_mode 1 ("match", for learning tests): team reward = fraction of the env's agents whose `action` (E,A) i32 -
 * taken on the previous observation - equals (that observation's first grid coordinate) mod n_actions. */
int mava_synth_rware_step(int E, int A, int O, int n_actions, int gs_tiles, int state_dim, int time_limit,
                          uint64_t seed, uint32_t t, const uint32_t* t_base, uint32_t env_offset, int is_reset,
                          int32_t* step_count, float* run_return, int32_t* run_length,
                          float* ep_return, int32_t* ep_length, float* agents_view,
                          float* global_state, uint8_t* action_mask, int32_t* obs_step_count,
                          float* reward, uint8_t* done, float* info_return, int32_t* info_length,
                          uint8_t* info_terminal, const int32_t* action, int reward_mode, mava_stream_t s);

/* ---- fused rollout: the whole `lax.scan(_env_step, length=T)` of mava/systems/ppo/ff_mappo.py:76-106 for one
 *      update-batch replica on the synthetic RWARE-shaped environment, plus the bootstrap value of :109-110, in ONE
 *      launch (mava_amd/csrc/rollout_h2.hip): every workgroup owns 64 / A environments for all T steps (environments
 *      are independent, the parameters fixed), weights register-resident, observations handed from the env phase to
 *      the next acting step through LDS.  Same Philox streams as mava_policy_step_f32 (seed policy_seed, counter
 *      (row_offset + row, t0 + t, ., "POLI")) and mava_synth_rware_step (seed env_seed, step t0 + t + 1): bit-identical
 *      observations / masks / rewards / dones / metrics; networks in split-f16 arithmetic (see MAVA_CTX_MATMUL_MODE).
 *      critic_shared 1: centralised critic on global_state (T+1, E, A*O), one value per env broadcast to its agents;
 *      0: decentralised critic on agents_view (global_state unused).  Slot 0 of agents_view / global_state /
 *      action_mask must hold the current observation; slots 1..T are written, the env state is advanced in place.
 *      adv / tgt (T, E, A) or both NULL: when given, every (env, agent) column's GAE (ff_mappo.py:112-139, sequential
 *      f32 recurrence) runs in the kernel's tail on the rewards / values / dones just written - no separate pass.
 *      Returns 0, a negative error, or 1 when the shape is not instantiated (the caller then runs mava_policy_step_f32
 *      + mava_synth_rware_step per step): needs 64 % A == 0, n_actions <= 8, the RWARE-style global state. */
int mava_rollout_ff_f32(const float* actor_params, int n_actions, const float* critic_params, int critic_shared,
                        int E, int A, int O, int T, int time_limit, uint64_t policy_seed, uint64_t env_seed,
                        uint32_t t0, uint32_t row_offset, uint32_t env_offset, int reward_mode,
                        int32_t* step_count, float* run_return, int32_t* run_length, float* ep_return,
                        int32_t* ep_length, float* agents_view, float* global_state, uint8_t* action_mask,
                        int32_t* obs_step_count, int32_t* action, float* value, float* reward, float* log_prob,
                        uint8_t* done, float* last_val, float* info_return, int32_t* info_length,
                        uint8_t* info_terminal, float* adv, float* tgt, float gamma, float gae_lambda, mava_stream_t s);

/* ---- recurrent systems (rec_ippo / rec_mappo): mava/networks.py:238-331 (ScannedRNN GRU with
 *      reset-on-done, RecurrentActor, RecurrentValueNet), mava/systems/ppo/rec_mappo.py:91-149,
 *      :210-266, :334-365.  Internal activations use the "T32" tile layout: element (row, f) of a
 *      (rows x N) matrix lives at ((row/32)*N + f)*32 + row%32; rows is a multiple of 32.  A sequence batch
 *      is time-major: row = t*Rm + m, m = (local env)*A + agent; external trajectory arrays are (T,E,A,..)
 *      and env ids come from idx (Rm/A entries, a slice of the env permutation) or the identity. */

/* Y = act(X W + b) [masked by gate > 0]; X is T32 (x_ld >= K features per 32-row tile, the first K used; x_ld <= 0
 * means K) or, with x_rowmajor, the external row-major source (row stride x_ld >= K, so a call can read a column
 * block) gathered per batch row; W (K x N) row-major with row stride ldw; Y T32 (rows x N).  accumulate: start from
 * the existing Y (K-chunked products for inputs wider than 384).  With MAVA_CTX_MATMUL_MODE = 1 T32 inputs run on
 * split-f16 operands (rec_dense_h2.hip: 3 f16 MFMAs per product, f32 accumulate, operands must sit in f16 range -
 * see grad_scale below); row-major inputs always run the exact-f32 kernel.  y_ld (<= 0: N) is the feature count of the
 * y / gate tiles: a call with a column block of W and y + 32 * n0 writes features [n0, n0 + N) of a wider matrix. */
int mava_rec_dense_f32(const mava_ctx* ctx, const float* x, int x_rowmajor, const int32_t* idx, int Rm, int E, int A,
                       int x_share, int x_ld, int accumulate, const float* w, int ldw, const float* bias,
                       const float* gate, float* y, int y_ld, int K, int N, int rows, int relu, mava_stream_t s);

/* per-block slabs of out_scale * dW = X^T Y (K x N row-major) followed by out_scale * db = colsum(Y) when want_bias;
 * y_ld (<= 0: N) = features per y tile (y + 32 * n0 reads a column block of a wider matrix).
 * out_scale undoes the grad_scale the backward chain was started with (1.0f when none). */
int mava_rec_xty_f32(const mava_ctx* ctx, const float* x, int x_rowmajor, const int32_t* idx, int Rm, int E, int A,
                     int x_share, int x_ld, const float* y, int y_ld, const float* y_tail, int y_split, int y_tail_ld, int K, int N,
                     int rows, int want_bias, float out_scale, float* slab, long slab_stride, int n_slab, mava_stream_t s);
/* (y_tail, round 3: when not NULL, features [y_split, N) of Y are features [0, N - y_split) of the T32 matrix y_tail with
 * y_tail_ld features per tile; y_split a multiple of 32.  The BPTT scan's dgh is [dgi's r and z thirds | its own n third].) */
/* Row-major, env-permuted observation slice of a minibatch (same gather description as mava_rec_dense_f32 with
 * x_rowmajor) -> T32 matrix `out` with k_pad >= K features per tile (zeros past K): done once per minibatch, read by
 * the pre-torso product and by its weight-gradient product as a plain T32 operand. */
int mava_rec_gather_t32_f32(const float* x, const int32_t* idx, int Rm, int E, int A, int x_share, int x_ld, int K,
                            int rows, int k_pad, float* out, mava_stream_t s);

/* GRU over T steps (flax GRUCell; hidden state zeroed where done enters the step).  gi = W_i x + b_i
 * precomputed (T32, T*Rm x 384); wh (128 x 384) = [hr|hz|hn]; outputs hs (T32, h after each step) and,
 * for training, hprev (masked h entering each step) and saved (T*Rm x 512 = [r|z|n|W_hn h + b_hn]). */
int mava_gru_scan_fwd_f32(const mava_ctx* ctx, int T, int Rm, int E, int A, const int32_t* idx, const uint8_t* done,
                          const float* h0, int h0_t32, const float* wh, const float* bhn,
                          const float* gi, float* hs, float* hprev, float* saved, mava_stream_t s);

/* BPTT through the same scan: dh_out (T32) is the gradient reaching each h_t from the output path;
 * writes dgi and dgh (T32, T*Rm x 384: gradients w.r.t. the input-side and hidden-side gate pre-activations). */
int mava_gru_scan_bwd_f32(const mava_ctx* ctx, int T, int Rm, int E, int A, const int32_t* idx, const uint8_t* done,
                          const float* wh, const float* saved, const float* hprev, const float* dh_out,
                          float* dgi, float* dgh, int dgh_n_only, mava_stream_t s);
/* (dgh_n_only != 0: dgh is T32 (T*Rm x 128) and receives the n third of the hidden-side gate gradient alone - its r and z
 * thirds equal dgi's, since the gates add the two pre-activations (flax GRUCell); the scan is HBM-bound, the two thirds
 * were a sixth of its traffic.) */

/* sequence losses on T32 logits / values: rec_mappo.py:210-242 and :244-266 (after the re-unroll);
 * loss_partials: (n_blocks, 2) partial sums already divided by the element count.  The gradient w.r.t. the network
 * outputs is written TIMES grad_scale: loss gradients are O(1 / rows), below f16's normal range at 10^6 rows, so the
 * backward chain (dense, BPTT scan - all linear in the gradient) runs in units of a power of two near `rows` and
 * mava_rec_xty_f32(out_scale = 1 / grad_scale) returns to true units; exact in f32, 1.0f for none. */
int mava_seq_actor_loss_f32(int T, int Rm, int E, int A, int n_actions, const int32_t* idx,
                            const float* logits, const uint8_t* mask, const int32_t* action,
                            const float* old_log_prob, const float* advantages, const double* adv_stats,
                            int n_stats, float clip_eps, float ent_coef, float grad_scale, float* dlogits,
                            float* loss_partials, int n_blocks, mava_stream_t s);
/* Continuous head on the recurrent systems (rec_mappo.py:210-242 with networks.py:127-169): `mean` / `dmean` are T32
 * (T*Rm x action_dim) like logits / dlogits of mava_seq_actor_loss_f32, `action` is the external (T, E, A, action_dim)
 * buffer, `log_std` the action_dim raw scales; dscale_partials (n_blocks x action_dim) receives d loss / d log_std
 * partials (already times sigmoid(log_std)).  Entropy noise: Philox counter (row_offset + trajectory row, ent_step, dim/2).
 * ContinuousActionHead(independent_std=False) (networks.py:140,161): log_std_rows is the T32 (T*Rm x action_dim) output
 * of the log_std layer (log_std may then be NULL) and dlog_std_rows receives its gradient (times grad_scale) instead of
 * dscale_partials; both NULL for the observation-independent scale. */
int mava_seq_actor_loss_continuous_f32(int T, int Rm, int E, int A, int action_dim, float min_scale, const int32_t* idx,
                                       const float* mean, const float* log_std, const float* log_std_rows,
                                       const float* action, const float* old_log_prob, const float* advantages,
                                       const double* adv_stats, int n_stats, float clip_eps, float ent_coef, uint64_t seed,
                                       uint32_t ent_step, uint32_t row_offset, float grad_scale, float* dmean,
                                       float* dlog_std_rows, float* loss_partials, float* dscale_partials, int n_blocks,
                                       mava_stream_t s);
/* rollout epilogue: T32 means of one step -> action (rows, action_dim) = tanh(loc + scale * noise), log_prob (rows) */
int mava_seq_sample_continuous_f32(int rows, int action_dim, float min_scale, const float* mean, const float* log_std,
                                   const float* log_std_rows, uint64_t seed,
                                   uint32_t step, uint32_t row_offset, int greedy, float* action, float* log_prob,
                                   mava_stream_t s);
/* agents_per_row = 1: one value per agent row (T, E, A).  agents_per_row = n > 1 (then A must be 1): the rows are
 * (t, env) rows whose n agents share the critic input (centralised critic on a tiled global state); old_value /
 * targets are (T, E, n) and the row's gradient is the sum of its agents' loss gradients - the same gradient as n
 * identical network passes. */
int mava_seq_critic_loss_f32(int T, int Rm, int E, int A, int agents_per_row, const int32_t* idx, const float* values,
                             const float* old_value, const float* targets, float clip_eps, float vf_coef,
                             float grad_scale, float* dvalues, float* loss_partials, int n_blocks, mava_stream_t s);

/* The OUTPUT PATH of a recurrent network in one launch on split-f16 operands (rec_out_h2.hip): post_torso -> head ->
 * PPO loss (rec_mappo.py:210-242 actor / :244-266 critic) -> backward through both layers.  hs: T32 (T*Rm x 128) hidden
 * states; params_post_head: [Wpost | bpost | Whead | bhead] (the tail of the flat parameters); is_actor: f0 / f1 =
 * old_log_prob / advantages (+ mask, action, adv_stats), else old_value / targets with agents_per_row as in
 * mava_seq_critic_loss_f32.  Writes dh (T32, d loss / d hs TIMES grad_scale) and per-block slabs [dWpost | dbpost | dWhead |
 * dbhead | loss sums (2)] in true units.  Returns 1 (nothing launched) for shapes it does not instantiate: more than 16
 * outputs, n_slab > tiles - the caller then runs the layer-wise kernels. */
int mava_rec_out_f32(int T, int Rm, int E, int A, int n_out, int agents_per_row, const int32_t* idx, const float* hs,
                     const float* params_post_head, const uint8_t* mask, const int32_t* action, const float* f0,
                     const float* f1, const double* adv_stats, int n_stats, float clip_eps, float coef, float grad_scale,
                     int is_actor, float* dh, float* slab, long slab_stride, int n_slab, mava_stream_t s);

/* rollout epilogue: masked Categorical sample + log_prob from T32 logits of one step (rows = E*A). */
int mava_seq_sample_f32(int rows, int n_actions, const float* logits, const uint8_t* mask, uint64_t seed,
                        uint32_t step, uint32_t row_offset, int greedy, int32_t* action, float* log_prob,
                        mava_stream_t s);

/* Fused recurrent ACTING step: rec_mappo.py:108-129 (_env_step) for both networks in one launch -
 * pre_torso -> GRU cell with reset-on-done (networks.py:238-266) -> post_torso -> head, then mask / Gumbel-max
 * sample / log-prob (actor; same Philox stream as mava_seq_sample_f32) or the value (critic).
 * Parameters: the recurrent flat layout [Wpre | bpre | Wi | bi | Wh | bhn | Wpost | bpost | Whead | bhead].
 * agents_view (rows_a, actor_din), critic_input (ceil(rows_c / critic_share), critic_din) row-major;
 * done_* u8 flags entering the step, one per row (critic row r reads done_c[r * done_c_stride]: stride A reads the
 * per-agent flags of an env's first agent when the critic runs once per env); h_*_in / h_*_out: T32 (rows x 128)
 * hidden states, distinct buffers;
 * rows_* multiples of 32 (0 skips that network); value (rows_c * value_broadcast). */
int mava_rec_step_f32(const float* actor_params, int actor_din, int n_actions, const float* agents_view,
                      const uint8_t* action_mask, const uint8_t* done_a, const float* h_actor_in,
                      float* h_actor_out, int rows_a, uint64_t seed, uint32_t step, uint32_t row_offset,
                      int greedy, int32_t* action, float* log_prob, const float* critic_params, int critic_din,
                      const float* critic_input, int critic_share, const uint8_t* done_c, int done_c_stride,
                      const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                      float* value, mava_stream_t s);
/* the same fused acting step with ContinuousActionHead: actor_params = [recurrent network | log_std(action_dim)],
 * action (rows_a, action_dim) float = tanh(loc + scale * noise) (noise stream of mava_seq_sample_continuous_f32). */
int mava_rec_step_continuous_f32(const float* actor_params, int actor_din, int action_dim, float min_scale, const float* agents_view,
                                 const uint8_t* done_a, const float* h_actor_in, float* h_actor_out, int rows_a,
                                 uint64_t seed, uint32_t step, uint32_t row_offset, int greedy, float* action,
                                 float* log_prob, const float* critic_params, int critic_din,
                                 const float* critic_input, int critic_share, const uint8_t* done_c, int done_c_stride,
                                 const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                                 float* value, mava_stream_t s);

/* The same acting step on split-f16 operands with PRE-PACKED weights (rec_step_h2.hip): mava_rec_step_pack_f32 splits
 * one network's [Wpre | Wi | Wh | Wpost] into f16 hi / lo planes in MFMA-fragment order (mava_rec_step_pack_bytes(din)
 * bytes); do it once per rollout, after the parameters changed.  mava_rec_step_packed_f32 takes both packs plus the
 * arguments of mava_rec_step_f32 (action_f != NULL selects the continuous head: then action / action_mask are unused and
 * n_actions is the action dimension); a block multiplies each weight fragment with up to three 32-row tiles.  More than
 * 16 head outputs run the exact-f32 kernel. */
long mava_rec_step_pack_bytes(int din);
int mava_rec_step_pack_f32(const float* params, int din, void* pack, mava_stream_t s);
int mava_rec_step_packed_f32(const void* pack_a, const void* pack_c, const float* actor_params, int actor_din,
                             int n_actions, float min_scale, const float* agents_view, const uint8_t* action_mask, const uint8_t* done_a,
                             const float* h_actor_in, float* h_actor_out, int rows_a, uint64_t seed, uint32_t step,
                             uint32_t row_offset, int greedy, int32_t* action, float* action_f, float* log_prob,
                             const float* critic_params, int critic_din, const float* critic_input, int critic_share,
                             const uint8_t* done_c, int done_c_stride, const float* h_critic_in, float* h_critic_out,
                             int rows_c, int value_broadcast, float* value, mava_stream_t s);

/* T32 <-> row-major conversion of a (rows x N) matrix. */
int mava_t32_convert_f32(const float* src, int N, int rows, int to_t32, float* dst, mava_stream_t s);

/* ---- general network path (torsos the fused kernels do not instantiate): mava/networks.py:39-58 MLPTorso with any
 *      layer_sizes, activation relu | tanh and use_layer_norm, mava/networks.py:61-85 CNNTorso.  Products run on
 *      mava_rec_dense_f32 / mava_rec_xty_f32; these are the kernels between them.  act: 0 none, 1 relu, 2 tanh. */
/* y = act(LayerNorm(x) + ln_bias) (flax LayerNorm over the last axis, eps 1e-6, no scale) or act(x); xhat (T32) and
 * rstd (rows) are saved for the backward pass when use_layer_norm. */
int mava_t32_norm_act_f32(const float* x, int N, long rows, int use_layer_norm, const float* ln_bias, int act, float* y,
                          float* xhat, float* rstd, mava_stream_t s);
/* dz = dy * act'(y) (the bias gradient is its column sum); dx = gradient w.r.t. the layer-norm input (= dz without it) */
int mava_t32_norm_act_bwd_f32(const float* dy, const float* y, int N, long rows, int use_layer_norm, const float* xhat,
                              const float* rstd, int act, float* dz, float* dx, mava_stream_t s);
/* GRU cell one time step at a time, for network.hidden_state_dim != 128 (mava/networks.py:222-266: ScannedRNN over flax
 * GRUCell; the register-resident scans mava_gru_scan_* serve 128).  All matrices T32 over the `rows` sequences of the step;
 * done_t / done_next: the external (E, A) u8 flags ENTERING that step, rows mapped through idx like the scans; the
 * recurrent products (h W_h, dgh W_h^T) are mava_rec_dense_f32 launches between these kernels.
 *   mask:      hprev = done_t ? 0 : h
 *   gates:     hs = GRUCell(gi, gh = hprev W_h, b_hn, hprev); saved (rows x 4 Hd) = [r | z | n | gh_n + b_hn] or NULL;
 *              hprev_next (or NULL) = done_next ? 0 : hs
 *   gates_bwd: dh = dh_out + (done_next ? 0 : acc_next + dhp_next) (acc_next = dgh_{t+1} W_h^T; NULL at the last step);
 *              dgi, dgh (rows x 3 Hd) and dhp = dh z */
int mava_t32_gru_mask_f32(const float* h, const uint8_t* done_t, const int32_t* idx, int E, int A, int Hd, long rows,
                          float* hprev, mava_stream_t s);
int mava_t32_gru_gates_f32(const float* gi, const float* gh, const float* bhn, const float* hprev, int Hd, long rows, float* hs,
                           float* saved, float* hprev_next, const uint8_t* done_next, const int32_t* idx, int E, int A,
                           mava_stream_t s);
int mava_t32_gru_gates_bwd_f32(const float* saved, const float* hprev, const float* dh_out, const float* acc_next,
                               const float* dhp_next, const uint8_t* done_next, const int32_t* idx, int E, int A, int Hd,
                               long rows, float* dgi, float* dgh, float* dhp, mava_stream_t s);
/* per-block partial column sums of a T32 matrix times `scale`: slab (n_slab, slab_stride >= N) */
int mava_t32_colsum_f32(const float* y, int N, long rows, float scale, float* slab, long slab_stride, int n_slab,
                        mava_stream_t s);
/* nn.Conv(padding='SAME') as a product: patches of (Hin, Win, C) images -> T32 (samples*Hout*Wout x k*k*C), feature
 * (ky*k + kx)*C + c = a row of the flattened (k, k, C, Cout) kernel.  src_flat = 1: source T32 (samples x Hin*Win*C) (the
 * gathered observations); 0: source T32 (samples*Hin*Win x C) (the previous conv layer).  col2im is its adjoint. */
int mava_t32_im2col_f32(const float* src, int src_flat, long samples, int Hin, int Win, int C, int k, int stride,
                        float* dst, mava_stream_t s);
int mava_t32_col2im_f32(const float* dcol, int src_flat, long samples, int Hin, int Win, int C, int k, int stride,
                        float* dsrc, mava_stream_t s);
/* jax.lax.collapse(x, -3): T32 (samples*P x C) -> T32 (samples x P*C) (to_flat = 1) and back (0) */
int mava_t32_flatten_f32(const float* src, long samples, int P, int C, int to_flat, float* dst, mava_stream_t s);

/* ---- exchange step: replaces the jax.lax.pmean calls of mava/systems/ppo/ff_mappo.py:224-238
 *      (actor and critic (grads, loss_info) over the "batch" and "device" axes) for a host without torch.distributed.
 *      One process per GPU: rank 0 obtains a 128-byte id (mava_comm_unique_id) and hands it to the other ranks by any
 *      host channel; every rank then creates its communicator on ITS current HIP device.  mava_allreduce_sum_f32 sums
 *      the flat [actor grads | critic grads | loss scalars] buffer over the ranks in place, asynchronously on stream s
 *      (RCCL over xGMI); the 1 / (update_batch_size * world) of the two pmeans is mava_clip_adam_f32's grad_scale.
 *      mava_broadcast_f32 replicates rank `root`'s parameters (flax.jax_utils.replicate, ff_mappo.py:426).
 *      librccl.so is dlopen-ed at the first call (MAVA_RCCL_LIB overrides the name), so a host that already carries an
 *      RCCL shares it.  RCCL errors come back as -2000 - ncclResult_t. */
int mava_comm_unique_id(uint8_t* id128);
int mava_comm_create(void** h, int rank, int world, const uint8_t* id128);
int mava_allreduce_sum_f32(void* h, float* buf, size_t n, mava_stream_t s);
int mava_broadcast_f32(void* h, float* buf, size_t n, int root, mava_stream_t s);
int mava_comm_destroy(void* h);

#ifdef __cplusplus
}
#endif
#endif /* MAVA_HIP_H */
