// Acting-side kernels (K1 + K2 + K3 of SURVEY §2.1): fused actor forward + action mask +
// categorical sample + log-prob, and critic forward, in ONE launch per environment step.
//
// Reference: mava/systems/ppo/ff_mappo.py:80-85 (_env_step: actor_apply, critic_apply,
// sample, log_prob), mava/networks.py:172-207, mava/distributions.py:146-165.
// Sampling is Gumbel-max like jax.random.categorical, but on this library's own Philox4x32-10
// stream (bit parity with JAX threefry is not a goal; parity tests compare against the oracle's
// restatement of the same Philox stream).
//
// Launch shape: 256-thread blocks (4 waves); each wave runs whole 32-row tiles through all
// three layers in registers (mlp_core.h).  Blocks [0, nblk_actor) serve the actor, the rest
// the critic, so one launch covers 2 * ceil(R/32) wave-tiles (1024 at the BASELINE config-2
// shape: one per SIMD).  W2/W3/biases are staged in LDS once per block.
#include "mlp_coop_body.h"
#include "ctx.h"
#include "tanh_normal.h"

namespace {

struct FwdTask {
  const float* params;
  const float* x;   // (rows_x, din)
  int din;
  int no;
  int xshare;       // x row = agent_row / xshare (A when all agents of an env share one input row)
  int xv;           // x load vector width (1/2/4)
  int R;            // agent rows to evaluate
};

template <int NO>
__device__ __forceinline__ void forward_tile(const FwdTask& tk, const float* lds, int row, bool valid,
                                             int h, int j, float (&y)[NO]) {
  const long xr = valid ? (long)(row / tk.xshare) : 0;
  const float* xrow = tk.x + xr * tk.din;
  const float* W1g = tk.params;
  f32x16 z1[4], z2[4];
  if (tk.xv == 4) {
    mlp_l1_forward<4>(xrow, tk.din, W1g, lds + MlpLds<NO>::B1, h, j, z1);
  } else if (tk.xv == 2) {
    mlp_l1_forward<2>(xrow, tk.din, W1g, lds + MlpLds<NO>::B1, h, j, z1);
  } else {
    mlp_l1_forward<1>(xrow, tk.din, W1g, lds + MlpLds<NO>::B1, h, j, z1);
  }
  mlp_relu(z1);
  mlp_l2_forward(z1, lds + MlpLds<NO>::W2, lds + MlpLds<NO>::B2, h, j, z2);
  mlp_relu(z2);
  mlp_head_forward<NO>(z2, lds + MlpLds<NO>::W3, lds + MlpLds<NO>::B3, h, y);
}

// Raw forward: out[row][o] = network(x[row / xshare])[o]
template <int NO>
__global__ __launch_bounds__(256, 2) void mlp_forward_kernel(FwdTask tk, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  mlp_fill_lds<NO>(lds, tk.params, tk.din, tk.no, 256);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
  const int ntiles = (tk.R + 31) / 32;
  for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += gridDim.x * 4) {
    const int row = tile * 32 + j;
    const bool valid = row < tk.R;
    float y[NO];
    forward_tile<NO>(tk, lds, row, valid, h, j, y);
    if (valid && h == 0) {
      for (int o = 0; o < tk.no && o < NO; ++o) out[(long)row * tk.no + o] = y[o];
    }
  }
}

struct StepOut {
  int32_t* action;      // (R)
  float* log_prob;      // (R)
  float* value;         // (R_critic * vbroadcast)
  float* logits;        // optional (R, no) raw (unmasked) logits, for parity tests
  const int32_t* forced_action;  // optional: evaluate log_prob of given actions instead of sampling
  const uint32_t* step_base;     // optional device word added to `step` (captured HIP graphs replay with a moving counter)
  // continuous head (tanh_normal.h) when action_f != nullptr: the network outputs are the means
  float* action_f;               // (R, no) sampled actions in (-1, 1)
  const float* log_std;          // (no) raw scale parameters
  const float* forced_action_f;  // optional (R, no): score these actions instead of sampling
  float min_scale;               // scale = softplus(log_std) + min_scale (networks.py:134,162)
};

// Per-wave actor: block `bid` of the `nblk` actor blocks; each wave carries whole 32-row tiles through the three
// layers in registers, then masks, samples (Gumbel-max on Philox) and scores the action.
template <int NOA>
__device__ __forceinline__ void actor_step_body(const FwdTask& actor, float* lds, int bid, int nblk,
                                                const uint8_t* __restrict__ mask, uint32_t seed_lo, uint32_t seed_hi,
                                                uint32_t step, uint32_t row_offset, int greedy, const StepOut& out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
  step += out.step_base ? *out.step_base : 0u;
  mlp_fill_lds<NOA>(lds, actor.params, actor.din, actor.no, 256);
  __syncthreads();
  const int ntiles = (actor.R + 31) / 32;
  for (int tile = bid * 4 + w; tile < ntiles; tile += nblk * 4) {
    const int row = tile * 32 + j;
    const bool valid = row < actor.R;
    float y[NOA];
    forward_tile<NOA>(actor, lds, row, valid, h, j, y);
    const int no = actor.no;
    if (out.action_f != nullptr) {  // uniform: continuous head, networks.py:127-169
      const uint32_t gid = row_offset + (uint32_t)row;
      float lp = 0.0f;
#pragma unroll
      for (int o = 0; o < NOA; ++o) {
        if (o < no) {
          const float sc = tn::scale_of(out.log_std[o], out.min_scale);
          float a;
          if (out.forced_action_f != nullptr) {
            a = valid ? out.forced_action_f[(long)row * no + o] : 0.0f;
          } else {
            // sample = tanh(loc + scale * eps); mode = tanh(loc) (distributions.py:75-77)
            const float eps = greedy ? 0.0f : tn::noise(gid, step, o, tn::STREAM_SAMPLE, seed_lo, seed_hi);
            a = tanhf(fmaf(sc, eps, y[o]));
          }
          lp += tn::log_prob(a, y[o], sc).lp;
          if (valid && h == 0) out.action_f[(long)row * no + o] = a;
        }
      }
      if (valid && h == 0) {
        out.log_prob[row] = lp;
        if (out.logits != nullptr)
          for (int o = 0; o < no && o < NOA; ++o) out.logits[(long)row * no + o] = y[o];
      }
      continue;
    }
    Categorical<NOA> cat;
    cat.build(y, (mask != nullptr && valid) ? (mask + (long)row * no) : nullptr, no);
    int a = 0;
    if (out.forced_action != nullptr) {
      a = valid ? out.forced_action[row] : 0;
    } else if (greedy) {
      float best = -FLT_MAX;
#pragma unroll
      for (int o = 0; o < NOA; ++o)
        if (o < no && cat.z[o] > best) { best = cat.z[o]; a = o; }
    } else {
      // Gumbel-max: argmax_o z[o] - log(-log(u_o)), first index wins ties
      float best = -FLT_MAX;
      const uint32_t gid = row_offset + (uint32_t)row;
#pragma unroll
      for (int c = 0; c < (NOA + 3) / 4; ++c) {
        Philox4 rnd = philox4x32_10(gid, step, (uint32_t)c, 0x504f4c49u /*"POLI"*/, seed_lo, seed_hi);
        const uint32_t wds[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o = 4 * c + q;
          if (o < NOA && o < no) {
            const float u = u01_open(wds[q]);
            const float g = -logf(-logf(u));
            const float sc = cat.z[o] + g;
            if (sc > best) { best = sc; a = o; }
          }
        }
      }
    }
    float lp = 0.0f;
#pragma unroll
    for (int o = 0; o < NOA; ++o)
      if (o == a) lp = cat.logp[o];
    if (valid && h == 0) {
      out.action[row] = a;
      out.log_prob[row] = lp;
      if (out.logits != nullptr)
        for (int o = 0; o < no && o < NOA; ++o) out.logits[(long)row * no + o] = y[o];
    }
  }
}

// Hybrid acting step: blocks [0, nblk_actor) run the per-wave actor, the remaining blocks the BLOCK-COOPERATIVE
// critic (4 waves per 32-row tile, coop_body) - for launches with few critic tiles (one value per env when the
// agents share the critic input: 128 tiles at the BASELINE shape).  A wave that carries a whole tile through the
// 264-wide critic is a ~26 us chain of ~900 MFMAs and used to set the step's duration while most SIMDs idled;
// split four ways it is shorter than the actor's chain, and with <= 256 blocks every block has its own CU.
template <int NOA, int KT1C>
__global__ __launch_bounds__(256, 1) void policy_hybrid_kernel(FwdTask actor, coop::CoopTask critic, coop::CoopLds Lc,
                                                               int nblk_actor, const uint8_t* __restrict__ mask,
                                                               uint32_t seed_lo, uint32_t seed_hi, uint32_t step,
                                                               uint32_t row_offset, int greedy, StepOut out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < nblk_actor) {
    actor_step_body<NOA>(actor, lds, (int)blockIdx.x, nblk_actor, mask, seed_lo, seed_hi, step, row_offset, greedy, out);
  } else {
    coop::coop_body<1, KT1C, coop::MODE_VALUE>(critic, Lc, lds, (int)blockIdx.x - nblk_actor, (int)gridDim.x - nblk_actor);
  }
}

template <int NOA>
__global__ __launch_bounds__(256, 2) void policy_step_kernel(FwdTask actor, FwdTask critic,
                                                             int nblk_actor,
                                                             const uint8_t* __restrict__ mask,
                                                             uint32_t seed_lo, uint32_t seed_hi,
                                                             uint32_t step, uint32_t row_offset,
                                                             int vbroadcast, int greedy,
                                                             StepOut out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bool is_actor = (int)blockIdx.x < nblk_actor;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
  if (is_actor) {
    actor_step_body<NOA>(actor, lds, (int)blockIdx.x, nblk_actor, mask, seed_lo, seed_hi, step, row_offset, greedy, out);
  } else {
    const int bid = blockIdx.x - nblk_actor;
    const int nblk = gridDim.x - nblk_actor;
    mlp_fill_lds<1>(lds, critic.params, critic.din, 1, 256);
    __syncthreads();
    const int ntiles = (critic.R + 31) / 32;
    for (int tile = bid * 4 + w; tile < ntiles; tile += nblk * 4) {
      const int row = tile * 32 + j;
      const bool valid = row < critic.R;
      float y[1];
      forward_tile<1>(critic, lds, row, valid, h, j, y);
      if (valid && h == 0) {
        for (int b = 0; b < vbroadcast; ++b) out.value[(long)row * vbroadcast + b] = y[0];
      }
    }
  }
}

int pick_xv(const float* x, int din) {
  const uintptr_t a = (uintptr_t)x;
  if (din % 4 == 0 && a % 16 == 0) return 4;
  if (din % 2 == 0 && a % 8 == 0) return 2;
  return 1;
}

template <int NO>
size_t lds_bytes() { return (size_t)MlpLds<NO>::END * sizeof(float); }

template <int NOA, int KT1C>
int launch_hybrid_t(const FwdTask& ta, const coop::CoopTask& ck, int nba, int nbc, const uint8_t* mask, uint32_t slo,
                    uint32_t shi, uint32_t step, uint32_t row_offset, int greedy, const StepOut& so, hipStream_t s) {
  const coop::CoopLds Lc = coop::make_coop_layout<1>(KT1C);
  size_t lb = (size_t)Lc.end * sizeof(float);
  if (lds_bytes<NOA>() > lb) lb = lds_bytes<NOA>();
  MAVA_ARG_CHECK(lb <= 163840, 8, "policy hybrid: %zu bytes of LDS needed exceed the 160 KiB of a CU", lb);
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)policy_hybrid_kernel<NOA, KT1C>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    attr_set = true;
  }
  hipLaunchKernelGGL((policy_hybrid_kernel<NOA, KT1C>), dim3(nba + nbc), dim3(256), lb, s, ta, ck, Lc, nba, mask, slo,
                     shi, step, row_offset, greedy, so);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

// returns 1 when the shape is not instantiated (caller falls back to the per-wave kernel)
int launch_hybrid(const FwdTask& ta, const coop::CoopTask& ck, int nba, int nbc, int n_actions, const uint8_t* mask,
                  uint32_t slo, uint32_t shi, uint32_t step, uint32_t row_offset, int greedy, const StepOut& so,
                  hipStream_t s) {
  const int kt = (ck.din + 31) / 32;
#define HY(NOA, KT) return launch_hybrid_t<NOA, KT>(ta, ck, nba, nbc, mask, slo, shi, step, row_offset, greedy, so, s)
#define HYK(NOA)                                   \
  switch (kt) {                                    \
    case 1: HY(NOA, 1);                            \
    case 2: HY(NOA, 2);                            \
    case 3: HY(NOA, 3);                            \
    case 4: HY(NOA, 4);                            \
    case 5: case 6: HY(NOA, 6);                    \
    case 7: case 8: case 9: HY(NOA, 9);            \
    default: return 1;                             \
  }
  if (n_actions <= 8) { HYK(8) }
  if (n_actions <= 16) { HYK(16) }
  HYK(32)
#undef HYK
#undef HY
}

}  // namespace

// block-cooperative kernels (mlp_coop.hip)
int mava_coop_actor(const float* params, int din, int n_actions, const float* agents_view, const uint8_t* mask,
                    int rows, uint64_t seed, uint32_t step, const uint32_t* step_base, uint32_t row_offset, int greedy,
                    const int32_t* forced_action, int32_t* action, float* log_prob, float* logits, hipStream_t s);
int mava_coop_value(const float* params, int din, const float* x, int x_share, int rows, int vbroadcast, float* value,
                    hipStream_t s);
int mava_coop_raw(const float* params, int din, int n_out, const float* x, int x_share, int rows, float* out,
                  hipStream_t s);

// 0 (default): per-wave register-resident kernel, one launch for actor + critic - as a HYBRID launch (cooperative
// critic blocks, policy_hybrid_kernel) when the critic has at most 128 tiles; 1: per-wave kernel always;
// 2: block-cooperative kernels, one launch per network (they pay a per-launch W2 staging cost that two 32-row
// tiles per block cannot amortise: 49.6 vs 43.9 us per step at 16384 rows per network on MI355X)
// (the variant is a field of the context handle: mava_ctx_set(ctx, MAVA_CTX_POLICY_VARIANT, v); NULL handle = 0)

extern "C" int mava_mlp_param_count(int din, int n_out) { return mlp_param_count(din, n_out); }

extern "C" int mava_mlp_forward_f32(const mava_ctx* ctx, const float* params, int din, int n_out, const float* x,
                                    int x_share, int rows, float* out, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && n_out >= 1 && n_out <= 32, 0,
                 "mava_mlp_forward_f32: din=%d n_out=%d unsupported (n_out <= 32)", din, n_out);
  MAVA_ARG_CHECK(rows >= 0 && x_share >= 1, 1, "mava_mlp_forward_f32: rows=%d x_share=%d", rows, x_share);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(params && x && out, 2, "mava_mlp_forward_f32: null pointer argument");
  if (mava_ctx_policy_variant(ctx) == 2 && din <= 288) return mava_coop_raw(params, din, n_out, x, x_share, rows, out, s);
  FwdTask tk = {params, x, din, n_out, x_share, pick_xv(x, din), rows};
  const int ntiles = mava_cdiv(rows, 32);
  int blocks = mava_cdiv(ntiles, 4);
  if (blocks > 512) blocks = 512;
#define LAUNCH_FWD(NO)                                                                        \
  do {                                                                                        \
    static bool attr_set = false;                                                             \
    if (!attr_set) {                                                                          \
      MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_forward_kernel<NO>,                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                         (int)lds_bytes<NO>()));                              \
      attr_set = true;                                                                        \
    }                                                                                         \
    hipLaunchKernelGGL(mlp_forward_kernel<NO>, dim3(blocks), dim3(256), lds_bytes<NO>(), s, tk, \
                       out);                                                                  \
  } while (0)
  if (n_out == 1) LAUNCH_FWD(1);
  else if (n_out <= 8) LAUNCH_FWD(8);
  else if (n_out <= 16) LAUNCH_FWD(16);
  else LAUNCH_FWD(32);
#undef LAUNCH_FWD
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

// Both acting entry points; action_f != nullptr selects the continuous head (action / forced_action unused then).
static int policy_step_impl(int variant, const float* actor_params, int actor_din, int n_actions,
                            const float* agents_view, const uint8_t* action_mask,
                            const float* critic_params, int critic_din,
                            const float* critic_input, int critic_share, int critic_rows,
                            int value_broadcast, int rows, uint64_t seed, uint32_t step, const uint32_t* step_base,
                            uint32_t row_offset, int greedy, const int32_t* forced_action,
                            int32_t* action, float* log_prob, float* value, float* logits,
                            float* action_f, const float* forced_action_f, float min_scale, hipStream_t s) {
  MAVA_ARG_CHECK(actor_din >= 1 && critic_din >= 1 && n_actions >= 1 && n_actions <= 32, 0,
                 "mava_policy_step_f32: actor_din=%d critic_din=%d n_actions=%d unsupported",
                 actor_din, critic_din, n_actions);
  MAVA_ARG_CHECK(rows >= 0 && critic_rows >= 0 && critic_share >= 1 && value_broadcast >= 1, 1,
                 "mava_policy_step_f32: bad row counts");
  if (rows == 0 && critic_rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(rows == 0 || (actor_params && agents_view && (action || action_f) && log_prob), 2,
                 "mava_policy_step_f32: null actor pointer argument");
  MAVA_ARG_CHECK(critic_rows == 0 || (critic_params && critic_input && value), 2,
                 "mava_policy_step_f32: null critic pointer argument");
  if (variant == 2 && actor_din <= 288 && critic_din <= 288 && action_f == nullptr) {
    int rc = MAVA_OK;
    if (rows > 0)
      rc = mava_coop_actor(actor_params, actor_din, n_actions, agents_view, action_mask, rows, seed, step, step_base, row_offset,
                           greedy, forced_action, action, log_prob, logits, s);
    if (rc != MAVA_OK) return rc;
    if (critic_rows > 0)
      rc = mava_coop_value(critic_params, critic_din, critic_input, critic_share, critic_rows, value_broadcast, value, s);
    return rc;
  }
  FwdTask ta = {actor_params, agents_view, actor_din, n_actions, 1, pick_xv(agents_view, actor_din), rows};
  const uint32_t slo = (uint32_t)seed, shi = (uint32_t)(seed >> 32);
  StepOut so = {action, log_prob, value, logits, forced_action, step_base, action_f,
                actor_params ? actor_params + mlp_param_count(actor_din, n_actions) : nullptr, forced_action_f, min_scale};
  // Few critic tiles (at most one per CU next to the actor's blocks): hybrid launch, cooperative critic blocks
  {
    const int tiles_c = mava_cdiv(critic_rows, 32);
    int nba_h = mava_cdiv(mava_cdiv(rows, 32), 4);
    if (nba_h > 128) nba_h = 128;
    if (variant == 0 && rows > 0 && critic_rows > 0 && tiles_c <= 128 && critic_din <= 287) {
      coop::CoopTask ck = {};
      ck.params = critic_params; ck.x = critic_input; ck.din = critic_din; ck.no = 1; ck.xshare = critic_share;
      ck.R = critic_rows; ck.value = value; ck.vbroadcast = value_broadcast;
      const int rc = launch_hybrid(ta, ck, nba_h, tiles_c, n_actions, action_mask, slo, shi, step, row_offset, greedy, so, s);
      if (rc != 1) return rc;  // 1 = shape not instantiated: fall through to the per-wave launch
    }
  }
  FwdTask tc = {critic_params, critic_input, critic_din, 1, critic_share,
                pick_xv(critic_input, critic_din), critic_rows};
  int nba = mava_cdiv(mava_cdiv(rows, 32), 4);
  int nbc = mava_cdiv(mava_cdiv(critic_rows, 32), 4);
  if (nba > 256) nba = 256;
  if (nbc > 256) nbc = 256;
  if (nba < 1 && rows > 0) nba = 1;
  if (nbc < 1 && critic_rows > 0) nbc = 1;
#define LAUNCH_STEP(NO)                                                                         \
  do {                                                                                          \
    const size_t lb = lds_bytes<NO>() > lds_bytes<1>() ? lds_bytes<NO>() : lds_bytes<1>();      \
    static bool attr_set = false; /* once per instantiation: the call costs host time on every launch */ \
    if (!attr_set) {                                                                            \
      MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)policy_step_kernel<NO>,                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb)); \
      attr_set = true;                                                                          \
    }                                                                                           \
    hipLaunchKernelGGL(policy_step_kernel<NO>, dim3(nba + nbc), dim3(256), lb, s, ta, tc, nba,  \
                       action_mask, slo, shi, step, row_offset, value_broadcast, greedy, so);   \
  } while (0)
  if (n_actions <= 8) LAUNCH_STEP(8);
  else if (n_actions <= 16) LAUNCH_STEP(16);
  else LAUNCH_STEP(32);
#undef LAUNCH_STEP
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_policy_step_f32(const mava_ctx* ctx, const float* actor_params, int actor_din, int n_actions,
                                    const float* agents_view, const uint8_t* action_mask,
                                    const float* critic_params, int critic_din,
                                    const float* critic_input, int critic_share, int critic_rows,
                                    int value_broadcast, int rows, uint64_t seed, uint32_t step, const uint32_t* step_base,
                                    uint32_t row_offset, int greedy, const int32_t* forced_action,
                                    int32_t* action, float* log_prob, float* value, float* logits,
                                    hipStream_t s) {
  return policy_step_impl(mava_ctx_policy_variant(ctx), actor_params, actor_din, n_actions, agents_view, action_mask, critic_params, critic_din,
                          critic_input, critic_share, critic_rows, value_broadcast, rows, seed, step, step_base,
                          row_offset, greedy, forced_action, action, log_prob, value, logits, nullptr, nullptr, 0.0f, s);
}

extern "C" int mava_policy_step_continuous_f32(const mava_ctx* ctx, const float* actor_params, int actor_din, int action_dim,
                                               float min_scale,
                                               const float* agents_view, const float* critic_params, int critic_din,
                                               const float* critic_input, int critic_share, int critic_rows,
                                               int value_broadcast, int rows, uint64_t seed, uint32_t step,
                                               const uint32_t* step_base, uint32_t row_offset, int greedy,
                                               const float* forced_action, float* action, float* log_prob,
                                               float* value, float* mean, hipStream_t s) {
  MAVA_ARG_CHECK(rows == 0 || action != nullptr, 2, "mava_policy_step_continuous_f32: null action pointer");
  return policy_step_impl(mava_ctx_policy_variant(ctx), actor_params, actor_din, action_dim, agents_view, nullptr, critic_params, critic_din,
                          critic_input, critic_share, critic_rows, value_broadcast, rows, seed, step, step_base,
                          row_offset, greedy, nullptr, nullptr, log_prob, value, mean, action, forced_action, min_scale, s);
}
