// Block-cooperative forward kernels for the acting side (K1 + K2 + K3): same tiling as the fused
// train kernel (ppo_train.hip) - a 256-thread block walks 32-row tiles, wave w owns feature tile
// [32w, 32w+32) of each layer, x is staged through LDS one tile ahead, W1 streams from L2 through a
// register ring, W2 sits in LDS - so every SIMD of the chip does the same amount of MFMA work and
// W1 is read once per block instead of once per wave.
//
// Reference: mava/systems/ppo/ff_mappo.py:80-85 (actor_apply, critic_apply, sample, log_prob),
// mava/networks.py:116-124, mava/distributions.py:146-165.  One launch per network per env step.
#include "mlp_coop_body.h"

namespace {

using namespace coop;

template <int NO, int KT1, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_coop_kernel(CoopTask tk, CoopLds L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  coop_body<NO, KT1, MODE>(tk, L, lds, (int)blockIdx.x, (int)gridDim.x);
}

template <int NO, int KT1, int MODE>
int launch_coop(const CoopTask& tk, hipStream_t s) {
  const CoopLds L = make_coop_layout<NO>(KT1);
  const size_t lb = (size_t)L.end * sizeof(float);
  MAVA_ARG_CHECK(lb <= 163840, 8, "mlp_coop: %zu bytes of LDS needed exceed the 160 KiB of a CU", lb);
  MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_coop_kernel<NO, KT1, MODE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
  int blocks = mava_cdiv(tk.R, 32);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL((mlp_coop_kernel<NO, KT1, MODE>), dim3(blocks), dim3(256), lb, s, tk, L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, int MODE>
int dispatch_coop(const CoopTask& tk, hipStream_t s) {
  const int kt = (tk.din + 31) / 32;
  switch (kt) {
    case 1: return launch_coop<NO, 1, MODE>(tk, s);
    case 2: return launch_coop<NO, 2, MODE>(tk, s);
    case 3: return launch_coop<NO, 3, MODE>(tk, s);
    case 4: return launch_coop<NO, 4, MODE>(tk, s);
    case 5: case 6: return launch_coop<NO, 6, MODE>(tk, s);
    case 7: case 8: case 9: return launch_coop<NO, 9, MODE>(tk, s);
    default:
      mava_set_error("mlp_coop: input width %d > 288 is not instantiated", tk.din);
      return MAVA_EARG(9);
  }
}

}  // namespace

// Internal entry points used by mava_policy_step_f32 / mava_mlp_forward_f32 (mlp_policy.hip).
int mava_coop_actor(const float* params, int din, int n_actions, const float* agents_view, const uint8_t* mask,
                    int rows, uint64_t seed, uint32_t step, const uint32_t* step_base, uint32_t row_offset, int greedy,
                    const int32_t* forced_action, int32_t* action, float* log_prob, float* logits, hipStream_t s) {
  CoopTask tk = {};
  tk.params = params; tk.x = agents_view; tk.din = din; tk.no = n_actions; tk.xshare = 1; tk.R = rows;
  tk.mask = mask; tk.forced_action = forced_action; tk.seed_lo = (uint32_t)seed; tk.seed_hi = (uint32_t)(seed >> 32);
  tk.step = step; tk.step_base = step_base; tk.row_offset = row_offset; tk.greedy = greedy; tk.action = action; tk.log_prob = log_prob;
  tk.logits = logits;
  if (n_actions <= 8) return dispatch_coop<8, MODE_SAMPLE>(tk, s);
  if (n_actions <= 16) return dispatch_coop<16, MODE_SAMPLE>(tk, s);
  return dispatch_coop<32, MODE_SAMPLE>(tk, s);
}

int mava_coop_value(const float* params, int din, const float* x, int x_share, int rows, int vbroadcast, float* value,
                    hipStream_t s) {
  CoopTask tk = {};
  tk.params = params; tk.x = x; tk.din = din; tk.no = 1; tk.xshare = x_share; tk.R = rows; tk.value = value;
  tk.vbroadcast = vbroadcast;
  return dispatch_coop<1, MODE_VALUE>(tk, s);
}

int mava_coop_raw(const float* params, int din, int n_out, const float* x, int x_share, int rows, float* out,
                  hipStream_t s) {
  CoopTask tk = {};
  tk.params = params; tk.x = x; tk.din = din; tk.no = n_out; tk.xshare = x_share; tk.R = rows; tk.out = out;
  if (n_out == 1) return dispatch_coop<1, MODE_RAW>(tk, s);
  if (n_out <= 8) return dispatch_coop<8, MODE_RAW>(tk, s);
  if (n_out <= 16) return dispatch_coop<16, MODE_RAW>(tk, s);
  return dispatch_coop<32, MODE_RAW>(tk, s);
}
