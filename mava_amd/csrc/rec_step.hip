// Fused recurrent ACTING step (rec_ippo / rec_mappo rollout): one launch per environment step runs, for both
// networks, pre_torso -> GRU cell (reset on done) -> post_torso -> head, then masks / samples / scores the action
// (actor) or writes the value (critic).
//
// Reference: mava/systems/ppo/rec_mappo.py:108-129 (_env_step: policy and critic applied to a length-1 sequence),
// mava/networks.py:238-331 (ScannedRNN: h = where(reset, 0, h); flax GRUCell; RecurrentActor / RecurrentValueNet),
// mava/distributions.py:146-165.  The layer-wise kernels (rec_dense.hip, rec_gru.hip) stay the training path; used
// for acting they are ~10 launches per step whose blocks each re-load a register-resident weight slice for two row
// tiles - latency, not arithmetic.
//
// MI355X mapping: a 256-thread block walks 32-row tiles; wave w owns features [32w, 32w+32) of every 128-wide layer
// and of each GRU gate.  All products are transposed exact-f32 MFMAs (out^T[feature][row] = W^T in^T); weights stream
// from L2 one 8-k-step batch ahead of the MFMAs that use them (no launch-long preload), activations cross waves
// through [feature][row] LDS tiles (stride 33).  Blocks [0, nblk_actor) serve the actor, the rest the critic - e.g.
// one critic sequence per ENV when the agents share the critic input - so both networks share one launch.
#include "mlp_core.h"
#include "rec_step_task.h"
#include "tanh_normal.h"

namespace {

constexpr int LDT = 33;
constexpr int G3 = 3 * MLP_H;

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// acc[g] += W[:, cols_g]^T . B^T over nb batches of 16 inputs.  W(k, c) = wbase[k * ldw + c]; lane (i = j, half h)
// supplies W(k + h, col_g + j).  B is an LDS tile: element (k, row j) at bt[k * bk + j * bj] (a [k][row] exchange tile:
// bk = 33, bj = 1; the row-major x tile: bk = 1, bj = ldx).  Operands run one batch ahead of the MFMAs (ping-pong).
template <int NG>
__device__ __forceinline__ void stream_mm(f32x16 (&acc)[NG], const float* __restrict__ wbase, int ldw, const int (&col)[NG],
                                          const float* bt, int bk, int bj, int nb, int h, int j) {
  const float* wp[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wp[g] = wbase + (long)h * ldw + col[g] + j;
  const float* bp = bt + h * bk + j * bj;
  // two static operand slots (ping-pong): batch b + 1 is requested while batch b multiplies.  (A third slot - two
  // batches of lead - was measured slower: 1.66 vs 1.73 M env-steps/s, it pushes the 16-output instantiation into
  // scratch.)
  float a0[NG][8], a1[NG][8], b0[8], b1[8];
  auto load = [&](int b, float (&a)[NG][8], float (&bb)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int g = 0; g < NG; ++g) a[g][s] = wp[g][(long)(16 * b + 2 * s) * ldw];
      bb[s] = bp[(16 * b + 2 * s) * bk];
    }
  };
  auto mul = [&](const float (&a)[NG][8], const float (&bb)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = MFMA32(a[g][s], bb[s], acc[g]);
  };
  load(0, a0, b0);
  // a rolled loop with fences: unrolled and free to reorder, the compiler hoists every batch's loads to the top
  // (the trip count is a constant 8 for the 128-wide layers) and spills hundreds of registers
#pragma unroll 1
  for (int b = 0; b < nb; b += 2) {
    if (b + 1 < nb) load(b + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mul(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (b + 2 < nb) load(b + 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (b + 1 < nb) mul(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NO, bool ACTOR>
__device__ __forceinline__ void rec_step_body(const RecNet& nt, const RecStepOut& out, float* lds, int bid, int nblk) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int srow = tid >> 3, l8 = tid & 7;
  const int din = nt.din, no = nt.no, ldx = nt.ldx;
  float* const XS = lds + nt.xs;
  float* const ET = lds + nt.et;
  float* const HT = lds + nt.ht;
  float* const H2T = lds + nt.h2t;
  float* const PT = lds + nt.pt;
  float* const W3s = lds + nt.w3;
  float* const YP = lds + nt.yp;
  const float* const Wpre = nt.params;
  const float* const bpre = Wpre + (long)din * MLP_H;
  const float* const Wi = bpre + MLP_H;
  const float* const bi = Wi + MLP_H * G3;
  const float* const Wh = bi + G3;
  const float* const bhn = Wh + MLP_H * G3;
  const float* const Wpost = bhn + MLP_H;
  const float* const bpost = Wpost + MLP_H * MLP_H;
  const float* const Whead = bpost + MLP_H;
  const float* const bhead = Whead + MLP_H * no;

  for (int i = tid; i < 32 * ldx; i += 256) XS[i] = 0.0f;  // padding columns [din, 16*nb1) stay zero
  for (int i = tid; i < MLP_H * NO; i += 256) {
    const int f = i / NO, o = i - f * NO;
    W3s[i] = (o < no) ? Whead[f * no + o] : 0.0f;
  }
  const int fb = 32 * w + 4 * h;
  const int ntiles = nt.rows / 32;
  __syncthreads();

  for (int it = bid; it < ntiles; it += nblk) {
    // ---- stage the x tile (row-major gather) and the masked hidden state
    {
      const int row = it * 32 + srow;
      const float* xrow = nt.x + (long)(row / nt.xshare) * din;
      for (int c = l8; c < din; c += 8) XS[srow * ldx + c] = xrow[c];
    }
    float hp[16];
    {
      const bool rs = nt.done[(long)(it * 32 + j) * nt.done_stride] != 0;  // networks.py:253-257
      const float* hin = nt.h_in + ((long)it * MLP_H + fb) * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = hin[((r & 3) + 8 * (r >> 2)) * 32];
        hp[r] = rs ? 0.0f : v;
        HT[(fb + (r & 3) + 8 * (r >> 2)) * LDT + j] = hp[r];
      }
    }
    __syncthreads();

    // ---- pre_torso: e = relu(x Wpre + bpre)
    {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = bpre[fb + (r & 3) + 8 * (r >> 2)];
      const int col[1] = {32 * w};
      stream_mm<1>(acc, Wpre, MLP_H, col, XS, 1, ldx, nt.nb1, h, j);
#pragma unroll
      for (int r = 0; r < 16; ++r) ET[(fb + (r & 3) + 8 * (r >> 2)) * LDT + j] = fmaxf(acc[0][r], 0.0f);
    }
    __syncthreads();

    // ---- GRU cell (flax GRUCell): r = s(W_ir e + b_ir + W_hr h), z likewise, n = tanh(W_in e + b_in + r (W_hn h + b_hn))
    float hn[16];
    {
      // accumulators {W_in e + b_in, r, z, W_hn h + b_hn}: the embedding product updates the first three, the
      // hidden-state product the last three (two contiguous windows of one array, no copies)
      f32x16 ga[4];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fb + (r & 3) + 8 * (r >> 2);
        ga[0][r] = bi[2 * MLP_H + f];
        ga[1][r] = bi[f];
        ga[2][r] = bi[MLP_H + f];
        ga[3][r] = bhn[f];
      }
      const int col_e[3] = {2 * MLP_H + 32 * w, 32 * w, MLP_H + 32 * w};
      const int col_h[3] = {32 * w, MLP_H + 32 * w, 2 * MLP_H + 32 * w};
      stream_mm<3>(*reinterpret_cast<f32x16(*)[3]>(&ga[0]), Wi, G3, col_e, ET, LDT, 1, 8, h, j);
      stream_mm<3>(*reinterpret_cast<f32x16(*)[3]>(&ga[1]), Wh, G3, col_h, HT, LDT, 1, 8, h, j);
      float* hout = nt.h_out + ((long)it * MLP_H + fb) * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rr = sigm(ga[1][r]);
        const float zz = sigm(ga[2][r]);
        const float nn = tanh_(ga[0][r] + rr * ga[3][r]);
        hn[r] = (1.0f - zz) * nn + zz * hp[r];
        hout[((r & 3) + 8 * (r >> 2)) * 32] = hn[r];
        H2T[(fb + (r & 3) + 8 * (r >> 2)) * LDT + j] = hn[r];
      }
    }
    __syncthreads();

    // ---- post_torso + partial head
    {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = bpost[fb + (r & 3) + 8 * (r >> 2)];
      const int col[1] = {32 * w};
      stream_mm<1>(acc, Wpost, MLP_H, col, H2T, LDT, 1, 8, h, j);
      float part[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) part[o] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fmaxf(acc[0][r], 0.0f);
        const float* w3 = W3s + (fb + (r & 3) + 8 * (r >> 2)) * NO;
#pragma unroll
        for (int o = 0; o < NO; ++o) part[o] = fmaf(pv, w3[o], part[o]);
      }
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const float v = part[o] + __shfl_xor(part[o], 32, 64);
        if (h == 0) YP[(w * NO + o) * 32 + j] = v;
      }
    }
    (void)PT;
    __syncthreads();

    // ---- epilogue: wave 0, one lane per row
    if (w == 0 && h == 0) {
      const int row = it * 32 + j;
      float y[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o)
        y[o] = (((YP[(0 * NO + o) * 32 + j] + YP[(1 * NO + o) * 32 + j]) + YP[(2 * NO + o) * 32 + j]) +
                YP[(3 * NO + o) * 32 + j]) + ((o < no) ? bhead[o] : 0.0f);
      if (!ACTOR) {
        for (int b = 0; b < out.vbroadcast; ++b) out.value[(long)row * out.vbroadcast + b] = y[0];
      } else if (out.action_f != nullptr) {
        // ContinuousActionHead (networks.py:127-169): same noise stream as mava_seq_sample_continuous_f32
        const float* const log_std = bhead + no;
        const uint32_t gid = out.row_offset + (uint32_t)row;
        float lp = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          if (o < no) {
            const float sc = tn::scale_of(log_std[o], out.min_scale);
            const float eps = out.greedy ? 0.0f : tn::noise(gid, out.step, o, tn::STREAM_SAMPLE, out.seed_lo, out.seed_hi);
            const float a = tanhf(fmaf(sc, eps, y[o]));
            lp += tn::log_prob(a, y[o], sc).lp;
            out.action_f[(long)row * no + o] = a;
          }
        }
        out.log_prob[row] = lp;
      } else {
        Categorical<NO> cat;
        cat.build(y, out.mask != nullptr ? (out.mask + (long)row * no) : nullptr, no);
        int a = 0;
        if (out.greedy) {
          float best = -FLT_MAX;
#pragma unroll
          for (int o = 0; o < NO; ++o)
            if (o < no && cat.z[o] > best) { best = cat.z[o]; a = o; }
        } else {
          // Gumbel-max: argmax_o z[o] - log(-log(u_o)), first index wins ties (same stream as mava_seq_sample_f32)
          float best = -FLT_MAX;
          const uint32_t gid = out.row_offset + (uint32_t)row;
#pragma unroll
          for (int c = 0; c < (NO + 3) / 4; ++c) {
            Philox4 rnd = philox4x32_10(gid, out.step, (uint32_t)c, 0x504f4c49u /*"POLI"*/, out.seed_lo, out.seed_hi);
            const uint32_t wds[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int o = 4 * c + q;
              if (o < NO && o < no) {
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
        for (int o = 0; o < NO; ++o)
          if (o == a) lp = cat.logp[o];
        out.action[row] = a;
        out.log_prob[row] = lp;
      }
    }
    __syncthreads();  // every LDS tile is free for the next row tile
  }
}

template <int NOA>
__global__ __launch_bounds__(256, 1) void rec_step_kernel(RecNet actor, RecNet critic, int nblk_actor, RecStepOut out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < nblk_actor) rec_step_body<NOA, true>(actor, out, lds, (int)blockIdx.x, nblk_actor);
  else rec_step_body<1, false>(critic, out, lds, (int)blockIdx.x - nblk_actor, (int)gridDim.x - nblk_actor);
}

void carve(RecNet& n, int no_pad) {
  n.nb1 = (n.din + 15) / 16;
  n.ldx = 16 * n.nb1 + 1;
  n.xs = 0;
  n.et = n.xs + 32 * n.ldx;
  n.ht = n.et + MLP_H * LDT;
  n.h2t = n.ht + MLP_H * LDT;
  n.pt = n.h2t + MLP_H * LDT;
  n.w3 = n.pt;  // (the post activations never leave registers: no PT tile)
  n.yp = n.w3 + MLP_H * no_pad;
  n.end = n.yp + 4 * no_pad * 32;
}

}  // namespace

// Both entry points; action_f != nullptr selects the continuous head (action / action_mask unused then).
static int rec_step_impl(const float* actor_params, int actor_din, int n_actions, const float* agents_view,
                         const uint8_t* action_mask, const uint8_t* done_a, const float* h_actor_in,
                         float* h_actor_out, int rows_a, uint64_t seed, uint32_t step, uint32_t row_offset,
                         int greedy, int32_t* action, float* action_f, float* log_prob, const float* critic_params,
                         int critic_din, const float* critic_input, int critic_share, const uint8_t* done_c,
                         int done_c_stride, const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                         float* value, hipStream_t s, float min_scale = 0.0f, const void* pack_a = nullptr, const void* pack_c = nullptr) {
  MAVA_ARG_CHECK(rows_a >= 0 && rows_c >= 0 && rows_a % 32 == 0 && rows_c % 32 == 0, 0,
                 "mava_rec_step_f32: rows must be multiples of 32 (rows_a=%d rows_c=%d)", rows_a, rows_c);
  if (rows_a == 0 && rows_c == 0) return MAVA_OK;
  MAVA_ARG_CHECK(rows_a == 0 || (actor_din >= 1 && n_actions >= 1 && n_actions <= 32 && actor_params && agents_view &&
                                 done_a && h_actor_in && h_actor_out && (action || action_f) && log_prob),
                 1, "mava_rec_step_f32: bad actor arguments");
  MAVA_ARG_CHECK(rows_c == 0 || (critic_din >= 1 && critic_share >= 1 && value_broadcast >= 1 && done_c_stride >= 1 && critic_params &&
                                 critic_input && done_c && h_critic_in && h_critic_out && value),
                 2, "mava_rec_step_f32: bad critic arguments");
  MAVA_ARG_CHECK(h_actor_in != h_actor_out && h_critic_in != h_critic_out, 3,
                 "mava_rec_step_f32: the hidden state cannot be updated in place (other tiles' lanes still read it)");
  const int noa = n_actions <= 8 ? 8 : (n_actions <= 16 ? 16 : 32);
  RecNet a = {}, c = {};
  a.params = actor_params; a.x = agents_view; a.done = done_a; a.h_in = h_actor_in; a.h_out = h_actor_out;
  a.din = rows_a ? actor_din : 1; a.no = n_actions; a.xshare = 1; a.rows = rows_a; a.done_stride = 1;
  c.params = critic_params; c.x = critic_input; c.done = done_c; c.h_in = h_critic_in; c.h_out = h_critic_out;
  c.din = rows_c ? critic_din : 1; c.no = 1; c.xshare = critic_share; c.rows = rows_c; c.done_stride = done_c_stride;
  carve(a, noa);
  carve(c, 1);
  const size_t lb = (size_t)(a.end > c.end ? a.end : c.end) * sizeof(float);
  MAVA_ARG_CHECK(lb <= 163840, 4, "mava_rec_step_f32: %zu bytes of LDS needed (input widths %d / %d) exceed 160 KiB", lb,
                 actor_din, critic_din);
  RecStepOut so = {};
  so.mask = action_mask; so.seed_lo = (uint32_t)seed; so.seed_hi = (uint32_t)(seed >> 32); so.step = step;
  so.row_offset = row_offset; so.greedy = greedy; so.action = action; so.log_prob = log_prob; so.value = value;
  so.vbroadcast = value_broadcast;
  so.action_f = action_f;
  so.min_scale = min_scale;
  if (pack_a != nullptr && pack_c != nullptr && rows_a > 0 && rows_c > 0) {  // split-f16 operands, pre-packed weights
    const int rc = mava_rec_step_h2_launch(a, c, pack_a, pack_c, so, s);
    if (rc <= 0) return rc;
  }
  const int ta = rows_a / 32, tc = rows_c / 32;
  // one block per CU; the CUs are shared out in proportion to the tiles of the two networks
  int nba = ta, nbc = tc;
  if (ta + tc > 256) {
    nbc = tc > 0 ? (int)((256L * tc + (ta + tc) - 1) / (ta + tc)) : 0;
    if (nbc > tc) nbc = tc;
    if (tc > 0 && nbc < 1) nbc = 1;
    nba = ta > 0 ? 256 - nbc : 0;
    if (nba > ta) nba = ta;
    if (ta > 0 && nba < 1) nba = 1;
  }
#define LAUNCH(NOA)                                                                                              \
  do {                                                                                                           \
    static bool attr_set = false; /* once: the call costs host time on every launch */                           \
    if (!attr_set) {                                                                                             \
      MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_step_kernel<NOA>,                                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 163840));                   \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL(rec_step_kernel<NOA>, dim3(nba + nbc), dim3(256), lb, s, a, c, nba, so);                  \
  } while (0)
  if (noa == 8) LAUNCH(8);
  else if (noa == 16) LAUNCH(16);
  else LAUNCH(32);
#undef LAUNCH
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_rec_step_f32(const float* actor_params, int actor_din, int n_actions, const float* agents_view,
                                 const uint8_t* action_mask, const uint8_t* done_a, const float* h_actor_in,
                                 float* h_actor_out, int rows_a, uint64_t seed, uint32_t step, uint32_t row_offset,
                                 int greedy, int32_t* action, float* log_prob, const float* critic_params,
                                 int critic_din, const float* critic_input, int critic_share, const uint8_t* done_c,
                                 int done_c_stride, const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                                 float* value, hipStream_t s) {
  return rec_step_impl(actor_params, actor_din, n_actions, agents_view, action_mask, done_a, h_actor_in, h_actor_out, rows_a,
                       seed, step, row_offset, greedy, action, nullptr, log_prob, critic_params, critic_din, critic_input,
                       critic_share, done_c, done_c_stride, h_critic_in, h_critic_out, rows_c, value_broadcast, value, s);
}

extern "C" int mava_rec_step_continuous_f32(const float* actor_params, int actor_din, int action_dim, float min_scale,
                                            const float* agents_view, const uint8_t* done_a, const float* h_actor_in,
                                            float* h_actor_out, int rows_a, uint64_t seed, uint32_t step,
                                            uint32_t row_offset, int greedy, float* action, float* log_prob,
                                            const float* critic_params, int critic_din, const float* critic_input,
                                            int critic_share, const uint8_t* done_c, int done_c_stride,
                                            const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                                            float* value, hipStream_t s) {
  MAVA_ARG_CHECK(action_dim <= 16 && (rows_a == 0 || action != nullptr), 1,
                 "mava_rec_step_continuous_f32: action_dim <= 16 and a non-null action buffer are required");
  return rec_step_impl(actor_params, actor_din, action_dim, agents_view, nullptr, done_a, h_actor_in, h_actor_out, rows_a,
                       seed, step, row_offset, greedy, nullptr, action, log_prob, critic_params, critic_din, critic_input,
                       critic_share, done_c, done_c_stride, h_critic_in, h_critic_out, rows_c, value_broadcast, value, s, min_scale);
}

// The same acting step on split-f16 operands (rec_step_h2.hip): pack_a / pack_c are the two networks' weights as written
// by mava_rec_step_pack_f32 from the CURRENT parameters (mava_rec_step_pack_bytes(din) bytes each; re-pack after every
// parameter update).  Shapes the f16x2 kernel does not instantiate (more than 16 outputs) run the exact-f32 kernel.
extern "C" int mava_rec_step_packed_f32(const void* pack_a, const void* pack_c, const float* actor_params, int actor_din,
                                        int n_actions, float min_scale, const float* agents_view, const uint8_t* action_mask,
                                        const uint8_t* done_a, const float* h_actor_in, float* h_actor_out, int rows_a,
                                        uint64_t seed, uint32_t step, uint32_t row_offset, int greedy, int32_t* action,
                                        float* action_f, float* log_prob, const float* critic_params, int critic_din,
                                        const float* critic_input, int critic_share, const uint8_t* done_c, int done_c_stride,
                                        const float* h_critic_in, float* h_critic_out, int rows_c, int value_broadcast,
                                        float* value, hipStream_t s) {
  MAVA_ARG_CHECK(pack_a && pack_c, 5, "mava_rec_step_packed_f32: null weight pack");
  MAVA_ARG_CHECK(action_f == nullptr || n_actions <= 16, 1, "mava_rec_step_packed_f32: action_dim <= 16 for the continuous head");
  return rec_step_impl(actor_params, actor_din, n_actions, agents_view, action_f ? nullptr : action_mask, done_a, h_actor_in,
                       h_actor_out, rows_a, seed, step, row_offset, greedy, action_f ? nullptr : action, action_f, log_prob,
                       critic_params, critic_din, critic_input, critic_share, done_c, done_c_stride, h_critic_in, h_critic_out,
                       rows_c, value_broadcast, value, s, min_scale, pack_a, pack_c);
}
