// Fused rollout: the whole `lax.scan(_env_step, length=T)` of one update-batch replica in ONE launch.
//
// Reference: mava/systems/ppo/ff_mappo.py:76-106 (_env_step under lax.scan: actor apply, critic apply, sample,
// log_prob, vmap(env.step), PPOTransition), :109-110 (bootstrap value of the last observation); networks
// mava/networks.py:39-58,88-124,172-207; masked Categorical mava/distributions.py:146-165; environment semantics as
// synth_rware.hip (wrappers observation.py:41-53, jumanji.py:53-59,128-143, auto_reset_wrapper.py:88-101,
// episode_metrics.py:78-111).
//
// Why one launch.  Environments are independent of each other and the parameters are fixed for the whole rollout, so a
// workgroup that owns a fixed set of environments can run all T steps for them with NO inter-workgroup dependency: no
// grid barrier, no per-step launch (the per-step kernels were latency-bound at ~33 us for 1.3 GFLOP of matrix work),
// and the weights are fetched and split ONCE.  Block b owns EB = 64 / A environments = 64 agent rows (two 32-row MFMA
// tiles of actor work, one tile of critic work when the agents share the critic input) and loops over t:
//   P1  layer 1 of actor (2 tiles) and critic, x from LDS images the env phase of the PREVIOUS step wrote
//   P2  layer 2 + heads (actor: partial logits from the accumulator as MFMA operand; critic: VALU)
//   S   mask, Gumbel-max sample on Philox (same stream as mava_policy_step_f32), log-prob, value -> trajectory slot t
//   E   env.step for the block's environments (same Philox stream as mava_synth_rware_step, bit for bit):
//       next observation -> trajectory slot t+1 (f32, HBM) AND straight into the LDS x images (split f16) - the
//       observations the policy reads never come back from memory
// then one more critic pass for the bootstrap value.  Four barriers per step.
//
// Arithmetic: split-f16 operands, three v_mfma_f32_32x32x16_f16 per product, f32 accumulation (h2_core.h).
// 512-thread workgroups in two roles with disjoint register sets (two waves per SIMD, <= 256 registers each): waves 0-3
// run the ACTOR (their slices of W1 / W2 / head weights stay in registers for the whole rollout: 120), waves 4-7 the
// CRITIC (W1 / W2 slices: 200) - the two networks' latency-bound chains overlap on every SIMD; the sample and env
// phases use all 512 threads.  LDS only holds activations.
#include "h2_core.h"

#ifdef MAVA_STAMPS
// phase stamps (diagnostic build): cycles per phase of wave 0 of each role of block 0, summed over the T steps
#define RSTAMP_DECL unsigned long long rs_prev = __builtin_readcyclecounter(), rs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RSTAMP(i)                                                   \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long rs_now = __builtin_readcyclecounter(); \
    rs_acc[i] += rs_now - rs_prev;                                  \
    rs_prev = rs_now;                                               \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
static __device__ unsigned long long g_rollout_stamps[16];
#else
#define RSTAMP_DECL
#define RSTAMP(i)
#endif

namespace {
using namespace h2;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr uint32_t ENV_STREAM = 0x454E5653u;   // "ENVS" (synth_rware.hip)
constexpr uint32_t POLICY_STREAM = 0x504f4c49u;  // "POLI" (mlp_policy.hip)

struct RolloutArgs {
  const float* pa;  // actor parameters (din_a -> 128 -> 128 -> no)
  const float* pc;  // critic parameters (din_c -> 128 -> 128 -> 1)
  int E, A, O, no, T, time_limit;
  uint32_t pseed_lo, pseed_hi, eseed_lo, eseed_hi;
  uint32_t t0;          // global step counter at the start of the rollout
  uint32_t row_offset;  // global agent-row id of row 0 (policy noise counter)
  uint32_t env_offset;  // global env id of env 0 (env stream counter)
  int reward_mode;
  // env state
  int32_t* step_count;  // (E, A)
  float* run_return; int32_t* run_length; float* ep_return; int32_t* ep_length;  // (E)
  // trajectory
  float* agents_view;      // (T+1, E, A, A+O)
  float* global_state;     // (T+1, E, A*O) or null (decentralised critic)
  uint8_t* action_mask;    // (T+1, E, A, no)
  int32_t* obs_step_count; // (T+1, E, A)
  int32_t* action; float* value; float* reward; float* log_prob; uint8_t* done;  // (T, E, A)
  float* last_val;         // (E, A)
  float* info_return; int32_t* info_length; uint8_t* info_terminal;  // (T, E)
  float* adv; float* tgt;  // (T, E, A) or null: GAE of ff_mappo.py:112-139 in the kernel's tail
  float gamma, lam;
};

struct RolloutLds {  // byte offsets
  int xa, xa_row, xa_plane, xc, xc_row, xc_plane, h1, wl, ypa, ypc, act, mask, small, gum, xlf, end;
};
constexpr int W1_REG_STEPS = 12;  // layer-1 steps of the critic kept in registers; further steps live in LDS as fragments

template <int NO, int S1A, int S1C, bool SHARED>
RolloutLds make_rollout_lds() {
  constexpr int KTA = (S1A + 1) / 2, KTC = (S1C + 1) / 2;
  RolloutLds L;
  L.xa = 0;
  L.xa_row = 2 * 32 * KTA + 16;
  L.xa_plane = 32 * L.xa_row;
  L.xc = L.xa + 2 * 2 * L.xa_plane;  // two actor tiles x (hi, lo)
  L.xc_row = 2 * 32 * KTC + 16;
  // a wide shared critic input (S1C > 12, needs >= 4 agents: <= 16 envs per block) keeps 16 image rows: the LDS it
  // saves holds the critic's layer-1 fragments beyond W1_REG_STEPS
  L.xc_plane = ((S1C > W1_REG_STEPS) ? 16 : 32) * L.xc_row;
  L.h1 = L.xc + (SHARED ? 2 * L.xc_plane : 0);
  L.wl = L.h1 + (SHARED ? 3 : 4) * IMG_BYTES;  // h1 images: actor tile 0, 1, critic tile(s)
  L.ypa = L.wl + ((S1C > W1_REG_STEPS) ? (S1C - W1_REG_STEPS) * 4 * 64 * 32 : 0);
  L.ypc = L.ypa + 2 * 4 * 32 * (NO + 1) * 4;
  L.act = L.ypc + 2 * 4 * 32 * 4;
  L.mask = L.act + 64 * 4;
  L.small = L.mask + 64 * 32;        // f32: b2a[128] b3a[32] b2c[128] b3c[4] w3c[128]
  L.gum = L.small + (128 + 32 + 128 + 4 + 128) * 4;  // f32 [64 rows][NO]: the step's Gumbel noise (see the S phase)
  L.gum = (L.gum + 15) & ~15;
  L.xlf = L.gum + 64 * NO * 4;  // u32: != 0 when some value of the slot-0 observations has a non-zero low f16 term
  L.end = L.xlf + 16;
  L.end = (L.end + 15) & ~15;
  return L;
}

// ROLE 0: actor group (threads 0-255), ROLE 1: critic group (threads 256-511).  Both run the same barrier sequence.
template <int NO, int S1A, int S1C, bool SHARED, int ROLE>
__device__ __forceinline__ void rollout_body(const RolloutArgs& a, const RolloutLds& L, u8* lds) {
  constexpr bool ACT_ROLE = ROLE == 0;
  const int t512 = threadIdx.x;          // thread of the workgroup (sample / env phases)
  const int tid = t512 & 255, lane = tid & 63, w = tid >> 6, h = lane >> 5, r = lane & 31;
  const int A = a.A, O = a.O, no = a.no, W = A + O, E = a.E;
  const int EB = 64 / A;                 // environments of this block
  const int e0 = blockIdx.x * EB;        // first environment
  const int din_a = W, din_c = SHARED ? A * O : W;
  constexpr int NTC = SHARED ? 1 : 2;    // critic tiles
  float* const B2a = reinterpret_cast<float*>(lds + L.small);
  float* const B3a = B2a + 128;
  float* const B2c = B3a + 32;
  float* const B3c = B2c + 128;
  float* const W3c = B3c + 4;
  int* const ACT = reinterpret_cast<int*>(lds + L.act);
  u8* const MASK = lds + L.mask;       // [64 rows][32]
  float* const GUM = reinterpret_cast<float*>(lds + L.gum);  // [64 rows][NO]
  unsigned* const XLF = reinterpret_cast<unsigned*>(lds + L.xlf);
  float* const YPA = reinterpret_cast<float*>(lds + L.ypa);
  float* const YPC = reinterpret_cast<float*>(lds + L.ypc);
  const int xa_row = L.xa_row, xa_plane = L.xa_plane, xc_row = L.xc_row, xc_plane = L.xc_plane;

  // ---------------------------------------------------------------- prologue
  for (int i = t512 * 16; i < L.end; i += 512 * 16) *reinterpret_cast<uint4*>(lds + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const int oW2a = mlp_off_w2(din_a), oW3a = mlp_off_w3(din_a), oW2c = mlp_off_w2(din_c), oW3c = mlp_off_w3(din_c);
  if (ACT_ROLE) {
    if (tid < 128) B2a[tid] = a.pa[mlp_off_b2(din_a) + tid];
    if (tid < no) B3a[tid] = a.pa[mlp_off_b3(din_a, no) + tid];
  } else {
    if (tid < 128) {
      B2c[tid] = a.pc[mlp_off_b2(din_c) + tid];
      W3c[tid] = a.pc[oW3c + tid];
    }
    if (tid == 0) B3c[0] = a.pc[mlp_off_b3(din_c, 1)];
  }
  // this role's weight fragments, split once with error diffusion along the summation index (h2_core.h) and kept in
  // registers for the whole rollout: W1[k = 16s + 8h + e][f = 32w + r] (k == din is b1), W2 likewise, head (actor)
  constexpr int S1R = ACT_ROLE ? S1A : S1C;
  constexpr int S1REG = (S1R > W1_REG_STEPS) ? W1_REG_STEPS : S1R;  // steps whose fragments stay in registers
  constexpr int XC_ROWS = (S1C > W1_REG_STEPS) ? 16 : 32;
  uint4* const WL = reinterpret_cast<uint4*>(lds + L.wl) + 2 * (w * 64 + lane);  // + 512 * (s - S1REG): this lane's fragment
  const float* const PR = ACT_ROLE ? a.pa : a.pc;
  const int din_r = ACT_ROLE ? din_a : din_c, oW2r = ACT_ROLE ? oW2a : oW2c;
  Frag W1r[S1REG], W2r[8], W3h[2];
  {
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int s = 0; s < S1R; ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * h + e;
        v[e] = (k <= din_r) ? PR[k * MLP_H + 32 * w + r] : 0.0f;
      }
      const Frag f = split8_carry(v, c1);
      if (s < S1REG) {
        W1r[s] = f;
      } else {
        WL[512 * (s - S1REG)] = __builtin_bit_cast(uint4, f.hi);
        WL[512 * (s - S1REG) + 1] = __builtin_bit_cast(uint4, f.lo);
      }
      __builtin_amdgcn_sched_barrier(0);  // one step's loads at a time: the hoisted loads of all steps spill otherwise
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = PR[oW2r + (16 * s + 8 * h + e) * MLP_H + 32 * w + r];
      W2r[s] = split8_carry(v, c2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < (ACT_ROLE ? 2 : 0); ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int f = 32 * w + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
        v[e] = (r < no) ? a.pa[oW3a + f * no + r] * W3_SCALE : 0.0f;
      }
      W3h[s] = split8(v);
    }
  }
  // x images of slot 0 (written by the previous rollout or the reset), ones (bias) columns, masks of slot 0
  {
    const long slot0_av = 0;
    uint32_t lo_any = 0;
    for (int i = t512; i < 64 * W; i += 512) {
      const int row = i / W, c = i - row * W;
      const int e = e0 + row / A;
      const float v = (e < E) ? a.agents_view[slot0_av + ((long)e0 * A + row) * W + c] : 0.0f;
      _Float16 x0, x1;
      split1(v, x0, x1);
      u8* p = lds + L.xa + (row >> 5) * 2 * xa_plane + (row & 31) * xa_row + 2 * c;
      *reinterpret_cast<_Float16*>(p) = x0;
      *reinterpret_cast<_Float16*>(p + xa_plane) = x1;
      lo_any |= (uint32_t)__builtin_bit_cast(uint16_t, x1) & 0x7FFFu;
    }
    if (t512 < 64) {
      u8* p = lds + L.xa + (t512 >> 5) * 2 * xa_plane + (t512 & 31) * xa_row + 2 * din_a;
      *reinterpret_cast<_Float16*>(p) = (_Float16)1.0f;
    }
    if (SHARED) {
      const int gw = A * O;
      for (int i = t512; i < EB * gw; i += 512) {
        const int le = i / gw, c = i - le * gw;
        const float v = (e0 + le < E) ? a.global_state[(long)(e0 + le) * gw + c] : 0.0f;
        _Float16 x0, x1;
        split1(v, x0, x1);
        u8* p = lds + L.xc + le * xc_row + 2 * c;
        *reinterpret_cast<_Float16*>(p) = x0;
        *reinterpret_cast<_Float16*>(p + xc_plane) = x1;
        lo_any |= (uint32_t)__builtin_bit_cast(uint16_t, x1) & 0x7FFFu;
      }
      if (t512 < XC_ROWS) *reinterpret_cast<_Float16*>(lds + L.xc + t512 * xc_row + 2 * din_c) = (_Float16)1.0f;
    }
    for (int i = t512; i < 64 * no; i += 512) {
      const int row = i / no, o = i - row * no;
      MASK[row * 32 + o] = (e0 + row / A < E) ? a.action_mask[((long)e0 * A + row) * no + o] : 1;
    }
    if (lo_any != 0) *XLF = 1u;  // (every writer stores the same value; zeroed with the rest of the LDS above)
  }
  __syncthreads();

  // bookkeeping: thread 256 + i (first wave of the critic group) owns agent row i (env e0 + i / A, agent i % A)
  const int bk_le = tid / A, bk_ag = tid - bk_le * A;
  const bool bk_on = !ACT_ROLE && tid < 64 && (e0 + bk_le) < E;
  const long bk_k = (long)(e0 + bk_le) * A + bk_ag;  // (e, a) index
  int sc_reg = bk_on ? a.step_count[bk_k] : 0;
  float rr_reg = 0.0f, er_reg = 0.0f;
  int rl_reg = 0, el_reg = 0;
  if (bk_on && bk_ag == 0) {
    rr_reg = a.run_return[e0 + bk_le]; rl_reg = a.run_length[e0 + bk_le];
    er_reg = a.ep_return[e0 + bk_le]; el_reg = a.ep_length[e0 + bk_le];
  }

  // The env phase below writes only the HIGH plane of the x images (its observations are exact in f16): when the slot-0
  // observations loaded above are exact too, the low plane is zero for the whole rollout, and layer 1 runs two MFMAs per
  // product instead of three without reading that plane (the skipped product is exactly 0: same bits).
  const bool x_lo = __builtin_amdgcn_readfirstlane((int)*XLF) != 0;
  const int rowB = r * IMG_ROW + 16 * h;  // + 32 s: features 16s + 8h .. + 7 of image row r
  const long EA = (long)E * A;
  const uint32_t nch = (O - 2 + 15) / 16 > 0 ? (O - 2 + 15) / 16 : 1;  // bit chunks per raw view

  // one pass of THIS ROLE's network over the x images: layer 1, layer 2, head partials (two barriers inside, the same
  // for both roles; `active` = false keeps only the barriers: the actor group during the bootstrap pass)
  constexpr int NT = ACT_ROLE ? 2 : NTC;  // 32-row tiles of this role
  RSTAMP_DECL
  auto forward = [&](bool active, bool noise, uint32_t noise_step) __attribute__((always_inline)) {
    f32x16 z[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) z[i][q] = 0.0f;
    if (active) {
      // the LDS operands of step s + 1 are read before the products of step s are issued (one step of prefetch: with the
      // reads inside their own step every step exposed an LDS round trip - 4.4 K cycles for 1.6 K of MFMAs in the critic)
      auto x_frag = [&](int s, int t) {
        return (!ACT_ROLE && SHARED) ? read_row_frag(lds + L.xc, xc_plane, (r & (XC_ROWS - 1)) * xc_row + 16 * h + 32 * s)
                                     : read_row_frag(lds + L.xa + t * 2 * xa_plane, xa_plane, r * xa_row + 16 * h + 32 * s);
      };
      auto w_frag = [&](int s) {
        Frag wf;
        wf.hi = __builtin_bit_cast(half8, WL[512 * (s - S1REG)]);
        wf.lo = __builtin_bit_cast(half8, WL[512 * (s - S1REG) + 1]);
        return wf;
      };
      auto x_hi = [&](int s, int t) -> half8 {
        return (!ACT_ROLE && SHARED)
                   ? *reinterpret_cast<const half8*>(lds + L.xc + (r & (XC_ROWS - 1)) * xc_row + 16 * h + 32 * s)
                   : *reinterpret_cast<const half8*>(lds + L.xa + t * 2 * xa_plane + r * xa_row + 16 * h + 32 * s);
      };
      if (x_lo) {
      Frag bn[NT], wn;
#pragma unroll
      for (int t = 0; t < NT; ++t) bn[t] = x_frag(0, t);
      if (S1REG == 0) wn = w_frag(0);
#pragma unroll
      for (int s = 0; s < S1R; ++s) {
        Frag b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bn[t];
        const Frag wf = (s < S1REG) ? W1r[s < S1REG ? s : 0] : wn;
        if (s + 1 < S1R) {
#pragma unroll
          for (int t = 0; t < NT; ++t) bn[t] = x_frag(s + 1, t);
          if (s + 1 >= S1REG) wn = w_frag(s + 1);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) z[t] = mfma3(wf, b[t], z[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {  // observations exact in f16: high plane only, two products
      half8 bn[NT];
      Frag wn;
#pragma unroll
      for (int t = 0; t < NT; ++t) bn[t] = x_hi(0, t);
      if (S1REG == 0) wn = w_frag(0);
#pragma unroll
      for (int s = 0; s < S1R; ++s) {
        half8 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bn[t];
        const Frag wf = (s < S1REG) ? W1r[s < S1REG ? s : 0] : wn;
        if (s + 1 < S1R) {
#pragma unroll
          for (int t = 0; t < NT; ++t) bn[t] = x_hi(s + 1, t);
          if (s + 1 >= S1REG) wn = w_frag(s + 1);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          z[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.lo, b[t], z[t], 0, 0, 0);
          z[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.hi, b[t], z[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) z[i][q] = fmaxf(z[i][q], 0.0f);
        half4 ph[4], pl[4];
        write_image(lds + L.h1 + ((ACT_ROLE ? 0 : 2) + i) * IMG_BYTES, r, 32 * w + 4 * h, z[i], ph, pl);
      }
    }
    if (ACT_ROLE && noise) {
      // The step's Gumbel noise -log(-log u) does not depend on the network: it is drawn HERE, where the actor group used
      // to wait ~750 cycles for the critic group's wider layer 1, instead of inside the sampling phase (two passes of a
      // Philox4x32-10 call and two logarithms per lane were ~1.8 K of that phase's 5.5 K cycles).  One call per
      // (row, four outputs): thread (row = tid / 4, c = tid % 4) draws outputs 4 c .. 4 c + 3; same counters, same values.
      const int grow = tid >> 2, c = tid & 3;
      if (c < NO / 4) {
        const uint32_t gid = a.row_offset + (uint32_t)(e0 * A + grow);
        const Philox4 rnd = philox4x32_10(gid, noise_step, (uint32_t)c, POLICY_STREAM, a.pseed_lo, a.pseed_hi);
        float4 g;
        g.x = -logf(-logf(u01_open(rnd.x)));
        g.y = -logf(-logf(u01_open(rnd.y)));
        g.z = -logf(-logf(u01_open(rnd.z)));
        g.w = -logf(-logf(u01_open(rnd.w)));
        *reinterpret_cast<float4*>(GUM + grow * NO + 4 * c) = g;
      }
    }
    RSTAMP(0);
    __syncthreads();  // barrier 1: h1 images complete
    RSTAMP(1);
    if (active) {
      const float* b2 = ACT_ROLE ? B2a : B2c;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) z[i][q] = b2[32 * w + (q & 3) + 8 * (q >> 2) + 4 * h];
      {
        auto h_frag = [&](int s, int t) {
          return read_row_frag(lds + L.h1 + ((ACT_ROLE ? 0 : 2) + t) * IMG_BYTES, IMG_PLANE, rowB + 32 * s);
        };
        Frag bn[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = h_frag(0, t);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          Frag b[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) b[t] = bn[t];
          if (s + 1 < 8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = h_frag(s + 1, t);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) z[t] = mfma3(W2r[s], b[t], z[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (ACT_ROLE) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          half4 ph[4], pl[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              _Float16 x0, x1;
              split1(fmaxf(z[t][4 * g + e], 0.0f), x0, x1);
              ph[g][e] = x0;
              pl[g][e] = x1;
            }
          f32x16 y;
#pragma unroll
          for (int q = 0; q < 16; ++q) y[q] = 0.0f;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            Frag b;
            b.hi = __builtin_shufflevector(ph[2 * s], ph[2 * s + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            b.lo = __builtin_shufflevector(pl[2 * s], pl[2 * s + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            y = mfma3(W3h[s], b, y);
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
            if ((q & 3) + 8 * (q >> 2) < NO) {
              if (o < NO) YPA[((t * 4 + w) * 32 + r) * (NO + 1) + o] = y[q] * W3_UNSCALE;
            }
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float4 w3g[4];  // this lane's 16 head weights: four aligned groups of four
#pragma unroll
          for (int g = 0; g < 4; ++g) w3g[g] = *reinterpret_cast<const float4*>(W3c + 32 * w + 4 * h + 8 * g);
          float part = 0.0f;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float4 tv = w3g[q >> 2];
            const float w3 = (q & 3) == 0 ? tv.x : (q & 3) == 1 ? tv.y : (q & 3) == 2 ? tv.z : tv.w;
            part = fmaf(fmaxf(z[t][q], 0.0f), w3, part);
          }
          part += __shfl_xor(part, 32, 64);
          if (h == 0) YPC[(t * 4 + w) * 32 + r] = part;
        }
      }
    }
    RSTAMP(2);
    __syncthreads();  // barrier 2: head partials complete
    RSTAMP(3);
  };
  auto value_of = [&](int crow) -> float {  // critic row (env for SHARED, agent row otherwise) of this block
    const float* yp = YPC + (crow >> 5) * 4 * 32 + (crow & 31);
    return (((yp[0] + yp[32]) + yp[64]) + yp[96]) + B3c[0];
  };

  // Trajectory copy of the observation the x images hold (critic group, 256 threads): the block's rows are contiguous in
  // every array, so consecutive lanes store consecutive floats - agents_view (64 rows x W), the global state (EB envs x
  // A * O, the shared critic's image) and the action mask (64 rows x no bytes).  Row / column cursors advance without
  // divisions.
  const int rows_ok = min(64, (E - e0) * A), envs_ok = min(EB, E - e0);  // valid rows / envs of this block
  const int AO = A * O;
  auto write_out = [&](long slot, bool part_gs) {  // part_gs: the global state (actor group, after sampling); else the rest
    if (!part_gs) {
      float* dst = a.agents_view + (slot * EA + (long)e0 * A) * W;
      if ((W & 1) == 0) {  // pairs: 4-byte LDS reads, 8-byte stores (row bases are multiples of 8 bytes)
        const int W2 = W >> 1, d_r = 256 / W2, d_f = 256 - d_r * W2;
        int row = tid / W2, f = tid - row * W2;
        for (int j = tid; j < 64 * W2; j += 256) {
          if (row < rows_ok) {
            const u8* xa = lds + L.xa + (row >> 5) * 2 * xa_plane + (row & 31) * xa_row;
            const half2v v = *reinterpret_cast<const half2v*>(xa + 4 * f);
            reinterpret_cast<float2*>(dst)[j] = make_float2((float)v[0], (float)v[1]);
          }
          row += d_r; f += d_f;
          if (f >= W2) { f -= W2; ++row; }
        }
      } else {
        const int d_r = 256 / W, d_f = 256 - d_r * W;
        int row = tid / W, f = tid - row * W;
        for (int j = tid; j < 64 * W; j += 256) {
          if (row < rows_ok) {
            const u8* xa = lds + L.xa + (row >> 5) * 2 * xa_plane + (row & 31) * xa_row;
            dst[j] = (float)*reinterpret_cast<const _Float16*>(xa + 2 * f);
          }
          row += d_r; f += d_f;
          if (f >= W) { f -= W; ++row; }
        }
      }
    }
    if (part_gs && SHARED && a.global_state != nullptr) {
      float* dst = a.global_state + (slot * E + e0) * (long)AO;
      if ((AO & 3) == 0) {  // quads: 8-byte LDS reads, 16-byte stores
        const int Q = AO >> 2, d_r = 256 / Q, d_f = 256 - d_r * Q;
        int le = tid / Q, k = tid - le * Q;
        for (int j = tid; j < EB * Q; j += 256) {
          if (le < envs_ok) {
            const half4 v = *reinterpret_cast<const half4*>(lds + L.xc + le * xc_row + 8 * k);
            reinterpret_cast<float4*>(dst)[j] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
          }
          le += d_r; k += d_f;
          if (k >= Q) { k -= Q; ++le; }
        }
      } else {
        const int d_r = 256 / AO, d_f = 256 - d_r * AO;
        int le = tid / AO, k = tid - le * AO;
        for (int j = tid; j < EB * AO; j += 256) {
          if (le < envs_ok) dst[j] = (float)*reinterpret_cast<const _Float16*>(lds + L.xc + le * xc_row + 2 * k);
          le += d_r; k += d_f;
          if (k >= AO) { k -= AO; ++le; }
        }
      }
    }
    if (!part_gs) {
      uint8_t* dst = a.action_mask + (slot * EA + (long)e0 * A) * no;
      for (int j = tid; j < rows_ok * no; j += 256) {
        const int row = j / no, o = j - row * no;
        dst[j] = MASK[row * 32 + o];
      }
    }
  };

  for (int t = 0; t < a.T; ++t) {
    const uint32_t step = a.t0 + (uint32_t)t;
    forward(true, true, step);
    // ---------------------------------------------------------------- S: sample, log-prob, value -> slot t
    // (the sample and observation-generation code runs on the actor group: the critic group's registers are full of
    // weights; it writes the values and does the per-agent bookkeeping meanwhile)
    if (ACT_ROLE) {
      constexpr int RPP = 256 / NO;  // rows per pass
#pragma unroll
      for (int pass = 0; pass < 64 / RPP; ++pass) {
        const int row = pass * RPP + tid / NO, o = tid & (NO - 1);
        const float* yp = YPA + (((row >> 5) * 4) * 32 + (row & 31)) * (NO + 1) + o;
        const float y = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) + B3a[o];
        const bool legal = (o < no) && (MASK[row * 32 + o] != 0);
        const float z = legal ? y : -FLT_MAX;  // networks.py:116-120
        auto fmax_op = [](float p, float q) { return fmaxf(p, q); };
        auto add_op = [](float p, float q) { return p + q; };
        const float mx = group_allreduce<NO>(z, fmax_op);
        const float se = group_allreduce<NO>(expf(z - mx), add_op);
        const float logp = z - (mx + logf(se));
        // Gumbel-max: argmax_o z[o] - log(-log(u_o)), first index wins ties (jax.random.categorical; the noise is
        // this library's Philox stream: counter (global row, step, o / 4, "POLI"))
        // (drawn before barrier 1 of this step: see forward())
        float sc = (o < no) ? (z + GUM[row * NO + o]) : -FLT_MAX;
        if (!(sc > -FLT_MAX)) sc = -FLT_MAX;  // padded / fully masked rows: index 0 wins like the sequential scan
        int best = o;
#pragma unroll
        for (int m = 1; m < NO; m <<= 1) {
          const float so = __shfl_xor(sc, m, 64);
          const int bo = __shfl_xor(best, m, 64);
          const bool take = (so > sc) || (so == sc && bo < best);
          sc = take ? so : sc;
          best = take ? bo : best;
        }
        const float lp = group_allreduce<NO>((o == best) ? logp : 0.0f, add_op);
        const int e = e0 + row / A;
        if (o == 0) {
          ACT[row] = best;
          if (e < E) {
            const long k = (long)t * EA + (long)e0 * A + row;
            a.action[k] = best;
            a.log_prob[k] = lp;
          }
        }
      }
    } else {
      if (tid < 64 && (e0 + tid / A) < E) {
        a.value[(long)t * EA + (long)e0 * A + tid] = value_of(SHARED ? tid / A : tid);
      }
      if (t > 0) write_out((long)t, false);  // slot t = what the env phase of step t - 1 produced (slot 0 came from memory)
    }
    if (ACT_ROLE && t > 0) write_out((long)t, true);
    RSTAMP(4);
    __syncthreads();  // barrier 3: actions visible to the env phase; every reader of the x images is done
    RSTAMP(5);
    // ---------------------------------------------------------------- E: env.step -> slot t + 1 (HBM f32 + LDS images)
    {
      const uint32_t tn = step + 1;
      const long slot = (long)(t + 1);
      const uint32_t n_view = (uint32_t)(64 * nch);  // (row, chunk) items of the block
      for (uint32_t i = ACT_ROLE ? (uint32_t)tid : n_view; i < n_view; i += 256) {
        const uint32_t row = i / nch, c = i - row * nch;
        const uint32_t le = row / A, ag = row - le * A, e = e0 + le;
        if ((int)e >= E) continue;
        const uint32_t ent = (a.env_offset + e) * A + ag;
        const Philox4 rv = philox4x32_10(ent, tn, c, ENV_STREAM, a.eseed_lo, a.eseed_hi);
        const uint32_t wds[4] = {rv.x, rv.y, rv.z, rv.w};
        // The new observation goes into the LDS x images only (every value - 0 / 1 bits, the two counters 0..9, the
        // one-hot agent id - is exact in f16, the lo plane stays zero).  Its f32 copy in the trajectory (slot t + 1) is
        // written from the images by the critic group during the NEXT step's sample phase (write_out below): stored
        // from here, one 4-byte piece per lane 64 bytes apart, every store instruction touched 64 cache lines and the
        // env phase was 15.4 K of a step's 28.8 K cycles (phase stamps).
        const uint32_t f0 = 2 + 16 * c;
        const int n = (int)min(16u, (uint32_t)O - f0);
        u8* xa = lds + L.xa + (row >> 5) * 2 * xa_plane + (row & 31) * xa_row;
        u8* xc = lds + L.xc + le * xc_row + 2 * (ag * O);
        // a full chunk on 4-byte aligned targets: eight 4-byte stores per image (pairs of f16 0.0 / 1.0) instead of sixteen
        // 2-byte ones
        const bool pk_a = (n == 16) && (((uint32_t)A + f0) & 1u) == 0;
        const bool pk_c = (n == 16) && ((ag * (uint32_t)O + f0) & 1u) == 0;
        uint32_t pr[8];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          const uint32_t b0 = (((wds[q >> 2] >> (8 * (q & 3))) & 0xFFu) < 51u) ? 0x3C00u : 0u;  // f16 1.0 / 0.0
          const uint32_t b1 = (((wds[(q + 1) >> 2] >> (8 * ((q + 1) & 3))) & 0xFFu) < 51u) ? 0x3C00u : 0u;
          pr[q >> 1] = b0 | (b1 << 16);
        }
        if (pk_a) {
#pragma unroll
          for (int q = 0; q < 8; ++q) *reinterpret_cast<uint32_t*>(xa + 2 * (A + f0) + 4 * q) = pr[q];
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (q < n) *reinterpret_cast<uint16_t*>(xa + 2 * (A + f0 + q)) = (uint16_t)(pr[q >> 1] >> (16 * (q & 1)));
          }
        }
        if (SHARED) {
          if (pk_c) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<uint32_t*>(xc + 2 * f0 + 4 * q) = pr[q];
          } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if (q < n) *reinterpret_cast<uint16_t*>(xc + 2 * (f0 + q)) = (uint16_t)(pr[q >> 1] >> (16 * (q & 1)));
            }
          }
        }
      }
      // the row's grid coordinates and the action mask of the new observation (a second Philox call per row): on the second
      // wave of the critic group, idle in this phase - on the actor group's lanes with c == 0 every one of its waves ran it
      if (!ACT_ROLE && tid >= 64 && tid < 128) {
        const uint32_t row = (uint32_t)(tid - 64);
        const uint32_t le = row / A, ag = row - le * A, e = e0 + le;
        if ((int)e < E) {
          const uint32_t ent = (a.env_offset + e) * A + ag;
          u8* xa = lds + L.xa + (row >> 5) * 2 * xa_plane + (row & 31) * xa_row;
          u8* xc = lds + L.xc + le * xc_row + 2 * (ag * O);
          const Philox4 cm = philox4x32_10(ent, tn, 0xFFFFu, ENV_STREAM, a.eseed_lo, a.eseed_hi);
          const float c0 = (float)(cm.x % 10u), c1 = (float)(cm.y % 10u);
          *reinterpret_cast<_Float16*>(xa + 2 * A) = (_Float16)c0;
          *reinterpret_cast<_Float16*>(xa + 2 * (A + 1)) = (_Float16)c1;
          if (SHARED) {
            *reinterpret_cast<_Float16*>(xc) = (_Float16)c0;
            *reinterpret_cast<_Float16*>(xc + 2) = (_Float16)c1;
          }
          // action mask of the new observation: all legal except action 1 w.p. 51/256
          const bool m1 = !((cm.z & 0xFFu) < 51u);
          for (int o = 0; o < no; ++o) MASK[row * 32 + o] = (o == 1 && no > 1) ? (m1 ? 1 : 0) : 1;
        }
      }
      if (bk_on) {
        const uint32_t env_id = a.env_offset + (uint32_t)(e0 + bk_le);
        const Philox4 ev = philox4x32_10(env_id, tn, 0u, ENV_STREAM ^ 1u, a.eseed_lo, a.eseed_hi);
        float rew = (u01_open(ev.x) < 0.02f) ? 1.0f : 0.0f;
        if (a.reward_mode == 1) {
          int hits = 0;
          for (int a2 = 0; a2 < A; ++a2) {
            const Philox4 pc2 = philox4x32_10(env_id * A + a2, tn - 1u, 0xFFFFu, ENV_STREAM, a.eseed_lo, a.eseed_hi);
            hits += (ACT[bk_le * A + a2] == (int)((pc2.x % 10u) % (uint32_t)no)) ? 1 : 0;
          }
          rew = (float)hits / (float)A;
        }
        const int sc_new = sc_reg + 1;
        const bool term = (sc_new >= a.time_limit) || (u01_open(ev.y) < 0.002f);
        const int sc_obs = term ? 0 : sc_new;
        sc_reg = sc_obs;
        a.obs_step_count[slot * EA + bk_k] = sc_obs;
        a.reward[(long)t * EA + bk_k] = rew;
        a.done[(long)t * EA + bk_k] = term ? 1 : 0;
        if (bk_ag == 0) {
          const float new_ret = rr_reg + rew;
          const int new_len = rl_reg + 1;
          er_reg = term ? new_ret : er_reg;
          el_reg = term ? new_len : el_reg;
          rr_reg = term ? 0.0f : new_ret;
          rl_reg = term ? 0 : new_len;
          const long ie = (long)t * E + e0 + bk_le;
          a.info_return[ie] = er_reg;
          a.info_length[ie] = el_reg;
          a.info_terminal[ie] = term ? 1 : 0;
        }
      }
    }
    RSTAMP(6);
    __syncthreads();  // barrier 4: x images of slot t + 1 complete
    RSTAMP(7);
  }
#ifdef MAVA_STAMPS
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) g_rollout_stamps[(ACT_ROLE ? 0 : 8) + i] = rs_acc[i];
#endif
  // ------------------------------------------------------------------ bootstrap value (ff_mappo.py:109-110)
  write_out((long)a.T, ACT_ROLE);  // the last observation
  forward(!ACT_ROLE, false, 0u);
  if (bk_on) {
    const float lv = value_of(SHARED ? tid / A : tid);
    a.last_val[bk_k] = lv;
    if (a.adv != nullptr) {
      // GAE (ff_mappo.py:117-136) for this thread's (env, agent) column, reading back the reward / value / done it
      // wrote itself during the rollout: delta = r + gamma V' (1 - d) - V; gae = delta + gamma lambda (1 - d) gae.
      // Eight steps' loads go out together, then the sequential f32 recurrence; no separate pass over the trajectory.
      float g = 0.0f, nv = lv;
      const float gl = a.gamma * a.lam;
      for (int tb = a.T; tb > 0; tb -= 8) {
        float rw[8], vv[8];
        uint8_t dd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = tb - 1 - i;
          const long k = (long)(t >= 0 ? t : 0) * EA + bk_k;
          rw[i] = a.reward[k]; vv[i] = a.value[k]; dd[i] = a.done[k];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = tb - 1 - i;
          if (t >= 0) {
            const float nd = dd[i] ? 0.0f : 1.0f;
            const float delta = rw[i] + a.gamma * nv * nd - vv[i];
            g = delta + gl * nd * g;
            const long k = (long)t * EA + bk_k;
            a.adv[k] = g;
            a.tgt[k] = g + vv[i];
            nv = vv[i];
          }
        }
      }
    }
  }
  if (bk_on) {
    a.step_count[bk_k] = sc_reg;
    if (bk_ag == 0) {
      a.run_return[e0 + bk_le] = rr_reg; a.run_length[e0 + bk_le] = rl_reg;
      a.ep_return[e0 + bk_le] = er_reg; a.ep_length[e0 + bk_le] = el_reg;
    }
  }
}

template <int NO, int S1A, int S1C, bool SHARED>
__global__ __launch_bounds__(512, 2) void rollout_h2_kernel(RolloutArgs a, RolloutLds L) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  if (threadIdx.x < 256) {
    rollout_body<NO, S1A, S1C, SHARED, 0>(a, L, lds);
  } else {
    rollout_body<NO, S1A, S1C, SHARED, 1>(a, L, lds);
  }
}

int g_rollout_last_instance = 0;  // diagnostic: which template instance the last launch used (see the export below)

template <int NO, int S1A, int S1C, bool SHARED>
int launch_rollout(const RolloutArgs& a, hipStream_t s) {
  g_rollout_last_instance = NO * 100000 + S1A * 1000 + S1C * 10 + (SHARED ? 1 : 0);
  const RolloutLds L = make_rollout_lds<NO, S1A, S1C, SHARED>();
  MAVA_ARG_CHECK(L.end <= 163840, 8, "mava_rollout_ff_f32: %d bytes of LDS exceed the 160 KiB of a CU", L.end);
  // (more than half a CU's LDS whatever the layout needs - the benchmarked layouts take ~150 KB anyway: two rollouts launched
  // side by side, the replicas of update_batch_size > 1, are never placed on one CU while others idle)
  const int lds_bytes = L.end > 83968 ? L.end : 83968;
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rollout_h2_kernel<NO, S1A, S1C, SHARED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  const int EB = 64 / a.A;
  hipLaunchKernelGGL((rollout_h2_kernel<NO, S1A, S1C, SHARED>), dim3(mava_cdiv(a.E, EB)), dim3(512), lds_bytes, s, a, L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

#ifdef MAVA_STAMPS
// diagnostic builds: copies the 2 x 8 phase sums of the last launch (actor role, critic role) to the host
extern "C" int mava_debug_get_rollout_stamps(unsigned long long* out16) {
  return -(int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rollout_stamps), 16 * sizeof(unsigned long long));
}
#endif

// Diagnostic (not part of include/mava_hip.h, like mava_debug_h2_launches): NO * 100000 + S1A * 1000 + S1C * 10 + SHARED
// of the rollout_h2_kernel<NO, S1A, S1C, SHARED> instance the last mava_rollout_ff_f32 call launched, so that a parity
// test can assert WHICH instantiation it checked (tests/test_gpu_learner.py).
extern "C" int mava_debug_rollout_last_instance(void) { return g_rollout_last_instance; }

extern "C" int mava_rollout_ff_f32(const float* actor_params, int n_actions, const float* critic_params, int critic_shared,
                                   int E, int A, int O, int T, int time_limit, uint64_t policy_seed, uint64_t env_seed,
                                   uint32_t t0, uint32_t row_offset, uint32_t env_offset, int reward_mode,
                                   int32_t* step_count, float* run_return, int32_t* run_length, float* ep_return,
                                   int32_t* ep_length, float* agents_view, float* global_state, uint8_t* action_mask,
                                   int32_t* obs_step_count, int32_t* action, float* value, float* reward, float* log_prob,
                                   uint8_t* done, float* last_val, float* info_return, int32_t* info_length,
                                   uint8_t* info_terminal, float* adv, float* tgt, float gamma, float gae_lambda,
                                   hipStream_t s) {
  MAVA_ARG_CHECK(E >= 1 && A >= 1 && O >= 2 && T >= 1 && n_actions >= 1 && time_limit >= 1, 0,
                 "mava_rollout_ff_f32: bad shape E=%d A=%d O=%d T=%d nA=%d", E, A, O, T, n_actions);
  MAVA_ARG_CHECK(actor_params && critic_params && step_count && run_return && run_length && ep_return && ep_length &&
                     agents_view && action_mask && obs_step_count && action && value && reward && log_prob && done &&
                     last_val && info_return && info_length && info_terminal && (!critic_shared || global_state),
                 1, "mava_rollout_ff_f32: null pointer argument");
  // shapes this kernel instantiates; 1 = not supported, the caller runs the per-step kernels
  if (64 % A != 0 || n_actions > 8) return 1;
  if (critic_shared && A < 2) return 1;
  const int s1a = (A + O + 1 + 15) / 16, s1c = critic_shared ? (A * O + 1 + 15) / 16 : s1a;
  RolloutArgs a = {};
  a.pa = actor_params; a.pc = critic_params; a.E = E; a.A = A; a.O = O; a.no = n_actions; a.T = T; a.time_limit = time_limit;
  a.pseed_lo = (uint32_t)policy_seed; a.pseed_hi = (uint32_t)(policy_seed >> 32);
  a.eseed_lo = (uint32_t)env_seed; a.eseed_hi = (uint32_t)(env_seed >> 32);
  a.t0 = t0; a.row_offset = row_offset; a.env_offset = env_offset; a.reward_mode = reward_mode;
  a.step_count = step_count; a.run_return = run_return; a.run_length = run_length; a.ep_return = ep_return;
  a.ep_length = ep_length; a.agents_view = agents_view; a.global_state = critic_shared ? global_state : nullptr;
  a.action_mask = action_mask; a.obs_step_count = obs_step_count; a.action = action; a.value = value; a.reward = reward;
  a.log_prob = log_prob; a.done = done; a.last_val = last_val; a.info_return = info_return; a.info_length = info_length;
  a.info_terminal = info_terminal;
  MAVA_ARG_CHECK((adv == nullptr) == (tgt == nullptr), 1, "mava_rollout_ff_f32: adv and tgt go together");
  a.adv = adv; a.tgt = tgt; a.gamma = gamma; a.lam = gae_lambda;
  if (critic_shared) {
    if (s1a == 5 && s1c == 17 && A >= 4) return launch_rollout<8, 5, 17, true>(a, s);  // RWARE tiny / small-4ag (O 66, A 4)
#ifndef MAVA_FAST_BUILD
    if (s1a <= 2 && s1c <= 2) return launch_rollout<8, 2, 2, true>(a, s);     // small test shapes
    if (s1a <= 2 && s1c <= 4) return launch_rollout<8, 2, 4, true>(a, s);
    if (s1a <= 5 && s1c <= 9) return launch_rollout<8, 5, 9, true>(a, s);     // tiny-2ag (O = 66, A = 2)
#endif
    return 1;
  }
  if (s1a == 5) return launch_rollout<8, 5, 5, false>(a, s);
#ifndef MAVA_FAST_BUILD
  if (s1a <= 2) return launch_rollout<8, 2, 2, false>(a, s);
#endif
  return 1;
}
