// Fused PPO minibatch kernels (K7 + K8 + K9 of SURVEY §2.1): row gather by permutation,
// network forward, clipped-PPO / entropy / clipped-value loss, and the full backward pass with
// weight-gradient accumulation, in ONE launch per network per minibatch.
//
// Reference: mava/systems/ppo/ff_mappo.py:150-180 (_actor_loss_fn), :182-201 (_critic_loss_fn),
// :204-218 (value_and_grad), :268-285 (shuffle: permutation over T*E, take, reshape into
// minibatches; mava/utils/jax_utils.py:33-49).  No shuffled copy of the trajectory is made:
// the permutation slice is an index vector and rows are gathered on load.
//
// MI355X mapping: persistent 256-thread blocks, ONE per CU (4 waves = one per SIMD), walking
// 32-row tiles of the gathered minibatch.  All products run TRANSPOSED on the exact-f32 MFMA
// (32x32x2): out^T[feature][row] = W^T . in^T, so the batch row sits on the lane and wave w owns
// feature tile [32w, 32w+32) of every layer:
//   P1  z1_w  = W1[:, w]^T x^T                 A: W1 from L2 through a register ring (resident up to 96 inputs),
//                                              B: the gathered x tile in LDS (16-byte rows, ds_read_b128); the tile's
//                                              ones column at index din multiplies "row din" of W1 = b1
//   P2  z2_w  = b2 + W2[:, w]^T h1^T           A: W2 in LDS, B: h1^T in LDS; partial head logits as 16 more MFMAs
//                                              (B = the layer-2 accumulator, A = W3 words in registers)
//   P3  loss + dlogits: critic per lane (one agent of the row per (wave, half) when the agents of a (t,e) row share
//       the input and are aggregated); actor once per row, lane-parallel (NO lanes per row, DPP reductions), dy
//       through an LDS tile; dz2_w = (W3 dy) * relu'(z2_w) (MFMA for the actor)
//   P4  dh1_w = W2[w, :] dz2^T                 A: W2 in LDS (row walk, odd stride), B: dz2^T in LDS;
//       gW3^T += dy h2^T (MFMA, actor); the critic's dW3 and every bias gradient are per-lane register partials
//   P5  gW2[:, w] += h1^T . dz2 ; gW1[:, w] += x^T . dz1   (A: h1^T / the x tile in LDS; row din of gW1 = db1)
// The weight-gradient accumulators (KT1*16 + 64 (+16) AGPRs per lane) stay resident for the whole
// launch; activations cross waves through three or four 16.5 KB LDS tiles ([feature][row], stride 33 =>
// conflict-free for both the B-operand row walk and the A-operand feature walk).  Each block
// writes ONE partial-gradient slab; mava_slab_reduce_f32 sums slabs in a fixed order, so the
// gradient is bitwise reproducible (no float atomics).
#include "mlp_core.h"
#include "ppo_train_task.h"
#include "ctx.h"
#include "tanh_normal.h"

namespace {

constexpr int STATS_BLOCKS = 128;
constexpr int LDT = 33;  // row stride of the [feature][32 rows] exchange tiles


#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                   \
  do {                                                             \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                                 \
    st_prev = st_now;                                              \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// LDS carve (floats) computed on the host per launch (din / NO dependent), checked against 160 KiB.
struct TrainLdsLayout {
  int h1t, h2t, dz2t, dz1t, yp, dy, xs, misc, end;  // dz1t == h2t: the dz1^T tile aliases h2^T (extra barrier D')
  int ldx;  // row stride of the staged x tile: 32*KT1 + 4 floats (16-byte rows, an odd number of 16-byte slots =>
            // conflict-free ds_read_b128 row walks in P1 and ds_write_b128 staging)
};

template <int NO>
TrainLdsLayout make_layout(int kt1) {
  TrainLdsLayout L;
  const int tile = MLP_H * LDT;
  L.h1t = MlpLds<NO>::END;
  L.h2t = L.h1t + tile;   // h2^T, later dz1^T (after the P4 sweeps, behind an extra barrier)
  L.dz2t = L.h2t + tile;
  L.yp = L.dz2t + tile;   // [4 waves][32 rows][NO + 1] partial logits
  L.dy = L.yp + 4 * 32 * (NO + 1);
  L.ldx = 32 * kt1 + 4;
  // dy: [32 outputs (rows >= NO stay zero)][33] - an MFMA operand both ways (k = output for dz2, k = row for dW3)
  L.xs = (L.dy + 32 * LDT + 3) & ~3;  // [32][ldx] gathered x tile, zero padded to 32*KT1 columns, 16-byte aligned
  L.misc = L.xs + 32 * L.ldx;  // 16-byte aligned (xs and 32*ldx are)
  L.end = L.misc + 16;
  // dz1^T gets its own tile when the CU's 160 KiB allow (narrow inputs): one barrier less per row tile
  L.dz1t = L.h2t;
  if ((size_t)(L.end + tile) * sizeof(float) <= 163840) {
    L.dz1t = L.end;
    L.end += tile;
  }
  return L;
}

// advantage statistics of one minibatch: partial (sum, sumsq) in f64 per block.  One thread per (t,e) index b (its A
// agent values are contiguous), four indices in flight per thread: the index gathers go out first, then the values
// that depend on them - not one dependent idx -> value chain per element with an integer division in front.
__global__ __launch_bounds__(256) void adv_stats_kernel(const float* __restrict__ adv,
                                                        const int32_t* __restrict__ idx,
                                                        long idx_base, int Rb, int A,
                                                        double* __restrict__ partials, long idx_stride) {
  __shared__ double sh[2][4];
  // blockIdx.y = minibatch of a batched launch (mava_adv_stats_batched_f64): its index slice and its partials
  if (idx) idx += (long)blockIdx.y * idx_stride;
  partials += (long)blockIdx.y * 2 * gridDim.x;
  double s1 = 0.0, s2 = 0.0;
  const long stride = (long)gridDim.x * 256;
  for (long b0 = (long)blockIdx.x * 256 + threadIdx.x; b0 < Rb; b0 += 4 * stride) {
    long p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long b = b0 + u * stride;
      const long bc = b < Rb ? b : b0;
      p[u] = idx ? (long)idx[bc] : idx_base + bc;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (b0 + u * stride < Rb) {
        const float* row = adv + p[u] * A;
        for (int a = 0; a < A; ++a) {
          const double v = (double)row[a];
          s1 += v;
          s2 += v * v;
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_down(s1, o, 64);
    s2 += __shfl_down(s2, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s1; sh[1][w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
    partials[2 * blockIdx.x + 1] = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  }
}

// Register-staged copy of one gathered 32-row x tile: 8 threads per row, thread l8 takes the V-float pieces
// l8, l8+8, ... of its row (V = 4 when the rows are 16-byte aligned: 128 bytes per row per instruction; else 1).
// Issued a whole tile ahead of its use (global -> registers now, registers -> LDS after the consumers of the
// previous tile have passed their last barrier).  Branch-free and single-path on purpose: a piece past the end
// of the row (nv = pieces per row) is loaded from the start of the row instead and stored to a dummy LDS slot,
// so there is no exec masking, and no merge of differently-allocated registers that would force the compiler to
// wait for the loads right after issuing them.
template <int V, int NR>
__device__ __forceinline__ void stage_load(const float* __restrict__ xrow, int nv, int l8, float (&xr)[NR]) {
#pragma unroll
  for (int i = 0; i < NR / V; ++i) {
    const int c = l8 + 8 * i;
    const int cc = (c < nv) ? c : 0;
    if (V == 4) {
      const float4 v = reinterpret_cast<const float4*>(xrow)[cc];
      xr[4 * i + 0] = v.x; xr[4 * i + 1] = v.y; xr[4 * i + 2] = v.z; xr[4 * i + 3] = v.w;
    } else {
      xr[i] = xrow[cc];
    }
  }
}
template <int V, int NR>
__device__ __forceinline__ void stage_write(float* xs_row, float* dummy, int nv, int l8, const float (&xr)[NR]) {
#pragma unroll
  for (int i = 0; i < NR / V; ++i) {
    const int c = l8 + 8 * i;
    float* q = (c < nv) ? (xs_row + V * c) : dummy;
    if (V == 4) {
      *reinterpret_cast<float4*>(q) = make_float4(xr[4 * i + 0], xr[4 * i + 1], xr[4 * i + 2], xr[4 * i + 3]);
    } else {
      *q = xr[i];
    }
  }
}

// All-reduce over aligned groups of G consecutive lanes (G = 8, 16, 32) on the VALU's DPP path: quad xor 1,
// quad xor 2, then mirrored halves (lane i <-> G-1-i pairs the two already-reduced halves) - no LDS round trips
// up to 16 lanes; the last step of a 32-lane group crosses DPP rows and uses a lane permute.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G, typename Op>
__device__ __forceinline__ float group_allreduce(float v, Op op) {
  v = op(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
  if (G >= 8) v = op(v, dpp_f<0x141>(v));   // row_half_mirror
  if (G >= 16) v = op(v, dpp_f<0x140>(v));  // row_mirror
  if (G >= 32) v = op(v, __shfl_xor(v, 16, 64));
  return v;
}

template <int NO, int KT1, bool ACTOR, int XV, bool CONT>
__global__ __launch_bounds__(256, 1) void ppo_train_kernel(TrainTask tk, TrainLdsLayout L) {
  static_assert(!CONT || ACTOR, "the continuous head belongs to the actor");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const W2s = lds + MlpLds<NO>::W2;
  float* const W3s = lds + MlpLds<NO>::W3;
  float* const H1T = lds + L.h1t;
  float* const H2T = lds + L.h2t;
  float* const DZ1T = lds + L.dz1t;  // own tile, or an alias of h2^T behind barrier D'
  float* const DZ2T = lds + L.dz2t;
  float* const YP = lds + L.yp;
  float* const DY = lds + L.dy;
  float* const XS = lds + L.xs;
  float* const misc = lds + L.misc;
  constexpr int ldx = 32 * KT1 + 4;  // == L.ldx

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int srow = tid >> 3, l8 = tid & 7;  // staging role: row srow of the tile, 8 threads per row
  const int din = tk.din, no = tk.no;
  // rows the network is evaluated on: agent rows, or (t,e) rows when the agents of a row are aggregated (the
  // .mean() of the reference still runs over all Rb*A agent rows)
  const long R = (!ACTOR && tk.agg > 1) ? (long)tk.Rb : (long)tk.Rb * tk.A;
  const float invR = 1.0f / (float)((long)tk.Rb * tk.A);
  constexpr int NR = 4 * KT1;  // staged floats per thread (32*KT1 columns / 8 threads)

  mlp_fill_lds<NO>(lds, tk.params, din, no, 256);
  // Padding columns stay zero for the whole launch except column din, which holds 1.0: "row din" of W1 in the
  // flat parameter vector IS b1, so layer 1 adds its bias and P5 accumulates db1 (row din of gW1) for free.
  for (int i = tid; i < 32 * ldx; i += 256) XS[i] = 0.0f;
  for (int i = tid; i < 32 * LDT; i += 256) DY[i] = 0.0f;  // output rows >= NO of the dy tile stay zero
  __syncthreads();
  if (tid < 32) XS[tid * ldx + din] = 1.0f;
  if (ACTOR && tid == 0) {
    // ff_mappo.py:164  gae = (gae - gae.mean()) / (gae.std() + 1e-8)   (population std)
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < STATS_BLOCKS; ++i) { s1 += tk.stats[2 * i]; s2 += tk.stats[2 * i + 1]; }
    const double mean = s1 / (double)R;
    double var = s2 / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    misc[0] = (float)mean;
    misc[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
  }
  __syncthreads();
  const float adv_mean = ACTOR ? misc[0] : 0.0f;
  const float adv_rstd = ACTOR ? misc[1] : 0.0f;

  const float* const wcol1 = tk.params + 32 * w + j;  // W1[k][32w + j] = wcol1[k*128]
  const int fbase = 32 * w + 4 * h;                   // + (r&3) + 8*(r>>2)

  // persistent MFMA accumulators: wave w owns output columns [32w, 32w+32) of dW1 and dW2
  f32x16 gW1[KT1], gW2[4];
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) gW1[t][r] = 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) gW2[t][r] = 0.0f;
  // per-thread accumulators of the small gradients
  f32x16 gW3;  // actor: gW3^T[o][32w + j] (MFMA accumulator, rows o < NO meaningful)
#pragma unroll
  for (int r = 0; r < 16; ++r) gW3[r] = 0.0f;
  // A operands of the head product logits^T[o][row] = sum_f W3[f][o] h2^T[f][row] over the wave's 32 features:
  // k-step r pairs the features of accumulator register r in both lane halves, so B is h2[r] straight from the
  // layer-2 accumulator.  Actor: lane i = output o (zero for o >= NO).  Critic: every lane carries W3[f] (all
  // result rows equal; row 0 is read), which is also the factor of dz2 = W3[f] * dy.
  float w3h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = fbase + (r & 3) + 8 * (r >> 2);
    w3h[r] = ACTOR ? ((j < NO) ? W3s[f * NO + j] : 0.0f) : W3s[f];
  }
  // actor: this lane's A operands of the dz2 product, W3[32w + j][2s + h]
  float w3a[ACTOR ? NO / 2 : 1];
#pragma unroll
  for (int s2 = 0; s2 < (ACTOR ? NO / 2 : 1); ++s2)
    w3a[s2] = ACTOR ? W3s[(32 * w + j) * NO + 2 * s2 + h] : 0.0f;
  float ab3 = 0.0f;   // per-lane partial of db3 (actor: output lo over this lane's rows; critic: wave 0, half 0)
  // per-lane partial sums over the rows this lane has seen (lane = row j of every tile), reduced across the 32
  // row lanes once in the epilogue: db2 for the wave's 16 (r, h) features and, for the one-output critic, dW3
  float ab2[16], aW3r[ACTOR ? 1 : 16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ab2[r] = 0.0f;
#pragma unroll
  for (int r = 0; r < (ACTOR ? 1 : 16); ++r) aW3r[r] = 0.0f;
  float loss_a = 0.f, loss_b = 0.f;  // wave 0, half 0 lanes: actor (pg, entropy) / critic (value loss)
  // continuous head: scale of this lane's action dimension, and the per-lane partial of d loss / d scale[lo]
  float als = 0.0f;
  const float ls_raw = CONT ? tk.params[mlp_param_count(tk.din, tk.no) + ((lane & (NO - 1)) < tk.no ? (lane & (NO - 1)) : 0)] : 0.0f;
  const float sc_lo = CONT ? tn::scale_of(ls_raw, tk.min_scale) : 1.0f;


  // per-row inputs of the loss (row j of the tile), prefetched one tile ahead
  // NOTHING in the prefetch path may do arithmetic on a value it has just loaded: the compiler places the
  // vmcnt wait at the first use, and vmcnt retires in order, so one early use exposes the whole HBM gather.
  // Actor: the loss of a tile is computed ONCE, lane-parallel: wave w owns rows [8w, 8w+8), NO lanes per row
  // (lane = (row, output)), NP passes of 64/NO rows.  Critic: one output, every lane computes row j itself.
  constexpr int NP = ACTOR ? NO / 8 : 1;       // passes
  constexpr int GR = ACTOR ? 64 / NO : 32;     // rows per pass (actor)
  const int lo = ACTOR ? (lane & (NO - 1)) : 0;  // this lane's output index (actor)
  auto loss_row = [&](int q) -> int { return ACTOR ? (8 * w + q * GR + lane / NO) : j; };
  // aggregated critic: the 8 (wave, half) copies of loss row j each take one agent of that row
  const int slot = 2 * w + h;
  const int slot_c = (!ACTOR && tk.agg > 1) ? (slot < tk.agg ? slot : 0) : 0;
  // per-row loss inputs (plain scalars, no aggregates: they must stay in registers)
  //   act: action ; f0, f1: actor old_logp, advantage / critic old_value, target ; m: raw mask byte of output lo
  auto load_row = [&](long fr, int& act, float& f0, float& f1, uint32_t& m) {
    if (ACTOR && CONT) {
      // component lo of the row's action vector (bit pattern, untouched here); m carries the row number, the
      // counter of the entropy noise
      act = __builtin_bit_cast(int, tk.action_f[fr * no + (lo < no ? lo : 0)]);
      f0 = tk.old_logp[fr];
      f1 = tk.adv[fr];
      m = (uint32_t)fr;
    } else if (ACTOR) {
      act = tk.action[fr];
      f0 = tk.old_logp[fr];
      f1 = tk.adv[fr];
      const uint8_t* mk = (tk.mask != nullptr && lo < no) ? (tk.mask + fr * no + lo) : nullptr;
      m = 1u;
      if (mk != nullptr) m = *mk;
    } else {
      act = 0;
      const long fa = (tk.agg > 1) ? fr * tk.agg + slot_c : fr;  // agent slot_c of (t,e) row fr
      f0 = tk.old_value[fa];
      f1 = tk.targets[fa];
      m = 0u;
    }
  };
  // Row cursors: agent-row q = 32*tile + r of the minibatch is (b, a) = (q / A, q % A), index idx[b], trajectory
  // row idx[b]*A + a.  A thread's q advances by the same 32*gridDim.x every tile, so (b, a) are advanced
  // incrementally - no integer division inside the tile loop.  Rows past the end clamp to row R-1.
  const uint32_t Au = (!ACTOR && tk.agg > 1) ? 1u : (uint32_t)tk.A;

  const uint32_t q_step = 32u * gridDim.x, b_step = q_step / Au, a_step = q_step % Au;
  const uint32_t b_last = (uint32_t)(R - 1) / Au, a_last = (uint32_t)(R - 1) % Au;
  struct Cursor { uint32_t q, b, a; };
  auto cursor_at = [&](int r) {
    Cursor c;
    c.q = 32u * blockIdx.x + (uint32_t)r;
    c.b = c.q / Au;
    c.a = c.q - c.b * Au;
    return c;
  };
  auto cursor_advance = [&](Cursor& c) {
    c.q += q_step; c.b += b_step; c.a += a_step;
    if (c.a >= Au) { c.a -= Au; c.b += 1u; }
  };
  // raw index gather at the cursor (the loaded value is NOT touched here) + the agent that goes with it
  auto cursor_gather = [&](const Cursor& c, int32_t& p_raw, uint32_t& a_out) {
    const bool in = c.q < (uint32_t)R;
    const uint32_t b = in ? c.b : b_last;
    a_out = in ? c.a : a_last;
    p_raw = tk.idx ? tk.idx[b] : (int32_t)(tk.idx_base + (long)b);
  };
  // input row of trajectory row fr = p*A + a is fr / xshare: fr itself (per-agent inputs) or p (one input row
  // per (t, env), shared by its A agents); 32-bit: TE*A < 2^31 is checked on the host
  auto stage_row = [&](int32_t p_raw, uint32_t a) -> uint32_t {
    const uint32_t fr = (uint32_t)p_raw * Au + a;
    return (tk.xshare == 1) ? fr : ((uint32_t)tk.xshare == Au ? (uint32_t)p_raw : fr / (uint32_t)tk.xshare);
  };
  float* const xs_dummy = misc + 4;  // 16 bytes nobody reads
  auto stage_issue = [&](uint32_t xrow_idx, float (&xr)[NR]) {
    stage_load<XV, NR>(tk.x + (long)xrow_idx * din, din / XV, l8, xr);
  };
  auto stage_commit = [&](const float (&xr)[NR]) {
    stage_write<XV, NR>(XS + srow * ldx, xs_dummy, din / XV, l8, xr);
  };

  // W1 operand ring depth (batches of 8 k-steps); must divide the batch count 2*KT1.  Up to 96 inputs the whole
  // slice of W1 a lane ever needs (<= 48 words) stays in registers for the launch: no refills at all.
  constexpr int RD = (2 * KT1 <= 6) ? (2 * KT1) : (((2 * KT1) % 6 == 0) ? 6 : 4);
  constexpr bool W1_RESIDENT = (RD == 2 * KT1);
  float wr[RD][8];
  // k order inside a batch of 16 inputs: MFMA step s multiplies k = 16b + s (lane half 0) and k = 16b + 8 + s
  // (half 1), so a lane's eight x operands of a batch are 32 contiguous bytes of its row (two ds_read_b128)
  const float* const wcol1h = wcol1 + 8 * h * MLP_H;  // W1[k + 8h][32w + j] = wcol1h[k * 128]
#pragma unroll
  for (int d = 0; d < RD; ++d)
#pragma unroll
    for (int s = 0; s < 8; ++s) wr[d][s] = wcol1h[(16 * d + s) * MLP_H];

  const long ntiles = (R + 31) / 32;
  long it = blockIdx.x;
  float xr[NR];
  int r_act[NP] = {}, n_act[NP] = {};
  float r_f0[NP] = {}, r_f1[NP] = {}, n_f0[NP] = {}, n_f1[NP] = {};
  uint32_t r_m[NP] = {}, n_m[NP] = {};
  // The index gather runs one tile ahead of the loads that depend on it, two tiles ahead of the compute:
  // cs / cl[] are the gather cursors (staging row srow / loss rows), p*_next, a*_next the raw idx value and agent
  // of the NEXT tile.
  Cursor cs = cursor_at(srow), cl[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) cl[q] = cursor_at(loss_row(q));
  int32_t ps_next = 0, pl_next[NP] = {};
  uint32_t as_next = 0, al_next[NP] = {};
  if (it < ntiles) {
    cursor_gather(cs, ps_next, as_next);
    stage_issue(stage_row(ps_next, as_next), xr);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      cursor_gather(cl[q], pl_next[q], al_next[q]);
      load_row((long)((uint32_t)pl_next[q] * Au + al_next[q]), r_act[q], r_f0[q], r_f1[q], r_m[q]);
    }
    stage_commit(xr);
    cursor_advance(cs);
#pragma unroll
    for (int q = 0; q < NP; ++q) cursor_advance(cl[q]);
    if (it + gridDim.x < ntiles) {
      cursor_gather(cs, ps_next, as_next);
#pragma unroll
      for (int q = 0; q < NP; ++q) cursor_gather(cl[q], pl_next[q], al_next[q]);
      cursor_advance(cs);
#pragma unroll
      for (int q = 0; q < NP; ++q) cursor_advance(cl[q]);
    }
  }
  __syncthreads();

  STAMP_DECL
  for (; it < ntiles; it += gridDim.x) {
    STAMP(9);
    const bool valid = (it * 32 + j) < R;
    const long itn = it + gridDim.x;
    const bool have_next = itn < ntiles;
    // Addresses of the next tile's rows, from the idx values gathered one tile ago.  Computed HERE, where every
    // outstanding load is a tile old: the first use of a loaded value waits for ALL younger loads too (vmcnt is
    // in order), so after barrier A it would wait for the W1 refills P1 has just issued.
    // (row numbers, not pointers, are pinned: a pointer through an asm loses its global address space)
    uint32_t xrow_next = stage_row(ps_next, as_next);
    uint32_t fr_next[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) fr_next[q] = (uint32_t)pl_next[q] * Au + al_next[q];
    asm volatile("" : "+v"(xrow_next));
#pragma unroll
    for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(fr_next[q]));

    // ---------------------------------------------------------------- P1: layer 1, tile w
    f32x16 h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = 0.0f;  // b1 enters through the ones column of the x tile
    {
      // W1 operands stream from L2 through a shifting ring of RD batches of 8 k-steps (prefetch distance
      // RD*512 MFMA cycles).  The refill index wraps, so when the loop ends the ring already holds
      // batches 0..RD-1 for the next row tile.  x operands come from the staged LDS tile one batch ahead.
      const float* xb = XS + j * ldx + 8 * h;  // x[row j][k + 8h]
      constexpr int NB = 2 * KT1;          // batches of 8 k-steps (16 inputs)
      static_assert(NB % RD == 0, "ring depth must divide the batch count");
      // Ring slot d holds batch g*RD + d (W1 operands from L2) and xo[d] its x operands (LDS).  Slots are
      // static inside a group of RD batches - no register copies next to the MFMAs - and groups form a
      // runtime loop.  A consumed slot is refilled with batch b+RD, wrapping to the head batches of the NEXT
      // row tile, so the ring is already primed when the next tile starts.
      // Rows k >= din of "W1" are the bias / W2 words that follow it in the flat parameter vector: finite
      // values that meet the zero padding of the x tile, so no clamp and no per-load address math.
      float xo[RD][8];
      {
        const float4 v0 = *reinterpret_cast<const float4*>(xb), v1 = *reinterpret_cast<const float4*>(xb + 4);
        xo[0][0] = v0.x; xo[0][1] = v0.y; xo[0][2] = v0.z; xo[0][3] = v0.w;
        xo[0][4] = v1.x; xo[0][5] = v1.y; xo[0][6] = v1.z; xo[0][7] = v1.w;
      }
#pragma unroll 1
      for (int g = 0; g < NB / RD; ++g) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
          const int b = g * RD + d;
          const int bn = (b + 1 < NB) ? (b + 1) : b;
          {
            const float4 v0 = *reinterpret_cast<const float4*>(xb + 16 * bn);
            const float4 v1 = *reinterpret_cast<const float4*>(xb + 16 * bn + 4);
            float(&xn)[8] = xo[(d + 1) % RD];
            xn[0] = v0.x; xn[1] = v0.y; xn[2] = v0.z; xn[3] = v0.w;
            xn[4] = v1.x; xn[5] = v1.y; xn[6] = v1.z; xn[7] = v1.w;
          }
          // a batch whose 16 inputs all lie past the ones column (k > din) multiplies zeros: skipped (uniform branch;
          // its operand loads still run, the ring stays in step).  din = 70 -> 5 of 6 batches, 264 -> 17 of 18.
          if (16 * b <= din) {
#pragma unroll
            for (int s = 0; s < 8; ++s) h1 = MFMA32(wr[d][s], xo[d][s], h1);
          }
          // refill the slot consumed ONE batch ago (its MFMAs have retired: no write-after-read wait on
          // operands still being read) with batch b-1+RD
          if (!W1_RESIDENT) {
            int bf = b - 1 + RD;
            bf = (bf >= NB) ? (bf - NB) : bf;
            const float* wb = wcol1h + bf * (16 * MLP_H);
#pragma unroll
            for (int s = 0; s < 8; ++s) wr[(d + RD - 1) % RD][s] = wb[s * MLP_H];
            // (measured: spreading these loads one behind each MFMA with sched_group_barrier is 9 % slower than
            // the compiler's clump of eight - left as is)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // trailing refill of the last slot (batch NB-1+RD wraps to head batch RD-1 of the next tile)
      if (!W1_RESIDENT) {
#pragma unroll
        for (int s = 0; s < 8; ++s) wr[RD - 1][s] = wcol1h[(16 * (RD - 1) + s) * MLP_H];
      }
    }
    STAMP(12);
    uint32_t relu1 = 0;  // bit r: z1 of accumulator register r is positive (relu' for P4b, no LDS re-read)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h1[r] = fmaxf(h1[r], 0.0f);
      relu1 |= (h1[r] > 0.0f) ? (1u << r) : 0u;
      H1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = h1[r];
    }
    STAMP(0);
    STAMP(13);
    // Next tile's gathers are issued only now: vmcnt retires in order, so issuing them before P1 would
    // put the HBM gather latency in front of every W1 operand wait of P1.  They do not depend on the h1^T tile, so
    // they go out BEFORE barrier A and the tile's LDS writes drain under them.
    if (have_next) {
      // idx values were gathered one iteration ago (complete by now); only now are they used
      stage_issue(xrow_next, xr);  // global -> registers, committed after barrier E
      STAMP(14);
#pragma unroll
      for (int q = 0; q < NP; ++q) load_row((long)fr_next[q], n_act[q], n_f0[q], n_f1[q], n_m[q]);
      STAMP(15);
      if (itn + gridDim.x < ntiles) {  // raw index gather for the tile after next (first used next iteration)
        cursor_gather(cs, ps_next, as_next);
#pragma unroll
        for (int q = 0; q < NP; ++q) cursor_gather(cl[q], pl_next[q], al_next[q]);
        cursor_advance(cs);
#pragma unroll
        for (int q = 0; q < NP; ++q) cursor_advance(cl[q]);
      }
    }
    __syncthreads();  // A: h1^T complete
    STAMP(10);

    // ---------------------------------------------------------------- P2: layer 2, tile w
    f32x16 h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = lds[MlpLds<NO>::B2 + fbase + (r & 3) + 8 * (r >> 2)];
    {
      const float* wl = W2s + h * MLP_LDW + 32 * w + j;  // W2[k + h][32w + j]
      const float* hb = H1T + h * LDT + j;               // h1^T[k + h][row j]
      float oa[2][8], ob[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { oa[0][s] = wl[(2 * s) * MLP_LDW]; ob[0][s] = hb[(2 * s) * LDT]; }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            oa[(g + 1) & 1][s] = wl[(16 * (g + 1) + 2 * s) * MLP_LDW];
            ob[(g + 1) & 1][s] = hb[(16 * (g + 1) + 2 * s) * LDT];
          }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) h2 = MFMA32(oa[g & 1][s], ob[g & 1][s], h2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(11);
    {
      f32x16 yacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[r] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        h2[r] = fmaxf(h2[r], 0.0f);
        H2T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = h2[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc = MFMA32(w3h[r], h2[r], yacc);
      // partial logits of this wave: register r of lane (row j, half h) is output (r&3) + 8(r>>2) + 4h
      if (ACTOR) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
          if ((r & 3) + 8 * (r >> 2) + 4 < NO || (r & 3) + 8 * (r >> 2) < NO) {
            if (o < NO) YP[(w * 32 + j) * (NO + 1) + o] = yacc[r];
          }
        }
      } else {
        if (h == 0) YP[(w * 32 + j) * (NO + 1)] = yacc[0];
      }
    }
    STAMP(1);
    __syncthreads();  // B

    // ---------------------------------------------------------------- P3: loss, dlogits, dz2
    f32x16 dz;  // dz2 of tile w
    if (ACTOR) {
      const float lo_c = 1.0f - tk.clip_eps, hi_c = 1.0f + tk.clip_eps;
      if (CONT) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int row = loss_row(q);
          const bool rvalid = (it * 32 + row) < R;
          const float* yp = YP + row * (NO + 1) + lo;
          const float mean = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) +
                             lds[MlpLds<NO>::B3 + lo];
          // Independent(TanhTransformed(Normal(mean, scale))) over the NO lanes of the row (networks.py:127-169,
          // distributions.py:24-91); entropy with a fresh reparameterised sample (ff_mappo.py:176-177)
          const bool live = lo < no;
          auto add_op = [](float a, float b) { return a + b; };
          const tn::LogProb lpd = tn::log_prob(__builtin_bit_cast(float, r_act[q]), mean, sc_lo);
          const float eps = tn::noise(tk.row_offset + r_m[q], tk.ent_step, lo, tn::STREAM_ENTROPY, tk.seed_lo, tk.seed_hi);
          const float xsmp = fmaf(sc_lo, eps, mean);
          const float th = tanhf(xsmp);
          const float ent_d = 0.5f + tn::HALF_LOG_2PI + logf(sc_lo) + tn::tanh_fldj(xsmp);
          const float lp = group_allreduce<NO>(live ? lpd.lp : 0.0f, add_op);
          const float ent = group_allreduce<NO>(live ? ent_d : 0.0f, add_op);
          const float gae = (r_f1[q] - adv_mean) * adv_rstd;
          const float ratio = expf(lp - r_f0[q]);
          const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
          const float l1 = ratio * gae, l2 = rc * gae;
          const float pg = -fminf(l1, l2);
          const bool inside = (ratio >= lo_c) && (ratio <= hi_c);
          const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
          const float g2 = inside ? (1.0f - g1) : 0.0f;
          const float dlp = rvalid ? (-(g1 + g2) * gae * ratio * invR) : 0.0f;
          const float ec = rvalid ? (tk.ent_coef * invR) : 0.0f;
          // d fldj / d x = -2 tanh(x); x = mean + scale * eps
          const float dyo = live ? (dlp * lpd.dmean + ec * 2.0f * th) : 0.0f;
          const float dsc = live ? (dlp * lpd.dscale - ec * (1.0f / sc_lo - 2.0f * th * eps)) : 0.0f;
          DY[lo * LDT + row] = dyo;
          ab3 += dyo;
          als += dsc;
          if (rvalid && lo == 0) {
            loss_a += pg * invR;
            loss_b += ent * invR;
          }
        }
      } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int row = loss_row(q);
        const bool rvalid = (it * 32 + row) < R;
        const float* yp = YP + row * (NO + 1) + lo;
        const float y = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) +
                        lds[MlpLds<NO>::B3 + lo];
        // masked Categorical over the NO lanes of the row (networks.py:116-124, distributions.py:146-165)
        const bool legal = (lo < no) && (r_m[q] != 0u);
        const float z = legal ? y : -FLT_MAX;
        auto fmax_op = [](float a, float b) { return fmaxf(a, b); };
        auto add_op = [](float a, float b) { return a + b; };
        const float mx = group_allreduce<NO>(z, fmax_op);
        const float se = group_allreduce<NO>(expf(z - mx), add_op);
        const float logp = z - (mx + logf(se));
        const float pr = expf(logp);
        const float ent = group_allreduce<NO>((pr > 0.0f) ? -(pr * logp) : 0.0f, add_op);
        const int act = r_act[q];
        const float lp = group_allreduce<NO>((lo == act) ? logp : 0.0f, add_op);
        const float gae = (r_f1[q] - adv_mean) * adv_rstd;
        const float ratio = expf(lp - r_f0[q]);
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float l1 = ratio * gae, l2 = rc * gae;
        const float pg = -fminf(l1, l2);
        // d(-min(l1,l2))/d ratio: ties split evenly (lax.min); clip passes gradient inside the range
        const bool inside = (ratio >= lo_c) && (ratio <= hi_c);
        const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        const float dlp = rvalid ? (-(g1 + g2) * gae * ratio * invR) : 0.0f;
        const float ec = rvalid ? (tk.ent_coef * invR) : 0.0f;
        const float oh = (lo == act) ? 1.0f : 0.0f;
        const float pl = (pr > 0.0f) ? logp : 0.0f;
        // -ent_coef * dH/dz_o = +ent_coef * p_o (log p_o + H)
        float dyo = dlp * (oh - pr) + ec * pr * (pl + ent);
        if (z == -FLT_MAX) dyo = 0.0f;
        DY[lo * LDT + row] = dyo;
        ab3 += dyo;
        if (rvalid && lo == 0) {
          loss_a += pg * invR;
          loss_b += ent * invR;
        }
      }
      }
      __syncthreads();  // B2: dy of all 32 rows visible
      // dz2^T[f][row] = sum_o W3[f][o] dy[o][row] on the MFMA (A: this lane's W3 words, kept in registers)
#pragma unroll
      for (int r = 0; r < 16; ++r) dz[r] = 0.0f;
      const float* dyb = DY + h * LDT + j;  // dy[o = 2s + h][row j]
#pragma unroll
      for (int s2 = 0; s2 < NO / 2; ++s2) dz = MFMA32(w3a[s2], dyb[(2 * s2) * LDT], dz);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fbase + (r & 3) + 8 * (r >> 2);
        dz[r] = (h2[r] > 0.0f) ? dz[r] : 0.0f;
        DZ2T[f * LDT + j] = dz[r];
        ab2[r] += dz[r];
      }
    } else {
      const float* yp = YP + j * (NO + 1);
      const float v = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) +
                      lds[MlpLds<NO>::B3];
      const float ov = r_f0[0], tg = r_f1[0];
      const float diff = v - ov;
      const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
      const float e1 = v - tg, e2 = vclip - tg;
      const float l1 = e1 * e1, l2 = e2 * e2;
      const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
      const float g1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
      const float g2 = inside ? (1.0f - g1) : 0.0f;
      float dy0 = valid ? (tk.vf_coef * (g1 * e1 + g2 * e2) * invR) : 0.0f;
      if (tk.agg > 1) {
        // this lane evaluated agent `slot` of row j: the value is shared by the row's agents, so the backward
        // pass runs once on the sum of their loss gradients (d loss / d theta = sum_a dy_a * d v / d theta)
        const bool mine = valid && slot < tk.agg;
        dy0 = mine ? dy0 : 0.0f;
        if (mine) {
          loss_a += 0.5f * fmaxf(l1, l2) * invR;
          ab3 += dy0;
        }
        DY[slot * LDT + j] = dy0;
        __syncthreads();  // B2
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < 8; ++a) sum += DY[a * LDT + j];  // fixed order: identical in every lane
        dy0 = sum;
      } else if (valid && w == 0 && h == 0) {
        loss_a += 0.5f * fmaxf(l1, l2) * invR;
        ab3 += dy0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fbase + (r & 3) + 8 * (r >> 2);
        dz[r] = (h2[r] > 0.0f) ? (w3h[r] * dy0) : 0.0f;
        DZ2T[f * LDT + j] = dz[r];
        ab2[r] += dz[r];
        aW3r[r] = fmaf(h2[r], dy0, aW3r[r]);
      }
    }
    STAMP(2);
    __syncthreads();  // C

    // ---------------------------------------------------------------- P4: dh1 tile w, small grads
    f32x16 d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
    {
      const float* wl = W2s + (32 * w + j) * MLP_LDW + h;  // W2[32w + j][n + h]
      const float* db = DZ2T + h * LDT + j;                // dz2^T[n + h][row j]
      float oa[2][8], ob[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { oa[0][s] = wl[2 * s]; ob[0][s] = db[(2 * s) * LDT]; }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            oa[(g + 1) & 1][s] = wl[16 * (g + 1) + 2 * s];
            ob[(g + 1) & 1][s] = db[(16 * (g + 1) + 2 * s) * LDT];
          }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) d1 = MFMA32(oa[g & 1][s], ob[g & 1][s], d1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ACTOR) {
      // gW3^T[o][f = 32w + j] += sum_rows dy[o][row] * h2^T[f][row]   (rows o >= NO of the dy tile are zero)
      const float* ea = DY + j * LDT + h;                  // dy[o = j][row 2s + h]
      const float* eb = H2T + (32 * w + j) * LDT + h;      // h2^T[32w + j][row 2s + h]
      float oa[16], ob[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) { oa[s2] = ea[2 * s2]; ob[s2] = eb[2 * s2]; }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) gW3 = MFMA32(oa[s2], ob[s2], gW3);
    }
    STAMP(3);
    if (L.dz1t == L.h2t) __syncthreads();  // D': every reader of h2^T is done; the tile is reused for dz1^T
    {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d1[r] = ((relu1 >> r) & 1u) ? d1[r] : 0.0f;
        DZ1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = d1[r];
      }
    }
    STAMP(4);

    // ---------------------------------------------------------------- P5: weight gradients
    // gW2 needs h1^T and dz2^T only, so it runs BEFORE barrier D: the dz1^T writes above drain under its MFMAs
    // instead of in front of the barrier
    {
      // gW2[k_in tile t][n = 32w + j] += sum_rows h1^T[k_in][row] * dz2^T[n][row]
      const float* ea = H1T + j * LDT + h;
      const float* eb = DZ2T + (32 * w + j) * LDT + h;
      float bz[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) bz[s] = eb[2 * s];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int s = 0; s < 16; ++s) gW2[t] = MFMA32(ea[(32 * t) * LDT + 2 * s], bz[s], gW2[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(5);
    __syncthreads();  // D: dz1^T complete
    {
      // gW1[k_in tile t][n] += sum_rows x[row][k_in] * dz1^T[n][row];  A from the staged x tile
      const float* eb = DZ1T + (32 * w + j) * LDT + h;
      const float* xa = XS + h * ldx + j;  // x[row 2s + h][32t + j]
      float bz[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) bz[s] = eb[2 * s];
#pragma unroll
      for (int t = 0; t < KT1; ++t) {
#pragma unroll
        for (int s = 0; s < 16; ++s) gW1[t] = MFMA32(xa[(2 * s) * ldx + 32 * t], bz[s], gW1[t]);
        __builtin_amdgcn_sched_barrier(0);  // keep the next tile's LDS reads from piling up in VGPRs
      }
    }
    STAMP(6);
    __syncthreads();  // E: every exchange tile and the x tile are free
    STAMP(7);
    if (have_next) stage_commit(xr);
#pragma unroll
    for (int q = 0; q < NP; ++q) { r_act[q] = n_act[q]; r_f0[q] = n_f0[q]; r_f1[q] = n_f1[q]; r_m[q] = n_m[q]; }
    __syncthreads();  // F: next x tile visible
    STAMP(8);
  }
#ifdef MAVA_STAMPS
  if (tk.stamps != nullptr && blockIdx.x == 0 && lane == 0) {
    for (int i = 0; i < 16; ++i) tk.stamps[w * 16 + i] = st_acc[i];
  }
#endif

  // ------------------------------------------------------------------ epilogue: one slab per block
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const int oB1 = mlp_off_b1(din), oW2 = mlp_off_w2(din), oB2 = mlp_off_b2(din), oW3 = mlp_off_w3(din),
            oB3 = mlp_off_b3(din, no);
  const int Pm = mlp_param_count(din, no);
  const int P = CONT ? Pm + no : Pm;  // the loss sums follow the parameters (continuous head: MLP, then the raw scales)
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = mlp_feat(t, r, h);
      if (k <= din) slab[k * MLP_H + 32 * w + j] = gW1[t][r];  // row din = db1 (oB1 == din * 128)
    }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[oW2 + mlp_feat(t, r, h) * MLP_H + 32 * w + j] = gW2[t][r];
  // per-lane partials -> sums over the 32 row lanes of each half (fixed xor tree: reproducible)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = ab2[r];
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if (j == 0) slab[oB2 + fbase + (r & 3) + 8 * (r >> 2)] = v;
  }
  float* red = lds + L.yp;  // epilogue scratch (the tile loop is over)
  if (ACTOR) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (o < no) slab[oW3 + (32 * w + j) * no + o] = gW3[r];
    }
    // db3[o]: sum over the lanes of output o (xor tree over the row bits), then over the 4 waves
    float v = ab3;
#pragma unroll
    for (int m = NO; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane < NO) red[w * NO + lane] = v;
    for (int o = 32; o > 0; o >>= 1) {
      loss_a += __shfl_down(loss_a, o, 64);
      loss_b += __shfl_down(loss_b, o, 64);
    }
    if (lane == 0) { red[4 * NO + 2 * w] = loss_a; red[4 * NO + 2 * w + 1] = loss_b; }
    if (CONT) {
      float u = als;
#pragma unroll
      for (int m = NO; m < 64; m <<= 1) u += __shfl_xor(u, m, 64);
      if (lane < NO) red[4 * NO + 8 + w * NO + lane] = u;
    }
    __syncthreads();
    if (tid < no) slab[oB3 + tid] = ((red[tid] + red[NO + tid]) + red[2 * NO + tid]) + red[3 * NO + tid];
    if (CONT && tid < no) {
      // d scale / d raw = softplus'(raw) = sigmoid(raw)
      const float* ra = red + 4 * NO + 8;
      slab[Pm + tid] = (((ra[tid] + ra[NO + tid]) + ra[2 * NO + tid]) + ra[3 * NO + tid]) * tn::sigmoid(tk.params[Pm + tid]);
    }
    if (tid == 0) {
      slab[P] = ((red[4 * NO] + red[4 * NO + 2]) + red[4 * NO + 4]) + red[4 * NO + 6];
      slab[P + 1] = ((red[4 * NO + 1] + red[4 * NO + 3]) + red[4 * NO + 5]) + red[4 * NO + 7];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = aW3r[r];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
      if (j == 0) slab[oW3 + fbase + (r & 3) + 8 * (r >> 2)] = v;
    }
    // db3 and the value loss: per-lane partials of every lane that took part (wave 0 / half 0 only, or one agent
    // per (wave, half) when aggregated) -> wave sums -> fixed-order sum over the 4 waves
    float v = ab3, l = loss_a;
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_down(v, o, 64);
      l += __shfl_down(l, o, 64);
    }
    if (lane == 0) { red[2 * w] = v; red[2 * w + 1] = l; }
    __syncthreads();
    if (tid == 0) {
      slab[oB3] = ((red[0] + red[2]) + red[4]) + red[6];
      slab[P] = ((red[1] + red[3]) + red[5]) + red[7];
      slab[P + 1] = 0.0f;
    }
  }
}

// 16-byte pieces when every input row is 16-byte aligned and the tile is wide enough for it to matter
int pick_xv(const float* x, int din) {
  const uintptr_t a = (uintptr_t)x;
  return (din % 4 == 0 && a % 16 == 0 && din >= 96) ? 4 : 1;
}

template <int NO, int KT1, bool ACTOR, int XV, bool CONT>
int launch_train(const TrainTask& tk, int n_slab, hipStream_t s) {
  const TrainLdsLayout L = make_layout<NO>(KT1);
  const size_t lb = (size_t)L.end * sizeof(float);
  MAVA_ARG_CHECK(lb <= 163840, 8,
                 "ppo_train: %zu bytes of LDS needed (n_out pad %d, input width %d) exceed the 160 KiB of a CU",
                 lb, NO, tk.din);
  static bool attr_set = false;  // once per instantiation (lb is a function of the template arguments only)
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)ppo_train_kernel<NO, KT1, ACTOR, XV, CONT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    attr_set = true;
  }
  hipLaunchKernelGGL((ppo_train_kernel<NO, KT1, ACTOR, XV, CONT>), dim3(n_slab), dim3(256), lb, s, tk, L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, int KT1, bool ACTOR, bool CONT>
int launch_xv(const TrainTask& tk, int n_slab, hipStream_t s) {
  if (KT1 >= 4 && tk.xv == 4) return launch_train<NO, KT1, ACTOR, (KT1 >= 4 ? 4 : 1), CONT>(tk, n_slab, s);
  return launch_train<NO, KT1, ACTOR, 1, CONT>(tk, n_slab, s);
}

template <int NO, bool ACTOR, bool CONT = false>
int dispatch_kt(const TrainTask& tk, int n_slab, hipStream_t s) {
  const int kt = tk.din / 32 + 1;  // 32*kt > din: the x tile always has a spare column for the ones (bias) input
#ifdef MAVA_FAST_BUILD  // developer iteration: only the BASELINE config-2 instantiations
  if (kt == 3) return launch_xv<NO, 3, ACTOR, CONT>(tk, n_slab, s);
  if (kt == 9) return launch_xv<NO, 9, ACTOR, CONT>(tk, n_slab, s);
  mava_set_error("fast build: input width %d not instantiated", tk.din);
  return MAVA_EARG(9);
#else
  switch (kt) {
    case 1: return launch_xv<NO, 1, ACTOR, CONT>(tk, n_slab, s);
    case 2: return launch_xv<NO, 2, ACTOR, CONT>(tk, n_slab, s);
    case 3: return launch_xv<NO, 3, ACTOR, CONT>(tk, n_slab, s);
    case 4: return launch_xv<NO, 4, ACTOR, CONT>(tk, n_slab, s);
    // (continuous head: the wide instantiations below spill some registers - tuned shapes are the ones above)
    case 5: case 6: return launch_xv<NO, 6, ACTOR, CONT>(tk, n_slab, s);
    case 7: case 8: case 9: return launch_xv<NO, 9, ACTOR, CONT>(tk, n_slab, s);
    default:
      mava_set_error("ppo_train: input width %d > 287 is not instantiated", tk.din);
      return MAVA_EARG(9);
  }
#endif
}

}  // namespace

#ifdef MAVA_STAMPS
// Diagnostic builds only (-DMAVA_STAMPS, tools/build_stamps.sh; not part of include/mava_hip.h): device buffer of 40 u64
// receiving block 0's per-phase cycle sums.
static unsigned long long* g_stamps = nullptr;
extern "C" int mava_debug_set_stamps(unsigned long long* p) {
  g_stamps = p;
  return MAVA_OK;
}
#else
static constexpr unsigned long long* g_stamps = nullptr;
#endif

// Settings of these kernels live in the caller's context handle (ctx.h; NULL = defaults):
//   critic aggregation, 1 (default): when the A agents of a (t,e) row share one critic input row (x_share == A <= 8) the
//     critic kernel evaluates the network once per row and back-propagates the sum of the agents' loss gradients - the
//     same gradient as A identical passes, at 1/A of the matrix work.  0: one pass per agent row (the reference's order).
//   matmul mode, 0: exact-f32 MFMA (v_mfma_f32_32x32x2_f32, this file); 1: split-f16 operands, three
//     v_mfma_f32_32x32x16_f16 per product, f32 accumulation (ppo_train_h2.hip) for the shapes that kernel instantiates
//     (discrete actor / critic, input width <= 287); everything else stays on mode 0.

extern "C" int mava_adv_stats_blocks(void) { return STATS_BLOCKS; }

extern "C" int mava_adv_stats_f64(const float* adv, const int32_t* idx, long idx_base, int Rb, int A,
                                  double* partials, hipStream_t s) {
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1, 0, "mava_adv_stats_f64: Rb=%d A=%d", Rb, A);
  MAVA_ARG_CHECK(adv && partials, 1, "mava_adv_stats_f64: null pointer argument");
  hipLaunchKernelGGL(adv_stats_kernel, dim3(STATS_BLOCKS), dim3(256), 0, s, adv, idx, idx_base, Rb, A,
                     partials, 0L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

// The statistics of n_batch minibatches in one launch: minibatch j takes the indices idx[j * idx_stride ...) and writes
// partials[j][mava_adv_stats_blocks()][2] - e.g. all ppo_epochs x num_minibatches slices of an update's permutations,
// known as soon as GAE has run (each 15 us launch on the critical path otherwise).
extern "C" int mava_adv_stats_batched_f64(const float* adv, const int32_t* idx, long idx_stride, int Rb, int A,
                                          int n_batch, double* partials, hipStream_t s) {
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_batch >= 1 && n_batch <= 65535 && idx_stride >= 0, 0,
                 "mava_adv_stats_batched_f64: Rb=%d A=%d n_batch=%d idx_stride=%ld", Rb, A, n_batch, idx_stride);
  MAVA_ARG_CHECK(adv && idx && partials, 1, "mava_adv_stats_batched_f64: null pointer argument");
  hipLaunchKernelGGL(adv_stats_kernel, dim3(STATS_BLOCKS, n_batch), dim3(256), 0, s, adv, idx, 0L, Rb, A, partials,
                     idx_stride);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_ppo_actor_grad_f32(mava_ctx* ctx, const float* params, int din, int n_actions,
                                       const float* agents_view, const uint8_t* action_mask,
                                       const int32_t* action, const float* old_log_prob,
                                       const float* advantages, const double* adv_stats,
                                       const int32_t* idx, long idx_base, int Rb, int A,
                                       float clip_eps, float ent_coef, float* slab, long slab_stride,
                                       int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && n_actions >= 1 && n_actions <= 32, 0,
                 "mava_ppo_actor_grad_f32: din=%d n_actions=%d unsupported", din, n_actions);
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024 && (long)Rb * A < (1L << 31), 1,
                 "mava_ppo_actor_grad_f32: Rb=%d A=%d n_slab=%d", Rb, A, n_slab);
  MAVA_ARG_CHECK(slab_stride >= mlp_param_count(din, n_actions) + 2, 2,
                 "mava_ppo_actor_grad_f32: slab_stride too small");
  MAVA_ARG_CHECK(params && agents_view && action && old_log_prob && advantages && adv_stats && slab, 3,
                 "mava_ppo_actor_grad_f32: null pointer argument");
  TrainTask tk = {};
  tk.params = params; tk.x = agents_view; tk.din = din; tk.no = n_actions; tk.xshare = 1; tk.agg = 1;
  tk.xv = pick_xv(agents_view, din); tk.A = A; tk.idx = idx; tk.idx_base = idx_base; tk.Rb = Rb;
  tk.mask = action_mask; tk.action = action; tk.old_logp = old_log_prob; tk.adv = advantages;
  tk.stats = adv_stats; tk.clip_eps = clip_eps; tk.ent_coef = ent_coef; tk.slab = slab;
  tk.slab_stride = slab_stride;
  tk.stamps = g_stamps;
  if (mava_ctx_matmul_mode(ctx) == 1) {
    const int rc = mava_train_h2_launch(ctx, tk, n_slab, true, s);
    if (rc <= 0) return rc;  // launched (0) or failed (< 0); 1: shape not instantiated there
  }
  if (n_actions <= 8) return dispatch_kt<8, true>(tk, n_slab, s);
  if (n_actions <= 16) return dispatch_kt<16, true>(tk, n_slab, s);
  return dispatch_kt<32, true>(tk, n_slab, s);
}

// Continuous action head (networks.py:127-169): params = [MLP(din -> 128 -> 128 -> action_dim) | log_std(action_dim)],
// slab row = gradient in the same layout, then (actor_loss, entropy) sums.  The entropy term uses one reparameterised
// sample per (row, dimension) from Philox(counter = (row_offset + trajectory row, ent_step, dim/2), key = seed).
extern "C" int mava_ppo_actor_grad_continuous_f32(const float* params, int din, int action_dim, float min_scale,
                                                  const float* agents_view, const float* action,
                                                  const float* old_log_prob, const float* advantages,
                                                  const double* adv_stats, const int32_t* idx, long idx_base, int Rb,
                                                  int A, float clip_eps, float ent_coef, uint64_t seed,
                                                  uint32_t ent_step, uint32_t row_offset, float* slab,
                                                  long slab_stride, int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && action_dim >= 1 && action_dim <= 16, 0,
                 "mava_ppo_actor_grad_continuous_f32: din=%d action_dim=%d unsupported (action_dim <= 16)", din,
                 action_dim);
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024 && (long)Rb * A < (1L << 31), 1,
                 "mava_ppo_actor_grad_continuous_f32: Rb=%d A=%d n_slab=%d", Rb, A, n_slab);
  MAVA_ARG_CHECK(slab_stride >= mlp_param_count(din, action_dim) + action_dim + 2, 2,
                 "mava_ppo_actor_grad_continuous_f32: slab_stride too small");
  MAVA_ARG_CHECK(params && agents_view && action && old_log_prob && advantages && adv_stats && slab, 3,
                 "mava_ppo_actor_grad_continuous_f32: null pointer argument");
  TrainTask tk = {};
  tk.params = params; tk.x = agents_view; tk.din = din; tk.no = action_dim; tk.xshare = 1; tk.agg = 1;
  tk.xv = pick_xv(agents_view, din); tk.A = A; tk.idx = idx; tk.idx_base = idx_base; tk.Rb = Rb;
  tk.action_f = action; tk.old_logp = old_log_prob; tk.adv = advantages;
  tk.stats = adv_stats; tk.clip_eps = clip_eps; tk.ent_coef = ent_coef; tk.slab = slab;
  tk.slab_stride = slab_stride;
  tk.seed_lo = (uint32_t)seed; tk.seed_hi = (uint32_t)(seed >> 32); tk.ent_step = ent_step; tk.row_offset = row_offset;
  tk.min_scale = min_scale;
  tk.stamps = g_stamps;
  if (action_dim <= 8) return dispatch_kt<8, true, true>(tk, n_slab, s);
  return dispatch_kt<16, true, true>(tk, n_slab, s);
}

extern "C" int mava_ppo_critic_grad_f32(mava_ctx* ctx, const float* params, int din, const float* critic_input,
                                        int x_share, const float* old_value, const float* targets,
                                        const int32_t* idx, long idx_base, int Rb, int A,
                                        float clip_eps, float vf_coef, float* slab, long slab_stride,
                                        int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && x_share >= 1, 0, "mava_ppo_critic_grad_f32: din=%d x_share=%d", din, x_share);
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024 && (long)Rb * A < (1L << 31), 1,
                 "mava_ppo_critic_grad_f32: Rb=%d A=%d n_slab=%d", Rb, A, n_slab);
  MAVA_ARG_CHECK(slab_stride >= mlp_param_count(din, 1) + 2, 2,
                 "mava_ppo_critic_grad_f32: slab_stride too small");
  MAVA_ARG_CHECK(params && critic_input && old_value && targets && slab, 3,
                 "mava_ppo_critic_grad_f32: null pointer argument");
  TrainTask tk = {};
  tk.params = params; tk.x = critic_input; tk.din = din; tk.no = 1; tk.xshare = x_share; tk.agg = 1;
  if (mava_ctx_critic_aggregation(ctx) && A > 1 && A <= 8 && x_share == A) {  // input row of index p is row p itself
    tk.agg = A;
    tk.xshare = 1;
  }
  tk.xv = pick_xv(critic_input, din); tk.A = A; tk.idx = idx; tk.idx_base = idx_base; tk.Rb = Rb;
  tk.old_value = old_value; tk.targets = targets; tk.clip_eps = clip_eps; tk.vf_coef = vf_coef;
  tk.slab = slab; tk.slab_stride = slab_stride;
  tk.stamps = g_stamps;
  if (mava_ctx_matmul_mode(ctx) == 1) {
    const int rc = mava_train_h2_launch(ctx, tk, n_slab, false, s);
    if (rc <= 0) return rc;
  }
  return dispatch_kt<1, false>(tk, n_slab, s);
}
