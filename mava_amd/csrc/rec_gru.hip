// GRU sequence kernels and the sequence losses of the recurrent PPO systems.
//
// Reference: mava/networks.py:238-266 (ScannedRNN: hidden state zeroed where `done` enters the step, then
// flax.linen.GRUCell), mava/systems/ppo/rec_mappo.py:108-117 (one-step use in the rollout), :210-266 (both
// losses re-unroll the whole sequence from hstates[0]).  GRUCell (restated; parity unpinned, see
// oracle/rec_oracle.py):  r = sig(gi_r + W_hr h)  z = sig(gi_z + W_hz h)  n = tanh(gi_n + r*(W_hn h + b_hn))
// h' = (1-z) n + z h,  with gi = W_i x + b_i computed beforehand for all steps by mava_rec_dense_f32.
//
// MI355X mapping: sequences are independent, time is sequential.  One 256-thread block owns 32
// sequences for all T steps; wave w owns hidden features [32w, 32w+32) of every gate and keeps ITS
// SLICE OF THE RECURRENT WEIGHTS IN REGISTERS for the whole scan (3 x 64 MFMA A-operands = 192 registers
// forward: columns of W_h; 192 registers backward: rows of W_h), so a time step is 192 exact-f32 MFMAs per
// wave fed by 64 LDS reads of the (transposed) previous hidden state - no weight traffic at all.
// Activations use the T32 tile layout (rec_dense.hip): accumulator-shaped loads/stores are coalesced.
#include "mlp_core.h"
#include "rec_task.h"
#include "tanh_normal.h"

#include "ctx.h"

namespace {

constexpr int LDT = 33;
constexpr int G3 = 3 * MLP_H;

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each; __builtin_amdgcn_rcpf:
// __frcp_rn expands to the ten-instruction correctly rounded division): the scan's
// elementwise phase runs with the MFMA pipe idle (one wave per SIMD), and libm's expf / tanhf / IEEE division cost
// ~8 K of the ~27 K cycles a time step took.  tanh(x) = 1 - 2 / (exp(2x) + 1) saturates correctly at both ends.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

__global__ __launch_bounds__(256, 1) void gru_scan_fwd_kernel(ScanTask tk) {
  __shared__ float HT[MLP_H * LDT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int mt = blockIdx.x;           // sequence tile
  const int m = mt * 32 + j;           // this lane's sequence
  const int tiles_per_t = tk.Rm / 32;
  const int fb = 32 * w + 4 * h;

  // resident slice of W_h: A[i = out feature 32w + j][k = in feature] = Wh[k][g*128 + 32w + j]
  float wr[3][8][8];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[g][b][s] = tk.wh[(long)(16 * b + 2 * s + h) * G3 + g * MLP_H + 32 * w + j];
  float bn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bn[r] = tk.bhn[fb + (r & 3) + 8 * (r >> 2)];

  // initial hidden state (masked by the reset flag of step 0)
  {
    const bool rs = tk.done[ext_row(tk, 0, m)] != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2);
      float v;
      if (tk.h0_t32) v = tk.h0[((long)mt * MLP_H + f) * 32 + j];
      else v = tk.h0[ext_row(tk, 0, m) * MLP_H + f];  // (E, A, 128): row = env * A + a (t = 0)
      HT[f * LDT + j] = rs ? 0.0f : v;
    }
  }

  // gate pre-activations of the CURRENT step (loaded one step ahead, under the previous step's MFMAs)
  float gr[16], gz[16], gin[16];
  const int lane_off = fb * 32 + j;  // element (feature fb, row j) of a T32 tile; the rest are immediates
#define OFFW(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)
  auto load_gi = [&](int t) {
    const float* git = tk.gi + ((long)t * tiles_per_t + mt) * G3 * 32 + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gr[r] = git[OFFW(r)];
      gz[r] = git[MLP_H * 32 + OFFW(r)];
      gin[r] = git[2 * MLP_H * 32 + OFFW(r)];
    }
  };
  load_gi(0);
  for (int t = 0; t < tk.T; ++t) {
    const long tile = (long)t * tiles_per_t + mt;
    f32x16 ar, az, an;
    float gn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ar[r] = gr[r]; az[r] = gz[r]; gn[r] = gin[r]; an[r] = bn[r]; }
    const bool rs_next = (t + 1 < tk.T) ? (tk.done[ext_row(tk, t + 1, m)] != 0) : false;
    if (t + 1 < tk.T) load_gi(t + 1);  // in flight during the 192 MFMAs below
    __syncthreads();  // HT (masked h entering step t) complete
    {
      const float* hb = HT + h * LDT + j;
      float xo[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xo[0][s] = hb[(2 * s) * LDT];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b + 1 < 8) {
#pragma unroll
          for (int s = 0; s < 8; ++s) xo[(b + 1) & 1][s] = hb[(16 * (b + 1) + 2 * s) * LDT];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          ar = MFMA32(wr[0][b][s], xo[b & 1][s], ar);
          az = MFMA32(wr[1][b][s], xo[b & 1][s], az);
          an = MFMA32(wr[2][b][s], xo[b & 1][s], an);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float hn[16], hp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hp[r] = HT[(fb + (r & 3) + 8 * (r >> 2)) * LDT + j];
    float* const hs_o = tk.hs + tile * (MLP_H * 32) + lane_off;
    float* const hp_o = tk.hprev ? tk.hprev + tile * (MLP_H * 32) + lane_off : nullptr;
    float* const sv = tk.saved ? tk.saved + tile * (4 * MLP_H) * 32 + lane_off : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rr = sigmoidf_(ar[r]);
      const float zz = sigmoidf_(az[r]);
      const float nn = tanhf_(gn[r] + rr * an[r]);
      hn[r] = (1.0f - zz) * nn + zz * hp[r];
      hs_o[OFFW(r)] = hn[r];
      if (hp_o != nullptr) hp_o[OFFW(r)] = hp[r];
      if (sv != nullptr) {
        sv[OFFW(r)] = rr;
        sv[MLP_H * 32 + OFFW(r)] = zz;
        sv[2 * MLP_H * 32 + OFFW(r)] = nn;
        sv[3 * MLP_H * 32 + OFFW(r)] = an[r];
      }
    }
    __syncthreads();  // every wave has read HT
#pragma unroll
    for (int r = 0; r < 16; ++r) HT[(fb + (r & 3) + 8 * (r >> 2)) * LDT + j] = rs_next ? 0.0f : hn[r];
  }
}

__global__ __launch_bounds__(256, 1) void gru_scan_bwd_kernel(ScanTask tk) {
  __shared__ float DG[G3 * LDT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int mt = blockIdx.x;
  const int m = mt * 32 + j;
  const int tiles_per_t = tk.Rm / 32;
  const int fb = 32 * w + 4 * h;

  // resident rows of W_h: A[i = in feature 32w + j][k = gate column n] = Wh[32w + j][n]
  float wb[192];
#pragma unroll
  for (int q = 0; q < 192; ++q) wb[q] = tk.wh[(long)(32 * w + j) * G3 + 2 * q + h];

  float dhc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dhc[r] = 0.0f;

  // per-step inputs of the CURRENT step, loaded one step ahead (under the previous step's 192 MFMAs)
  float i_r[16], i_z[16], i_n[16], i_hl[16], i_hp[16], i_dh[16];
  // One per-lane base pointer per array and step, every element at a COMPILE-TIME offset from it
  // (OFF(r) floats, < 4 KiB: an instruction immediate).  Written as sv[(long)f * 32 + j] the sign extension of
  // f = fb + const does not distribute, the compiler forms one 64-bit address per element, hoists the ~150
  // loop-invariant halves out of the time loop and spills them (scratch traffic shares vmcnt with the prefetch).
#define OFF(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)
  const int lane_off = fb * 32 + j;
  auto load_step = [&](int t) {
    const long tile = (long)t * tiles_per_t + mt;
    const float* sv = tk.saved + tile * (4 * MLP_H) * 32 + lane_off;
    const float* hpv = tk.hprev + tile * (MLP_H * 32) + lane_off;
    const float* dho = tk.dh_out + tile * (MLP_H * 32) + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      i_r[r] = sv[OFF(r)];
      i_z[r] = sv[MLP_H * 32 + OFF(r)];
      i_n[r] = sv[2 * MLP_H * 32 + OFF(r)];
      i_hl[r] = sv[3 * MLP_H * 32 + OFF(r)];
      i_hp[r] = hpv[OFF(r)];
      i_dh[r] = dho[OFF(r)];
    }
  };
  load_step(tk.T - 1);
  for (int t = tk.T - 1; t >= 0; --t) {
    const long tile = (long)t * tiles_per_t + mt;
    const bool rs = tk.done[ext_row(tk, t, m)] != 0;
    float dhp[16];
    float* const gi_o = tk.dgi + tile * G3 * 32 + lane_off;
    float* const gh_o = tk.dgh + tile * (tk.dgh_n_only ? MLP_H : G3) * 32 + lane_off;
    float* const dgl = DG + fb * LDT + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      constexpr int LOFF = 0;  // (LDS offsets are 32-bit: no hoisting problem)
      const int fo = ((r & 3) + 8 * (r >> 2)) * LDT + LOFF;
      const float rr = i_r[r], zz = i_z[r], nn = i_n[r], hl = i_hl[r], hp = i_hp[r];
      const float dh = i_dh[r] + dhc[r];
      const float dn = dh * (1.0f - zz);
      const float dz = dh * (hp - nn);
      dhp[r] = dh * zz;
      const float dn_pre = dn * (1.0f - nn * nn);
      const float dr = dn_pre * hl;
      const float dghn = dn_pre * rr;
      const float dz_pre = dz * zz * (1.0f - zz);
      const float dr_pre = dr * rr * (1.0f - rr);
      gi_o[OFF(r)] = dr_pre;
      gi_o[MLP_H * 32 + OFF(r)] = dz_pre;
      gi_o[2 * MLP_H * 32 + OFF(r)] = dn_pre;
      if (tk.dgh_n_only) {  // (block-uniform) the n third alone: r and z thirds are dgi's
        gh_o[OFF(r)] = dghn;
      } else {
        gh_o[OFF(r)] = dr_pre;
        gh_o[MLP_H * 32 + OFF(r)] = dz_pre;
        gh_o[2 * MLP_H * 32 + OFF(r)] = dghn;
      }
      dgl[fo] = dr_pre;
      dgl[MLP_H * LDT + fo] = dz_pre;
      dgl[2 * MLP_H * LDT + fo] = dghn;
      // keep each feature's nine stores next to its arithmetic: clustered, the 96 results are all live at once
      // on top of 192 resident weight words and the 96-word prefetch, and spill (scratch shares vmcnt with the prefetch)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t > 0) load_step(t - 1);  // in flight during the MFMAs below
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
      const float* db = DG + h * LDT + j;
      float xo[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xo[0][s] = db[(2 * s) * LDT];
#pragma unroll
      for (int b = 0; b < 24; ++b) {
        if (b + 1 < 24) {
#pragma unroll
          for (int s = 0; s < 8; ++s) xo[(b + 1) & 1][s] = db[(16 * (b + 1) + 2 * s) * LDT];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = MFMA32(wb[8 * b + s], xo[b & 1][s], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // gradient into the hidden state entering step t; a reset cuts the chain (networks.py:253-257)
#pragma unroll
    for (int r = 0; r < 16; ++r) dhc[r] = rs ? 0.0f : (acc[r] + dhp[r]);
    __syncthreads();  // DG free
  }
}

// ----------------------------------------------------------------------------------------------------------
// Sequence losses on T32 logits / values (rec_mappo.py:210-266 after the network re-unroll).
struct SeqLossTask {
  int T, Rm, E, A, no;
  int agg;                   // critic: agent slots per row when the rows are (t, env) rows shared by the agents (else 1)
  const int32_t* idx;
  const float* y;            // T32 (T*Rm x no) logits, or (T*Rm x 1) values
  float* dy;                 // T32 same shape
  const uint8_t* mask;       // external (T, E, A, no) or null
  const int32_t* action;     // external (T, E, A)
  const float* f0;           // actor: old_log_prob ; critic: old_value     (T, E, A)
  const float* f1;           // actor: advantages   ; critic: targets
  const double* stats;       // adv stats partials (actor)
  int n_stats;
  float clip_eps, coef;      // ent_coef (actor) / vf_coef (critic)
  float grad_scale;          // dy is written in units of grad_scale (a power of two; see rec_dense_h2.hip)
  float* loss_partials;      // (gridDim.x, 2)
  // sampling (rollout)
  uint32_t seed_lo, seed_hi, step, row_offset;
  int greedy;
  int32_t* action_out;       // external (E, A) slot
  float* logp_out;
  float* value_out;
  // continuous head (tanh_normal.h): y holds the means
  float min_scale;           // scale = softplus(raw) + min_scale (networks.py:134,162)
  const float* log_std;      // (no) raw scales
  const float* log_std_rows; // T32 (rows x no) raw scales per row (ContinuousActionHead(independent_std=False)) or null
  float* dlog_std_rows;      // T32: d loss / d raw scale per row (times grad_scale), with log_std_rows
  const float* action_f;     // external (T, E, A, no) actions in (-1, 1)
  float* action_f_out;       // external (E, A, no) slot
  float* dscale_partials;    // (gridDim.x, no): d loss / d log_std partials
  uint32_t ent_step;
};

template <int NO, bool ACTOR>
__global__ __launch_bounds__(256) void seq_loss_kernel(SeqLossTask tk) {
  __shared__ float red[2][4];
  __shared__ float st[2];
  const long R = (long)tk.T * tk.Rm;
  const float invR = 1.0f / (float)(R * (ACTOR ? 1 : (tk.agg > 1 ? tk.agg : 1)));  // mean over all agent row-steps
  const float gs = tk.grad_scale;
  if (ACTOR && threadIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < tk.n_stats; ++i) { s1 += tk.stats[2 * i]; s2 += tk.stats[2 * i + 1]; }
    const double mean = s1 / (double)R;
    double var = s2 / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    st[0] = (float)mean;
    st[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
  }
  __syncthreads();
  float la = 0.0f, lb = 0.0f;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < R; q += (long)gridDim.x * 256) {
    const int t = (int)(q / tk.Rm), m = (int)(q - (long)t * tk.Rm);
    const int e_local = m / tk.A, a = m - e_local * tk.A;
    const int env = tk.idx ? tk.idx[e_local] : e_local;
    const long er = ((long)t * tk.E + env) * tk.A + a;
    const long tile = q >> 5;
    const int jj = (int)(q & 31);
    if (ACTOR) {
      float y[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) y[o] = (o < tk.no) ? tk.y[(tile * tk.no + o) * 32 + jj] : 0.0f;
      Categorical<NO> cat;
      cat.build(y, tk.mask ? (tk.mask + er * tk.no) : nullptr, tk.no);
      const int act = tk.action[er];
      float lp = 0.0f;
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (o == act) lp = cat.logp[o];
      const float gae = (tk.f1[er] - st[0]) * st[1];
      const float ratio = expf(lp - tk.f0[er]);
      const float lo = 1.0f - tk.clip_eps, hi = 1.0f + tk.clip_eps;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float l1 = ratio * gae, l2 = rc * gae;
      const bool inside = (ratio >= lo) && (ratio <= hi);
      const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
      const float g2 = inside ? (1.0f - g1) : 0.0f;
      const float dlp = -(g1 + g2) * gae * ratio * invR;
      const float ec = tk.coef * invR;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        if (o < tk.no) {
          const float oh = (o == act) ? 1.0f : 0.0f;
          const float pl = (cat.p[o] > 0.0f) ? cat.logp[o] : 0.0f;
          float d = dlp * (oh - cat.p[o]) + ec * cat.p[o] * (pl + cat.entropy);
          if (cat.z[o] == -FLT_MAX) d = 0.0f;
          tk.dy[(tile * tk.no + o) * 32 + jj] = d * gs;
        }
      }
      la += -fminf(l1, l2) * invR;
      lb += cat.entropy * invR;
    } else {
      // one value per row; with agg > 1 the row is a (t, env) row whose agents share the critic input, and the
      // loss gradients of its agg agent slots add up (d loss / d v = sum_a dy_a)
      const float v = tk.y[tile * 32 + jj];
      const int na = tk.agg > 1 ? tk.agg : 1;
      float dsum = 0.0f;
      for (int a2 = 0; a2 < na; ++a2) {
        const long ea = er * na + a2;
        const float ov = tk.f0[ea], tg = tk.f1[ea];
        const float diff = v - ov;
        const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
        const float e1 = v - tg, e2 = vclip - tg;
        const float l1 = e1 * e1, l2 = e2 * e2;
        const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
        const float g1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        dsum += tk.coef * (g1 * e1 + g2 * e2) * invR;
        la += 0.5f * fmaxf(l1, l2) * invR;
      }
      tk.dy[tile * 32 + jj] = dsum * gs;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    la += __shfl_down(la, o, 64);
    lb += __shfl_down(lb, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = la; red[1][w] = lb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    tk.loss_partials[2 * blockIdx.x + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    tk.loss_partials[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}

// Rollout epilogue: T32 logits of ONE step (Rm = E*A rows, identity env order) -> sampled action + log-prob
// written row-major into the trajectory slot; or T32 values -> row-major.
template <int NO>
__global__ __launch_bounds__(256) void seq_sample_kernel(SeqLossTask tk) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= tk.Rm) return;
  const long tile = q >> 5;
  const int jj = q & 31;
  float y[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) y[o] = (o < tk.no) ? tk.y[(tile * tk.no + o) * 32 + jj] : 0.0f;
  Categorical<NO> cat;
  cat.build(y, tk.mask ? (tk.mask + (long)q * tk.no) : nullptr, tk.no);
  int a = 0;
  float best = -FLT_MAX;
  if (tk.greedy) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < tk.no && cat.z[o] > best) { best = cat.z[o]; a = o; }
  } else {
    const uint32_t gid = tk.row_offset + (uint32_t)q;
#pragma unroll
    for (int c = 0; c < (NO + 3) / 4; ++c) {
      Philox4 rnd = philox4x32_10(gid, tk.step, (uint32_t)c, 0x504f4c49u, tk.seed_lo, tk.seed_hi);
      const uint32_t wds[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = 4 * c + k;
        if (o < NO && o < tk.no) {
          const float sc = cat.z[o] - logf(-logf(u01_open(wds[k])));
          if (sc > best) { best = sc; a = o; }
        }
      }
    }
  }
  float lp = 0.0f;
#pragma unroll
  for (int o = 0; o < NO; ++o)
    if (o == a) lp = cat.logp[o];
  tk.action_out[q] = a;
  tk.logp_out[q] = lp;
}

// Continuous head of the sequence loss (rec_mappo.py:210-242 with networks.py:127-169 / distributions.py:24-91): the
// same row walk as seq_loss_kernel<NO, true>; a thread owns all `no` action dimensions of its row.
template <int NO>
__global__ __launch_bounds__(256) void seq_loss_cont_kernel(SeqLossTask tk) {
  __shared__ float red[2 + NO][4];
  __shared__ float st[2];
  const long R = (long)tk.T * tk.Rm;
  const float invR = 1.0f / (float)R;
  if (threadIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < tk.n_stats; ++i) { s1 += tk.stats[2 * i]; s2 += tk.stats[2 * i + 1]; }
    const double mean = s1 / (double)R;
    double var = s2 / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    st[0] = (float)mean;
    st[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
  }
  __syncthreads();
  float sc[NO], ds[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    sc[o] = tk.log_std != nullptr ? tn::scale_of(tk.log_std[o < tk.no ? o : 0], tk.min_scale) : 1.0f;
    ds[o] = 0.0f;
  }
  float la = 0.0f, lb = 0.0f;
  const float lo = 1.0f - tk.clip_eps, hi = 1.0f + tk.clip_eps;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < R; q += (long)gridDim.x * 256) {
    const int t = (int)(q / tk.Rm), m = (int)(q - (long)t * tk.Rm);
    const int e_local = m / tk.A, a = m - e_local * tk.A;
    const int env = tk.idx ? tk.idx[e_local] : e_local;
    const long er = ((long)t * tk.E + env) * tk.A + a;
    const long tile = q >> 5;
    const int jj = (int)(q & 31);
    float dm[NO], dsl[NO], th[NO], ep[NO], lsr[NO];
    float lp = 0.0f, ent = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      dm[o] = dsl[o] = th[o] = ep[o] = lsr[o] = 0.0f;
      if (o < tk.no) {
        const float mean = tk.y[(tile * tk.no + o) * 32 + jj];
        if (tk.log_std_rows != nullptr) {  // state-dependent scale: networks.py:140,161
          lsr[o] = tk.log_std_rows[(tile * tk.no + o) * 32 + jj];
          sc[o] = tn::scale_of(lsr[o], tk.min_scale);
        }
        const tn::LogProb l = tn::log_prob(tk.action_f[er * tk.no + o], mean, sc[o]);
        lp += l.lp;
        dm[o] = l.dmean;
        dsl[o] = l.dscale;
        ep[o] = tn::noise(tk.row_offset + (uint32_t)er, tk.ent_step, o, tn::STREAM_ENTROPY, tk.seed_lo, tk.seed_hi);
        const float x = fmaf(sc[o], ep[o], mean);
        th[o] = tanhf(x);
        ent += 0.5f + tn::HALF_LOG_2PI + logf(sc[o]) + tn::tanh_fldj(x);
      }
    }
    const float gae = (tk.f1[er] - st[0]) * st[1];
    const float ratio = expf(lp - tk.f0[er]);
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float l1 = ratio * gae, l2 = rc * gae;
    const bool inside = (ratio >= lo) && (ratio <= hi);
    const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
    const float g2 = inside ? (1.0f - g1) : 0.0f;
    const float dlp = -(g1 + g2) * gae * ratio * invR;
    const float ec = tk.coef * invR;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      if (o < tk.no) {
        tk.dy[(tile * tk.no + o) * 32 + jj] = (dlp * dm[o] + ec * 2.0f * th[o]) * tk.grad_scale;
        const float dsc = dlp * dsl[o] - ec * (1.0f / sc[o] - 2.0f * th[o] * ep[o]);
        if (tk.dlog_std_rows != nullptr) tk.dlog_std_rows[(tile * tk.no + o) * 32 + jj] = dsc * tn::sigmoid(lsr[o]) * tk.grad_scale;
        else ds[o] += dsc;
      }
    }
    la += -fminf(l1, l2) * invR;
    lb += ent * invR;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    la += __shfl_down(la, o, 64);
    lb += __shfl_down(lb, o, 64);
#pragma unroll
    for (int k = 0; k < NO; ++k) ds[k] += __shfl_down(ds[k], o, 64);
  }
  if (lane == 0) {
    red[0][w] = la;
    red[1][w] = lb;
#pragma unroll
    for (int k = 0; k < NO; ++k) red[2 + k][w] = ds[k];
  }
  __syncthreads();
  if (threadIdx.x < 2) tk.loss_partials[2 * blockIdx.x + threadIdx.x] =
      ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
  if ((int)threadIdx.x < tk.no && tk.log_std != nullptr) {
    const int k = threadIdx.x;  // d scale / d log_std = sigmoid(log_std)
    tk.dscale_partials[(long)blockIdx.x * tk.no + k] =
        (((red[2 + k][0] + red[2 + k][1]) + red[2 + k][2]) + red[2 + k][3]) * tn::sigmoid(tk.log_std[k]);
  }
}

// Rollout epilogue of the continuous head: T32 means of ONE step -> tanh(loc + scale * noise) and its log-prob.
template <int NO>
__global__ __launch_bounds__(256) void seq_sample_cont_kernel(SeqLossTask tk) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= tk.Rm) return;
  const long tile = q >> 5;
  const int jj = q & 31;
  const uint32_t gid = tk.row_offset + (uint32_t)q;
  float lp = 0.0f;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    if (o < tk.no) {
      const float mean = tk.y[(tile * tk.no + o) * 32 + jj];
      const float sc = tn::scale_of(tk.log_std_rows != nullptr ? tk.log_std_rows[(tile * tk.no + o) * 32 + jj] : tk.log_std[o], tk.min_scale);
      const float eps = tk.greedy ? 0.0f : tn::noise(gid, tk.step, o, tn::STREAM_SAMPLE, tk.seed_lo, tk.seed_hi);
      const float a = tanhf(fmaf(sc, eps, mean));
      lp += tn::log_prob(a, mean, sc).lp;
      tk.action_f_out[(long)q * tk.no + o] = a;
    }
  }
  tk.logp_out[q] = lp;
}

__global__ __launch_bounds__(256) void t32_to_rows_kernel(const float* __restrict__ src, int N, int rows,
                                                          float* __restrict__ dst) {
  // dst[row][f] (row-major) <- T32 src ; one thread per element, reads coalesced along rows
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * N) return;
  const long tile = i / (32L * N);
  const int rem = (int)(i - tile * 32L * N);
  const int f = rem >> 5, jj = rem & 31;
  dst[(tile * 32 + jj) * N + f] = src[i];
}

__global__ __launch_bounds__(256) void rows_to_t32_kernel(const float* __restrict__ src, int N, int rows,
                                                          float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * N) return;
  const long tile = i / (32L * N);
  const int rem = (int)(i - tile * 32L * N);
  const int f = rem >> 5, jj = rem & 31;
  dst[i] = src[(tile * 32 + jj) * N + f];
}

}  // namespace

extern "C" int mava_gru_scan_fwd_f32(const mava_ctx* ctx, int T, int Rm, int E, int A, const int32_t* idx, const uint8_t* done,
                                     const float* h0, int h0_t32, const float* wh, const float* bhn,
                                     const float* gi, float* hs, float* hprev, float* saved, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm >= 32 && Rm % 32 == 0 && E >= 1 && A >= 1 && Rm % A == 0, 0,
                 "mava_gru_scan_fwd_f32: T=%d Rm=%d E=%d A=%d (Rm must be a multiple of 32 and of A)", T, Rm, E, A);
  MAVA_ARG_CHECK(done && h0 && wh && bhn && gi && hs, 1, "mava_gru_scan_fwd_f32: null pointer argument");
  MAVA_ARG_CHECK((hprev == nullptr) == (saved == nullptr), 2, "mava_gru_scan_fwd_f32: hprev and saved come together (training) or not at all");
  ScanTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.idx = idx; tk.done = done; tk.h0 = h0; tk.h0_t32 = h0_t32;
  tk.wh = wh; tk.bhn = bhn; tk.gi = gi; tk.hs = hs; tk.hprev = hprev; tk.saved = saved;
  if (mava_ctx_matmul_mode(ctx) == 1) return mava_gru_scan_fwd_h2_launch(tk, s);  // rec_gru_h2.hip
  hipLaunchKernelGGL(gru_scan_fwd_kernel, dim3(Rm / 32), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_gru_scan_bwd_f32(const mava_ctx* ctx, int T, int Rm, int E, int A, const int32_t* idx, const uint8_t* done,
                                     const float* wh, const float* saved, const float* hprev,
                                     const float* dh_out, float* dgi, float* dgh, int dgh_n_only, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm >= 32 && Rm % 32 == 0 && E >= 1 && A >= 1 && Rm % A == 0, 0,
                 "mava_gru_scan_bwd_f32: T=%d Rm=%d E=%d A=%d", T, Rm, E, A);
  MAVA_ARG_CHECK(done && wh && saved && hprev && dh_out && dgi && dgh, 1,
                 "mava_gru_scan_bwd_f32: null pointer argument");
  ScanTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.idx = idx; tk.done = done; tk.wh = wh;
  tk.saved = const_cast<float*>(saved); tk.hprev = const_cast<float*>(hprev); tk.dh_out = dh_out;
  tk.dgi = dgi; tk.dgh = dgh; tk.dgh_n_only = dgh_n_only != 0;
  if (mava_ctx_matmul_mode(ctx) == 1) return mava_gru_scan_bwd_h2_launch(tk, s);  // rec_gru_h2.hip
  hipLaunchKernelGGL(gru_scan_bwd_kernel, dim3(Rm / 32), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_seq_actor_loss_f32(int T, int Rm, int E, int A, int n_actions, const int32_t* idx,
                                       const float* logits, const uint8_t* mask, const int32_t* action,
                                       const float* old_log_prob, const float* advantages, const double* adv_stats,
                                       int n_stats, float clip_eps, float ent_coef, float grad_scale, float* dlogits,
                                       float* loss_partials, int n_blocks, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm % 32 == 0 && n_actions >= 1 && n_actions <= 32 && n_blocks >= 1, 0,
                 "mava_seq_actor_loss_f32: bad shape");
  MAVA_ARG_CHECK(logits && action && old_log_prob && advantages && adv_stats && dlogits && loss_partials, 1,
                 "mava_seq_actor_loss_f32: null pointer argument");
  SeqLossTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.no = n_actions; tk.idx = idx; tk.y = logits; tk.dy = dlogits;
  tk.mask = mask; tk.action = action; tk.f0 = old_log_prob; tk.f1 = advantages; tk.stats = adv_stats;
  tk.n_stats = n_stats; tk.clip_eps = clip_eps; tk.coef = ent_coef; tk.loss_partials = loss_partials;
  tk.grad_scale = grad_scale;
  if (n_actions <= 8) hipLaunchKernelGGL((seq_loss_kernel<8, true>), dim3(n_blocks), dim3(256), 0, s, tk);
  else if (n_actions <= 16) hipLaunchKernelGGL((seq_loss_kernel<16, true>), dim3(n_blocks), dim3(256), 0, s, tk);
  else hipLaunchKernelGGL((seq_loss_kernel<32, true>), dim3(n_blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_seq_actor_loss_continuous_f32(int T, int Rm, int E, int A, int action_dim, float min_scale, const int32_t* idx,
                                                  const float* mean, const float* log_std, const float* log_std_rows,
                                                  const float* action,
                                                  const float* old_log_prob, const float* advantages,
                                                  const double* adv_stats, int n_stats, float clip_eps, float ent_coef,
                                                  uint64_t seed, uint32_t ent_step, uint32_t row_offset,
                                                  float grad_scale, float* dmean, float* dlog_std_rows,
                                                  float* loss_partials, float* dscale_partials, int n_blocks,
                                                  hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm % 32 == 0 && action_dim >= 1 && action_dim <= 16 && n_blocks >= 1, 0,
                 "mava_seq_actor_loss_continuous_f32: bad shape (action_dim <= 16)");
  MAVA_ARG_CHECK(mean && (log_std || log_std_rows) && action && old_log_prob && advantages && adv_stats && dmean && loss_partials &&
                 (dscale_partials || log_std_rows), 1, "mava_seq_actor_loss_continuous_f32: null pointer argument");
  SeqLossTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.no = action_dim; tk.idx = idx; tk.y = mean; tk.dy = dmean;
  MAVA_ARG_CHECK((log_std_rows == nullptr) == (dlog_std_rows == nullptr), 2,
                 "mava_seq_actor_loss_continuous_f32: log_std_rows and dlog_std_rows come together");
  tk.min_scale = min_scale;
  tk.log_std = log_std; tk.log_std_rows = log_std_rows; tk.dlog_std_rows = dlog_std_rows; tk.action_f = action; tk.f0 = old_log_prob; tk.f1 = advantages; tk.stats = adv_stats;
  tk.n_stats = n_stats; tk.clip_eps = clip_eps; tk.coef = ent_coef; tk.loss_partials = loss_partials;
  tk.dscale_partials = dscale_partials; tk.seed_lo = (uint32_t)seed; tk.seed_hi = (uint32_t)(seed >> 32);
  tk.ent_step = ent_step; tk.row_offset = row_offset; tk.grad_scale = grad_scale;
  if (action_dim <= 8) hipLaunchKernelGGL((seq_loss_cont_kernel<8>), dim3(n_blocks), dim3(256), 0, s, tk);
  else hipLaunchKernelGGL((seq_loss_cont_kernel<16>), dim3(n_blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_seq_sample_continuous_f32(int rows, int action_dim, float min_scale, const float* mean, const float* log_std,
                                              const float* log_std_rows, uint64_t seed, uint32_t step, uint32_t row_offset, int greedy,
                                              float* action, float* log_prob, hipStream_t s) {
  MAVA_ARG_CHECK(rows >= 1 && rows % 32 == 0 && action_dim >= 1 && action_dim <= 16, 0,
                 "mava_seq_sample_continuous_f32: rows=%d action_dim=%d", rows, action_dim);
  MAVA_ARG_CHECK(mean && (log_std || log_std_rows) && action && log_prob, 1, "mava_seq_sample_continuous_f32: null pointer argument");
  SeqLossTask tk = {};
  tk.Rm = rows; tk.no = action_dim; tk.min_scale = min_scale; tk.y = mean; tk.log_std = log_std; tk.log_std_rows = log_std_rows; tk.seed_lo = (uint32_t)seed;
  tk.seed_hi = (uint32_t)(seed >> 32); tk.step = step; tk.row_offset = row_offset; tk.greedy = greedy;
  tk.action_f_out = action; tk.logp_out = log_prob;
  const int blocks = mava_cdiv(rows, 256);
  if (action_dim <= 8) hipLaunchKernelGGL((seq_sample_cont_kernel<8>), dim3(blocks), dim3(256), 0, s, tk);
  else hipLaunchKernelGGL((seq_sample_cont_kernel<16>), dim3(blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_seq_critic_loss_f32(int T, int Rm, int E, int A, int agents_per_row, const int32_t* idx, const float* values,
                                        const float* old_value, const float* targets, float clip_eps,
                                        float vf_coef, float grad_scale, float* dvalues, float* loss_partials,
                                        int n_blocks, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm % 32 == 0 && n_blocks >= 1 && agents_per_row >= 1 && (agents_per_row == 1 || A == 1), 0,
                 "mava_seq_critic_loss_f32: bad shape (agents_per_row > 1 needs A == 1: rows are (t, env) rows)");
  MAVA_ARG_CHECK(values && old_value && targets && dvalues && loss_partials, 1,
                 "mava_seq_critic_loss_f32: null pointer argument");
  SeqLossTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.agg = agents_per_row; tk.no = 1; tk.idx = idx; tk.y = values; tk.dy = dvalues;
  tk.f0 = old_value; tk.f1 = targets; tk.clip_eps = clip_eps; tk.coef = vf_coef; tk.loss_partials = loss_partials;
  tk.grad_scale = grad_scale;
  hipLaunchKernelGGL((seq_loss_kernel<1, false>), dim3(n_blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_seq_sample_f32(int rows, int n_actions, const float* logits, const uint8_t* mask, uint64_t seed,
                                   uint32_t step, uint32_t row_offset, int greedy, int32_t* action,
                                   float* log_prob, hipStream_t s) {
  MAVA_ARG_CHECK(rows >= 1 && rows % 32 == 0 && n_actions >= 1 && n_actions <= 32, 0,
                 "mava_seq_sample_f32: rows=%d n_actions=%d", rows, n_actions);
  MAVA_ARG_CHECK(logits && action && log_prob, 1, "mava_seq_sample_f32: null pointer argument");
  SeqLossTask tk = {};
  tk.Rm = rows; tk.no = n_actions; tk.y = logits; tk.mask = mask; tk.seed_lo = (uint32_t)seed;
  tk.seed_hi = (uint32_t)(seed >> 32); tk.step = step; tk.row_offset = row_offset; tk.greedy = greedy;
  tk.action_out = action; tk.logp_out = log_prob;
  const int blocks = mava_cdiv(rows, 256);
  if (n_actions <= 8) hipLaunchKernelGGL((seq_sample_kernel<8>), dim3(blocks), dim3(256), 0, s, tk);
  else if (n_actions <= 16) hipLaunchKernelGGL((seq_sample_kernel<16>), dim3(blocks), dim3(256), 0, s, tk);
  else hipLaunchKernelGGL((seq_sample_kernel<32>), dim3(blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_convert_f32(const float* src, int N, int rows, int to_t32, float* dst, hipStream_t s) {
  MAVA_ARG_CHECK(N >= 1 && rows >= 0 && rows % 32 == 0, 0, "mava_t32_convert_f32: N=%d rows=%d", N, rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(src && dst, 1, "mava_t32_convert_f32: null pointer argument");
  const int blocks = mava_cdiv((long)rows * N, 256);
  if (to_t32) hipLaunchKernelGGL(rows_to_t32_kernel, dim3(blocks), dim3(256), 0, s, src, N, rows, dst);
  else hipLaunchKernelGGL(t32_to_rows_kernel, dim3(blocks), dim3(256), 0, s, src, N, rows, dst);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
