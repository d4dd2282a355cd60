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
//   P1  z1_w  = b1 + W1[:, w]^T x^T            A: W1 from L2 (coalesced), B: x from HBM
//   P2  z2_w  = b2 + W2[:, w]^T h1^T           A: W2 in LDS, B: h1^T in LDS; partial head logits
//   P3  loss + dlogits on the VALU (every lane owns one row); dz2_w = (W3 dy) * relu'(z2_w)
//   P4  dh1_w = W2[w, :] dz2^T                 A: W2 in LDS (row walk, odd stride), B: dz2^T in LDS
//       small gradients (dW3, db3, db2) as per-thread LDS sweeps with register accumulators
//   P5  gW2[:, w] += h1^T . dz2 ; gW1[:, w] += x^T . dz1   (A: h1^T in LDS / x from HBM)
// The weight-gradient accumulators (KT1*16 + 64 AGPRs per lane) stay resident for the whole
// launch; activations cross waves through four 16.5 KB LDS tiles ([feature][row], stride 33 =>
// conflict-free for both the B-operand row walk and the A-operand feature walk).  Each block
// writes ONE partial-gradient slab; mava_slab_reduce_f32 sums slabs in a fixed order, so the
// gradient is bitwise reproducible (no float atomics).
#include "mlp_core.h"

namespace {

constexpr int STATS_BLOCKS = 128;
constexpr int LDT = 33;  // row stride of the [feature][32 rows] exchange tiles

struct TrainTask {
  const float* params;
  const float* x;          // (rows_x, din)
  int din, no, xshare, xv;
  int A;                   // agent rows per (t,e) index
  const int32_t* idx;      // minibatch (t*E+e) indices, or null => idx_base + b
  long idx_base;
  int Rb;                  // (t,e) rows in the minibatch; agent rows R = Rb * A
  const uint8_t* mask;     // (TE*A, no) or null
  const int32_t* action;   // (TE*A)
  const float* old_logp;   // (TE*A)
  const float* adv;        // (TE*A)
  const double* stats;     // STATS_BLOCKS x {sum, sumsq} partials of the minibatch advantages
  const float* old_value;  // (TE*A)
  const float* targets;    // (TE*A)
  float clip_eps, ent_coef, vf_coef;
  float* slab;
  long slab_stride;
};

template <int NO>
struct TrainLds {
  static constexpr int TILE = MLP_H * LDT;
  static constexpr int H1T = MlpLds<NO>::END;
  static constexpr int H2T = H1T + TILE;
  static constexpr int DZ2T = H2T + TILE;
  static constexpr int DZ1T = DZ2T + TILE;  // doubles as YP[4][NO][32] (partial logits) until P4
  static constexpr int DZ1T_SIZE = (TILE > 4 * NO * 32) ? TILE : 4 * NO * 32;
  static constexpr int DY = DZ1T + DZ1T_SIZE;  // [NO][32]
  static constexpr int ROWX = DY + NO * 32;    // 32 x int64
  static constexpr int MISC = ROWX + 64;
  static constexpr int END = MISC + 8;
};

// advantage statistics of one minibatch: partial (sum, sumsq) in f64 per block
__global__ __launch_bounds__(256) void adv_stats_kernel(const float* __restrict__ adv,
                                                        const int32_t* __restrict__ idx,
                                                        long idx_base, int Rb, int A,
                                                        double* __restrict__ partials) {
  __shared__ double sh[2][4];
  double s1 = 0.0, s2 = 0.0;
  const long R = (long)Rb * A;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < R; q += (long)gridDim.x * 256) {
    const long b = q / A;
    const int a = (int)(q - b * A);
    const long p = idx ? (long)idx[b] : idx_base + b;
    const double v = (double)adv[p * A + a];
    s1 += v;
    s2 += v * v;
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

// Layer-1 product for ONE 32-feature tile with a PD-deep register ring of operand prefetches
// (one wave per SIMD: nothing else hides the L2 / HBM latency).
template <int XV, int PD>
__device__ __forceinline__ void l1_tile(const float* __restrict__ xrow, int din,
                                        const float* __restrict__ wcol, int h, f32x16& z) {
  constexpr int STEP = 2 * XV;
  const int nfull = din / STEP;
  float xb[PD][XV], wb[PD][XV];
  auto load = [&](int c, float (&xo)[XV], float (&wo)[XV]) {
    if (c < nfull) {
      const int k0 = c * STEP + XV * h;
      typename XVec<XV>::T xv = *reinterpret_cast<const typename XVec<XV>::T*>(xrow + k0);
      const float* xs = reinterpret_cast<const float*>(&xv);
#pragma unroll
      for (int m = 0; m < XV; ++m) {
        xo[m] = xs[m];
        wo[m] = wcol[(k0 + m) * MLP_H];
      }
    } else {
#pragma unroll
      for (int m = 0; m < XV; ++m) { xo[m] = 0.f; wo[m] = 0.f; }
    }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) load(u, xb[u], wb[u]);
  for (int c0 = 0; c0 < nfull; c0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (c0 + u < nfull) {
#pragma unroll
        for (int m = 0; m < XV; ++m) z = MFMA32(wb[u][m], xb[u][m], z);
      }
      load(c0 + u + PD, xb[u], wb[u]);
    }
  }
  for (int k0 = nfull * STEP; k0 < din; k0 += 2) {
    const int k = k0 + h;
    const bool ok = k < din;
    const int kc = ok ? k : (din - 1);
    const float xv = ok ? xrow[kc] : 0.0f;
    z = MFMA32(wcol[kc * MLP_H], xv, z);
  }
}

template <int NO, int KT1, bool ACTOR>
__global__ __launch_bounds__(256, 1) void ppo_train_kernel(TrainTask tk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const W2s = lds + MlpLds<NO>::W2;
  float* const W3s = lds + MlpLds<NO>::W3;
  float* const H1T = lds + TrainLds<NO>::H1T;
  float* const H2T = lds + TrainLds<NO>::H2T;
  float* const DZ2T = lds + TrainLds<NO>::DZ2T;
  float* const DZ1T = lds + TrainLds<NO>::DZ1T;
  float* const YP = DZ1T;
  float* const DY = lds + TrainLds<NO>::DY;
  long* const rowx = reinterpret_cast<long*>(lds + TrainLds<NO>::ROWX);
  float* const misc = lds + TrainLds<NO>::MISC;

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int din = tk.din, no = tk.no;
  const long R = (long)tk.Rb * tk.A;
  const float invR = 1.0f / (float)R;

  mlp_fill_lds<NO>(lds, tk.params, din, no, 256);
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
  constexpr int NOH = (NO + 1) / 2;
  const int sf = tid & 127, og = tid >> 7;  // dW3: feature sf, outputs o = 2i + og
  float aW3[NOH];
#pragma unroll
  for (int i = 0; i < NOH; ++i) aW3[i] = 0.0f;
  float ab3 = 0.0f;   // tid < NO: db3[tid]
  float ab21 = 0.0f;  // tid < 128: db2[tid]; tid >= 128: db1[tid-128]
  float loss_a = 0.f, loss_b = 0.f;  // wave 0, half 0 lanes: actor (pg, entropy) / critic (value loss)

  const float* const wcol1 = tk.params + 32 * w + j;  // W1[k][32w + j] = wcol1[k*128]
  const int fbase = 32 * w + 4 * h;                   // + (r&3) + 8*(r>>2)

  const long ntiles = (R + 31) / 32;
  for (long it = blockIdx.x; it < ntiles; it += gridDim.x) {
    const long q = it * 32 + j;
    const bool valid = q < R;
    long fr = 0;
    if (valid) {
      const long b = q / tk.A;
      const int a = (int)(q - b * tk.A);
      const long p = tk.idx ? (long)tk.idx[b] : tk.idx_base + b;
      fr = p * tk.A + a;
    }
    const long xr = fr / tk.xshare;
    const float* xrow = tk.x + xr * din;
    if (w == 0 && h == 0) rowx[j] = xr;

    // ---------------------------------------------------------------- P1: layer 1, tile w
    f32x16 h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = lds[MlpLds<NO>::B1 + fbase + (r & 3) + 8 * (r >> 2)];
    if (tk.xv == 4) l1_tile<4, 4>(xrow, din, wcol1, h, h1);
    else if (tk.xv == 2) l1_tile<2, 6>(xrow, din, wcol1, h, h1);
    else l1_tile<1, 8>(xrow, din, wcol1, h, h1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h1[r] = fmaxf(h1[r], 0.0f);
      H1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = h1[r];
    }
    __syncthreads();  // A

    // ---------------------------------------------------------------- P2: layer 2, tile w
    f32x16 h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = lds[MlpLds<NO>::B2 + fbase + (r & 3) + 8 * (r >> 2)];
    {
      const float* wl = W2s + h * MLP_LDW + 32 * w + j;  // W2[k + h][32w + j]
      const float* hb = H1T + h * LDT + j;               // h1^T[k + h][row j]
#pragma unroll 16
      for (int k = 0; k < MLP_H; k += 2) h2 = MFMA32(wl[k * MLP_LDW], hb[k * LDT], h2);
    }
    {
      float part[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) part[o] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        h2[r] = fmaxf(h2[r], 0.0f);
        const int f = fbase + (r & 3) + 8 * (r >> 2);
        H2T[f * LDT + j] = h2[r];
        const float* w3 = W3s + f * NO;
#pragma unroll
        for (int o = 0; o < NO; ++o) part[o] = fmaf(h2[r], w3[o], part[o]);
      }
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const float v = part[o] + __shfl_xor(part[o], 32, 64);
        if (h == 0) YP[(w * NO + o) * 32 + j] = v;
      }
    }
    __syncthreads();  // B

    // ---------------------------------------------------------------- P3: loss, dlogits, dz2
    float dy[NO];
    {
      float y[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o)
        y[o] = (((YP[(0 * NO + o) * 32 + j] + YP[(1 * NO + o) * 32 + j]) + YP[(2 * NO + o) * 32 + j]) +
                YP[(3 * NO + o) * 32 + j]) + lds[MlpLds<NO>::B3 + o];
      if (ACTOR) {
        Categorical<NO> cat;
        cat.build(y, (tk.mask != nullptr && valid) ? (tk.mask + fr * no) : nullptr, no);
        const int act = valid ? tk.action[fr] : 0;
        float lp = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o)
          if (o == act) lp = cat.logp[o];
        const float old_lp = valid ? tk.old_logp[fr] : 0.0f;
        const float gae = valid ? (tk.adv[fr] - adv_mean) * adv_rstd : 0.0f;
        const float ratio = expf(lp - old_lp);
        const float lo = 1.0f - tk.clip_eps, hi = 1.0f + tk.clip_eps;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float l1 = ratio * gae, l2 = rc * gae;
        const float pg = -fminf(l1, l2);
        // d(-min(l1,l2))/d ratio: ties split evenly (lax.min); clip passes gradient inside the range
        const bool inside = (ratio >= lo) && (ratio <= hi);
        const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        const float dlp = valid ? (-(g1 + g2) * gae * ratio * invR) : 0.0f;
        const float ec = valid ? (tk.ent_coef * invR) : 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          const float oh = (o == act) ? 1.0f : 0.0f;
          const float pl = (cat.p[o] > 0.0f) ? cat.logp[o] : 0.0f;
          // -ent_coef * dH/dz_o = +ent_coef * p_o (log p_o + H)
          dy[o] = dlp * (oh - cat.p[o]) + ec * cat.p[o] * (pl + cat.entropy);
          if (cat.z[o] == -FLT_MAX) dy[o] = 0.0f;
        }
        if (valid && w == 0 && h == 0) {
          loss_a += pg * invR;
          loss_b += cat.entropy * invR;
        }
      } else {
        const float v = y[0];
        const float ov = valid ? tk.old_value[fr] : 0.0f;
        const float tg = valid ? tk.targets[fr] : 0.0f;
        const float diff = v - ov;
        const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
        const float e1 = v - tg, e2 = vclip - tg;
        const float l1 = e1 * e1, l2 = e2 * e2;
        const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
        const float g1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        dy[0] = valid ? (tk.vf_coef * (g1 * e1 + g2 * e2) * invR) : 0.0f;
        if (valid && w == 0 && h == 0) loss_a += 0.5f * fmaxf(l1, l2) * invR;
      }
      if (w == 0 && h == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) DY[o * 32 + j] = dy[o];
      }
    }
    f32x16 dz;  // dz2 of tile w
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fbase + (r & 3) + 8 * (r >> 2);
      const float* w3 = W3s + f * NO;
      float acc = 0.0f;
#pragma unroll
      for (int o = 0; o < NO; ++o) acc = fmaf(w3[o], dy[o], acc);
      dz[r] = (h2[r] > 0.0f) ? acc : 0.0f;
      DZ2T[f * LDT + j] = dz[r];
    }
    __syncthreads();  // C  (YP is dead from here; DZ1T may be written)

    // ---------------------------------------------------------------- P4: dh1 tile w, small grads
    f32x16 d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
    {
      const float* wl = W2s + (32 * w + j) * MLP_LDW + h;  // W2[32w + j][n + h]
      const float* db = DZ2T + h * LDT + j;                // dz2^T[n + h][row j]
#pragma unroll 16
      for (int n = 0; n < MLP_H; n += 2) d1 = MFMA32(wl[n], db[n * LDT], d1);
    }
    {
      // dW3[f][o] += sum_rows h2[f][row] * dy[o][row];  db3, db2
      const float* hrow = H2T + sf * LDT;
#pragma unroll 8
      for (int row = 0; row < 32; ++row) {
        const float hv = hrow[row];
#pragma unroll
        for (int i = 0; i < NOH; ++i) {
          const int o = 2 * i + og;
          if (o < NO) aW3[i] = fmaf(hv, DY[o * 32 + row], aW3[i]);
        }
      }
      if (tid < NO) {
        float s = 0.0f;
        for (int row = 0; row < 32; ++row) s += DY[tid * 32 + row];
        ab3 += s;
      }
      if (tid < 128) {
        float s = 0.0f;
        const float* zr = DZ2T + tid * LDT;
#pragma unroll 8
        for (int row = 0; row < 32; ++row) s += zr[row];
        ab21 += s;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      d1[r] = (h1[r] > 0.0f) ? d1[r] : 0.0f;
      DZ1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = d1[r];
    }
    __syncthreads();  // D

    // ---------------------------------------------------------------- P5: weight gradients
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
      }
    }
    {
      // gW1[k_in tile t][n] += sum_rows x[row][k_in] * dz1^T[n][row];  A straight from HBM/L2
      const float* eb = DZ1T + (32 * w + j) * LDT + h;
      float bz[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) bz[s] = eb[2 * s];
      const float* xp[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) xp[s] = tk.x + rowx[2 * s + h] * din + j;
      float ac[16], an[16];
      auto loadA = [&](int t, float (&ao)[16]) {
        const bool ok = (32 * t + j) < din;
#pragma unroll
        for (int s = 0; s < 16; ++s) ao[s] = ok ? xp[s][32 * t] : 0.0f;
      };
      loadA(0, ac);
#pragma unroll
      for (int t = 0; t < KT1; ++t) {
        if (t + 1 < KT1) loadA(t + 1, an);
#pragma unroll
        for (int s = 0; s < 16; ++s) gW1[t] = MFMA32(ac[s], bz[s], gW1[t]);
#pragma unroll
        for (int s = 0; s < 16; ++s) ac[s] = an[s];
      }
      if (tid >= 128) {
        float s = 0.0f;
        const float* zr = DZ1T + (tid - 128) * LDT;
#pragma unroll 8
        for (int row = 0; row < 32; ++row) s += zr[row];
        ab21 += s;
      }
    }
    __syncthreads();  // E: exchange tiles are free for the next row tile
  }

  // ------------------------------------------------------------------ epilogue: one slab per block
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const int oB1 = mlp_off_b1(din), oW2 = mlp_off_w2(din), oB2 = mlp_off_b2(din), oW3 = mlp_off_w3(din),
            oB3 = mlp_off_b3(din, no);
  const int P = mlp_param_count(din, no);
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = mlp_feat(t, r, h);
      if (k < din) slab[k * MLP_H + 32 * w + j] = gW1[t][r];
    }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[oW2 + mlp_feat(t, r, h) * MLP_H + 32 * w + j] = gW2[t][r];
#pragma unroll
  for (int i = 0; i < NOH; ++i) {
    const int o = 2 * i + og;
    if (o < no) slab[oW3 + sf * no + o] = aW3[i];
  }
  if (tid < no) slab[oB3 + tid] = ab3;
  if (tid < 128) slab[oB2 + tid] = ab21;
  else slab[oB1 + (tid - 128)] = ab21;
  if (w == 0) {
    for (int o = 32; o > 0; o >>= 1) {
      loss_a += __shfl_down(loss_a, o, 64);
      loss_b += __shfl_down(loss_b, o, 64);
    }
    if (lane == 0) {
      slab[P] = loss_a;
      slab[P + 1] = loss_b;
    }
  }
}

int pick_xv(const float* x, int din) {
  const uintptr_t a = (uintptr_t)x;
  if (din % 4 == 0 && a % 16 == 0) return 4;
  if (din % 2 == 0 && a % 8 == 0) return 2;
  return 1;
}

template <int NO, int KT1, bool ACTOR>
int launch_train(const TrainTask& tk, int n_slab, hipStream_t s) {
  static_assert(TrainLds<NO>::END * sizeof(float) <= 163840, "LDS budget exceeded");
  const size_t lb = (size_t)TrainLds<NO>::END * sizeof(float);
  MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)ppo_train_kernel<NO, KT1, ACTOR>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
  hipLaunchKernelGGL((ppo_train_kernel<NO, KT1, ACTOR>), dim3(n_slab), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, bool ACTOR>
int dispatch_kt(const TrainTask& tk, int n_slab, hipStream_t s) {
  const int kt = (tk.din + 31) / 32;
  switch (kt) {
    case 1: return launch_train<NO, 1, ACTOR>(tk, n_slab, s);
    case 2: return launch_train<NO, 2, ACTOR>(tk, n_slab, s);
    case 3: return launch_train<NO, 3, ACTOR>(tk, n_slab, s);
    case 4: return launch_train<NO, 4, ACTOR>(tk, n_slab, s);
    case 5: case 6: return launch_train<NO, 6, ACTOR>(tk, n_slab, s);
    case 7: case 8: case 9: return launch_train<NO, 9, ACTOR>(tk, n_slab, s);
    default:
      mava_set_error("ppo_train: input width %d > 288 is not instantiated", tk.din);
      return MAVA_EARG(9);
  }
}

}  // namespace

extern "C" int mava_adv_stats_blocks(void) { return STATS_BLOCKS; }

extern "C" int mava_adv_stats_f64(const float* adv, const int32_t* idx, long idx_base, int Rb, int A,
                                  double* partials, hipStream_t s) {
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1, 0, "mava_adv_stats_f64: Rb=%d A=%d", Rb, A);
  MAVA_ARG_CHECK(adv && partials, 1, "mava_adv_stats_f64: null pointer argument");
  hipLaunchKernelGGL(adv_stats_kernel, dim3(STATS_BLOCKS), dim3(256), 0, s, adv, idx, idx_base, Rb, A,
                     partials);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_ppo_actor_grad_f32(const float* params, int din, int n_actions,
                                       const float* agents_view, const uint8_t* action_mask,
                                       const int32_t* action, const float* old_log_prob,
                                       const float* advantages, const double* adv_stats,
                                       const int32_t* idx, long idx_base, int Rb, int A,
                                       float clip_eps, float ent_coef, float* slab, long slab_stride,
                                       int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && n_actions >= 1 && n_actions <= 32, 0,
                 "mava_ppo_actor_grad_f32: din=%d n_actions=%d unsupported", din, n_actions);
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024, 1,
                 "mava_ppo_actor_grad_f32: Rb=%d A=%d n_slab=%d", Rb, A, n_slab);
  MAVA_ARG_CHECK(slab_stride >= mlp_param_count(din, n_actions) + 2, 2,
                 "mava_ppo_actor_grad_f32: slab_stride too small");
  MAVA_ARG_CHECK(params && agents_view && action && old_log_prob && advantages && adv_stats && slab, 3,
                 "mava_ppo_actor_grad_f32: null pointer argument");
  TrainTask tk = {};
  tk.params = params; tk.x = agents_view; tk.din = din; tk.no = n_actions; tk.xshare = 1;
  tk.xv = pick_xv(agents_view, din); tk.A = A; tk.idx = idx; tk.idx_base = idx_base; tk.Rb = Rb;
  tk.mask = action_mask; tk.action = action; tk.old_logp = old_log_prob; tk.adv = advantages;
  tk.stats = adv_stats; tk.clip_eps = clip_eps; tk.ent_coef = ent_coef; tk.slab = slab;
  tk.slab_stride = slab_stride;
  if (n_actions <= 8) return dispatch_kt<8, true>(tk, n_slab, s);
  if (n_actions <= 16) return dispatch_kt<16, true>(tk, n_slab, s);
  return dispatch_kt<32, true>(tk, n_slab, s);
}

extern "C" int mava_ppo_critic_grad_f32(const float* params, int din, const float* critic_input,
                                        int x_share, const float* old_value, const float* targets,
                                        const int32_t* idx, long idx_base, int Rb, int A,
                                        float clip_eps, float vf_coef, float* slab, long slab_stride,
                                        int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(din >= 1 && x_share >= 1, 0, "mava_ppo_critic_grad_f32: din=%d x_share=%d", din, x_share);
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024, 1,
                 "mava_ppo_critic_grad_f32: Rb=%d A=%d n_slab=%d", Rb, A, n_slab);
  MAVA_ARG_CHECK(slab_stride >= mlp_param_count(din, 1) + 2, 2,
                 "mava_ppo_critic_grad_f32: slab_stride too small");
  MAVA_ARG_CHECK(params && critic_input && old_value && targets && slab, 3,
                 "mava_ppo_critic_grad_f32: null pointer argument");
  TrainTask tk = {};
  tk.params = params; tk.x = critic_input; tk.din = din; tk.no = 1; tk.xshare = x_share;
  tk.xv = pick_xv(critic_input, din); tk.A = A; tk.idx = idx; tk.idx_base = idx_base; tk.Rb = Rb;
  tk.old_value = old_value; tk.targets = targets; tk.clip_eps = clip_eps; tk.vf_coef = vf_coef;
  tk.slab = slab; tk.slab_stride = slab_stride;
  return dispatch_kt<1, false>(tk, n_slab, s);
}
