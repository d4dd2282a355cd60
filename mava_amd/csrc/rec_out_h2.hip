// The OUTPUT PATH of the recurrent networks as one fused kernel on split-f16 operands:
//   post_torso (128 -> 128, relu) -> head (128 -> n_out) -> PPO loss -> d loss / d outputs -> backward through head and
//   post_torso to the gradient reaching the GRU's hidden states, plus the weight / bias gradients of both layers.
// Reference: mava/networks.py:283-294 / :322-331 (post_torso, action head / Dense(1) after the ScannedRNN),
// mava/systems/ppo/rec_mappo.py:210-242 (actor loss), :244-266 (critic loss) and their jax.grad.
//
// Layer-wise, this path was eight launches per network and minibatch (two dense layers, the loss, two backward dense
// layers, two X^T Y products and a few reductions) that move ~5.3 KB per row-step; fused it reads the hidden states once
// (512 B) and writes their gradient once (512 B).  Structure = the fused feed-forward gradient kernel
// (ppo_train_h2.hip) around the streaming input stage of rec_dense_h2.hip:
//  * the T32 tile of hidden states is read with 16-byte loads a tile ahead, split once, staged as f16 hi / lo planes
//    [feature][32 rows] (swizzled, double-buffered); the post_torso product reads it through ds_read_b64_tr_b16, the
//    Wpost gradient through ds_read_b128;
//  * wave w owns features [32w, 32w + 32) of post / dpost / dh; W_post (both orientations) and the head fragments stay in
//    registers as split fragments; the head product uses the layer accumulator itself as its operand;
//  * activations cross waves as LDS images [row][feature] (h2_core.h); the loss runs one (row, output) pair per lane;
//  * gradients accumulate in registers over the block's tiles and leave as per-block slabs
//    [dWpost | dbpost | dWhead | dbhead | loss sums] = the tail of the flat parameter layout.
// Gradient units: d loss / d outputs is formed times grad_scale (rec_dense_h2.hip), dh is written in those units, the
// slabs are multiplied by 1 / grad_scale.  Discrete actor heads up to 16 actions and the critic (one value, optionally the
// sum of `agg` agents' loss gradients per row); the continuous head stays on the layer-wise kernels.
#include <float.h>

#include "h2_core.h"
#include "rec_task.h"

namespace {

using namespace h2;

#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                    \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                                  \
    st_prev = st_now;                                               \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif

struct OutTask {
  int T, Rm, E, A, no, agg;
  const int32_t* idx;
  const float* hs;          // T32 (T*Rm x 128)
  const float* params;      // [Wpost (128 x 128) | bpost | Whead (128 x no) | bhead]
  const uint8_t* mask;      // external (T, E, A, no) or null
  const int32_t* action;    // external (T, E, A)
  const float* f0;          // actor: old_log_prob ; critic: old_value
  const float* f1;          // actor: advantages   ; critic: targets
  const double* stats;      // advantage statistics partials (actor)
  int n_stats;
  float clip_eps, coef, grad_scale;
  float* dh;                // T32 (T*Rm x 128): d loss / d hs, times grad_scale
  float* slab;              // (gridDim.x, slab_stride)
  long slab_stride;
  unsigned long long* stamps;  // diagnostic builds (-DMAVA_STAMPS): per-phase cycle sums of block 0, wave 0
};

constexpr int XPL = 128 * 64;       // bytes per staged plane
constexpr int XBUF = 2 * XPL;

__device__ __forceinline__ int stage_off(int q) {
  const int f = q >> 3, c = (q & 7) >> 1;
  return f * 64 + ((c ^ ((f >> 2) & 3)) << 4) + 8 * (q & 1);
}
__device__ __forceinline__ void stage4(u8* plane_hi, int off, const float4& v) {
  half4 ph, pl;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 a, b;
    split1(vv[e], a, b);
    ph[e] = a;
    pl[e] = b;
  }
  *reinterpret_cast<half4*>(plane_hi + off) = ph;
  *reinterpret_cast<half4*>(plane_hi + XPL + off) = pl;
}
__device__ __forceinline__ Frag rows_frag(const u8* plane_hi, int f, int c) {
  const int off = f * 64 + ((c ^ ((f >> 2) & 3)) << 4);
  Frag r;
  r.hi = *reinterpret_cast<const half8*>(plane_hi + off);
  r.lo = *reinterpret_cast<const half8*>(plane_hi + XPL + off);
  return r;
}

template <int NO, bool ACTOR>
__global__ __launch_bounds__(256, 1) void rec_out_h2_kernel(OutTask tk) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  u8* const STG = lds;                              // 2 x XBUF
  u8* const POSTI = STG + 2 * XBUF;                 // IMG_BYTES
  u8* const DPOSTI = POSTI + IMG_BYTES;             // IMG_BYTES
  u8* const DYI = DPOSTI + IMG_BYTES;               // 2 x DY_PLANE
  float* const YP = reinterpret_cast<float*>(DYI + 2 * DY_PLANE);  // [4][32][NO + 1]
  float* const misc = YP + 4 * 32 * (NO + 1);       // [0] adv mean, [1] adv rstd, [2 ..] reductions, then bhead
  float* const B3s = misc + 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int no = tk.no;
  const float* const Wpost = tk.params;
  const float* const bpost = Wpost + MLP_H * MLP_H;
  const float* const Whead = bpost + MLP_H;
  const float* const bhead = Whead + MLP_H * no;
  const long R = (long)tk.T * tk.Rm;
  const int na = (!ACTOR && tk.agg > 1) ? tk.agg : 1;
  const float invR = 1.0f / (float)(R * na);
  const float gsc = invR * tk.grad_scale;  // d loss / d output per row, in grad_scale units

  // ---------------------------------------------------------------- resident fragments
  Frag wfa[8];  // post forward:  element e = Wpost[16b + 8h + e][32w + r]
  Frag wfb[8];  // dh = Wpost . dpost^T:  element e = Wpost[32w + r][16s + 8h + e]
  {
    float ca = 0.0f, cb = 0.0f;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      float va[8], vb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        va[e] = Wpost[(16 * b + 8 * h + e) * MLP_H + 32 * w + r];
        vb[e] = Wpost[(32 * w + r) * MLP_H + 16 * b + 8 * h + e];
      }
      wfa[b] = split8_carry(va, ca);
      wfb[b] = split8_carry(vb, cb);
    }
  }
  Frag W3h[2];  // head: logits^T[o = r][row] over this wave's 32 features, operand = the layer accumulator itself
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 32 * w + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
      v[e] = (r < no) ? Whead[f * no + r] * W3_SCALE : 0.0f;
    }
    W3h[s] = split8(v);
  }
  Frag W3d;  // dpost = Whead . dy:  element e = Whead[32w + r][o = 8h + e]
  {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = 8 * h + e;
      v[e] = (o < no) ? Whead[(32 * w + r) * no + o] * W3_SCALE : 0.0f;
    }
    W3d = split8(v);
  }
  float bp[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bp[q] = bpost[32 * w + 4 * h + (q & 3) + 8 * (q >> 2)];
  if (tid < NO) B3s[tid] = tid < no ? bhead[tid] : 0.0f;
  if (ACTOR && tid == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < tk.n_stats; ++i) { s1 += tk.stats[2 * i]; s2 += tk.stats[2 * i + 1]; }
    const double mean = s1 / (double)R;
    double var = s2 / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    misc[0] = (float)mean;
    misc[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
  }
  for (int i = tid; i < 2 * DY_PLANE / 4; i += 256) reinterpret_cast<uint32_t*>(DYI)[i] = 0u;  // outputs >= NO stay zero

  // persistent accumulators (grad_scale units)
  f32x16 gWp[4], gW3;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) gWp[t][q] = 0.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) gW3[q] = 0.0f;
  float ab2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ab2[q] = 0.0f;
  float ab3[NO / 8], loss_a = 0.0f, loss_b = 0.0f;
#pragma unroll
  for (int k = 0; k < NO / 8; ++k) ab3[k] = 0.0f;

  // ---------------------------------------------------------------- lane mappings
  // loss: eight lanes per row (all 32 rows of the tile in one pass), lane l8 of a row owns outputs l8, l8 + 8, ...
  constexpr int NP = 1;
  constexpr int OPL = NO / 8;
  const int l8 = lane & 7;
  auto loss_row = [&](int) -> int { return 8 * w + (lane >> 3); };
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, g1 = (lane >> 4) & 1;
  const int trI = (8 * h + tq) * IMG_ROW + 2 * (16 * g1 + 4 * tp);
  const int trD = (8 * h + tq) * DY_ROW + 2 * (16 * g1 + 4 * tp);
  const int chunk = 2 * g1 + (tp >> 1), o8 = 8 * (tp & 1);
  const int tr0 = (8 * h + tq) * 64 + ((chunk ^ (2 * h)) << 4) + o8;            // staged planes: + 1024 b
  const int tr1 = (8 * h + tq + 4) * 64 + ((chunk ^ (2 * h + 1)) << 4) + o8;
  auto x_frag = [&](const u8* buf, int b) {
    Frag f;
    const u8* p0 = buf + tr0 + 1024 * b;
    const u8* p1 = buf + tr1 + 1024 * b;
    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0));
    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1));
    const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0 + XPL));
    const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1 + XPL));
    f.hi = __builtin_bit_cast(half8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
    f.lo = __builtin_bit_cast(half8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  };
  float4 raw[4];
  int soff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) soff[k] = stage_off(tid + 256 * k);
  auto issue = [&](long it, int k) { raw[k] = reinterpret_cast<const float4*>(tk.hs + it * MLP_H * 32)[tid + 256 * k]; };

  const long ntiles = R / 32;
  const long G = gridDim.x;
  long it = blockIdx.x;
  long n_er[NP];
  int n_act[NP];
  float n_f0[NP], n_f1[NP];
  uint32_t n_m[OPL];
  // two-deep pipeline: the env ids (idx) of a tile are loaded two tiles ahead, the values behind them one tile ahead -
  // neither address waits for memory
  uint32_t p_t[NP], p_a[NP], p_env[NP];
  auto load_env = [&](long itx) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const uint32_t qq = (uint32_t)itx * 32u + (uint32_t)loss_row(q);
      const uint32_t t = qq / (uint32_t)tk.Rm, m = qq - t * (uint32_t)tk.Rm;
      const uint32_t e_local = m / (uint32_t)tk.A;
      p_t[q] = t;
      p_a[q] = m - e_local * (uint32_t)tk.A;
      p_env[q] = tk.idx ? (uint32_t)tk.idx[e_local] : e_local;
    }
  };
  auto load_rows = [&]() {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      n_er[q] = ((long)p_t[q] * tk.E + p_env[q]) * tk.A + p_a[q];
      n_act[q] = 0;
      n_f0[q] = n_f1[q] = 0.0f;
#pragma unroll
      for (int k = 0; k < OPL; ++k) n_m[k] = 1u;
      if (ACTOR) {
        n_act[q] = tk.action[n_er[q]];
        n_f0[q] = tk.f0[n_er[q]];
        n_f1[q] = tk.f1[n_er[q]];
#pragma unroll
        for (int k = 0; k < OPL; ++k)
          if (tk.mask != nullptr && l8 + 8 * k < no) n_m[k] = tk.mask[n_er[q] * no + l8 + 8 * k];
      }
    }
  };
  load_env(it < ntiles ? it : 0);
  load_rows();
  load_env(it + G < ntiles ? it + G : (ntiles - 1));
  __syncthreads();
  const float adv_mean = ACTOR ? misc[0] : 0.0f, adv_rstd = ACTOR ? misc[1] : 0.0f;
  if (it < ntiles) {
#pragma unroll
    for (int k = 0; k < 4; ++k) issue(it, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) stage4(STG, soff[k], raw[k]);
  }
  {
    const long itn = it + G;
#pragma unroll
    for (int k = 0; k < 4; ++k) issue(itn < ntiles ? itn : (ntiles - 1), k);
  }
  __syncthreads();
  int cur = 0;
  STAMP_DECL
  for (; it < ntiles; it += G) {
    STAMP(7);
    const u8* const rb = STG + cur * XBUF;
    u8* const wbuf = STG + (cur ^ 1) * XBUF;
    const long itnn = it + 2 * G;
    const long it_issue = itnn < ntiles ? itnn : (ntiles - 1);
    // loss inputs of this lane's (row, output) pairs: they sit behind a dependent index load, so they are requested a
    // whole tile ahead (n_*: the next tile's, loaded below) and rotate in here
    long er[NP];
    int r_act[NP];
    float r_f0[NP], r_f1[NP];
    uint32_t r_m[OPL];
#pragma unroll
    for (int q = 0; q < NP; ++q) { er[q] = n_er[q]; r_act[q] = n_act[q]; r_f0[q] = n_f0[q]; r_f1[q] = n_f1[q]; }
#pragma unroll
    for (int k = 0; k < OPL; ++k) r_m[k] = n_m[k];
    load_rows();                                                   // tile it + G (its env ids arrived a tile ago)
    load_env(it + 2 * G < ntiles ? it + 2 * G : (ntiles - 1));     // tile it + 2G

    // ---------------------------------------------------------------- P1: post = relu(Wpost^T hs^T + b), head partials
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = bp[q];
    {
      Frag xf[2];
      xf[0] = x_frag(rb, 0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b + 1 < 8) xf[(b + 1) & 1] = x_frag(rb, b + 1);
        acc = mfma3(wfa[b], xf[b & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    uint32_t relu2 = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q] = fmaxf(acc[q], 0.0f);
      relu2 |= (acc[q] > 0.0f) ? (1u << q) : 0u;
    }
    {
      half4 ph[4], pl[4];
      write_image(POSTI, r, 32 * w + 4 * h, acc, ph, pl);
      f32x16 yacc;
#pragma unroll
      for (int q = 0; q < 16; ++q) yacc[q] = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag b;
        b.hi = __builtin_shufflevector(ph[2 * s], ph[2 * s + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        b.lo = __builtin_shufflevector(pl[2 * s], pl[2 * s + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        yacc = mfma3(W3h[s], b, yacc);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
        if ((q & 3) + 8 * (q >> 2) < NO) {
          if (o < NO) YP[(w * 32 + r) * (NO + 1) + o] = yacc[q] * W3_UNSCALE;
        }
      }
    }
    STAMP(0);
    __syncthreads();  // B: partial logits and the post image complete
    STAMP(1);

    // ---------------------------------------------------------------- P3: loss, d loss / d outputs (grad_scale units)
    {
      const float lo_c = 1.0f - tk.clip_eps, hi_c = 1.0f + tk.clip_eps;
      const int row = loss_row(0);
      float y[OPL], dyo[OPL];
#pragma unroll
      for (int k = 0; k < OPL; ++k) {
        const int o = l8 + 8 * k;
        const float* yp = YP + row * (NO + 1) + o;
        y[k] = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) + B3s[o];
        dyo[k] = 0.0f;
      }
      if (ACTOR) {
        auto fmax_op = [](float a, float b) { return fmaxf(a, b); };
        auto add_op = [](float a, float b) { return a + b; };
        // masked Categorical over the row's outputs (networks.py:116-124, distributions.py:146-165)
        float z[OPL], logp[OPL], pr[OPL];
        float mx = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
          const bool legal = (l8 + 8 * k < no) && (r_m[k] != 0u);
          z[k] = legal ? y[k] : -FLT_MAX;
          mx = fmaxf(mx, z[k]);
        }
        mx = group_allreduce<8>(mx, fmax_op);
        float se = 0.0f;
#pragma unroll
        for (int k = 0; k < OPL; ++k) se += expf(z[k] - mx);
        se = group_allreduce<8>(se, add_op);
        const float lse = mx + logf(se);
        const int act = r_act[0];
        float ent = 0.0f, lp = 0.0f;
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
          logp[k] = z[k] - lse;
          pr[k] = expf(logp[k]);
          ent += (pr[k] > 0.0f) ? -(pr[k] * logp[k]) : 0.0f;
          lp += (l8 + 8 * k == act) ? logp[k] : 0.0f;
        }
        ent = group_allreduce<8>(ent, add_op);
        lp = group_allreduce<8>(lp, add_op);
        const float gae = (r_f1[0] - adv_mean) * adv_rstd;
        const float ratio = expf(lp - r_f0[0]);
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float l1 = ratio * gae, l2 = rc * gae;
        const bool inside = (ratio >= lo_c) && (ratio <= hi_c);
        const float g1w = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2w = inside ? (1.0f - g1w) : 0.0f;
        const float dlp = -(g1w + g2w) * gae * ratio * gsc;
        const float ec = tk.coef * gsc;
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
          const float oh = (l8 + 8 * k == act) ? 1.0f : 0.0f;
          const float pl2 = (pr[k] > 0.0f) ? logp[k] : 0.0f;
          dyo[k] = dlp * (oh - pr[k]) + ec * pr[k] * (pl2 + ent);
          if (z[k] == -FLT_MAX) dyo[k] = 0.0f;
        }
        if (l8 == 0) {
          loss_a += -fminf(l1, l2) * invR;
          loss_b += ent * invR;
        }
      } else if (l8 == 0) {
        // one value per row; with agg > 1 the row is shared by its agg agents: their loss gradients add up
        for (int a2 = 0; a2 < na; ++a2) {
          const long ea = er[0] * na + a2;
          const float ov = tk.f0[ea], tg = tk.f1[ea];
          const float diff = y[0] - ov;
          const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
          const float e1 = y[0] - tg, e2 = vclip - tg;
          const float l1 = e1 * e1, l2 = e2 * e2;
          const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
          const float g1w = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
          const float g2w = inside ? (1.0f - g1w) : 0.0f;
          dyo[0] += tk.coef * (g1w * e1 + g2w * e2) * gsc;
          loss_a += 0.5f * fmaxf(l1, l2) * invR;
        }
      }
#pragma unroll
      for (int k = 0; k < OPL; ++k) {
        _Float16 da, db;
        split1(dyo[k], da, db);
        *reinterpret_cast<_Float16*>(DYI + row * DY_ROW + 2 * (l8 + 8 * k)) = da;
        *reinterpret_cast<_Float16*>(DYI + DY_PLANE + row * DY_ROW + 2 * (l8 + 8 * k)) = db;
        ab3[k] += dyo[k];
      }
    }
    STAMP(2);
    __syncthreads();  // B2: dy of all 32 rows visible; every reader of the partial logits is done

    // ---------------------------------------------------------------- P4: dWhead, dpost
    f32x16 dz;
    {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag a = read_tr_frag(DYI + trD + 16 * s * DY_ROW, DY_PLANE, DY_ROW);
        const Frag b = read_tr_frag(POSTI + trI + 16 * s * IMG_ROW + 2 * (32 * w), IMG_PLANE, IMG_ROW);
        gW3 = mfma3(a, b, gW3);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) dz[q] = 0.0f;
      const Frag b = read_row_frag(DYI, DY_PLANE, r * DY_ROW + 16 * h);
      dz = mfma3(W3d, b, dz);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dz[q] = ((relu2 >> q) & 1u) ? dz[q] * W3_UNSCALE : 0.0f;
        ab2[q] += dz[q];
      }
      half4 ph[4], pl[4];
      write_image(DPOSTI, r, 32 * w + 4 * h, dz, ph, pl);
    }
    STAMP(3);
    __syncthreads();  // C: dpost image complete
    STAMP(4);

    // ---------------------------------------------------------------- P5: dh = Wpost dpost^T (stored), dWpost; next tile staged
    {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
      // two independent chains share the phase: dh (8 steps on one accumulator) and dWpost (2 row steps x 4 feature tiles);
      // step s of the first is issued next to product s of the second
      Frag bn = read_row_frag(DPOSTI, IMG_PLANE, r * IMG_ROW + 16 * h);
      Frag gb = read_tr_frag(DPOSTI + trI + 2 * (32 * w), IMG_PLANE, IMG_ROW);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const Frag b = bn;
        if (s + 1 < 8) bn = read_row_frag(DPOSTI, IMG_PLANE, r * IMG_ROW + 16 * h + 32 * (s + 1));
        const int s2 = s >> 2, t = s & 3;
        const Frag ga = rows_frag(rb, 32 * t + r, 2 * s2 + h);
        acc = mfma3(wfb[s], b, acc);
        gWp[t] = mfma3(ga, gb, gWp[t]);
        if (s == 3) gb = read_tr_frag(DPOSTI + trI + 16 * IMG_ROW + 2 * (32 * w), IMG_PLANE, IMG_ROW);
        if (s & 1) {  // half a slot of the next tile's commit per two steps, and the loads of the tile after it
          const int k = s >> 1;
          stage4(wbuf, soff[k], raw[k]);
          issue(it_issue, k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float* const dho = tk.dh + ((it * MLP_H + 32 * w + 4 * h) * 32 + r);
#pragma unroll
      for (int q = 0; q < 16; ++q) dho[((q & 3) + 8 * (q >> 2)) * 32] = acc[q];
    }
    STAMP(5);
    __syncthreads();  // D: images and the staging buffers change hands
    STAMP(6);
    cur ^= 1;
  }

#ifdef MAVA_STAMPS
  if (tk.stamps != nullptr && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) tk.stamps[i] = st_acc[i];
#endif
  // ---------------------------------------------------------------- epilogue: slabs (true units)
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const float inv = 1.0f / tk.grad_scale;
  const int oWp = 0, oBp = MLP_H * MLP_H, oW3 = oBp + MLP_H, oB3 = oW3 + MLP_H * no, oL = oB3 + no;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      slab[oWp + (32 * t + (q & 3) + 8 * (q >> 2) + 4 * h) * MLP_H + 32 * w + r] = gWp[t][q] * inv;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float v = ab2[q];
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if (r == 0) slab[oBp + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h] = v * inv;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
    if (o < no) slab[oW3 + (32 * w + r) * no + o] = gW3[q] * inv;
  }
  float* red = misc + 2;
  {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NO / 8; ++k) {  // lanes with the same l8 hold output l8 + 8k of different rows
      float v = ab3[k];
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
      if (lane < 8) red[w * NO + lane + 8 * k] = v;
    }
    for (int o = 32; o > 0; o >>= 1) {
      loss_a += __shfl_down(loss_a, o, 64);
      loss_b += __shfl_down(loss_b, o, 64);
    }
    float* redl = reinterpret_cast<float*>(POSTI);
    if (lane == 0) { redl[2 * w] = loss_a; redl[2 * w + 1] = loss_b; }
    __syncthreads();
    if (tid < no) slab[oB3 + tid] = (((red[tid] + red[NO + tid]) + red[2 * NO + tid]) + red[3 * NO + tid]) * inv;
    if (tid == 0) {
      slab[oL] = ((redl[0] + redl[2]) + redl[4]) + redl[6];
      slab[oL + 1] = ((redl[1] + redl[3]) + redl[5]) + redl[7];
    }
  }
}

template <int NO, bool ACTOR>
int launch_out(const OutTask& tk, int n_slab, hipStream_t s) {
  const size_t lb = (size_t)2 * XBUF + 2 * IMG_BYTES + 2 * DY_PLANE + (size_t)(4 * 32 * (NO + 1) + 64 + 32) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_out_h2_kernel<NO, ACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    attr_set = true;
  }
  hipLaunchKernelGGL((rec_out_h2_kernel<NO, ACTOR>), dim3(n_slab), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

static unsigned long long* g_out_stamps = nullptr;
// Diagnostic hook (not part of include/mava_hip.h): device buffer of 8 u64 for -DMAVA_STAMPS builds.
extern "C" int mava_debug_set_out_stamps(unsigned long long* p) {
  g_out_stamps = p;
  return MAVA_OK;
}

// Returns 1 when the shape is not instantiated (the caller runs the layer-wise kernels).
extern "C" int mava_rec_out_f32(int T, int Rm, int E, int A, int n_out, int agents_per_row, const int32_t* idx, const float* hs,
                                const float* params_post_head, const uint8_t* mask, const int32_t* action, const float* f0,
                                const float* f1, const double* adv_stats, int n_stats, float clip_eps, float coef, float grad_scale,
                                int is_actor, float* dh, float* slab, long slab_stride, int n_slab, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 1 && Rm >= 32 && Rm % 32 == 0 && E >= 1 && A >= 1 && n_out >= 1 && n_slab >= 1, 0, "mava_rec_out_f32: bad shape");
  MAVA_ARG_CHECK(hs && params_post_head && f0 && f1 && dh && slab && (!is_actor || (action && adv_stats)), 1,
                 "mava_rec_out_f32: null pointer argument");
  MAVA_ARG_CHECK(slab_stride >= (long)MLP_H * MLP_H + MLP_H + (long)MLP_H * n_out + n_out + 2, 2, "mava_rec_out_f32: slab_stride too small");
  MAVA_ARG_CHECK(is_actor || agents_per_row == 1 || A == 1, 3, "mava_rec_out_f32: agents_per_row > 1 needs A == 1");
  if (n_out > 16 || (!is_actor && n_out != 1) || (long)n_slab > (long)T * Rm / 32) return 1;
  OutTask tk = {};
  tk.T = T; tk.Rm = Rm; tk.E = E; tk.A = A; tk.no = n_out; tk.agg = agents_per_row; tk.idx = idx; tk.hs = hs;
  tk.params = params_post_head; tk.mask = mask; tk.action = action; tk.f0 = f0; tk.f1 = f1; tk.stats = adv_stats;
  tk.n_stats = n_stats; tk.clip_eps = clip_eps; tk.coef = coef; tk.grad_scale = grad_scale; tk.dh = dh; tk.slab = slab;
  tk.slab_stride = slab_stride;
  tk.stamps = g_out_stamps;
  if (!is_actor) return launch_out<8, false>(tk, n_slab, s);
  if (n_out <= 8) return launch_out<8, true>(tk, n_slab, s);
  return launch_out<16, true>(tk, n_slab, s);
}
