// Block-cooperative forward body shared by mlp_coop.hip (stand-alone kernels) and mlp_policy.hip (hybrid acting
// step: per-wave actor blocks + cooperative critic blocks in ONE launch).  See mlp_coop.hip for the design notes.
#pragma once
#include "mlp_core.h"

namespace coop {

constexpr int LDT = 33;

struct CoopTask {
  const float* params;
  const float* x;     // (rows_x, din)
  int din, no, xshare;
  int R;              // agent rows
  // actor extras
  const uint8_t* mask;          // (R, no) or null
  const int32_t* forced_action; // (R) or null
  uint32_t seed_lo, seed_hi, step, row_offset;
  const uint32_t* step_base;    // optional device word added to step
  int greedy;
  int32_t* action;    // (R)
  float* log_prob;    // (R)
  float* logits;      // (R, no) or null
  // value extras
  float* value;       // (R * vbroadcast)
  int vbroadcast;
  // raw extras
  float* out;         // (R, no)
};

struct CoopLds {
  int h1t, yp, xs, end, ldx;
};

template <int NO>
CoopLds make_coop_layout(int kt1) {
  CoopLds L;
  L.h1t = MlpLds<NO>::END;
  L.yp = L.h1t + MLP_H * LDT;
  L.ldx = 32 * kt1 + 1;
  L.xs = L.yp + 4 * NO * 32;
  L.end = L.xs + 32 * L.ldx + 32;
  return L;
}

enum { MODE_RAW = 0, MODE_SAMPLE = 1, MODE_VALUE = 2 };

// Body of the block-cooperative forward: block `bid` of `nblk` walks the 32-row tiles bid, bid + nblk, ...
// (called by mlp_coop_kernel with the whole grid, and by policy_hybrid_kernel with the critic's share of a grid
// whose other blocks run the per-wave actor)
template <int NO, int KT1, int MODE>
__device__ __forceinline__ void coop_body(const CoopTask& tk, const CoopLds& L, float* lds, int bid, int nblk) {
  float* const W2s = lds + MlpLds<NO>::W2;
  float* const W3s = lds + MlpLds<NO>::W3;
  float* const H1T = lds + L.h1t;
  float* const YP = lds + L.yp;
  float* const XS = lds + L.xs;
  const int ldx = L.ldx;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int srow = tid >> 3, l8 = tid & 7;
  const int din = tk.din, no = tk.no;
  const int R = tk.R;
  constexpr int NR = 4 * KT1;
  constexpr int NB = 2 * KT1;
  constexpr int RD = (NB % 3 == 0) ? 3 : ((NB % 4 == 0) ? 4 : 2);

  mlp_fill_lds<NO>(lds, tk.params, din, no, 256);
  for (int i = tid; i < 32 * ldx + 32; i += 256) XS[i] = 0.0f;

  const float* const wcol1h = tk.params + h * MLP_H + 32 * w + j;
  float wr[RD][8];
#pragma unroll
  for (int d = 0; d < RD; ++d)
#pragma unroll
    for (int s = 0; s < 8; ++s) wr[d][s] = wcol1h[(16 * d + 2 * s) * MLP_H];
  const int fbase = 32 * w + 4 * h;
  const int nfull = din >> 3;

  auto stage_issue = [&](int tile, float (&xr)[NR]) {
    int q = tile * 32 + srow;
    q = q < R ? q : (R - 1);
    const float* xrow = tk.x + (long)((uint32_t)q / (uint32_t)tk.xshare) * din + l8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i < nfull) xr[i] = xrow[8 * i];
      else if (i == nfull && l8 + 8 * i < din) xr[i] = xrow[8 * i];
      else xr[i] = 0.0f;
    }
  };
  auto stage_commit = [&](const float (&xr)[NR]) {
    float* xs = XS + srow * ldx + l8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i < nfull) xs[8 * i] = xr[i];
      else if (i == nfull && l8 + 8 * i < din) xs[8 * i] = xr[i];
    }
  };

  const int ntiles = (R + 31) / 32;
  int it = bid;
  float xr[NR];
  __syncthreads();  // XS zero fill done
  if (it < ntiles) {
    stage_issue(it, xr);
    stage_commit(xr);
  }
  __syncthreads();

  for (; it < ntiles; it += nblk) {
    const int itn = it + nblk;
    const bool have_next = itn < ntiles;

    // ---------------------------------------------------------------- layer 1, tile w
    f32x16 h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = lds[MlpLds<NO>::B1 + fbase + (r & 3) + 8 * (r >> 2)];
    {
      const float* xb = XS + j * ldx + h;
      float xo[RD][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xo[0][s] = xb[2 * s];
#pragma unroll 1
      for (int g = 0; g < NB / RD; ++g) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
          const int b = g * RD + d;
          const int bn = (b + 1 < NB) ? (b + 1) : b;
#pragma unroll
          for (int s = 0; s < 8; ++s) xo[(d + 1) % RD][s] = xb[16 * bn + 2 * s];
#pragma unroll
          for (int s = 0; s < 8; ++s) h1 = MFMA32(wr[d][s], xo[d][s], h1);
          int bf = b - 1 + RD;
          bf = (bf >= NB) ? (bf - NB) : bf;
          const float* wb = wcol1h + bf * (16 * MLP_H);
#pragma unroll
          for (int s = 0; s < 8; ++s) wr[(d + RD - 1) % RD][s] = wb[(2 * s) * MLP_H];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[RD - 1][s] = wcol1h[(16 * (RD - 1) + 2 * s) * MLP_H];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) H1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = fmaxf(h1[r], 0.0f);
    __syncthreads();  // A
    if (have_next) stage_issue(itn, xr);

    // ---------------------------------------------------------------- layer 2, tile w + partial head
    f32x16 h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = lds[MlpLds<NO>::B2 + fbase + (r & 3) + 8 * (r >> 2)];
    {
      const float* wl = W2s + h * MLP_LDW + 32 * w + j;
      const float* hb = H1T + h * LDT + j;
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
    {
      float part[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) part[o] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float hv = fmaxf(h2[r], 0.0f);
        const float* w3 = W3s + (fbase + (r & 3) + 8 * (r >> 2)) * NO;
#pragma unroll
        for (int o = 0; o < NO; ++o) part[o] = fmaf(hv, w3[o], part[o]);
      }
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const float v = part[o] + __shfl_xor(part[o], 32, 64);
        if (h == 0) YP[(w * NO + o) * 32 + j] = v;
      }
    }
    __syncthreads();  // B

    // ---------------------------------------------------------------- epilogue (wave 0, one lane per row)
    if (w == 0 && h == 0) {
      const int row = it * 32 + j;
      const bool valid = row < R;
      float y[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o)
        y[o] = (((YP[(0 * NO + o) * 32 + j] + YP[(1 * NO + o) * 32 + j]) + YP[(2 * NO + o) * 32 + j]) +
                YP[(3 * NO + o) * 32 + j]) + lds[MlpLds<NO>::B3 + o];
      if (MODE == MODE_RAW) {
        if (valid)
          for (int o = 0; o < no && o < NO; ++o) tk.out[(long)row * no + o] = y[o];
      } else if (MODE == MODE_VALUE) {
        if (valid)
          for (int b = 0; b < tk.vbroadcast; ++b) tk.value[(long)row * tk.vbroadcast + b] = y[0];
      } else {
        Categorical<NO> cat;
        cat.build(y, (tk.mask != nullptr && valid) ? (tk.mask + (long)row * no) : nullptr, no);
        int a = 0;
        if (tk.forced_action != nullptr) {
          a = valid ? tk.forced_action[row] : 0;
        } else if (tk.greedy) {
          float best = -FLT_MAX;
#pragma unroll
          for (int o = 0; o < NO; ++o)
            if (o < no && cat.z[o] > best) { best = cat.z[o]; a = o; }
        } else {
          // Gumbel-max: argmax_o z[o] - log(-log(u_o)), first index wins ties
          float best = -FLT_MAX;
          const uint32_t gid = tk.row_offset + (uint32_t)row;
#pragma unroll
          for (int c = 0; c < (NO + 3) / 4; ++c) {
            Philox4 rnd = philox4x32_10(gid, tk.step + (tk.step_base ? *tk.step_base : 0u), (uint32_t)c,
                                        0x504f4c49u /*"POLI"*/, tk.seed_lo, tk.seed_hi);
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
        if (valid) {
          tk.action[row] = a;
          tk.log_prob[row] = lp;
          if (tk.logits != nullptr)
            for (int o = 0; o < no && o < NO; ++o) tk.logits[(long)row * no + o] = y[o];
        }
      }
    }
    __syncthreads();  // E: H1T / YP / XS free
    if (have_next) stage_commit(xr);
    __syncthreads();  // F
  }
}


}  // namespace coop
