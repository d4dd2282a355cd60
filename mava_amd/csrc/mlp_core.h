// Register-resident 3-layer MLP building blocks on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Networks (reference: mava/networks.py:39-58 MLPTorso, :88-124 DiscreteActionHead,
// :172-207 FeedForwardActor / FeedForwardValueNet):  x -> Dense(Din,128) -> ReLU ->
// Dense(128,128) -> ReLU -> Dense(128,nO).  Flax Dense is y = x @ kernel + bias with kernel
// stored (in, out); flat parameter layout here is [W1 | b1 | W2 | b2 | W3 | b3], each W
// row-major (in, out).
//
// MI355X mapping (wave64, one wave owns 32 batch rows end to end):
//   Everything is computed TRANSPOSED: Z^T[feature][row] = W^T . X^T, so the MFMA accumulator
//   holds the batch row on the lane (col = lane & 31) and the feature in the register
//   (feature = 32*tile + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)).  That is exactly the B-operand
//   shape of the next layer's MFMA (B[k][col]: one f32 per lane, k chosen by lane >> 5), so the
//   activations of all three layers - and the back-propagated deltas - never leave registers:
//   accumulator register r of tile t feeds k-step r of the next product directly, with the
//   A operand (the weight) fetched for k = 32*t + (r&3) + 8*(r>>2) + 4*(lane>>5).
//   f32 MFMA issues every 64 cycles per SIMD, so one ds_read_b32 (A) per MFMA is far below the
//   LDS rate; W2 lives in LDS with an odd row stride (129) so both the forward pattern
//   (lanes walk the output index) and the backward pattern (lanes walk the input index) are
//   bank-conflict free.  W1 (up to 135 KB for the centralised critic) is streamed from L2 with
//   coalesced 128-B half-wave reads; X is read straight from HBM with per-lane vector loads.
#pragma once
#include <float.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_H 128     // hidden width (network/mlp.yaml: layer_sizes [128, 128])
#define MLP_LDW 129   // LDS row stride of W2 (floats), odd => conflict-free both ways
#define MLP_LDE 65    // LDS row stride of the 64-row exchange buffers

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// feature index held by accumulator register r of tile t on lane-half h
__device__ __forceinline__ int mlp_feat(int t, int r, int h) {
  return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

struct MlpDims {
  int din;  // input width
  int no;   // real number of outputs (<= NO template pad)
};

__host__ __device__ inline int mlp_param_count(int din, int no) {
  return din * MLP_H + MLP_H + MLP_H * MLP_H + MLP_H + MLP_H * no + no;
}
__host__ __device__ inline int mlp_off_b1(int din) { return din * MLP_H; }
__host__ __device__ inline int mlp_off_w2(int din) { return din * MLP_H + MLP_H; }
__host__ __device__ inline int mlp_off_b2(int din) { return din * MLP_H + MLP_H + MLP_H * MLP_H; }
__host__ __device__ inline int mlp_off_w3(int din) { return mlp_off_b2(din) + MLP_H; }
__host__ __device__ inline int mlp_off_b3(int din, int no) { return mlp_off_w3(din) + MLP_H * no; }

// LDS carve (floats) shared by the policy and train kernels.
template <int NO>
struct MlpLds {
  static constexpr int W2 = 0;
  static constexpr int W3 = W2 + MLP_H * MLP_LDW;
  static constexpr int B1 = W3 + MLP_H * NO;
  static constexpr int B2 = B1 + MLP_H;
  static constexpr int B3 = B2 + MLP_H;
  static constexpr int END = B3 + ((NO + 3) & ~3);
};

// Cooperative fill of W2 / W3 / biases (all threads of the block; caller syncs afterwards).
template <int NO>
__device__ __forceinline__ void mlp_fill_lds(float* lds, const float* __restrict__ params, int din,
                                             int no, int nthreads) {
  const float* W2g = params + mlp_off_w2(din);
  for (int i = threadIdx.x; i < MLP_H * MLP_H; i += nthreads) {
    const int k = i >> 7, n = i & 127;
    lds[MlpLds<NO>::W2 + k * MLP_LDW + n] = W2g[i];
  }
  const float* W3g = params + mlp_off_w3(din);
  for (int i = threadIdx.x; i < MLP_H * NO; i += nthreads) {
    const int f = i / NO, o = i - f * NO;
    lds[MlpLds<NO>::W3 + i] = (o < no) ? W3g[f * no + o] : 0.0f;
  }
  for (int i = threadIdx.x; i < MLP_H; i += nthreads) {
    lds[MlpLds<NO>::B1 + i] = params[mlp_off_b1(din) + i];
    lds[MlpLds<NO>::B2 + i] = params[mlp_off_b2(din) + i];
  }
  for (int i = threadIdx.x; i < NO; i += nthreads)
    lds[MlpLds<NO>::B3 + i] = (i < no) ? params[mlp_off_b3(din, no) + i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Layer 1:  z[t] = b1 + W1^T . x^T   (x row of this lane's batch row; W1 from global/L2)
// XV = per-lane vector width of the x loads (needs din % XV == 0 and 4*XV-byte aligned rows).
// The k order inside a chunk of 2*XV inputs is permuted (lane-half h takes inputs
// kb + XV*h .. kb + XV*h + XV-1); the A operand follows the same permutation.
// ---------------------------------------------------------------------------------------------
template <int XV>
struct XVec;
template <>
struct XVec<1> { typedef float T; };
template <>
struct XVec<2> { typedef float2 T; };
template <>
struct XVec<4> { typedef float4 T; };

template <int XV>
__device__ __forceinline__ void mlp_l1_forward(const float* __restrict__ xrow, int din,
                                               const float* __restrict__ W1g, const float* b1s,
                                               int h, int j, f32x16 (&z)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) z[t][r] = b1s[mlp_feat(t, r, h)];

  constexpr int STEP = 2 * XV;
  const int nfull = din / STEP;
  const float* wbase = W1g + j;

  // Two static operand buffers in ping-pong (no register copies next to the MFMAs, every load
  // unconditional so the compiler can keep counted vmcnt waits): chunk c+1 streams in while chunk c computes.
  float xa[XV], wa[XV][4], xb2[XV], wb2[XV][4];
  auto load = [&](int c, float (&xo)[XV], float (&wo)[XV][4]) {
    const int k0 = c * STEP + XV * h;
    typename XVec<XV>::T xv = *reinterpret_cast<const typename XVec<XV>::T*>(xrow + k0);
    const float* xs = reinterpret_cast<const float*>(&xv);
#pragma unroll
    for (int m = 0; m < XV; ++m) {
      xo[m] = xs[m];
#pragma unroll
      for (int t = 0; t < 4; ++t) wo[m][t] = wbase[(k0 + m) * MLP_H + 32 * t];
    }
  };
  auto compute = [&](const float (&xo)[XV], const float (&wo)[XV][4]) {
#pragma unroll
    for (int m = 0; m < XV; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) z[t] = MFMA32(wo[m][t], xo[m], z[t]);
  };
  if (nfull > 0) {
    load(0, xa, wa);
    int c = 0;
    for (; c + 1 < nfull; c += 2) {
      load(c + 1, xb2, wb2);
      compute(xa, wa);
      load((c + 2 < nfull) ? (c + 2) : (nfull - 1), xa, wa);  // clamped: the last one is a harmless reload
      compute(xb2, wb2);
    }
    if (c < nfull) compute(xa, wa);
  }
  // tail: remaining inputs in natural pairs (k, k+1), guarded
  for (int k0 = nfull * STEP; k0 < din; k0 += 2) {
    const int k = k0 + h;
    const bool ok = k < din;
    const int kc = ok ? k : (din - 1);
    const float xb = ok ? xrow[kc] : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float a = wbase[kc * MLP_H + 32 * t];
      z[t] = MFMA32(a, xb, z[t]);
    }
  }
}

__device__ __forceinline__ void mlp_relu(f32x16 (&z)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) z[t][r] = fmaxf(z[t][r], 0.0f);
}

// Layer 2:  z2[t'] = b2 + W2^T . h1^T, h1 supplied from accumulator registers (B operand).
__device__ __forceinline__ void mlp_l2_forward(const f32x16 (&h1)[4], const float* W2s,
                                               const float* b2s, int h, int j, f32x16 (&z2)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) z2[t][r] = b2s[mlp_feat(t, r, h)];
  const float* wl = W2s + (4 * h) * MLP_LDW + j;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = 32 * t + (r & 3) + 8 * (r >> 2);  // + 4h folded into wl
      const float b = h1[t][r];
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2) {
        const float a = wl[k * MLP_LDW + 32 * t2];
        z2[t2] = MFMA32(a, b, z2[t2]);
      }
    }
  }
}

// Back-propagation through layer 2:  dh1[t'] = W2 . dz2^T  (dz2 from registers as B operand),
// A[i = input feature 32t'+i][k = output feature n] = W2[32t'+i][n]  (lanes walk the W2 row index).
__device__ __forceinline__ void mlp_l2_backward(const f32x16 (&dz2)[4], const float* W2s, int h,
                                                int j, f32x16 (&dh1)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh1[t][r] = 0.0f;
  const float* wl = W2s + j * MLP_LDW + 4 * h;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * t + (r & 3) + 8 * (r >> 2);  // + 4h folded into wl
      const float b = dz2[t][r];
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2) {
        const float a = wl[(32 * t2) * MLP_LDW + n];
        dh1[t2] = MFMA32(a, b, dh1[t2]);
      }
    }
  }
}

// Output head on the VALU: y[o] = b3[o] + sum_f h2[f] * W3[f][o].  Each lane-half holds 64 of
// the 128 features of its row; the halves are combined with one cross-half exchange.
template <int NO>
__device__ __forceinline__ void mlp_head_forward(const f32x16 (&h2)[4], const float* W3s,
                                                 const float* b3s, int h, float (&y)[NO]) {
  float part[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) part[o] = 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float hv = h2[t][r];
      const float* w = W3s + mlp_feat(t, r, h) * NO;
#pragma unroll
      for (int o = 0; o < NO; ++o) part[o] = fmaf(hv, w[o], part[o]);
    }
#pragma unroll
  for (int o = 0; o < NO; ++o) y[o] = (part[o] + __shfl_xor(part[o], 32, 64)) + b3s[o];
}

// dh2[f] = sum_o W3[f][o] * dy[o], masked by ReLU'(h2) and written over h2 (=> dz2).
template <int NO>
__device__ __forceinline__ void mlp_head_backward_inplace(f32x16 (&h2)[4], const float* W3s, int h,
                                                          const float (&dy)[NO]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* w = W3s + mlp_feat(t, r, h) * NO;
      float acc = 0.0f;
#pragma unroll
      for (int o = 0; o < NO; ++o) acc = fmaf(w[o], dy[o], acc);
      h2[t][r] = (h2[t][r] > 0.0f) ? acc : 0.0f;
    }
}

// Transposing reduction over the 32 lanes of each wave half: given 32 per-lane values v[0..31],
// lane q (q = lane & 31) of each half returns sum over that half's lanes of v[q].
// 31 exchanges instead of 32*5; fixed tree => deterministic.
__device__ __forceinline__ float half_transpose_reduce32(float (&v)[32], int j) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const bool up = (j & 16) != 0;
    const float send = up ? v[i] : v[i + 16];
    const float keep = up ? v[i + 16] : v[i];
    v[i] = keep + __shfl_xor(send, 16, 64);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool up = (j & 8) != 0;
    const float send = up ? v[i] : v[i + 8];
    const float keep = up ? v[i + 8] : v[i];
    v[i] = keep + __shfl_xor(send, 8, 64);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool up = (j & 4) != 0;
    const float send = up ? v[i] : v[i + 4];
    const float keep = up ? v[i + 4] : v[i];
    v[i] = keep + __shfl_xor(send, 4, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool up = (j & 2) != 0;
    const float send = up ? v[i] : v[i + 2];
    const float keep = up ? v[i + 2] : v[i];
    v[i] = keep + __shfl_xor(send, 2, 64);
  }
  {
    const bool up = (j & 1) != 0;
    const float send = up ? v[0] : v[1];
    const float keep = up ? v[1] : v[0];
    v[0] = keep + __shfl_xor(send, 1, 64);
  }
  return v[0];
}
// After the call lane q of half h holds the sum for value index q; with values taken as
// v[16*(t - 2g) + r] = x[t][r] (g = tile pair), that is feature mlp_feat(2g + (q >> 4), q & 15, h).

// Masked categorical over NO padded logits (reference: networks.py:116-124 +
// distributions.py:146-165 / tfd.Categorical).  Illegal or padded actions get finfo(f32).min.
template <int NO>
struct Categorical {
  float z[NO];    // masked logits
  float logp[NO];
  float p[NO];
  float entropy;

  __device__ __forceinline__ void build(const float (&y)[NO], const uint8_t* mask, int no) {
    uint32_t bits = 0xFFFFFFFFu;
    if (mask != nullptr) {
      bits = 0;
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (o < no && mask[o] != 0) bits |= (1u << o);
    }
    build_bits(y, bits, no);
  }

  // legality given as a bit per action
  __device__ __forceinline__ void build_bits(const float (&y)[NO], uint32_t bits, int no) {
    float mx = -FLT_MAX;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const bool legal = (o < no) && ((bits >> o) & 1u);
      z[o] = legal ? y[o] : -FLT_MAX;
      mx = fmaxf(mx, z[o]);
    }
    float se = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) se += expf(z[o] - mx);
    const float lse = mx + logf(se);
    entropy = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      logp[o] = z[o] - lse;
      p[o] = expf(logp[o]);
      entropy -= (p[o] > 0.0f) ? p[o] * logp[o] : 0.0f;
    }
  }
};
