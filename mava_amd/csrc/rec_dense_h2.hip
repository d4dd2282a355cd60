// Recurrent path, split-f16 ("f16x2") forms of the dense layer and the X^T Y weight-gradient product
// (rec_dense.hip holds the exact-f32 forms and the ABI entry points that dispatch here; mava/networks.py:238-331).
//
// Every f32 operand is split into two f16 terms (hi + lo, ~22 bits) where it enters a matrix product and the product
// is three v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi, f32 accumulation): 96 instead of 512 matrix-pipe cycles per
// 16 inputs.  Both kernels read their T32 operands STRAIGHT from memory in MFMA operand layout - no LDS, no barriers:
//  * dense:  lane (row j, half h) of a 16-input batch reads features 16b + 2e + h, e = 0..7 (eight coalesced
//    128-byte accesses per wave-half), exactly the accesses of the f32 kernel; the weight slice of a wave is split
//    once per launch (error diffusion along the summation index) and stays in registers.
//  * X^T Y:  the product sums over batch rows, and a T32 tile holds the 32 rows of a feature contiguously: lane
//    (feature j, half h) reads rows 16s + 8h .. + 7 as two 16-byte loads - the operand fragment of k-step s.
// Backward operands (loss gradients) are O(1/rows): the caller runs the backward chain in units scaled by a power of
// two (mava_seq_*_loss_f32 grad_scale) so that they sit in f16's normal range, and the X^T Y epilogue multiplies the
// inverse back in (XtyTask::out_scale); powers of two are exact in f32, so the f32 kernels are unaffected.
// mava_rec_gather_t32_f32 turns the row-major, env-permuted observation slice of a minibatch into a T32 matrix once
// per minibatch (features padded to a multiple of 16 with zeros), which the pre-torso product AND its weight-gradient
// product then read as plain T32 operands.
#include "h2_core.h"
#include "rec_task.h"

namespace {

using h2::Frag;
using h2::half8;

#define DOFF(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)

// NB = 16-input batches of K, NTW = 32-feature output tiles per wave (tiles w, w+4, w+8).
// FULLK: the x tiles hold exactly 16*NB features (compile-time operand offsets); else (NB <= 2) clamped loads.
template <int NB, int NTW, bool FULLK>
__global__ __launch_bounds__(256, 1) void rec_dense_h2_kernel(DenseTask tk) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int K = tk.K, N = tk.N;
  const int KX = tk.x_ld;  // features per x tile (>= 16 NB when FULLK)
  const int ntile_n = (N + 31) / 32;

  // ---- resident weight slice, split once: wf[tw][b] element e = W[16b + 2e + h][32(w + 4tw) + j]
  Frag wf[NTW][NB];
  {
    float wreg[NTW][NB][8];
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int col = 32 * (w + 4 * tw) + j;
      const int colc = col < N ? col : (N - 1);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * b + 2 * e + h;
          const int kc = k < K ? k : (K - 1);
          wreg[tw][b][e] = tk.w[(long)kc * tk.ldw + colc];
        }
    }
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int col = 32 * (w + 4 * tw) + j;
      float carry = 0.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * b + 2 * e + h;
          wreg[tw][b][e] = (k < K && col < N) ? wreg[tw][b][e] : 0.0f;
        }
        wf[tw][b] = h2::split8_carry(wreg[tw][b], carry);
      }
    }
  }

  const int ntiles = tk.rows / 32;
  auto epilogue = [&](int it, f32x16 (&acc)[NTW], const float (&gpre)[16], bool use_pre) {
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int fb = 32 * (w + 4 * tw) + 4 * h;
      const long base = ((long)it * N + fb) * 32 + j;
      float* const yo = tk.y + base;
      const float* const go = tk.gate != nullptr ? tk.gate + base : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fb + (r & 3) + 8 * (r >> 2);
        if (f < N) {
          float v = acc[tw][r];
          if (tk.relu) v = fmaxf(v, 0.0f);
          if (go != nullptr) {
            const float gv = (use_pre && tw == 0) ? gpre[r] : go[DOFF(r)];
            v = (gv > 0.0f) ? v : 0.0f;
          }
          yo[DOFF(r)] = v;
        }
      }
    }
  };
  auto load_gate = [&](int it, float (&g)[16]) {
    const int fb = 32 * w + 4 * h;
    const float* const go = tk.gate + ((long)it * N + fb) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = go[(fb + (r & 3) + 8 * (r >> 2) < N) ? DOFF(r) : 0];
  };
  // the bias stays in registers for the whole launch: a per-tile reload sits in front of the tile's first MFMA, and
  // with in-order memory returns it also waits for the whole operand prefetch ring
  float breg[NTW][16];
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * (w + 4 * tw) + 4 * h + (r & 3) + 8 * (r >> 2);
      breg[tw][r] = (tk.bias != nullptr && f < N) ? tk.bias[f] : 0.0f;
    }
  auto init_acc = [&](int it, f32x16 (&acc)[NTW]) {
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int fb = 32 * (w + 4 * tw) + 4 * h;
      const float* const yo = tk.y + ((long)it * N + fb) * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fb + (r & 3) + 8 * (r >> 2);
        float a0 = breg[tw][r];
        if (tk.accumulate && f < N) a0 += yo[DOFF(r)];
        acc[tw][r] = a0;
      }
    }
  };
  auto tile_ptr = [&](int it) { return tk.x + ((long)it * KX + (FULLK ? h : 0)) * 32 + j; };
  auto load_batch = [&](const float* xp, int bq, float (&dst)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (FULLK) {
        dst[e] = xp[(16 * bq + 2 * e) * 32];  // elem(row j, k + h): h folded into the tile pointer
      } else {
        int k = 16 * bq + 2 * e + h;
        k = k < KX ? k : (KX - 1);  // weights past K are zero
        dst[e] = xp[(long)k * 32];
      }
    }
  };
  const bool pf_gate = (NTW == 1) && tk.gate != nullptr && (32 * w < N);

  if (NB <= 2) {
    // narrow inputs (the n_out-wide loss gradient of the head backward): a tile is 3 NB MFMAs, far shorter than a
    // memory round trip - the whole x tile and the gate values run PF tiles ahead through rotating registers
    constexpr int PF = 3;
    float xq[PF + 1][NB][8], gq[PF + 1][16];
    int it = blockIdx.x;
#pragma unroll
    for (int d = 1; d <= PF; ++d) {
      const int itd = it + (d - 1) * (int)gridDim.x;
      const int itc = itd < ntiles ? itd : (ntiles - 1);
#pragma unroll
      for (int b = 0; b < NB; ++b) load_batch(tile_ptr(itc), b, xq[d][b]);
      if (pf_gate) load_gate(itc, gq[d]);
    }
    for (; it < ntiles; it += gridDim.x) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 8; ++e) xq[d][b][e] = xq[d + 1][b][e];
#pragma unroll
        for (int r = 0; r < 16; ++r) gq[d][r] = gq[d + 1][r];
      }
      {
        const int itd = it + PF * (int)gridDim.x;
        const int itc = itd < ntiles ? itd : (ntiles - 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) load_batch(tile_ptr(itc), b, xq[PF][b]);
        if (pf_gate) load_gate(itc, gq[PF]);
      }
      f32x16 acc[NTW];
      init_acc(it, acc);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const Frag xf = h2::split8(xq[0][b]);
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw)
          acc[tw] = h2::mfma3(wf[tw][b], xf, acc[tw]);  // n-tiles past N run on clamped operands and are never stored
      }
      epilogue(it, acc, gq[0], pf_gate);
    }
    return;
  }

  // ---- operand batches stream through a ring that runs ACROSS row tiles (prefetch distance RDX - 1 batches)
  constexpr int RDX = (NB <= 2) ? 1 : (NB % 8 == 0) ? 8 : ((NB % 6 == 0) ? 6 : ((NB % 5 == 0) ? 5 : ((NB % 4 == 0) ? 4 : 2)));
  constexpr int PD = RDX > 1 ? RDX - 1 : 0;
  static_assert(NB % RDX == 0, "ring depth must divide the batch count");
  float xo[RDX][8];
  float gcur[16], gnext[16];
  int it = blockIdx.x;
  const float* xt_cur = tile_ptr(it < ntiles ? it : 0);
#pragma unroll
  for (int d = 0; d < PD; ++d) load_batch(xt_cur, d % NB, xo[d]);
  if (pf_gate && it < ntiles) load_gate(it, gnext);
  for (; it < ntiles; it += gridDim.x) {
    const int itn = it + gridDim.x;
    const float* xt_next = (itn < ntiles) ? tile_ptr(itn) : xt_cur;
    if (pf_gate) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gcur[r] = gnext[r];
      load_gate(itn < ntiles ? itn : it, gnext);
    }
    f32x16 acc[NTW];
    init_acc(it, acc);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      {
        const int bq = b + PD;
        if (bq < NB) load_batch(xt_cur, bq, xo[bq % RDX]);
        else load_batch(xt_next, bq - NB, xo[bq % RDX]);
      }
      const Frag xf = h2::split8(xo[b % RDX]);
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw)
        acc[tw] = h2::mfma3(wf[tw][b], xf, acc[tw]);
      __builtin_amdgcn_sched_barrier(0);
    }
    epilogue(it, acc, gcur, pf_gate);
    xt_cur = xt_next;
  }
}

// dW = X^T Y, db = colsum(Y) for T32 X (KX features per tile, the first K used) and T32 Y.
// KT = 32-feature tiles of X (accumulator rows), NTW = 32-feature tiles of Y per wave (tiles w, w+4, w+8).
template <int KT, int NTW>
__global__ __launch_bounds__(256, 1) void rec_xty_h2_kernel(XtyTask tk) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int K = tk.K, N = tk.N, KX = tk.x_ld;
  const int ntile_n = (N + 31) / 32;

  f32x16 acc[KT][NTW];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][tw][r] = 0.0f;
  float bsum[NTW];
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) bsum[tw] = 0.0f;

  // per-lane float4 offsets inside a tile: feature (clamped to the matrix), rows 8h .. 8h + 7 of k-step 0
  int xoff[KT], yoff[NTW];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int f = 32 * kt + j;
    xoff[kt] = ((f < KX ? f : (KX - 1)) * 32 + 8 * h) / 4;
  }
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
    const int f = 32 * (w + 4 * tw) + j;
    yoff[tw] = ((f < N ? f : (N - 1)) * 32 + 8 * h) / 4;
  }
  // raw operands of one tile: [k-step s][feature tile][2 x float4]
  float4 xr[2][KT][2], yr[2][NTW][2];
  auto issue = [&](int it, float4 (&xd)[2][KT][2], float4 (&yd)[2][NTW][2]) {
    const float4* xs = reinterpret_cast<const float4*>(tk.x + (long)it * KX * 32);
    const float4* ys = reinterpret_cast<const float4*>(tk.y + (long)it * N * 32);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        xd[s][kt][0] = xs[xoff[kt] + 4 * s];
        xd[s][kt][1] = xs[xoff[kt] + 4 * s + 1];
      }
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) {
        yd[s][tw][0] = ys[yoff[tw] + 4 * s];
        yd[s][tw][1] = ys[yoff[tw] + 4 * s + 1];
      }
    }
  };
  auto to8 = [](const float4& a, const float4& b, float (&v)[8]) {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };

  const int ntiles = tk.rows / 32;
  int it = blockIdx.x;
  if (it < ntiles) issue(it, xr, yr);
  for (; it < ntiles; it += gridDim.x) {
    const int itn = it + gridDim.x;
    // this tile's operands leave the landing registers as split fragments (X) / a raw copy (Y) before the next tile's
    // loads are issued into them
    Frag xf[2][KT];
    float4 yc[2][NTW][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        float v[8];
        to8(xr[s][kt][0], xr[s][kt][1], v);
        xf[s][kt] = h2::split8(v);
      }
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) { yc[s][tw][0] = yr[s][tw][0]; yc[s][tw][1] = yr[s][tw][1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(itn < ntiles ? itn : it, xr, yr);  // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) {
        float v[8];
        to8(yc[s][tw][0], yc[s][tw][1], v);
        if (tk.want_bias) bsum[tw] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const Frag yf = h2::split8(v);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[kt][tw] = h2::mfma3(xf[s][kt], yf, acc[kt][tw]);  // tiles past N: never stored
      }
  }
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const float sc = tk.out_scale;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = 32 * (w + 4 * tw) + j;
        if (k < K && n < N) slab[(long)k * N + n] = acc[kt][tw][r] * sc;
      }
  if (tk.want_bias) {
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const float b = bsum[tw] + __shfl_xor(bsum[tw], 32, 64);
      const int n = 32 * (w + 4 * tw) + j;
      if (h == 0 && n < N) slab[(long)K * N + n] = b * sc;
    }
  }
}

// Row-major, env-permuted (time-major) source -> T32 matrix with KP >= K features per tile (zeros past K).
// One block per 32-row tile pass: rows are read in 256-byte runs, transposed through LDS, written feature-major.
__global__ __launch_bounds__(256) void rec_gather_t32_kernel(DenseTask tk, int KP) {
  extern __shared__ float xs[];  // [32][ld]
  const int K = tk.K;
  const int ld = KP | 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ntiles = tk.rows / 32;
  for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * w + rr;
      const float* xrow = tk.x + gather_row(tk, it * 32 + r) * tk.x_ld;
      for (int c = lane; c < KP; c += 64) xs[r * ld + c] = c < K ? xrow[c] : 0.0f;
    }
    __syncthreads();
    float* out = tk.y + (long)it * KP * 32;
    for (int i = tid; i < KP * 32; i += 256) out[i] = xs[(i & 31) * ld + (i >> 5)];
    __syncthreads();
  }
}

template <int NB, int NTW, bool FULLK>
int launch_dense_h2(const DenseTask& tk, hipStream_t s) {
  int blocks = tk.rows / 32;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL((rec_dense_h2_kernel<NB, NTW, FULLK>), dim3(blocks), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int KT, int NTW>
int launch_xty_h2(const XtyTask& tk, int n_slab, hipStream_t s) {
  hipLaunchKernelGGL((rec_xty_h2_kernel<KT, NTW>), dim3(n_slab), dim3(256), 0, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

int mava_rec_dense_h2_launch(const DenseTask& tk, hipStream_t s) {
  if (tk.x_rowmajor) return 1;
  const int nb = (tk.K + 15) / 16;
  const int ntw = ((tk.N + 31) / 32 + 3) / 4;
  if (nb <= 2 && ntw == 1) {  // x tiles of exactly tk.x_ld features, any K <= 32
    if (nb == 1) return launch_dense_h2<1, 1, false>(tk, s);
    return launch_dense_h2<2, 1, false>(tk, s);
  }
  // NB batches read 16 NB features of every x tile: the tile must hold them (weights past K are zero)
#define DENSE_H2(NBv, NTWv) \
  if (nb <= NBv && ntw == NTWv && tk.x_ld >= 16 * NBv) return launch_dense_h2<NBv, NTWv, true>(tk, s)
  DENSE_H2(4, 1); DENSE_H2(6, 1); DENSE_H2(8, 1); DENSE_H2(10, 1); DENSE_H2(12, 1); DENSE_H2(18, 1); DENSE_H2(24, 1);
  DENSE_H2(8, 3);
#undef DENSE_H2
  return 1;
}

int mava_rec_xty_h2_launch(const XtyTask& tk, int n_slab, hipStream_t s) {
  if (tk.x_rowmajor) return 1;
  const int kt = (tk.K + 31) / 32;
  const int ntw = ((tk.N + 31) / 32 + 3) / 4;
#define XTY_H2(KTv, NTWv) \
  if (kt <= KTv && ntw == NTWv) return launch_xty_h2<KTv, NTWv>(tk, n_slab, s)
  XTY_H2(3, 1); XTY_H2(4, 1); XTY_H2(5, 1); XTY_H2(6, 1); XTY_H2(9, 1); XTY_H2(4, 3);
#undef XTY_H2
  return 1;
}

extern "C" int mava_rec_gather_t32_f32(const float* x, const int32_t* idx, int Rm, int E, int A, int x_share, int x_ld,
                                       int K, int rows, int k_pad, float* out, hipStream_t s) {
  MAVA_ARG_CHECK(K >= 1 && k_pad >= K && k_pad <= 1024 && x_ld >= K, 0, "mava_rec_gather_t32_f32: K=%d k_pad=%d x_ld=%d", K,
                 k_pad, x_ld);
  MAVA_ARG_CHECK(rows >= 0 && rows % 32 == 0, 1, "mava_rec_gather_t32_f32: rows=%d must be a multiple of 32", rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(x && out, 2, "mava_rec_gather_t32_f32: null pointer argument");
  MAVA_ARG_CHECK(Rm >= 1 && A >= 1 && E >= 1 && x_share >= 1 && rows % Rm == 0, 3,
                 "mava_rec_gather_t32_f32: bad gather description Rm=%d E=%d A=%d", Rm, E, A);
  DenseTask tk = {};
  tk.x = x; tk.x_rowmajor = 1; tk.idx = idx; tk.Rm = Rm; tk.E = E; tk.A = A; tk.xshare = x_share; tk.x_ld = x_ld;
  tk.y = out; tk.K = K; tk.rows = rows;
  const size_t lb = (size_t)32 * (k_pad | 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_gather_t32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  int blocks = rows / 32;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rec_gather_t32_kernel, dim3(blocks), dim3(256), lb, s, tk, k_pad);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
