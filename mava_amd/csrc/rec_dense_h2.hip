// Recurrent path, split-f16 ("f16x2") forms of the dense layer and the X^T Y weight-gradient product
// (rec_dense.hip holds the exact-f32 forms and the ABI entry points that dispatch here; mava/networks.py:238-331).
//
// Every f32 operand is split into two f16 terms (hi + lo, ~22 bits) where it enters a matrix product and the product
// is three v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi, f32 accumulation): 96 instead of 512 matrix-pipe cycles per
// 16 inputs - at that rate both kernels are HBM-bound, and they are built as streaming kernels:
//  * a T32 tile (32 rows) is ONE contiguous run of K*32 floats: the block reads it with 16-byte loads, each byte once
//    per CU, a full tile ahead (the loads of tile i+2 are issued while tile i is multiplied);
//  * the loading thread splits its four values once and stores them as f16 hi / lo planes in LDS, [feature][32 rows]
//    (64 bytes per feature, 16-byte chunks XOR-swizzled by the feature index so that neither access below conflicts);
//  * X^T Y sums over batch rows: an operand fragment (8 consecutive rows of a feature) is one ds_read_b128 per plane;
//    the dense layer sums over features: its fragment (8 consecutive features of a row) comes from the same layout
//    through the hardware-transposed ds_read_b64_tr_b16 - one staging format serves both;
//  * LDS is double-buffered: the commit of tile i+1 is sliced between the MFMA groups of tile i (VALU and LDS writes
//    run under the matrix pipe) and a tile costs one block barrier.
// The weight slice of a dense wave is split once per launch (error diffusion along the summation index) and stays in
// registers.  Backward operands (loss gradients) are O(1/rows): the caller runs the backward chain in units scaled by a
// power of two (mava_seq_*_loss_f32 grad_scale) so that they sit in f16's normal range, and the X^T Y epilogue
// multiplies the inverse back in (XtyTask::out_scale); powers of two are exact in f32.
// mava_rec_gather_t32_f32 turns the row-major, env-permuted observation slice of a minibatch into a T32 matrix once
// per minibatch (features padded to a multiple of 32 with zeros), which the pre-torso product AND its weight-gradient
// product then read as plain T32 operands.
#include <float.h>

#include "h2_core.h"
#include "rec_task.h"

namespace {

using h2::Frag;
using h2::half4;
using h2::half8;
using h2::u8;
using h2::s16x4;

#define DOFF(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)

// ---- staging: float4 number q of a T32 tile = rows 4 (q & 7) .. + 3 of feature q >> 3 -> 8 bytes in each plane
__device__ __forceinline__ int stage_off(int q) {
  const int f = q >> 3, c = (q & 7) >> 1;
  return f * 64 + ((c ^ ((f >> 2) & 3)) << 4) + 8 * (q & 1);
}
__device__ __forceinline__ void stage4(u8* plane_hi, int plane_bytes, int off, const float4& v) {
  half4 ph, pl;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 a, b;
    h2::split1(vv[e], a, b);
    ph[e] = a;
    pl[e] = b;
  }
  *reinterpret_cast<half4*>(plane_hi + off) = ph;
  *reinterpret_cast<half4*>(plane_hi + plane_bytes + off) = pl;
}
// rows 8c .. 8c + 7 of feature f (the fragment of a product that sums over batch rows)
__device__ __forceinline__ Frag rows_frag(const u8* plane_hi, int plane_bytes, int f, int c) {
  const int off = f * 64 + ((c ^ ((f >> 2) & 3)) << 4);
  Frag r;
  r.hi = *reinterpret_cast<const half8*>(plane_hi + off);
  r.lo = *reinterpret_cast<const half8*>(plane_hi + plane_bytes + off);
  return r;
}

// Y = act(X W + b) [gated] for T32 X (x_ld features per tile, the first K used).
// NB = 16-input batches of K, NTW = 32-feature output tiles per wave (tiles w, w+4, w+8).
// LEAN (the K = 384 instantiation, whose 192 weight registers leave no slack): no bias registers (the caller passes no
// bias), staging offsets recomputed instead of held, the gate values of a tile requested at its start instead of a tile
// ahead - the 48 registers this frees were spilled and reloaded every tile otherwise (and a reload drains the prefetch).
template <int NB, int NTW, bool LEAN = false>
__global__ __launch_bounds__(256, 1) void rec_dense_h2_kernel(DenseTask tk) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  constexpr int XF = 16 * NB;          // staged features
  constexpr int XPL = XF * 64;         // bytes per plane
  constexpr int BUF = 2 * XPL;         // hi, lo
  constexpr int XS = (XF * 8 + 255) / 256;  // float4 slots per thread and tile
  const int tid = threadIdx.x;
  const int lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: conditions on it are scalar branches
  const int K = tk.K, N = tk.N;
  const int nx4 = K * 8;

  // ---- resident weight slice, split once: wf[tw][b] element e = W[16b + 8h + e][32(w + 4tw) + j]
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
          const int k = 16 * b + 8 * h + e;
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
          const int k = 16 * b + 8 * h + e;
          wreg[tw][b][e] = (k < K && col < N) ? wreg[tw][b][e] : 0.0f;
        }
        wf[tw][b] = h2::split8_carry(wreg[tw][b], carry);
      }
    }
  }
  // features past K are never staged: they must read as zeros (their weights are zero, 0 * garbage may be NaN)
  for (int i = tid; i < 2 * BUF / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  float breg[LEAN ? 1 : NTW][16];
#pragma unroll
  for (int tw = 0; tw < (LEAN ? 0 : NTW); ++tw)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * (w + 4 * tw) + 4 * h + (r & 3) + 8 * (r >> 2);
      breg[tw][r] = (tk.bias != nullptr && f < N) ? tk.bias[f] : 0.0f;
    }

  const int ntiles = tk.rows / 32;
  // transposed operand reads: lane (row j, half h) receives features 16b + 8h + 0..7 of row j from two 4-feature blocks
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, g1 = (lane >> 4) & 1;
  const int chunk = 2 * g1 + (tp >> 1), o8 = 8 * (tp & 1);
  const int tr0 = (8 * h + tq) * 64 + ((chunk ^ (2 * h)) << 4) + o8;            // + 1024 b
  const int tr1 = (8 * h + tq + 4) * 64 + ((chunk ^ (2 * h + 1)) << 4) + o8;
  auto x_frag = [&](const u8* buf, int b) {
    Frag f;
    const u8* p0 = buf + tr0 + 1024 * b;
    const u8* p1 = buf + tr1 + 1024 * b;
    const h2::s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0));
    const h2::s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1));
    const h2::s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0 + XPL));
    const h2::s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1 + XPL));
    f.hi = __builtin_bit_cast(half8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
    f.lo = __builtin_bit_cast(half8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  };

  float4 raw[XS];
  int soff[LEAN ? 1 : XS];
#pragma unroll
  for (int k = 0; k < (LEAN ? 0 : XS); ++k) soff[k] = stage_off(tid + 256 * k);
  auto issue = [&](int it, int k) {
    const float4* xs = reinterpret_cast<const float4*>(tk.x + (long)it * tk.x_ld * 32);
    const int q = tid + 256 * k;
    raw[k] = xs[q < nx4 ? q : (nx4 - 1)];
  };
  // branch-free: a slot past the K used features stages zeros (its loads are clamped duplicates)
  auto commit = [&](u8* buf, int k) {
    const bool in = tid + 256 * k < nx4;
    float4 v = raw[k];
    v.x = in ? v.x : 0.0f; v.y = in ? v.y : 0.0f; v.z = in ? v.z : 0.0f; v.w = in ? v.w : 0.0f;
    // an odd NB stages 128 NB float4: the upper half of the block has no feature row in the last slot
    if (!(NB & 1) || k < XS - 1 || tid < 128) stage4(buf, XPL, LEAN ? stage_off(tid + 256 * k) : soff[k], v);
  };
  // Output tile: activation, optional relu-mask gate, coalesced T32 stores.  A wave whose 32 features all exist takes a
  // straight-line path (wave-uniform scalar branches only): per-element bounds checks and pointer tests cost a tile more
  // issue slots than its matrix products.
  auto epilogue = [&](int it, f32x16 (&acc)[NTW], const float (&gpre)[16], bool use_pre) {
    const float floor_v = tk.relu ? 0.0f : -FLT_MAX;
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int fb = 32 * (w + 4 * tw) + 4 * h;
      const long base = ((long)it * tk.y_ld + fb) * 32 + j;
      float* const yo = tk.y + base;
      const float* const go = tk.gate != nullptr ? tk.gate + base : nullptr;
      if (32 * (w + 4 * tw) + 32 <= N) {
        if (go == nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) yo[DOFF(r)] = fmaxf(acc[tw][r], floor_v);
        } else {
          float gv[16];
          if (use_pre && tw == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[r] = gpre[r];
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[r] = go[DOFF(r)];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) yo[DOFF(r)] = (gv[r] > 0.0f) ? fmaxf(acc[tw][r], floor_v) : 0.0f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = fb + (r & 3) + 8 * (r >> 2);
          if (f < N) {
            float v = fmaxf(acc[tw][r], floor_v);
            if (go != nullptr) {
              const float gv = (use_pre && tw == 0) ? gpre[r] : go[DOFF(r)];
              v = (gv > 0.0f) ? v : 0.0f;
            }
            yo[DOFF(r)] = v;
          }
        }
      }
    }
  };
  auto load_gate = [&](int it, float (&g)[16]) {
    const int fb = 32 * w + 4 * h;
    const float* const go = tk.gate + ((long)it * tk.y_ld + fb) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = go[(fb + (r & 3) + 8 * (r >> 2) < N) ? DOFF(r) : 0];
  };
  const bool pf_gate = (NTW == 1) && tk.gate != nullptr && (32 * w < N);

  const int G = gridDim.x;
  int it = blockIdx.x;
  __syncthreads();  // zero fill done
  if (it < ntiles) {
#pragma unroll
    for (int k = 0; k < XS; ++k) issue(it, k);
#pragma unroll
    for (int k = 0; k < XS; ++k) commit(lds, k);
  }
  {
    const int itn = it + G;
#pragma unroll
    for (int k = 0; k < XS; ++k) issue(itn < ntiles ? itn : (ntiles - 1), k);
  }
  float gcur[16], gnext[LEAN ? 1 : 16];
  if constexpr (!LEAN) {
    if (pf_gate && it < ntiles) load_gate(it, gnext);
  }
  __syncthreads();
  int cur = 0;
  for (; it < ntiles; it += G) {
    const u8* const rb = lds + cur * BUF;
    u8* const wbuf = lds + (cur ^ 1) * BUF;
    const int itn = it + G, itnn = it + 2 * G;
    const bool has_next = itn < ntiles;
    const int it_issue = itnn < ntiles ? itnn : (ntiles - 1);
    if (pf_gate) {
      if constexpr (LEAN) {
        load_gate(it, gcur);  // used a whole tile of matrix work later
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) gcur[r] = gnext[r];
        load_gate(has_next ? itn : it, gnext);
      }
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tw][r] = LEAN ? 0.0f : breg[tw][r];
    if (tk.accumulate) {  // K-chunked products (inputs wider than 384): start from the existing output
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) {
        const int fb = 32 * (w + 4 * tw) + 4 * h;
        const float* const yo = tk.y + ((long)it * tk.y_ld + fb) * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (fb + (r & 3) + 8 * (r >> 2) < N) acc[tw][r] += yo[DOFF(r)];
      }
    }
    Frag xf[2];
    xf[0] = x_frag(rb, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) xf[(b + 1) & 1] = x_frag(rb, b + 1);
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) acc[tw] = h2::mfma3(wf[tw][b], xf[b & 1], acc[tw]);  // tiles past N: never stored
      // this batch's share of the next tile's commit, and the loads of the tile after it into the freed registers
#pragma unroll
      for (int k = (b * XS) / NB; k < ((b + 1) * XS) / NB; ++k) {
        commit(wbuf, k);  // (after the last tile: a duplicate nobody reads)
        issue(it_issue, k);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    epilogue(it, acc, gcur, pf_gate);
    __syncthreads();
    cur ^= 1;
  }
}

// dW = X^T Y, db = colsum(Y) for T32 X (x_ld features per tile, the first K used) and T32 Y.
// KT = 32-feature tiles of X (accumulator rows), NTW = 32-feature tiles of Y per wave (tiles w, w+4, w+8).
template <int KT, int NTW>
__global__ __launch_bounds__(256, 1) void rec_xty_h2_kernel(XtyTask tk) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  constexpr int NT = 4 * NTW;
  constexpr int XPL = 32 * KT * 64, YPL = 32 * NT * 64;  // bytes per plane
  constexpr int BUF = 2 * XPL + 2 * YPL;                 // X hi, X lo, Y hi, Y lo
  constexpr int S = KT + NT;                             // float4 slots per thread and tile: X then Y
  constexpr int NG = 2 * NTW;                            // MFMA groups per tile: (k-step s, y tile tw)
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int K = tk.K, N = tk.N;
  const int nx4 = K * 8, ny4 = N * 8, ysp4 = tk.y_split * 8;

  f32x16 acc[KT][NTW];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][tw][r] = 0.0f;
  float bacc[NT];  // lanes with (tid & 7) == 0: column sum of y feature (tid >> 3) + 32 i
#pragma unroll
  for (int i = 0; i < NT; ++i) bacc[i] = 0.0f;

  float4 raw[S];
  int soff[S];
#pragma unroll
  for (int k = 0; k < S; ++k) soff[k] = stage_off(tid + 256 * (k < KT ? k : k - KT));
  auto issue = [&](int it, int k) {
    if (k < KT) {
      const float4* xs = reinterpret_cast<const float4*>(tk.x + (long)it * tk.x_ld * 32);
      const int q = tid + 256 * k;
      raw[k] = xs[q < nx4 ? q : (nx4 - 1)];
    } else {
      const int q = tid + 256 * (k - KT);
      const int qq = q < ny4 ? q : (ny4 - 1);
      if (tk.y_tail != nullptr && qq >= ysp4) {  // (wave-uniform: y_split is a multiple of 32 features = 256 float4)
        raw[k] = reinterpret_cast<const float4*>(tk.y_tail + (long)it * tk.y_tail_ld * 32)[qq - ysp4];
      } else {
        raw[k] = reinterpret_cast<const float4*>(tk.y + (long)it * tk.y_ld * 32)[qq];
      }
    }
  };
  // Branch-free: slots past K / N stage clamped duplicates into feature rows whose products are never stored.
  // `live` is 1.0f for a real tile, 0.0f for the duplicate committed after a block's last tile.
  auto commit = [&](u8* buf, int k, float live) {
    if (k < KT) {
      stage4(buf, XPL, soff[k], raw[k]);
    } else {
      stage4(buf + 2 * XPL, YPL, soff[k], raw[k]);
      // this thread's four rows of its feature; the 8 threads of a feature are added up once, after the last tile
      bacc[k - KT] += live * ((raw[k].x + raw[k].y) + (raw[k].z + raw[k].w));
    }
  };

  const int ntiles = tk.rows / 32;
  const int G = gridDim.x;
  int it = blockIdx.x;
  if (it < ntiles) {
#pragma unroll
    for (int k = 0; k < S; ++k) issue(it, k);
#pragma unroll
    for (int k = 0; k < S; ++k) commit(lds, k, 1.0f);
  }
  {
    const int itn = it + G;
#pragma unroll
    for (int k = 0; k < S; ++k) issue(itn < ntiles ? itn : (ntiles - 1), k);
  }
  __syncthreads();
  int cur = 0;
  for (; it < ntiles; it += G) {
    const u8* const rb = lds + cur * BUF;
    u8* const wbuf = lds + (cur ^ 1) * BUF;
    const int itn = it + G, itnn = it + 2 * G;
    const float live = itn < ntiles ? 1.0f : 0.0f;
    const int it_issue = itnn < ntiles ? itnn : (ntiles - 1);
    Frag xf[KT];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int s = g / NTW, tw = g % NTW;
      if (tw == 0) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) xf[kt] = rows_frag(rb, XPL, 32 * kt + j, 2 * s + h);
      }
      const Frag yf = rows_frag(rb + 2 * XPL, YPL, 32 * (w + 4 * tw) + j, 2 * s + h);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[kt][tw] = h2::mfma3(xf[kt], yf, acc[kt][tw]);  // tiles past K / N: never stored
      // this group's share of the next tile's commit, and the loads of the tile after it into the freed registers
#pragma unroll
      for (int k = (g * S) / NG; k < ((g + 1) * S) / NG; ++k) {
        commit(wbuf, k, live);
        issue(it_issue, k);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
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
    for (int i = 0; i < NT; ++i) {
      float b = bacc[i];
      b += __shfl_xor(b, 1, 64);
      b += __shfl_xor(b, 2, 64);
      b += __shfl_xor(b, 4, 64);
      const int n = (tid >> 3) + 32 * i;
      if ((tid & 7) == 0 && n < N) slab[(long)K * N + n] = b * sc;
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
    // all eight row addresses of this wave first (each sits behind an index load), then all row loads: no round trip per row
    const float* xrow[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) xrow[rr] = tk.x + gather_row(tk, it * 32 + 8 * w + rr) * tk.x_ld;
    if (KP <= 256) {
      float v[8][4];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = lane + 64 * cc;
          v[rr][cc] = xrow[rr][c < K ? c : 0];
        }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = lane + 64 * cc;
          if (c < KP) xs[(8 * w + rr) * ld + c] = c < K ? v[rr][cc] : 0.0f;
        }
    } else {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        for (int c = lane; c < KP; c += 64) xs[(8 * w + rr) * ld + c] = c < K ? xrow[rr][c] : 0.0f;
    }
    __syncthreads();
    float* out = tk.y + (long)it * KP * 32;
    for (int i = tid; i < KP * 32; i += 256) out[i] = xs[(i & 31) * ld + (i >> 5)];
    __syncthreads();
  }
}

template <int NB, int NTW, bool LEAN = false>
int launch_dense_h2(const DenseTask& tk, hipStream_t s) {
  constexpr int lb = 2 * 2 * 16 * NB * 64;  // two buffers of hi + lo planes
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_dense_h2_kernel<NB, NTW, LEAN>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    attr_set = true;
  }
  int blocks = tk.rows / 32;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL((rec_dense_h2_kernel<NB, NTW, LEAN>), dim3(blocks), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int KT, int NTW>
int launch_xty_h2(const XtyTask& tk, int n_slab, hipStream_t s) {
  constexpr int lb = 2 * 2 * 32 * (KT + 4 * NTW) * 64;
  static_assert(lb <= 163840, "double-buffered staging must fit the CU's LDS");
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_xty_h2_kernel<KT, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    attr_set = true;
  }
  if (n_slab > tk.rows / 32) return 1;  // every block must own a tile (its prologue stages one)
  hipLaunchKernelGGL((rec_xty_h2_kernel<KT, NTW>), dim3(n_slab), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

int mava_rec_dense_h2_launch(const DenseTask& tk, hipStream_t s) {
  if (tk.x_rowmajor) return 1;
  const int nb = (tk.K + 15) / 16;
  const int ntw = ((tk.N + 31) / 32 + 3) / 4;
  // only the first K features of a tile are read (x_ld >= K is checked by the caller); staged features past K are zeros
#define DENSE_H2(NBv, NTWv) \
  if (nb <= NBv && ntw == NTWv) return launch_dense_h2<NBv, NTWv>(tk, s)
  DENSE_H2(1, 1); DENSE_H2(2, 1); DENSE_H2(4, 1); DENSE_H2(6, 1); DENSE_H2(8, 1); DENSE_H2(10, 1); DENSE_H2(12, 1);
  DENSE_H2(18, 1); DENSE_H2(8, 3);
  if (nb <= 24 && ntw == 1) {
    if (tk.bias != nullptr) return 1;  // (the lean K = 384 form carries no bias registers: the exact-f32 kernel takes it)
    return launch_dense_h2<24, 1, true>(tk, s);
  }
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
