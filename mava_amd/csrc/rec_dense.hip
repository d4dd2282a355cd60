// Building blocks of the recurrent PPO path (rec_ippo / rec_mappo; mava/networks.py:238-331):
// dense layers with register-resident weights and the X^T Y weight-gradient product.
//
// Internal activation layout "T32": a (rows x N) matrix is stored as 32-row tiles, feature-major
// inside a tile:  elem(row, f) at ((row / 32) * N + f) * 32 + (row % 32).  In this layout an MFMA
// accumulator of the transposed product (feature in the register, batch row on the lane) is read and
// written with fully coalesced 128-byte accesses and can be used as the B operand of the next layer
// straight from memory; the X^T Y product stages tiles in LDS as [feature][33] and walks features on
// the lanes (conflict-free).  External tensors (observations, masks, ...) stay row-major and are
// gathered on load.  rows must be a multiple of 32 (checked on the host).
//
// mava_rec_dense_f32:  Y = act(X W + b) [* (G > 0)]   - every wave keeps its slice of W (<= 192
//   registers) for the whole launch; persistent blocks walk row tiles, no LDS, no barriers when the
//   input is T32.
// mava_rec_xty_f32:    dW = X^T Y, db = colsum(Y) as per-block slabs (fixed-order reduction elsewhere).
#include "mlp_core.h"
#include "rec_task.h"

#include "ctx.h"

namespace {

// NB = K padded to 16-input batches; NTW = 32-feature output tiles per wave (tiles w, w+4, w+8)
// RM: row-major gathered input (else T32).  FULLK (T32 only): K == 16*NB, so operand addresses are
// compile-time offsets (a per-load clamp makes the compiler hoist one 64-bit address per load out of the tile loop).
template <int NB, int NTW, bool RM, bool FULLK>
__global__ __launch_bounds__(256, 1) void rec_dense_kernel(DenseTask tk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // row-major input only: XS[32][ldx]
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int K = tk.K, N = tk.N;
  const int ldx = 16 * NB + 1;
  const int ntile_n = (N + 31) / 32;

  // ---- resident weight slice: wreg[tw][b][s] = W[16b + 2s + h][32*(w + 4tw) + j]
  // Two passes: every load first (clamped to a valid element, no arithmetic on a loaded value), the zeroing of
  // the padding afterwards - a conditional load per word makes the compiler wait for memory once per word, which
  // is what a 16 K-row acting-step launch then spends its time on.
  float wreg[NTW][NB][8];
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
    const int col = 32 * (w + 4 * tw) + j;
    const int colc = col < N ? col : (N - 1);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = 16 * b + 2 * s + h;
        const int kc = k < K ? k : (K - 1);
        wreg[tw][b][s] = tk.w[(long)kc * tk.ldw + colc];
      }
  }
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
    const int col = 32 * (w + 4 * tw) + j;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = 16 * b + 2 * s + h;
        wreg[tw][b][s] = (k < K && col < N) ? wreg[tw][b][s] : 0.0f;
      }
  }
  if (RM) {
    for (int i = tid; i < 32 * ldx + 4; i += 256) lds[i] = 0.0f;
    __syncthreads();
  }

  const int ntiles = tk.rows / 32;
  // element (it, f, j) of the T32 output sits at ((it*N + f)*32 + j): one per-lane base per (tile, n-tile), the 16
  // accumulator registers at compile-time offsets from it (no 64-bit address per element)
#define DOFF(r) ((((r) & 3) + 8 * ((r) >> 2)) * 32)
  // gate values of n-tile 0 can be handed in already loaded (gpre, prefetched a tile or more ahead)
  auto epilogue = [&](int it, f32x16 (&acc)[NTW], const float (&gpre)[16], bool use_pre) {
    // activation, optional relu-mask gate, coalesced T32 store
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int fb = 32 * (w + 4 * tw) + 4 * h;
      const long base = ((long)it * tk.y_ld + fb) * 32 + j;
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
  // raw gate loads of n-tile 0 for tile `it` (features past N clamp to the tile's first feature)
  auto load_gate = [&](int it, float (&g)[16]) {
    const int fb = 32 * w + 4 * h;
    const float* const go = tk.gate + ((long)it * tk.y_ld + fb) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = go[(fb + (r & 3) + 8 * (r >> 2) < N) ? DOFF(r) : 0];
  };
  auto init_acc = [&](int it, f32x16 (&acc)[NTW]) {
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int fb = 32 * (w + 4 * tw) + 4 * h;
      const float* const yo = tk.y + ((long)it * tk.y_ld + fb) * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fb + (r & 3) + 8 * (r >> 2);
        float a0 = (tk.bias != nullptr && f < N) ? tk.bias[f] : 0.0f;
        if (tk.accumulate && f < N) a0 += yo[DOFF(r)];
        acc[tw][r] = a0;
      }
    }
  };

  if (RM) {
    // ---- row-major source: the next tile's 32 gathered rows travel global -> registers during this tile's MFMAs
    // and are written to the LDS tile after its last reader (branch-free single path: a piece past the row end
    // is loaded from the row start and stored to a dummy slot); B operands are read one batch ahead.
    const int srow = tid >> 3, l8 = tid & 7;
    float* const xs = lds + srow * ldx;
    float* const dummy = lds + 32 * ldx;
    float xr[2 * NB];
    const float gnone[16] = {};
    auto issue = [&](int it) {
      const float* xrow = tk.x + gather_row(tk, it * 32 + srow) * tk.x_ld;
#pragma unroll
      for (int i = 0; i < 2 * NB; ++i) {
        const int c = l8 + 8 * i;
        xr[i] = xrow[c < K ? c : 0];
      }
    };
    int it = blockIdx.x;
    if (it < ntiles) issue(it);
    for (; it < ntiles; it += gridDim.x) {
      __syncthreads();  // previous tile's readers done
#pragma unroll
      for (int i = 0; i < 2 * NB; ++i) {
        const int c = l8 + 8 * i;
        float* q = c < K ? xs + c : dummy;
        *q = xr[i];
      }
      __syncthreads();
      if (it + (int)gridDim.x < ntiles) issue(it + gridDim.x);  // in flight during the MFMAs below
      const float* xb = lds + j * ldx + h;  // x[row j][k + h], zero padded
      f32x16 acc[NTW];
      init_acc(it, acc);
      float xo[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xo[0][s] = xb[2 * s];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB) {
#pragma unroll
          for (int s = 0; s < 8; ++s) xo[(b + 1) & 1][s] = xb[16 * (b + 1) + 2 * s];
        }
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw) {
          if (w + 4 * tw < ntile_n) {
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[tw] = MFMA32(wreg[tw][b][s], xo[b & 1][s], acc[tw]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      epilogue(it, acc, gnone, false);
    }
  } else {
    // ---- T32 source: B operands stream straight from memory through a ring of RDX batches that runs
    // ACROSS row tiles (prefetch distance RDX-1 batches, also over the tile boundary), so neither the HBM
    // latency of a batch nor the start-up latency of a tile is exposed.  Inputs past K are clamped to a
    // valid feature (their weights are zero).
    // ring depth: deep when the weight slice is small, shallow when it already fills most of the register file
    constexpr bool BIGW = (NB * NTW * 8 > 128);
    constexpr int RDX = BIGW ? ((NB % 4 == 0) ? 4 : ((NB % 3 == 0) ? 3 : 2))
                             : ((NB % 8 == 0) ? 8 : ((NB % 6 == 0) ? 6 : ((NB % 4 == 0) ? 4 : 2)));
    constexpr int PD = RDX - 1;
    static_assert(NB % RDX == 0, "ring depth must divide the batch count");
    const int KX = tk.x_ld;  // features per x tile (>= K; the padded gather of rec_dense_h2.hip has more)
    auto tile_ptr = [&](int it) { return tk.x + ((long)it * KX + (FULLK ? h : 0)) * 32 + j; };
    auto load_batch = [&](const float* xp, int bq, float (&dst)[8]) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (FULLK) {
          dst[s] = xp[(16 * bq + 2 * s) * 32];  // elem(row j, k + h): h folded into the tile pointer
        } else {
          int k = 16 * bq + 2 * s + h;
          k = k < K ? k : (K - 1);
          dst[s] = xp[(long)k * 32];
        }
      }
    };
    const bool pf_gate = (NTW == 1) && tk.gate != nullptr && (32 * w < N);
    if (NB <= 2) {
      // Narrow inputs (the n_out-wide dlogits of the head backward): a tile is only NB*8 MFMAs, far shorter than a
      // memory round trip, so the whole x tile AND the gate values run PF tiles ahead through rotating registers.
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
        for (int d = 0; d < PF; ++d) {  // rotate: slot 0 = this tile
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < 8; ++s) xq[d][b][s] = xq[d + 1][b][s];
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
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int tw = 0; tw < NTW; ++tw) {
            if (w + 4 * tw < ntile_n) {
#pragma unroll
              for (int s = 0; s < 8; ++s) acc[tw] = MFMA32(wreg[tw][b][s], xq[0][b][s], acc[tw]);
            }
          }
        epilogue(it, acc, gq[0], pf_gate);
      }
      return;
    }
    float xo[RDX][8];
    float gcur[16], gnext[16];
    int it = blockIdx.x;
    const float* xt_cur = tile_ptr(it < ntiles ? it : 0);
#pragma unroll
    for (int d = 0; d < PD; ++d) load_batch(xt_cur, d % NB, xo[d]);  // PD <= NB - 1 always (RDX divides NB)
    if (pf_gate && it < ntiles) load_gate(it, gnext);
    for (; it < ntiles; it += gridDim.x) {
      const int itn = it + gridDim.x;
      const float* xt_next = (itn < ntiles) ? tile_ptr(itn) : xt_cur;
      if (pf_gate) {  // this tile's gate values arrived during the previous tile; the next tile's go out now
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
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw) {
          if (w + 4 * tw < ntile_n) {
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[tw] = MFMA32(wreg[tw][b][s], xo[b % RDX][s], acc[tw]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      epilogue(it, acc, gcur, pf_gate);
      xt_cur = xt_next;
    }
  }
}

// KT = K tiles of 32, NTW = N tiles per wave (n-tiles w, w+4, w+8)
template <int KT, int NTW>
__global__ __launch_bounds__(256, 1) void rec_xty_kernel(XtyTask tk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LD = 33;
  const int K = tk.K, N = tk.N;
  float* const XT = lds;                  // [32*KT][33] x^T tile (feature-major)
  float* const YT = lds + 32 * KT * LD;   // [N pad 32][33]
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int ntile_n = (N + 31) / 32;
  const int npad = ntile_n * 32;

  f32x16 acc[KT][NTW];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][tw][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};  // thread handles columns tid and tid + 256

  for (int i = tid; i < (32 * KT + npad) * LD; i += 256) lds[i] = 0.0f;
  __syncthreads();

  // Tiles are register-prefetched one row tile ahead with 16-byte loads (a T32 tile is one contiguous
  // run of K*32 / N*32 floats) and committed to LDS after the MFMAs of the current tile.
  constexpr int NT_ALL = 4 * NTW;  // n tiles handled by the block
  float4 xq[KT], yq[NT_ALL];
  const int nx4 = K * 8, ny4 = N * 8;  // float4 per tile
  auto issue = [&](int it) {
    const float4* ysrc = reinterpret_cast<const float4*>(tk.y + ((long)it * tk.y_ld) * 32);
    const float4* ytail = reinterpret_cast<const float4*>(tk.y_tail + ((long)it * tk.y_tail_ld) * 32);
    const int ysp4 = tk.y_tail ? tk.y_split * 8 : ny4;  // features past y_split come from y_tail (a multiple of 32 features: warp-uniform)
#pragma unroll
    for (int i = 0; i < NT_ALL; ++i) {
      const int q = tid + 256 * i;
      const int qq = q < ny4 ? q : (ny4 - 1);
      yq[i] = qq < ysp4 ? ysrc[qq] : ytail[qq - ysp4];
    }
    if (!tk.x_rowmajor) {
      const float4* xsrc = reinterpret_cast<const float4*>(tk.x + ((long)it * tk.x_ld) * 32);
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const int q = tid + 256 * i;
        xq[i] = xsrc[q < nx4 ? q : (nx4 - 1)];
      }
    }
  };
  auto commit = [&](int it) {
#pragma unroll
    for (int i = 0; i < NT_ALL; ++i) {
      const int q = tid + 256 * i;
      if (q < ny4) {
        const int e = 4 * q, f = e >> 5, r = e & 31;
        float* d = YT + f * LD + r;
        d[0] = yq[i].x; d[1] = yq[i].y; d[2] = yq[i].z; d[3] = yq[i].w;
      }
    }
    if (!tk.x_rowmajor) {
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const int q = tid + 256 * i;
        if (q < nx4) {
          const int e = 4 * q, f = e >> 5, r = e & 31;
          float* d = XT + f * LD + r;
          d[0] = xq[i].x; d[1] = xq[i].y; d[2] = xq[i].z; d[3] = xq[i].w;
        }
      }
    } else {
      const int srow = tid >> 3, l8 = tid & 7;
      DenseTask g;
      g.Rm = tk.Rm; g.E = tk.E; g.A = tk.A; g.xshare = tk.xshare; g.idx = tk.idx;
      const float* xrow = tk.x + gather_row(g, it * 32 + srow) * tk.x_ld;
      for (int k = l8; k < K; k += 8) XT[k * LD + srow] = xrow[k];
    }
  };

  const int ntiles = tk.rows / 32;
  int it = blockIdx.x;
  if (it < ntiles) {
    issue(it);
    commit(it);
  }
  __syncthreads();
  for (; it < ntiles; it += gridDim.x) {
    const int itn = it + gridDim.x;
    if (itn < ntiles) issue(itn);
    // ---- dW[k][n] += sum_rows x[row][k] * y[row][n]
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      if (w + 4 * tw < ntile_n) {
        const float* eb = YT + (32 * (w + 4 * tw) + j) * LD + h;
        float bz[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bz[s] = eb[2 * s];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const float* ea = XT + (32 * kt + j) * LD + h;
#pragma unroll
          for (int s = 0; s < 16; ++s) acc[kt][tw] = MFMA32(ea[2 * s], bz[s], acc[kt][tw]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (tk.want_bias) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int n = tid + 256 * u;
        if (n < N) {
          float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = YT[n * LD + 16 * c + r];
#pragma unroll
            for (int r = 0; r < 16; r += 2) { s0 += v[r]; s1 += v[r + 1]; }
          }
          bsum[u] += s0 + s1;
        }
      }
    }
    __syncthreads();  // all readers of the staged tiles are done
    if (itn < ntiles) commit(itn);
    __syncthreads();
  }
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = 32 * (w + 4 * tw) + j;
        if (k < K && n < N) slab[(long)k * N + n] = acc[kt][tw][r] * tk.out_scale;
      }
  if (tk.want_bias) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = tid + 256 * u;
      if (n < N) slab[(long)K * N + n] = bsum[u] * tk.out_scale;
    }
  }
}

template <int NB, int NTW, bool RM, bool FULLK>
int launch_dense_rm(const DenseTask& tk, hipStream_t s) {
  const size_t lb = RM ? ((size_t)32 * (16 * NB + 1) + 4) * sizeof(float) : 0;
  static bool attr_set = false;  // once per instantiation
  if (lb > 0 && !attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_dense_kernel<NB, NTW, RM, FULLK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    attr_set = true;
  }
  int blocks = tk.rows / 32;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL((rec_dense_kernel<NB, NTW, RM, FULLK>), dim3(blocks), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NB, int NTW>
int launch_dense(const DenseTask& tk, hipStream_t s) {
  if (tk.x_rowmajor) {
    if (NTW == 1) return launch_dense_rm<NB, 1, true, false>(tk, s);  // external observations feed 128-wide torsos only
    mava_set_error("mava_rec_dense_f32: row-major input is instantiated for N <= 128 only");
    return MAVA_EARG(9);
  }
  if (tk.K == 16 * NB) return launch_dense_rm<NB, NTW, false, true>(tk, s);
  if (NTW == 1) return launch_dense_rm<NB, 1, false, false>(tk, s);  // any width: clamped operand addresses
  mava_set_error("mava_rec_dense_f32: T32 input width K=%d is not instantiated for N=%d (T32 inputs: K <= 32, or K in "
                 "{64,96,128,192,288,384} with N <= 128, or K = 128 / 192 with N <= 384 / 256)", tk.K, tk.N);
  return MAVA_EARG(9);
}

template <int KT, int NTW>
int launch_xty(const XtyTask& tk, int n_slab, hipStream_t s) {
  const int npad = ((tk.N + 31) / 32) * 32;
  const size_t lb = (size_t)(32 * KT + npad) * 33 * sizeof(float);
  MAVA_ARG_CHECK(lb <= 163840, 8, "mava_rec_xty_f32: %zu bytes of LDS needed", lb);
  static bool attr_set = false;  // once per instantiation: allow the whole 160 KiB (lb varies with N)
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_xty_kernel<KT, NTW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((rec_xty_kernel<KT, NTW>), dim3(n_slab), dim3(256), lb, s, tk);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

extern "C" int mava_rec_dense_f32(const mava_ctx* ctx, const float* x, int x_rowmajor, const int32_t* idx, int Rm, int E, int A,
                                  int x_share, int x_ld, int accumulate, const float* w, int ldw,
                                  const float* bias, const float* gate, float* y, int y_ld, int K, int N, int rows,
                                  int relu, hipStream_t s) {
  if (y_ld <= 0) y_ld = N;
  if (!x_rowmajor && x_ld <= 0) x_ld = K;
  MAVA_ARG_CHECK(K >= 1 && K <= 384 && N >= 1 && N <= 384 && ldw >= N, 0,
                 "mava_rec_dense_f32: K=%d N=%d ldw=%d unsupported (K, N <= 384)", K, N, ldw);
  MAVA_ARG_CHECK(rows >= 0 && rows % 32 == 0, 1, "mava_rec_dense_f32: rows=%d must be a multiple of 32", rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(x && w && y, 2, "mava_rec_dense_f32: null pointer argument");
  MAVA_ARG_CHECK(!x_rowmajor || (Rm >= 1 && A >= 1 && E >= 1 && x_share >= 1 && rows % Rm == 0 && x_ld >= K), 3,
                 "mava_rec_dense_f32: bad gather description Rm=%d E=%d A=%d x_ld=%d", Rm, E, A, x_ld);
  MAVA_ARG_CHECK(x_rowmajor || x_ld >= K, 5, "mava_rec_dense_f32: T32 input with %d features per tile < K=%d", x_ld, K);
  MAVA_ARG_CHECK(y_ld >= N, 7, "mava_rec_dense_f32: y_ld=%d < N=%d", y_ld, N);
  DenseTask tk = {x, x_rowmajor, idx, Rm, E, A, x_share, x_ld, accumulate, w, ldw, bias, gate, y, K, N, rows, relu, y_ld};
  if (mava_ctx_matmul_mode(ctx) == 1) {
    const int rc = mava_rec_dense_h2_launch(tk, s);
    if (rc <= 0) return rc;
  }
  const int nb = (K + 15) / 16;
  const int ntw = ((N + 31) / 32 + 3) / 4;  // n-tiles per wave
  MAVA_ARG_CHECK(nb * ntw * 8 <= 192, 4, "mava_rec_dense_f32: weight slice of %d registers does not fit (K=%d N=%d)",
                 nb * ntw * 8, K, N);
#define DENSE_CASE(NBv, NTWv) \
  if (nb <= NBv && ntw == NTWv) return launch_dense<NBv, NTWv>(tk, s)
  DENSE_CASE(2, 1); DENSE_CASE(4, 1); DENSE_CASE(6, 1); DENSE_CASE(8, 1); DENSE_CASE(12, 1); DENSE_CASE(18, 1);
  DENSE_CASE(24, 1); DENSE_CASE(8, 2); DENSE_CASE(12, 2); DENSE_CASE(8, 3);
#undef DENSE_CASE
  mava_set_error("mava_rec_dense_f32: shape K=%d N=%d is not instantiated", K, N);
  return MAVA_EARG(9);
}

extern "C" int mava_rec_xty_f32(const mava_ctx* ctx, const float* x, int x_rowmajor, const int32_t* idx, int Rm, int E, int A, int x_share,
                                int x_ld, const float* y, int y_ld, const float* y_tail, int y_split, int y_tail_ld, int K, int N,
                                int rows, int want_bias, float out_scale, float* slab, long slab_stride, int n_slab, hipStream_t s) {
  if (!x_rowmajor && x_ld <= 0) x_ld = K;
  if (y_ld <= 0) y_ld = N;
  MAVA_ARG_CHECK(y_ld >= N, 6, "mava_rec_xty_f32: y_ld=%d < N=%d", y_ld, N);
  MAVA_ARG_CHECK(K >= 1 && K <= 384 && N >= 1 && N <= 384, 0, "mava_rec_xty_f32: K=%d N=%d unsupported", K, N);
  MAVA_ARG_CHECK(rows >= 32 && rows % 32 == 0 && n_slab >= 1 && n_slab <= 1024, 1,
                 "mava_rec_xty_f32: rows=%d n_slab=%d", rows, n_slab);
  MAVA_ARG_CHECK(slab_stride >= (long)K * N + (want_bias ? N : 0), 2, "mava_rec_xty_f32: slab_stride too small");
  MAVA_ARG_CHECK(x && y && slab, 3, "mava_rec_xty_f32: null pointer argument");
  MAVA_ARG_CHECK(x_ld >= K, 5, "mava_rec_xty_f32: x_ld=%d < K=%d", x_ld, K);
  MAVA_ARG_CHECK(y_tail == nullptr || (y_split >= 32 && y_split % 32 == 0 && y_split < N && y_tail_ld >= N - y_split), 10,
                 "mava_rec_xty_f32: y_tail with y_split=%d y_tail_ld=%d N=%d", y_split, y_tail_ld, N);
  XtyTask tk = {x, x_rowmajor, idx, Rm, E, A, x_share, x_ld, y, K, N, rows, y_ld, slab, slab_stride, want_bias, out_scale,
                y_tail, y_tail ? y_split : N, y_tail_ld};
  if (mava_ctx_matmul_mode(ctx) == 1) {
    const int rc = mava_rec_xty_h2_launch(tk, n_slab, s);
    if (rc <= 0) return rc;
  }
  const int kt = (K + 31) / 32;
  const int ntw = ((N + 31) / 32 + 3) / 4;
  MAVA_ARG_CHECK(kt * ntw <= 12, 4, "mava_rec_xty_f32: %d accumulator tiles per wave do not fit (K=%d N=%d)", kt * ntw, K, N);
#define XTY_CASE(KTv, NTWv) \
  if (kt <= KTv && ntw == NTWv) return launch_xty<KTv, NTWv>(tk, n_slab, s)
  XTY_CASE(1, 1); XTY_CASE(2, 1); XTY_CASE(3, 1); XTY_CASE(4, 1); XTY_CASE(6, 1); XTY_CASE(9, 1); XTY_CASE(12, 1);
  XTY_CASE(4, 2); XTY_CASE(6, 2); XTY_CASE(4, 3);
#undef XTY_CASE
  mava_set_error("mava_rec_xty_f32: shape K=%d N=%d is not instantiated", K, N);
  return MAVA_EARG(9);
}
