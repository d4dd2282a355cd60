// Fused PPO actor minibatch kernel, f16x2 arithmetic, EIGHT waves per workgroup (two per SIMD).
//
// Reference: mava/systems/ppo/ff_mappo.py:150-180 (_actor_loss_fn), :204-214 (value_and_grad), :268-285 (shuffle as an
// index vector) - the same math and the same split-f16 products as ppo_train_h2.hip, phase by phase.
//
// Why a second mapping.  ppo_train_h2.hip runs four waves (one per SIMD, 492 registers each) on 32x32x16 MFMAs, wave w
// owning 32 features of every layer.  Its phase stamps (profiles/r02_v5_train_phase_stamps.txt) put a 32-row tile at
// ~12.2 K cycles for 3.8 K cycles of MFMAs: between the matrix phases sit ~1300 vector instructions per wave (splits,
// ReLU masks, image stores, the loss, the gather's address arithmetic), and ONE wave on a SIMD issues a vector
// instruction every 4 cycles where the SIMD could take one every 2 (MI355X_MICROARCH.md, cycle constants) - the kernel
// is bound by vector ISSUE with nothing to overlap it.  Here the same tile is worked by eight waves, wave v owning 16
// features of every layer on v_mfma_f32_16x16x32_f16 (M = 16 features, N = 16 batch rows, two N-tiles per 32-row tile):
// the per-wave register set halves (weights, accumulators: <= 256 registers, so two waves fit a SIMD), each SIMD
// interleaves the vector work of one wave with the matrix work of the other, and the barrier count per tile drops from
// five to four because the weight-gradient products read only the wave's OWN columns of the dz images.
//
// Operand maps (pinned on the hardware by tools/microbench/w8_probe.hip): lane (i = l & 15, kg = l >> 4) holds
// A[i][8 kg + j], B[8 kg + j][i] (j = 0..7) of the 16x16x32 product and C[4 kg + r][i] (r = 0..3); for 16x16x16:
// A[i][4 kg + j], B[4 kg + j][i] - so an accumulator tile IS the B operand of a 16-deep product (the head).
//
// LDS images: [rows][128 features] f16, hi and lo plane, 256-byte rows, 16-byte chunk c of row r at chunk c ^ sw(r),
// sw(r) = ((r & 3) << 1) | (9 * ((r >> 3) & 1)): conflict-free for the row reads of the 16x16x32 B operand
// (ds_read_b128, lanes (n, kg): row n, chunk 4 s + kg) AND for its hardware-transposed reads (ds_read_b64_tr_b16, two
// 4-row blocks 8 rows apart per 32-lane half) - the swizzle (b) of cdna_hip_programming.md T10 is 2-way on the former -
// and 2-way (8 LDS-array cycles for a 6-cycle instruction) on the 8-byte image stores of an accumulator tile (16 lanes =
// 16 rows of one 8-byte column; banks are mod 32 dwords for stores), where sw without the bit 0 term is 4-way (16 cycles:
// with eight waves storing at once that was ~1.3 K LDS-array cycles per tile; tools/lds_swizzle_search.py has the bank
// model and the search over the linear maps).
#include "h2_core.h"
#include "ppo_train_task.h"
#include "ctx.h"

namespace {

using namespace h2;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STATS_BLOCKS = 128;  // == ppo_train.hip (mava_adv_stats_blocks)
#ifndef MAVA_W8_GWD
#define MAVA_W8_GWD 1
#endif
constexpr int GWD = MAVA_W8_GWD;  // operand prefetch depth of the weight-gradient products

#ifdef MAVA_STAMPS
#define WSTAMP_DECL unsigned long long ws_prev = __builtin_readcyclecounter(), ws_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WSTAMP(i)                                                   \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long ws_now = __builtin_readcyclecounter(); \
    ws_acc[i] += ws_now - ws_prev;                                  \
    ws_prev = ws_now;                                               \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#else
#define WSTAMP_DECL
#define WSTAMP(i)
#endif

struct Frag4 {  // split operand of the 16-deep product: 4 k-values per lane as hi + lo
  half4 hi, lo;
};
__device__ __forceinline__ f32x4 mfma3w(const Frag& a, const Frag& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);  // small terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
}
// the same sum when one operand's low term is known to be zero (its product would add exactly 0: same bits, one MFMA less)
__device__ __forceinline__ f32x4 mfma2w_blo0(const Frag& a, const half8& bhi, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, bhi, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, bhi, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma2w_alo0(const half8& ahi, const Frag& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, b.hi, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma3k16(const Frag4& a, const Frag4& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, c, 0, 0, 0);
}
__device__ __forceinline__ Frag4 split4(const f32x4& v) {
  Frag4 f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    _Float16 a, b;
    split1(v[r], a, b);
    f.hi[r] = a;
    f.lo[r] = b;
  }
  return f;
}

// ---- image addressing (bytes) -------------------------------------------------------------------------------------------
constexpr int WROW = 256;              // bytes per image row
constexpr int WPLANE32 = 32 * WROW;    // one plane of a 32-row image
constexpr int WIMG = 2 * WPLANE32;     // hi + lo
constexpr int W2PLANE = 128 * WROW;
constexpr int DYROW = 48;              // [32 rows][16 outputs] f16 + 16 bytes: conflict-free 8-byte row reads
constexpr int DYPLANE = 32 * DYROW;
__host__ __device__ inline int w8_sw(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) * 9); }
__host__ __device__ inline int w8_off(int row, int chunk) { return WROW * row + 16 * (chunk ^ w8_sw(row)); }

struct W8Layout {
  int h1, dz2, dz1, w2, xs, dy, small, end;
};
inline W8Layout make_w8_layout() {
  W8Layout L;
  L.h1 = 0;
  L.dz2 = L.h1 + WIMG;   // also the partial logits (f32), idle between barriers A and B2
  L.dz1 = L.dz2 + WIMG;  // also the h2 image: every wave reads back only its OWN 16 columns of either
  L.w2 = L.dz1 + WIMG;   // W2 [128 k][128 n]: row reads for dh1 = W2 . dz2 (the forward operand lives in registers)
  L.xs = L.w2 + 2 * W2PLANE;
  L.dy = L.xs + 2 * WIMG;  // two x buffers
  L.small = L.dy + 2 * DYPLANE;
  L.end = L.small + (32 + 16 + 16) * 4;  // f32: b3[32] | misc[16] | dummy
  return L;
}

// NO: padded action count of the loss lanes (8 or 16); S1: 32-input steps of layer 1 (din + 1 <= 32 S1 <= 128);
// XV: floats per staged piece (2 when the rows are 8-byte aligned and din is even).
// W2R: steps (of four) of the layer-2 forward operand - this wave's 16 columns of W2, 8 registers a step - that stay in
// registers; the others are read from the W2 image with hardware-transposed reads (4 per step, tile and wave).  All four
// resident need 271 registers and spill inside the tile loop; w8_w2r() below picks two where they fit.
// ROLE: with 8 action lanes per row the loss occupies 256 threads, so the per-row duties are divided between the two wave
// groups (separate functions: separate register sets): ROLE 1 = waves 0-3, the loss of the tile's 32 rows; ROLE 2 = waves 4-7,
// the x staging (gather of the next tile's rows, split + commit into the other x buffer) - in the first version, where
// every wave staged and waves 0-3 also ran the loss, waves 4-7 waited ~900 of a tile's 10.7 K cycles at barrier B2
// (phase stamps).  ROLE 0 (16 action lanes: every thread is a loss lane) does both.  All roles run the same barrier sequence
// and the same matrix work on their own 16 features.  (A static priority raise for waves 4-7, the younger half that loses
// issue arbitration against its SIMD partners - s_setprio 1 - was measured: 65.5 M env-steps/s either way.)
template <int NO, int S1, int XV, int W2R, bool ACTOR, int ROLE>
__device__ __forceinline__ void w8_body(const TrainTask& tk, const W8Layout& L, u8* lds) {
  constexpr bool DO_LOSS = ROLE != 2, DO_STAGE = ROLE != 1;
  // ACTOR = false: the value network on narrow inputs (ff_ippo's critic, ff_mappo.py:183-213): one output, the clipped value
  // loss on lane 0 of a row's NO loss lanes; its weights are split as WS * w and the layer accumulators unscaled by WU
  // (h2_core.h W_SCALE_CRITIC: the low terms of the weights leave f16's subnormal range)
  constexpr float WS = ACTOR ? 1.0f : W_SCALE_CRITIC, WU = 1.0f / WS;
  constexpr int KT1 = 2 * S1;  // 16-input tiles of the layer-1 weight gradient
  u8* const H1I = lds + L.h1;
  u8* const DZ2I = lds + L.dz2;
  u8* const DZ1I = lds + L.dz1;
  u8* const W2I = lds + L.w2;
  u8* const DYI = lds + L.dy;
  float* const YP = reinterpret_cast<float*>(lds + L.dz2);
  constexpr int YSTR = (NO == 8) ? 9 : NO;  // floats per (wave, row) of partial logits: 8 x 32 x YSTR x 4 bytes <= one image
  float* const B3s = reinterpret_cast<float*>(lds + L.small);
  float* const misc = B3s + 32;
  // xflag[b] != 0: some staged value of x buffer b has a non-zero low f16 term.  Observations that are exact in f16 - flags,
  // one-hot ids, small integer coordinates: RobotWarehouse's and the synthetic env's whole agents_view - leave it 0, and the
  // layer-1 product and its weight gradient then skip the x_lo . W product (it would add exactly 0) and the reads of that plane.
  unsigned* const xflag = reinterpret_cast<unsigned*>(misc + 2);

  const int tid = threadIdx.x;
  const int l = tid & 63, v = tid >> 6, i = l & 15, kg = l >> 4;
  const int din = tk.din, no = tk.no;
  const long R = (long)tk.Rb * tk.A;
  const float invR = 1.0f / (float)R;
  const float* const P = tk.params;
  const int oW2 = mlp_off_w2(din), oW3 = mlp_off_w3(din);

  // ---------------------------------------------------------------- prologue: LDS images, small vectors
  for (int o = tid * 16; o < L.end; o += 512 * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < no) B3s[tid] = P[mlp_off_b3(din, no) + tid];
  if (tid < 64) {  // the ones column of both x buffers: "row din" of W1 in the flat parameter vector is b1
    const int b = tid >> 5, row = tid & 31;
    *reinterpret_cast<_Float16*>(lds + L.xs + b * WIMG + w8_off(row, din >> 3) + 2 * (din & 7)) = (_Float16)1.0f;
  }
  {
    // W2 image: thread = (output feature n, quarter of the k range), sequential over k with the error-diffusion carry of
    // split1_carry (h2_core.h); coalesced over n
    const int n = tid & 127, k0 = 32 * (tid >> 7);
    float carry = 0.0f;
#pragma unroll 8
    for (int k = k0; k < k0 + 32; ++k) {
      _Float16 x0, x1;
      split1_carry(P[oW2 + k * MLP_H + n] * WS, carry, x0, x1);
      const int o = w8_off(k, n >> 3) + 2 * (n & 7);
      *reinterpret_cast<_Float16*>(W2I + o) = x0;
      *reinterpret_cast<_Float16*>(W2I + W2PLANE + o) = x1;
    }
  }
  if (ACTOR && tid == 0) {
    // ff_mappo.py:164  gae = (gae - gae.mean()) / (gae.std() + 1e-8)   (population std)
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < STATS_BLOCKS; ++b) { s1 += tk.stats[2 * b]; s2 += tk.stats[2 * b + 1]; }
    const double mean = s1 / (double)R;
    double var = s2 / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    misc[0] = (float)mean;
    misc[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
  }
  // ---------------------------------------------------------------- weight fragments kept in registers
  // layer 1 (A operand): W1[k = 32 s + 8 kg + e][f = 16 v + i]; k == din is b1, k > din zero
  Frag W1f[S1], W2f[W2R > 0 ? W2R : 1];
  {
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * s + 8 * kg + e;
        w[e] = (k <= din) ? P[k * MLP_H + 16 * v + i] * WS : 0.0f;
      }
      W1f[s] = split8_carry(w, c1);
    }
#pragma unroll
    for (int s = 0; s < W2R; ++s) {  // layer 2 (A operand), the first W2R steps: W2[k = 32 s + 8 kg + e][n = 16 v + i]
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = P[oW2 + (32 * s + 8 * kg + e) * MLP_H + 16 * v + i] * WS;
      W2f[s] = split8_carry(w, c2);
    }
  }
  Frag4 W3h, W3d;  // head forward A[m = o = i][k = 4 kg + j] = W3[16 v + 4 kg + j][o]; backward A[m = f = i][k = o = 4 kg + j]
  {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = (i < no) ? P[oW3 + (16 * v + 4 * kg + j) * no + i] * W3_SCALE : 0.0f;
      b[j] = (4 * kg + j < no) ? P[oW3 + (16 * v + i) * no + 4 * kg + j] * W3_SCALE : 0.0f;
    }
    W3h = split4(a);
    W3d = split4(b);
  }
  float b2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) b2r[r] = P[mlp_off_b2(din) + 16 * v + 4 * kg + r] * WS;  // (unscaled with the products)
  __syncthreads();
  const float adv_mean = ACTOR ? misc[0] : 0.0f, adv_rstd = ACTOR ? misc[1] : 0.0f;

  // persistent accumulators (R x gradient units): wave v owns output columns [16 v, 16 v + 16) of dW1 and dW2
  f32x4 gW1[KT1], gW2[8], gW3;
#pragma unroll
  for (int t = 0; t < KT1; ++t) gW1[t] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t) gW2[t] = f32x4{0, 0, 0, 0};
  gW3 = f32x4{0, 0, 0, 0};
  float ab2[4] = {0, 0, 0, 0}, ab3 = 0.0f, loss_a = 0.0f, loss_b = 0.0f;

  // ---------------------------------------------------------------- rows: cursors, staging, loss inputs
  constexpr int LOSS_THREADS = 32 * NO;   // one lane per (row, action slot): threads [0, 32 NO)
  static_assert(ROLE == 0 ? LOSS_THREADS == 512 : LOSS_THREADS == 256, "role split belongs to 8 action lanes");
  const int lrow = tid / NO, lo = tid & (NO - 1);
  constexpr int TPR = (ROLE == 2) ? 8 : 16;        // staging threads per row
  const int st = (ROLE == 2) ? (tid - 256) : tid;
  const int srow = st / TPR, l16 = st % TPR;
  const int nv = din / XV;                        // pieces per row
  constexpr int NPC = (32 * S1 / XV + TPR - 1) / TPR;   // pieces per thread
  constexpr int NR = DO_STAGE ? NPC * XV : 1;
  const uint32_t Au = (uint32_t)tk.A;
  const uint32_t q_step = 32u * gridDim.x, b_step = q_step / Au, a_step = q_step % Au;
  const uint32_t b_last = (uint32_t)(R - 1) / Au, a_last = (uint32_t)(R - 1) % Au;
  struct Cursor { uint32_t q, b, a; };
  auto cursor_at = [&](int rr) {
    Cursor c;
    c.q = 32u * blockIdx.x + (uint32_t)rr;
    c.b = c.q / Au;
    c.a = c.q - c.b * Au;
    return c;
  };
  auto cursor_advance = [&](Cursor& c) {
    c.q += q_step; c.b += b_step; c.a += a_step;
    if (c.a >= Au) { c.a -= Au; c.b += 1u; }
  };
  // The (t,e) index of a cursor's row is LOADED here and only USED a tile later (row = p_raw * A + a): an index load whose
  // value is consumed in the same block makes the compiler wait for it - and, loads retiring in order, for the x gathers
  // issued just before it: 1.9 K of a tile's 11.7 K cycles in the first version of this kernel (phase stamps).
  auto cursor_gather = [&](const Cursor& c, int32_t& p_raw, uint32_t& a_out) {
    const bool in = c.q < (uint32_t)R;
    const uint32_t b = in ? c.b : b_last;
    a_out = in ? c.a : a_last;
    p_raw = tk.idx ? tk.idx[b] : (int32_t)(tk.idx_base + (long)b);
  };
  // input row of agent row (index p, agent a): the critic's inputs may be shared by groups of xshare agent rows
  auto stage_row = [&](int32_t p_raw, uint32_t a) -> uint32_t {
    const uint32_t fr = (uint32_t)p_raw * Au + a;
    if constexpr (ACTOR) return fr;
    else return (tk.xshare == 1) ? fr : ((uint32_t)tk.xshare == Au ? (uint32_t)p_raw : fr / (uint32_t)tk.xshare);
  };
  auto stage_issue = [&](uint32_t fr, float (&xr)[NR]) {
    const float* xrow = tk.x + (long)fr * din;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int c = l16 + TPR * k;
      const int cc = (c < nv) ? c : 0;  // a piece past the row end is loaded from the row start and stored to a dummy slot
      if (XV == 2) {
        const float2 t = reinterpret_cast<const float2*>(xrow)[cc];
        xr[2 * k] = t.x; xr[2 * k + 1] = t.y;
      } else {
        xr[k] = xrow[cc];
      }
    }
  };
  u8* const xs_dummy = reinterpret_cast<u8*>(misc + 8);
  const int xsw = 16 * w8_sw(srow);
  auto stage_commit = [&](int buf, const float (&xr)[NR]) {
    u8* base = lds + L.xs + buf * WIMG + WROW * srow;
    uint32_t lo_any = 0;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int c = l16 + TPR * k;
      const bool ok = c < nv;
      const int col = XV * c;  // first feature of the piece
      u8* qa = ok ? (base + ((16 * (col >> 3)) ^ xsw) + 2 * (col & 7)) : xs_dummy;
      u8* qb = ok ? (qa + WPLANE32) : xs_dummy;
      if (XV == 2) {
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        half2v a, b;
#pragma unroll
        for (int e = 0; e < 2; ++e) { _Float16 x0, x1; split1(xr[2 * k + e], x0, x1); a[e] = x0; b[e] = x1; }
        *reinterpret_cast<half2v*>(qa) = a;
        *reinterpret_cast<half2v*>(qb) = b;
        lo_any |= __builtin_bit_cast(uint32_t, b) & 0x7FFF7FFFu;
      } else {
        _Float16 x0, x1;
        split1(xr[k], x0, x1);
        *reinterpret_cast<_Float16*>(qa) = x0;
        *reinterpret_cast<_Float16*>(qb) = x1;
        lo_any |= (uint32_t)__builtin_bit_cast(uint16_t, x1) & 0x7FFFu;
      }
    }
    if (lo_any != 0) xflag[buf] = 1u;  // (every writer stores the same value)
  };
  auto load_row = [&](uint32_t fr, int& act, float& f0, float& f1, uint32_t& m) {
    if constexpr (ACTOR) {
      act = tk.action[fr];
      f0 = tk.old_logp[fr];
      f1 = tk.adv[fr];
      const uint8_t* mk = (tk.mask != nullptr && lo < no) ? (tk.mask + (long)fr * no + lo) : nullptr;
      m = 1u;
      if (mk != nullptr) m = *mk;
    } else {
      act = 0;
      f0 = tk.old_value[fr];
      f1 = tk.targets[fr];
      m = 0u;
    }
  };

  const long ntiles = (R + 31) / 32;
  long it = blockIdx.x;
  float xr[NR];
  int r_act = 0, n_act = 0;
  float r_f0 = 0.0f, r_f1 = 0.0f, n_f0 = 0.0f, n_f1 = 0.0f;
  uint32_t r_m = 1u, n_m = 1u;
  Cursor cs = cursor_at(DO_STAGE ? srow : 0), cl = cursor_at(DO_LOSS ? lrow : 0);
  int32_t ps_next = 0, pl_next = 0;
  uint32_t as_next = 0, al_next = 0;
  if (it < ntiles) {
    if constexpr (DO_STAGE) {
      cursor_gather(cs, ps_next, as_next);
      stage_issue(stage_row(ps_next, as_next), xr);
    }
    if constexpr (DO_LOSS) {
      cursor_gather(cl, pl_next, al_next);
      load_row((uint32_t)pl_next * Au + al_next, r_act, r_f0, r_f1, r_m);
    }
    if constexpr (DO_STAGE) {
      stage_commit(0, xr);
      cursor_advance(cs);
    }
    if constexpr (DO_LOSS) cursor_advance(cl);
    if (it + gridDim.x < ntiles) {
      if constexpr (DO_STAGE) { cursor_gather(cs, ps_next, as_next); cursor_advance(cs); }
      if constexpr (DO_LOSS) { cursor_gather(cl, pl_next, al_next); cursor_advance(cl); }
    }
  }
  __syncthreads();

  // per-lane LDS byte offsets
  const int swi = w8_sw(i);
  int rdA[4];  // row read of image row (16 nt + i), chunk 4 s + kg: + 4096 nt (+ image base, + plane) as immediates
#pragma unroll
  for (int s = 0; s < 4; ++s) rdA[s] = WROW * i + 16 * ((4 * s + kg) ^ swi);
  const int wrA = WROW * i + 16 * ((2 * v + (kg >> 1)) ^ swi) + 8 * (kg & 1);  // image write: features 16 v + 4 kg .. + 3 of row i
  const int tq = i >> 2, tp = i & 3;
  const int trow = 8 * kg + tq;                       // transposed reads: block row of the first 4-row block
  const int trb = WROW * trow + 8 * (tp & 1);         // (second block: 4 rows = 1024 bytes on; same swizzle)
  const int thx = 16 * ((tp >> 1) ^ w8_sw(trow));     // chunk 2 t + (tp >> 1) of column tile t: ((32 t) ^ thx)
  int trA[8];  // tile-invariant: kept in registers (an xor + add per use otherwise: ~30 vector instructions per tile)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    trA[t] = trb + ((32 * t) ^ thx);
    asm volatile("" : "+v"(trA[t]));
  }
  auto tr_addr = [&](int t) -> int { return trA[t]; };
  const int trOwn = tr_addr(v);                       // this wave's own 16 columns
  auto read_tr = [&](const u8* img, int plane, int a0) -> Frag {
    Frag f;
    f.hi = read_tr8(img + a0, WROW);
    f.lo = read_tr8(img + plane + a0, WROW);
    return f;
  };
  auto read_row = [&](const u8* img, int plane, int a0) -> Frag { return read_row_frag(img, plane, a0); };
  // dy image: row reads (lane (n, kg): outputs 4 kg .. + 3 of row 16 nt + n) and transposed reads (outputs x rows)
  const int dyRd = DYROW * i + 8 * kg;
  const int dyTr = DYROW * trow + 8 * tp;
  int buf = 0;

  WSTAMP_DECL
  for (; it < ntiles; it += gridDim.x, buf ^= 1) {
    WSTAMP(14);
    const long itn = it + gridDim.x;
    const bool have_next = itn < ntiles;
    const u8* const XSI = lds + L.xs + buf * WIMG;
    // rows of the next tile from the indices loaded a tile ago
    uint32_t xrow_next = DO_STAGE ? stage_row(ps_next, as_next) : 0u, lrow_next = DO_LOSS ? ((uint32_t)pl_next * Au + al_next) : 0u;
    if constexpr (DO_STAGE) asm volatile("" : "+v"(xrow_next));
    if constexpr (DO_LOSS) asm volatile("" : "+v"(lrow_next));

    // ---------------------------------------------------------------- P1: z1 = W1^T x^T (+ b1 through the ones column)
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const bool x_lo = __builtin_amdgcn_readfirstlane((int)xflag[buf]) != 0 || tk.force_xlo != 0;  // (wave-uniform: one branch per phase)
    if (x_lo) {
      Frag xb[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) xb[nt] = read_row(XSI + 4096 * nt, WPLANE32, rdA[0]);
#pragma unroll
      for (int s = 0; s < S1; ++s) {
        Frag b[2] = {xb[0], xb[1]};
        if (s + 1 < S1) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) xb[nt] = read_row(XSI + 4096 * nt, WPLANE32, rdA[(s + 1) & 3]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma3w(W1f[s], b[nt], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {  // x is exact in f16: hi plane only, two products
      half8 xb[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) xb[nt] = *reinterpret_cast<const half8*>(XSI + 4096 * nt + rdA[0]);
#pragma unroll
      for (int s = 0; s < S1; ++s) {
        const half8 b[2] = {xb[0], xb[1]};
        if (s + 1 < S1) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) xb[nt] = *reinterpret_cast<const half8*>(XSI + 4096 * nt + rdA[(s + 1) & 3]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma2w_blo0(W1f[s], b[nt], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(0);
    uint32_t relu1 = 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[nt][r] = fmaxf(ACTOR ? acc[nt][r] : acc[nt][r] * WU, 0.0f);
        relu1 |= (acc[nt][r] > 0.0f) ? (1u << (4 * nt + r)) : 0u;
      }
      const Frag4 f = split4(acc[nt]);
      *reinterpret_cast<half4*>(H1I + 4096 * nt + wrA) = f.hi;
      *reinterpret_cast<half4*>(H1I + 4096 * nt + WPLANE32 + wrA) = f.lo;
    }
    WSTAMP(1);
    // next tile's gathers: their latency hides under P2 .. P3
    if (have_next) {
      if constexpr (DO_STAGE) stage_issue(xrow_next, xr);
      if constexpr (DO_LOSS) load_row(lrow_next, n_act, n_f0, n_f1, n_m);
      if (itn + gridDim.x < ntiles) {
        if constexpr (DO_STAGE) { cursor_gather(cs, ps_next, as_next); cursor_advance(cs); }
        if constexpr (DO_LOSS) { cursor_gather(cl, pl_next, al_next); cursor_advance(cl); }
      }
    }
    WSTAMP(2);
    __syncthreads();  // A: h1 image complete
    WSTAMP(3);
    // the other x buffer's flag: its last readers (the previous tile's dW1) are behind this barrier, its next writers (this
    // tile's commit) behind barrier B
    if constexpr (DO_STAGE) {
      if (tid == (ROLE == 2 ? 256 : 0)) xflag[buf ^ 1] = 0u;
    }

    // ---------------------------------------------------------------- P2: z2 = b2 + W2^T h1^T ; head partial logits
    f32x4 h2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) h2[nt] = f32x4{b2r[0], b2r[1], b2r[2], b2r[3]};
    {
      Frag hb[2], wn;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) hb[nt] = read_row(H1I + 4096 * nt, WPLANE32, rdA[0]);
      // (not resident: W2[k = 32 s + 8 kg + e][n = 16 v + i] = rows 32 s + 8 kg .. of this wave's column tile of the W2 image)
      if constexpr (W2R < 1) wn = read_tr(W2I, W2PLANE, trOwn);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag b[2] = {hb[0], hb[1]};
        const Frag a = (s < W2R) ? W2f[s < W2R ? s : 0] : wn;
        if (s + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) hb[nt] = read_row(H1I + 4096 * nt, WPLANE32, rdA[s + 1]);
          if (s + 1 >= W2R) wn = read_tr(W2I + 32 * (s + 1) * WROW, W2PLANE, trOwn);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) h2[nt] = mfma3w(a, b[nt], h2[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(4);
    uint32_t relu2 = 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h2[nt][r] = fmaxf(ACTOR ? h2[nt][r] : h2[nt][r] * WU, 0.0f);
        relu2 |= (h2[nt][r] > 0.0f) ? (1u << (4 * nt + r)) : 0u;
      }
      const Frag4 f = split4(h2[nt]);
      // h2 image: only this wave's own columns are ever read back (dy^T . h2 below)
      *reinterpret_cast<half4*>(DZ1I + 4096 * nt + wrA) = f.hi;
      *reinterpret_cast<half4*>(DZ1I + 4096 * nt + WPLANE32 + wrA) = f.lo;
      // partial logits of this wave's 16 features: the accumulator tile is the B operand of the 16-deep product
      const f32x4 y = mfma3k16(W3h, f, f32x4{0, 0, 0, 0});
      // register r of lane (row i, kg) is output 4 kg + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * kg + r < NO) YP[(v * 32 + 16 * nt + i) * YSTR + 4 * kg + r] = y[r] * W3_UNSCALE;
      }
    }
    WSTAMP(5);
    __syncthreads();  // B: partial logits complete
    WSTAMP(6);

    // ---------------------------------------------------------------- P3: loss, d loss / d logits (x R) -> dy image
    if constexpr (DO_LOSS) {
      const float lo_c = 1.0f - tk.clip_eps, hi_c = 1.0f + tk.clip_eps;
      const bool rvalid = (it * 32 + lrow) < R;
      const float* yp = YP + lrow * YSTR + lo;
      float y = B3s[lo];
#pragma unroll
      for (int u = 0; u < 8; ++u) y += yp[u * 32 * YSTR];
      float dyo = 0.0f, pg_row = 0.0f, ent_row = 0.0f;
      if constexpr (ACTOR) {
        // masked Categorical over the NO lanes of the row (networks.py:116-124, distributions.py:146-165)
        const bool legal = (lo < no) && (r_m != 0u);
        const float z = legal ? y : -FLT_MAX;
        auto fmax_op = [](float a, float b) { return fmaxf(a, b); };
        auto add_op = [](float a, float b) { return a + b; };
        const float mx = group_allreduce<NO>(z, fmax_op);
        const float se = group_allreduce<NO>(expf(z - mx), add_op);
        const float logp = z - (mx + logf(se));
        const float pr = expf(logp);
        const float ent = group_allreduce<NO>((pr > 0.0f) ? -(pr * logp) : 0.0f, add_op);
        const int act = r_act;
        const float lp = group_allreduce<NO>((lo == act) ? logp : 0.0f, add_op);
        const float gae = (r_f1 - adv_mean) * adv_rstd;
        const float ratio = expf(lp - r_f0);
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float l1 = ratio * gae, l2 = rc * gae;
        const float pg = -fminf(l1, l2);
        const bool inside = (ratio >= lo_c) && (ratio <= hi_c);
        const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        const float dlp = rvalid ? (-(g1 + g2) * gae * ratio) : 0.0f;  // x R: the 1/R of .mean() is applied at the end
        const float ec = rvalid ? tk.ent_coef : 0.0f;
        const float oh = (lo == act) ? 1.0f : 0.0f;
        const float pl2 = (pr > 0.0f) ? logp : 0.0f;
        dyo = dlp * (oh - pr) + ec * pr * (pl2 + ent);
        if (z == -FLT_MAX) dyo = 0.0f;
        pg_row = pg;
        ent_row = ent;
      } else {
        // clipped value loss (ff_mappo.py:198-213) on lane 0 of the row: y is the value
        const float ov = r_f0, tg = r_f1;
        const float diff = y - ov;
        const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
        const float e1 = y - tg, e2 = vclip - tg;
        const float l1 = e1 * e1, l2 = e2 * e2;
        const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
        const float g1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        dyo = (rvalid && lo == 0) ? (tk.vf_coef * (g1 * e1 + g2 * e2)) : 0.0f;  // x R
        pg_row = 0.5f * fmaxf(l1, l2);
      }
      _Float16 da, db;
      split1(dyo, da, db);
      *reinterpret_cast<_Float16*>(DYI + lrow * DYROW + 2 * lo) = da;
      *reinterpret_cast<_Float16*>(DYI + DYPLANE + lrow * DYROW + 2 * lo) = db;
      ab3 += dyo;
      if (rvalid && lo == 0) {
        loss_a += pg_row * invR;
        loss_b += ent_row * invR;
      }
      r_act = n_act; r_f0 = n_f0; r_f1 = n_f1; r_m = n_m;
    }
    // the next tile's x rows have arrived: split + store them into the other buffer (its last readers, the dW1 product of
    // the previous tile, finished before barrier A; it is read again from the next tile's P1 on, three barriers away)
    if constexpr (DO_STAGE) {
      if (have_next) stage_commit(buf ^ 1, xr);
    }
    WSTAMP(7);
    __syncthreads();  // B2: dy of all 32 rows visible; every reader of the partial logits is done
    WSTAMP(8);

    // ---------------------------------------------------------------- P3b: dW3, dz2 = W3 dy, dW2
    {
      // gW3^T[o][f = 16 v + i] += sum_rows dy[row][o] h2[row][f]   (outputs >= no of the dy image are zero)
      Frag a, b;
      a.hi = read_tr8(DYI + dyTr, DYROW);
      a.lo = read_tr8(DYI + DYPLANE + dyTr, DYROW);
      b = read_tr(DZ1I, WPLANE32, trOwn);
      gW3 = mfma3w(a, b, gW3);
    }
    f32x4 dz[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      Frag4 b;
      b.hi = *reinterpret_cast<const half4*>(DYI + 16 * nt * DYROW + dyRd);
      b.lo = *reinterpret_cast<const half4*>(DYI + DYPLANE + 16 * nt * DYROW + dyRd);
      dz[nt] = mfma3k16(W3d, b, f32x4{0, 0, 0, 0});
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dz[nt][r] = ((relu2 >> (4 * nt + r)) & 1u) ? dz[nt][r] * W3_UNSCALE : 0.0f;
        ab2[r] += dz[nt][r];
      }
      const Frag4 f = split4(dz[nt]);
      *reinterpret_cast<half4*>(DZ2I + 4096 * nt + wrA) = f.hi;
      *reinterpret_cast<half4*>(DZ2I + 4096 * nt + WPLANE32 + wrA) = f.lo;
    }
    WSTAMP(9);
    {
      // gW2[k][n = 16 v + i] += sum_rows h1[row][k] dz2[row][n]: B = this wave's own columns of the dz2 image (written
      // above by this very wave: no barrier), A = column tile t of the h1 image
      // (operand reads run GWD products ahead of their MFMAs; measured on one box, headline env-steps/s: depth 1 68.6 M,
      // 2 68.2 M, 3 67.9 M - the phase is not waiting on its reads; MAVA_W8_GWD, default 1)
      const Frag b = read_tr(DZ2I, WPLANE32, trOwn);
      Frag an[GWD];
#pragma unroll
      for (int t = 0; t < GWD; ++t) an[t] = read_tr(H1I, WPLANE32, tr_addr(t));
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const Frag a = an[t % GWD];
        if (t + GWD < 8) an[t % GWD] = read_tr(H1I, WPLANE32, tr_addr(t + GWD));
        gW2[t] = mfma3w(a, b, gW2[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(10);
    __syncthreads();  // C: dz2 image complete
    WSTAMP(11);

    // ---------------------------------------------------------------- P4: dh1 = W2 dz2^T -> dz1 ; dW1
    {
      acc[0] = f32x4{0, 0, 0, 0};
      acc[1] = f32x4{0, 0, 0, 0};
      Frag an = read_row(W2I + 4096 * v, W2PLANE, rdA[0]);  // W2[16 v + i][32 s + 8 kg .. + 7]
      Frag bn[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bn[nt] = read_row(DZ2I + 4096 * nt, WPLANE32, rdA[0]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const Frag a = an;
        Frag b[2] = {bn[0], bn[1]};
        if (s + 1 < 4) {
          an = read_row(W2I + 4096 * v, W2PLANE, rdA[s + 1]);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) bn[nt] = read_row(DZ2I + 4096 * nt, WPLANE32, rdA[s + 1]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma3w(a, b[nt], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nt][r] = ((relu1 >> (4 * nt + r)) & 1u) ? (ACTOR ? acc[nt][r] : acc[nt][r] * WU) : 0.0f;
        const Frag4 f = split4(acc[nt]);
        *reinterpret_cast<half4*>(DZ1I + 4096 * nt + wrA) = f.hi;
        *reinterpret_cast<half4*>(DZ1I + 4096 * nt + WPLANE32 + wrA) = f.lo;
      }
    }
    WSTAMP(12);
    {
      // gW1[k][n = 16 v + i] += sum_rows x[row][k] dz1[row][n]   (row din of gW1 = db1 through the ones column)
      const Frag b = read_tr(DZ1I, WPLANE32, trOwn);
      constexpr int D1 = GWD < KT1 ? GWD : KT1;
      if (x_lo) {
        Frag an[D1];
#pragma unroll
        for (int t = 0; t < D1; ++t) an[t] = read_tr(XSI, WPLANE32, tr_addr(t));
#pragma unroll
        for (int t = 0; t < KT1; ++t) {
          const Frag a = an[t % D1];
          if (t + D1 < KT1) an[t % D1] = read_tr(XSI, WPLANE32, tr_addr(t + D1));
          gW1[t] = mfma3w(a, b, gW1[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        half8 an[D1];
#pragma unroll
        for (int t = 0; t < D1; ++t) an[t] = read_tr8(XSI + tr_addr(t), WROW);
#pragma unroll
        for (int t = 0; t < KT1; ++t) {
          const half8 a = an[t % D1];
          if (t + D1 < KT1) an[t % D1] = read_tr8(XSI + tr_addr(t + D1), WROW);
          gW1[t] = mfma2w_alo0(a, b, gW1[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    WSTAMP(13);
    // no barrier here: the next tile's P1 reads the other x buffer (complete since B2) and writes the h1 image, whose
    // last readers (dW2) sit before barrier C; every other image is rewritten only behind barriers A .. B2
  }
#ifdef MAVA_STAMPS
  if (tk.stamps != nullptr && blockIdx.x == 0 && l == 0) {
    for (int k = 0; k < 16; ++k) tk.stamps[v * 16 + k] = ws_acc[k];
  }
#endif

  // ------------------------------------------------------------------ epilogue: one slab per block (x 1/R)
  __syncthreads();
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const int oB2 = mlp_off_b2(din), oB3 = mlp_off_b3(din, no);
  const int Pn = mlp_param_count(din, no);
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * t + 4 * kg + r;
      if (k <= din) slab[k * MLP_H + 16 * v + i] = gW1[t][r] * invR;  // row din = db1
    }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[oW2 + (16 * t + 4 * kg + r) * MLP_H + 16 * v + i] = gW2[t][r] * invR;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = ab2[r];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) x += __shfl_xor(x, m, 64);
    if (i == 0) slab[oB2 + 16 * v + 4 * kg + r] = x * invR;
    const int o = 4 * kg + r;
    if (o < no) slab[oW3 + (16 * v + i) * no + o] = gW3[r] * invR;
  }
  float* red = reinterpret_cast<float*>(lds + L.h1);  // epilogue scratch (the tile loop is over)
  if constexpr (DO_LOSS) {
    float x = ab3;
#pragma unroll
    for (int m = NO; m < 64; m <<= 1) x += __shfl_xor(x, m, 64);
    if (l < NO) red[v * NO + l] = x;
    for (int o = 32; o > 0; o >>= 1) {
      loss_a += __shfl_down(loss_a, o, 64);
      loss_b += __shfl_down(loss_b, o, 64);
    }
    if (l == 0) { red[8 * NO + 2 * v] = loss_a; red[8 * NO + 2 * v + 1] = loss_b; }
  }
  __syncthreads();
  constexpr int LW = LOSS_THREADS / 64;  // waves that ran the loss
  if (tid < no) {
    float x = 0.0f;
#pragma unroll
    for (int u = 0; u < LW; ++u) x += red[u * NO + tid];
    slab[oB3 + tid] = x * invR;
  }
  if (tid == 0) {
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int u = 0; u < LW; ++u) { a += red[8 * NO + 2 * u]; b += red[8 * NO + 2 * u + 1]; }
    slab[Pn] = a;
    slab[Pn + 1] = b;
  }
}

// Steps of the layer-2 forward operand kept in registers, per instantiation: two of the four where the register file has room
// (kernel-resource remarks: 254 of 256 registers at <8, 3, 2>, no spills), none in the wide instantiations
constexpr int w8_w2r(int no, int s1, int xv) {
#ifdef MAVA_W8_W2R
  return MAVA_W8_W2R;
#else
  return (no == 8) ? ((s1 <= 2) ? 2 : ((s1 == 3 && xv == 2) ? 2 : 0)) : ((s1 <= 2) ? 2 : 0);
#endif
}
template <int NO, int S1, int XV, int W2R, bool ACTOR>
__global__ __launch_bounds__(512, 2) void ppo_train_w8_kernel(TrainTask tk, W8Layout L) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  if constexpr (NO == 8) {
    if (threadIdx.x < 256) {
      w8_body<NO, S1, XV, W2R, ACTOR, 1>(tk, L, lds);
    } else {
      w8_body<NO, S1, XV, W2R, ACTOR, 2>(tk, L, lds);
    }
  } else {
    w8_body<NO, S1, XV, W2R, ACTOR, 0>(tk, L, lds);
  }
}

template <int NO, int S1, int XV, bool ACTOR>
int launch_w8(const TrainTask& tk, int n_slab, hipStream_t s) {
  const W8Layout L = make_w8_layout();
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)ppo_train_w8_kernel<NO, S1, XV, w8_w2r(NO, S1, XV), ACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       L.end));
    attr_set = true;
  }
  hipLaunchKernelGGL((ppo_train_w8_kernel<NO, S1, XV, w8_w2r(NO, S1, XV), ACTOR>), dim3(n_slab), dim3(512), L.end, s, tk, L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, int S1, bool ACTOR>
int dispatch_w8_xv(const TrainTask& tk, int n_slab, hipStream_t s) {
  if (tk.din % 2 == 0 && ((uintptr_t)tk.x) % 8 == 0) return launch_w8<NO, S1, 2, ACTOR>(tk, n_slab, s);
  // (97 .. 127 inputs staged one float at a time need more staging registers than the eight-wave kernel has: hundreds of
  // spilled registers - the four-wave kernel takes that shape)
  if constexpr (S1 == 4) return 1;
  else return launch_w8<NO, S1, 1, ACTOR>(tk, n_slab, s);
}

template <int NO, bool ACTOR>
int dispatch_w8(const TrainTask& tk, int n_slab, hipStream_t s) {
  const int s1 = (tk.din + 1 + 31) / 32;  // 32-input steps of layer 1, including the ones (bias) column
  switch (s1) {
#ifndef MAVA_FAST_BUILD
    case 1: return dispatch_w8_xv<NO, 1, ACTOR>(tk, n_slab, s);
    case 2: return dispatch_w8_xv<NO, 2, ACTOR>(tk, n_slab, s);
    case 4: return dispatch_w8_xv<NO, 4, ACTOR>(tk, n_slab, s);
#endif
    case 3: return dispatch_w8_xv<NO, 3, ACTOR>(tk, n_slab, s);
    default: return 1;
  }
}

}  // namespace

// Discrete actor (<= 16 actions) or the value network without input aggregation, input width <= 127.  Returns MAVA_OK, 1 when
// the shape is not instantiated here (the caller then runs ppo_train_h2.hip's four-wave kernels), or a negative error code.
int mava_train_w8_launch(const TrainTask& tk, int n_slab, bool actor, hipStream_t s) {
  if (tk.action_f != nullptr || tk.din + 1 > 128) return 1;
  if (!actor) {
    if (tk.agg > 1 || tk.no != 1) return 1;
    return dispatch_w8<8, false>(tk, n_slab, s);
  }
  if (tk.no <= 8) return dispatch_w8<8, true>(tk, n_slab, s);
#ifndef MAVA_FAST_BUILD
  if (tk.no <= 16) return dispatch_w8<16, true>(tk, n_slab, s);
#endif
  return 1;
}
