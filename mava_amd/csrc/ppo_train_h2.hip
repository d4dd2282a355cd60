// Fused PPO minibatch kernels on the f16 matrix pipe with SPLIT operands (the "f16x2" form of ppo_train.hip).
//
// Reference: mava/systems/ppo/ff_mappo.py:150-180 (_actor_loss_fn), :182-201 (_critic_loss_fn), :204-218
// (value_and_grad), :268-285 (shuffle as an index vector) - the same math as ppo_train.hip, phase by phase.
//
// Arithmetic.  Every matrix operand x (f32) is split into two f16 terms, hi = f16(x), lo = f16(x - hi) (round to
// nearest: x = hi + lo to ~2^-22 |x|), and a product a.b is evaluated as a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on
// v_mfma_f32_32x32x16_f16 with f32 accumulation (the dropped a_lo.b_lo term is 2^-22 relative): three 32-cycle
// MFMAs per 16 inputs instead of eight 64-cycle exact-f32 MFMAs - 5.3x less matrix-pipe time at ~22 mantissa bits
// per operand.  The backward pass runs in units of R x gradient (R = rows of the minibatch: the 1/R of the losses'
// .mean() is applied once when the slab is written), so loss gradients are O(1) - far from f16's subnormal range -
// and the tiny W3 of a freshly initialised actor head (orthogonal(0.01)) is scaled by 2^7 inside the kernel.
// Operands must stay below f16's 65504 (observations / activations / weights of these networks are orders of
// magnitude smaller); the gradient parity tests (1e-4 against the float64 oracle) run on this kernel too.
//
// MI355X mapping.  Persistent 256-thread blocks, one per CU, 32-row tiles, wave w owns feature slice [32w, 32w+32) of
// every layer; products are transposed (feature on the accumulator register, batch row on the lane) like ppo_train.hip.
// What changed with the 16-bit MFMA (A[m][8 consecutive k], B[8 consecutive k][n] per lane):
//   * the WEIGHTS never touch LDS: each lane keeps its pre-split fragments of W1 / W2 (column slice for layer 2, row
//     slice for the backward product) / W3 in registers for the whole launch (8 registers per 16-input step);
//   * activations cross waves through ONE LDS image per tensor, [32 rows][128 features] f16 in a hi and a lo plane
//     (row stride 272 B = an odd number of 16-byte slots: conflict-free ds_read_b128 row reads), written once by the
//     producer (which splits) and read two ways: ds_read_b128 along a row where the product sums over FEATURES
//     (layer 2, dh1, dz2), ds_read_b64_tr_b16 (hardware transpose) where it sums over BATCH ROWS (the weight
//     gradients h1^T.dz2, x^T.dz1, dy^T.h2) - no second layout, no consumer-side splitting;
//   * the head's partial logits take the layer-2 accumulator straight as their B operand (accumulator registers
//     8s..8s+7 of a lane ARE the k-step-s fragment of the next product, in a permuted k order the W3 fragment follows);
//   * the x tile is double-buffered in LDS, which removes two of the seven barriers per tile.
#include "h2_core.h"
#include "ppo_train_task.h"
#include "ctx.h"

// Operand prefetch depth of the weight-gradient products (0 = pairs of reads followed by pairs of products).  A rolling
// prefetch (operand of product i + depth read while product i runs) was measured at depth 3 (narrow kernels) and 2 / 3 (the
// wide kernel's loader role): actor 5.12 vs 5.10 ms per update - nothing -, critic 2.31 / 2.43 vs 2.20 ms - the extra live
// fragments spill in the wide roles (44 / 76 bytes of scratch per lane).  Kept selectable; off.
#ifndef MAVA_GW_DEPTH
#define MAVA_GW_DEPTH 0
#endif
#define GW_DEPTH MAVA_GW_DEPTH
#ifndef MAVA_GW_DEPTH_WIDE
#define MAVA_GW_DEPTH_WIDE 0
#endif

namespace {

using namespace h2;

constexpr int STATS_BLOCKS = 128;  // == ppo_train.hip (mava_adv_stats_blocks)

#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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

struct H2Layout {  // byte offsets into the dynamic LDS array (all multiples of 16)
  int h1, dz2, dz1, w2, xs, xs_plane, xs_row, dy, yp, agg, small, end;
};

// WIDE (input width > 95): the layer-1 weights stream from a pre-split global copy through a register ring and the
// x tile is single-buffered; otherwise W1 is register-resident and the x tile double-buffered.
template <int NO, int S1, bool WIDE>
H2Layout make_h2_layout(bool actor) {
  constexpr int KT1 = (S1 + 1) / 2;
  H2Layout L;
  L.h1 = 0;
  L.dz2 = L.h1 + IMG_BYTES;
  L.dz1 = L.dz2 + IMG_BYTES;  // also the h2 image (actor): h2's last reader (dy^T.h2) runs before barrier C, dz1 is written behind it
  L.w2 = L.dz1 + IMG_BYTES;   // W2 [128 k][128 n]: row reads for dh1 = W2.dz2, transposed reads for layer 2
  L.xs = L.w2 + 4 * IMG_BYTES;
  L.xs_row = 2 * 32 * KT1 + 16;  // [32 rows][32*KT1 inputs] f16 + 16: an odd number of 16-byte slots
  L.xs_plane = 32 * L.xs_row;
  L.dy = L.xs + (WIDE ? 1 : 2) * 2 * L.xs_plane;  // buffers x (hi, lo)
  // partial logits (f32) live inside an image that is idle between barriers A and B2: dz2's (actor), dz1's (critic)
  L.yp = actor ? L.dz2 : L.dz1;
  L.agg = L.dy + (actor ? 2 * DY_PLANE : 0);
  L.small = L.agg + 8 * 33 * 4;  // f32: b2[128] | b3[32] | misc[16] | W3[128] (critic)
  L.small = (L.small + 15) & ~15;
  L.end = L.small + (128 + 32 + 16 + 128) * 4;
  return L;
}
constexpr int W2_PLANE = 128 * SW_ROW;  // (the images of this kernel are the swizzled ones of h2_core.h; regions keep IMG_BYTES)
constexpr int W1_RING = 3;  // WIDE: register ring depth in 16-input steps (S1 is padded to a multiple of it)

// WIDE: pre-split copy of W1 in fragment order (h2_core.h pack_w1_body), written by this kernel ahead of a launch - or by
// the Adam kernel of the minibatch before it (adam.hip, mava_ppo_finish_f32), which then marks the handle's copy fresh.
template <int STEPS>
__global__ __launch_bounds__(256) void pack_w1_kernel(const float* __restrict__ P, int din, uint4* __restrict__ out, float scale) {
  pack_w1_body<STEPS>(P, din, out, threadIdx.x, scale);
}

// ROLE 0: one 4-wave group does everything (narrow inputs).  WIDE launches run 8 waves in two roles with disjoint register
// sets (two waves per SIMD, 256 registers each): ROLE 1, the CHAIN group (waves 0-3: layers, loss, backward chain, dW2,
// small gradients; W1 through a register ring), and ROLE 2, the LOADER group (waves 4-7: gathers and splits the x tiles,
// owns the 9 x 16 accumulator registers of dW1).  Every role executes the same barrier sequence.
template <int NO, int S1, bool ACTOR, bool WIDE, int XV, int ROLE>
__device__ __forceinline__ void h2_body(const TrainTask& tk, const H2Layout& L, const uint4* __restrict__ w1p, u8* lds) {
  static_assert(!WIDE || S1 % W1_RING == 0, "WIDE: S1 must be a multiple of the W1 ring depth");
  static_assert((ROLE == 0) == !WIDE, "roles 1 / 2 belong to WIDE launches");
  constexpr bool CHAIN = ROLE != 2, LOADER = ROLE != 1;
#if defined(MAVA_NO_W2_RESIDENT) || defined(MAVA_W2_RESIDENT_P4)
  constexpr bool W2_RESIDENT = false;
#else
  constexpr bool W2_RESIDENT = !WIDE && NO <= 16;
#endif
#ifdef MAVA_W2_RESIDENT_P4
  constexpr bool W2_RES4 = !WIDE && ACTOR && NO <= 16;
#else
  constexpr bool W2_RES4 = false;
#endif
  constexpr int NTHR = WIDE ? 512 : 256;
  // the critic's weights are split as WS * w, its layer accumulators unscaled by WU (h2_core.h W_SCALE_CRITIC); actor: 1
  constexpr float WS = ACTOR ? 1.0f : W_SCALE_CRITIC, WU = 1.0f / WS;
  constexpr int KT1 = (S1 + 1) / 2;  // 32-input tiles of the layer-1 weight gradient
  u8* const H1I = lds + L.h1;
  u8* const DZ2I = lds + L.dz2;
  u8* const DZ1I = lds + L.dz1;
  u8* const H2I = lds + L.dz1;  // alias, see make_h2_layout
  u8* const W2I = lds + L.w2;
  u8* const DYI = lds + L.dy;
  float* const YP = reinterpret_cast<float*>(lds + L.yp);
  float* const AGG = reinterpret_cast<float*>(lds + L.agg);
  float* const B2s = reinterpret_cast<float*>(lds + L.small);
  float* const B3s = B2s + 128;
  float* const misc = B3s + 32;
  float* const W3s = misc + 16;  // critic: the 128 head weights (read per tile: head on the VALU, dz2 = W3[f] * dy)
  const int xs_row = L.xs_row, xs_plane = L.xs_plane;

  const int tid = threadIdx.x & 255;  // thread of its 4-wave group
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, r = lane & 31;
  const int srow = tid >> 3, l8 = tid & 7;  // staging role: row srow of the tile, 8 threads per row
  const int din = tk.din, no = tk.no;
  const long R = (!ACTOR && tk.agg > 1) ? (long)tk.Rb : (long)tk.Rb * tk.A;
  const float invR = 1.0f / (float)((long)tk.Rb * tk.A);
  constexpr int NPC = (32 * KT1 / XV + 7) / 8;  // staged pieces (XV floats each) per thread, 8 threads per row
  constexpr int NR = NPC * XV;

  // ---------------------------------------------------------------- prologue: LDS images, small vectors
  for (int i = threadIdx.x * 16; i < L.end; i += NTHR * 16) *reinterpret_cast<uint4*>(lds + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const float* const P = tk.params;
  const int oW2 = mlp_off_w2(din), oW3 = mlp_off_w3(din);
  if (CHAIN) {
    if (tid < 128) B2s[tid] = P[mlp_off_b2(din) + tid];
    if (tid < no) B3s[tid] = P[mlp_off_b3(din, no) + tid];
    if (!ACTOR && tid < 128) W3s[tid] = P[oW3 + tid];
    // the ones column of the x buffers: "row din" of W1 in the flat parameter vector is b1
    if (tid < (WIDE ? 32 : 64)) {
      const int b = tid >> 5, row = tid & 31;
      *reinterpret_cast<_Float16*>(lds + L.xs + b * 2 * xs_plane + row * xs_row + 2 * din) = (_Float16)1.0f;
    }
    // W2 image: thread = (output feature n, half of the k range); sequential over k with the error-diffusion carry
    // of split1_carry (layer 2 sums over k: the representation errors of a column then sum to two final carries)
    {
      const int n = tid & 127, k0 = 64 * (tid >> 7);
      float carry = 0.0f;
#pragma unroll 8
      for (int k = k0; k < k0 + 64; ++k) {
        _Float16 x0, x1;
        split1_carry(P[oW2 + k * MLP_H + n] * WS, carry, x0, x1);
        const int o = sw_off(k, n >> 3) + 2 * (n & 7);
        *reinterpret_cast<_Float16*>(W2I + o) = x0;
        *reinterpret_cast<_Float16*>(W2I + W2_PLANE + o) = x1;
      }
    }
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
  }
  // ---------------------------------------------------------------- weight fragments kept in registers
  // layer 1 (A operand): W1[k = 16s + 8h + e][f = 32w + r]; k == din is b1, k > din is zero.  Narrow inputs: all S1
  // steps resident; WIDE: a ring of W1_RING steps fed from the pre-split global copy.
  // WIDE: layer 1 is split along its input index between the two groups - the chain group takes steps [0, S1 / 2), the
  // loader group (idle during P1 otherwise) [S1 / 2, S1) - each through its own ring; the loader's partial sums cross in
  // LDS (barrier A0).  The ring is latency-bound (3 steps x 2 KB in flight per wave against a ~700-cycle L2 round trip:
  // 6.0 K cycles per tile for 1.7 K of MFMAs with one group); the second group doubles the bytes in flight.
  constexpr int S_LO = (WIDE && !CHAIN) ? S1 / 2 : 0, S_HI = (WIDE && CHAIN) ? S1 / 2 : S1, RL = S_HI - S_LO;
  static_assert(!WIDE || ((S1 / 2) % W1_RING == 0 && (S1 - S1 / 2) % W1_RING == 0), "WIDE: both halves must be multiples of the ring");
  constexpr int NW1 = WIDE ? W1_RING : (CHAIN ? S1 : 1);
  Frag W1f[NW1];
  // The fragment of step s sits 8 KB x s behind this lane's base: too far for an instruction immediate.  Written against a
  // loop-invariant base the compiler forms one 64-bit address per step, hoists all of them out of the tile loop and spills
  // them - and every reload (scratch shares vmcnt with the ring's loads) drains the ring: the wide critic spent 513 cycles
  // per step on 96 cycles of matrix work.  A base made opaque once per tile keeps the per-step addresses inside the loop:
  // two adds from a live register.
  const uint4* w1p_lane = w1p + 2 * (w * 64 + lane);  // + 2 * 256 * s
  auto w1_fetch = [&](int s) -> Frag {
    Frag f;
    f.hi = __builtin_bit_cast(half8, w1p_lane[512 * s]);
    f.lo = __builtin_bit_cast(half8, w1p_lane[512 * s + 1]);
    return f;
  };
  float w1_carry = 0.0f;
#pragma unroll
  for (int s = 0; s < (CHAIN ? NW1 : 0); ++s) {  // (the loader group of WIDE launches fills its ring at the start of each P1)
    if (WIDE) {
      W1f[s] = w1_fetch(S_LO + s);
    } else {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * h + e;
        v[e] = (k <= din) ? P[k * MLP_H + 32 * w + r] * WS : 0.0f;
      }
      W1f[s] = split8_carry(v, w1_carry);
    }
  }
  // head: logits^T[o = r][row] += sum over the wave's 32 features, B = the layer-2 accumulator itself: element e of
  // lane half h in k-step s is feature 32w + 16s + 8(e>>2) + 4h + (e&3)
  Frag W3h[2];
#pragma unroll
  for (int s = 0; s < ((CHAIN && ACTOR) ? 2 : 0); ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 32 * w + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
      v[e] = (r < no) ? P[oW3 + f * no + r] * W3_SCALE : 0.0f;
    }
    W3h[s] = split8(v);
  }
  Frag W3d;  // actor: A of dz2 = W3 . dy: W3[f = 32w + r][o = 8h + e]
  if (CHAIN && ACTOR) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = 8 * h + e;
      v[e] = (o < no) ? P[oW3 + (32 * w + r) * no + o] * W3_SCALE : 0.0f;
    }
    W3d = split8(v);
  }
  __syncthreads();
  const float adv_mean = ACTOR ? misc[0] : 0.0f;
  const float adv_rstd = ACTOR ? misc[1] : 0.0f;

  // persistent accumulators (R x gradient units): wave w owns output columns [32w, 32w+32) of dW1 and dW2
  f32x16 gW1[LOADER ? KT1 : 1], gW2[4], gW3;
#pragma unroll
  for (int t = 0; t < (LOADER ? KT1 : 1); ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) gW1[t][q] = 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) gW2[t][q] = 0.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) gW3[q] = 0.0f;
  float ab2[16], aW3r[ACTOR ? 1 : 16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ab2[q] = 0.0f;
#pragma unroll
  for (int q = 0; q < (ACTOR ? 1 : 16); ++q) aW3r[q] = 0.0f;
  float ab3 = 0.0f, loss_a = 0.0f, loss_b = 0.0f;

  // ---------------------------------------------------------------- row cursors and prefetch (as ppo_train.hip)
  constexpr int NP = ACTOR ? NO / 8 : 1;
  constexpr int GR = ACTOR ? 64 / NO : 32;
  const int lo = ACTOR ? (lane & (NO - 1)) : 0;
  auto loss_row = [&](int q) -> int { return ACTOR ? (8 * w + q * GR + lane / NO) : r; };
  const int slot = 2 * w + h;
  uint32_t slot_c = (!ACTOR && tk.agg > 1) ? (uint32_t)(slot < tk.agg ? slot : 0) : 0u;
  auto load_row = [&](long fr, int& act, float& f0, float& f1, uint32_t& m) {
    if (ACTOR) {
      act = tk.action[fr];
      f0 = tk.old_logp[fr];
      f1 = tk.adv[fr];
      const uint8_t* mk = (tk.mask != nullptr && lo < no) ? (tk.mask + fr * no + lo) : nullptr;
      m = 1u;
      if (mk != nullptr) m = *mk;
    } else {
      act = 0;
      // (32-bit row arithmetic, rows < 2^31: a 64-bit lane constant here was hoisted, spilled and reloaded per tile)
      const long fa = (tk.agg > 1) ? (long)((uint32_t)fr * (uint32_t)tk.agg + slot_c) : fr;
      f0 = tk.old_value[fa];
      f1 = tk.targets[fa];
      m = 0u;
    }
  };
  const uint32_t Au = (!ACTOR && tk.agg > 1) ? 1u : (uint32_t)tk.A;
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
  auto cursor_gather = [&](const Cursor& c, int32_t& p_raw, uint32_t& a_out) {
    const bool in = c.q < (uint32_t)R;
    const uint32_t b = in ? c.b : b_last;
    a_out = in ? c.a : a_last;
    p_raw = tk.idx ? tk.idx[b] : (int32_t)(tk.idx_base + (long)b);
  };
  auto stage_row = [&](int32_t p_raw, uint32_t a) -> uint32_t {
    const uint32_t fr = (uint32_t)p_raw * Au + a;
    return (tk.xshare == 1) ? fr : ((uint32_t)tk.xshare == Au ? (uint32_t)p_raw : fr / (uint32_t)tk.xshare);
  };
  // x staging: thread l8 of a row takes the XV-float pieces l8, l8 + 8, ...; a piece past the row end is loaded from the
  // row start and stored to a dummy slot (branch-free, see ppo_train.hip stage_load / stage_write)
  const int nv = din / XV;  // pieces per row (XV > 1 only when din % XV == 0 and the rows are XV*4-byte aligned)
  auto stage_issue = [&](uint32_t xrow_idx, float (&xr)[NR]) {
    const float* xrow = tk.x + (long)xrow_idx * din;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int c = l8 + 8 * i;
      const int cc = (c < nv) ? c : 0;
      if (XV == 4) {
        const float4 v = reinterpret_cast<const float4*>(xrow)[cc];
        xr[4 * i] = v.x; xr[4 * i + 1] = v.y; xr[4 * i + 2] = v.z; xr[4 * i + 3] = v.w;
      } else if (XV == 2) {
        const float2 v = reinterpret_cast<const float2*>(xrow)[cc];
        xr[2 * i] = v.x; xr[2 * i + 1] = v.y;
      } else {
        xr[i] = xrow[cc];
      }
    }
  };
  u8* const xs_dummy = reinterpret_cast<u8*>(misc + 8);  // 16 bytes nobody reads
  // wide kernel: xflag[tile parity] != 0 when some staged value of that tile's x rows has a non-zero low f16 term.  Inputs
  // exact in f16 (flags, one-hot ids, small integers: the whole global state of RobotWarehouse and of the synthetic env) leave
  // it 0; the loader group's dW1 product then skips x_lo . dz1 (exactly 0) and the reads of that plane - same bits.
  unsigned* const xflag = reinterpret_cast<unsigned*>(misc + 2);
  int tpar = 0;  // parity of the tile being processed
  auto stage_commit = [&](int buf, const float (&xr)[NR], int flag_slot = 0) {
    u8* base = lds + L.xs + buf * 2 * xs_plane + srow * xs_row;
    uint32_t lo_any = 0;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int c = l8 + 8 * i;
      const bool ok = c < nv;
      u8* qa = ok ? (base + 2 * XV * c) : xs_dummy;
      u8* qb = ok ? (base + xs_plane + 2 * XV * c) : xs_dummy;
      if (XV == 4) {
        half4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 x0, x1; split1(xr[4 * i + e], x0, x1); a[e] = x0; b[e] = x1; }
        *reinterpret_cast<half4*>(qa) = a;
        *reinterpret_cast<half4*>(qb) = b;
        if constexpr (WIDE) {
          const uint2 bb = __builtin_bit_cast(uint2, b);
          lo_any |= (bb.x | bb.y) & 0x7FFF7FFFu;
        }
      } else if (XV == 2) {
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        half2v a, b;
#pragma unroll
        for (int e = 0; e < 2; ++e) { _Float16 x0, x1; split1(xr[2 * i + e], x0, x1); a[e] = x0; b[e] = x1; }
        *reinterpret_cast<half2v*>(qa) = a;
        *reinterpret_cast<half2v*>(qb) = b;
        if constexpr (WIDE) lo_any |= __builtin_bit_cast(uint32_t, b) & 0x7FFF7FFFu;
      } else {
        _Float16 x0, x1;
        split1(xr[i], x0, x1);
        *reinterpret_cast<_Float16*>(qa) = x0;
        *reinterpret_cast<_Float16*>(qb) = x1;
        if constexpr (WIDE) lo_any |= (uint32_t)__builtin_bit_cast(uint16_t, x1) & 0x7FFFu;
      }
    }
    if constexpr (WIDE) {
      if (lo_any != 0) xflag[flag_slot] = 1u;  // (every writer stores the same value)
    }
  };

  const long ntiles = (R + 31) / 32;
  long it = blockIdx.x;
  float xr[NR];
  int r_act[NP] = {}, n_act[NP] = {};
  float r_f0[NP] = {}, r_f1[NP] = {}, n_f0[NP] = {}, n_f1[NP] = {};
  uint32_t r_m[NP] = {}, n_m[NP] = {};
  Cursor cs = cursor_at(srow), cl[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) cl[q] = cursor_at(loss_row(q));
  int32_t ps_next = 0, pl_next[NP] = {};
  uint32_t as_next = 0, al_next[NP] = {};
  if (it < ntiles) {
    if (LOADER) {
      cursor_gather(cs, ps_next, as_next);
      stage_issue(stage_row(ps_next, as_next), xr);
    }
    if (CHAIN) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        cursor_gather(cl[q], pl_next[q], al_next[q]);
        load_row((long)((uint32_t)pl_next[q] * Au + al_next[q]), r_act[q], r_f0[q], r_f1[q], r_m[q]);
      }
    }
    if (LOADER) {
      stage_commit(0, xr);
      cursor_advance(cs);
    }
    if (CHAIN) {
#pragma unroll
      for (int q = 0; q < NP; ++q) cursor_advance(cl[q]);
    }
    if (it + gridDim.x < ntiles) {
      if (LOADER) {
        cursor_gather(cs, ps_next, as_next);
        cursor_advance(cs);
      }
      if (CHAIN) {
#pragma unroll
        for (int q = 0; q < NP; ++q) cursor_gather(cl[q], pl_next[q], al_next[q]);
#pragma unroll
        for (int q = 0; q < NP; ++q) cursor_advance(cl[q]);
      }
    }
  }
  __syncthreads();

  // per-lane LDS byte offsets
  const SwRow rowB = sw_row(r, h);                             // step s: features 16s + 8h .. + 7 of image row r
  const SwTr trS = sw_tr(lane);                                // transposed reads of the swizzled images
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, g1 = (lane >> 4) & 1;
  // transposed reads: block row (8h + tq) of a 16-row step, columns (16 g1 + 4 tp) .. + 3 of a 32-column tile
  const int trX = (8 * h + tq) * xs_row + 2 * (16 * g1 + 4 * tp);
  const int trD = (8 * h + tq) * DY_ROW + 2 * (16 * g1 + 4 * tp);
  const SwRow w2row = sw_row(32 * w + r, h);                   // step s: W2[32w + r][16s + 8h .. + 7]
  int buf = 0;
  // narrow inputs: this wave's layer-2 operand (its 32 columns of W2, all 8 steps: 64 registers of the ~110 the role
  // leaves free) stays in registers - P2 then reads only the h1 image, half the LDS bytes of the phase
  Frag w2r[(W2_RESIDENT || W2_RES4) ? 8 : 1];
  if constexpr (W2_RESIDENT) {
#pragma unroll
    for (int s = 0; s < 8; ++s) w2r[s] = sw_read_tr(W2I, W2_PLANE, trS, s, w);
  }
  if constexpr (W2_RES4) {  // (measured alternative: the backward operand W2[32w + r][.] instead)
#pragma unroll
    for (int s = 0; s < 8; ++s) w2r[s] = sw_read_row(W2I, W2_PLANE, w2row, s);
  }

  STAMP_DECL
  for (; it < ntiles; it += gridDim.x, buf ^= (WIDE ? 0 : 1), tpar ^= 1) {
    STAMP(14);
    const bool valid = (it * 32 + r) < R;
    const long itn = it + gridDim.x;
    const bool have_next = itn < ntiles;
    uint32_t xrow_next = LOADER ? stage_row(ps_next, as_next) : 0u;
    uint32_t fr_next[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) fr_next[q] = CHAIN ? ((uint32_t)pl_next[q] * Au + al_next[q]) : 0u;
    if (LOADER) asm volatile("" : "+v"(xrow_next));
    if (CHAIN) {
#pragma unroll
      for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(fr_next[q]));
    }
    const u8* const XSI = lds + L.xs + buf * 2 * xs_plane;

    // ---------------------------------------------------------------- P1: z1 = W1^T x^T (+ b1 through the ones column)
    f32x16 acc;
    uint32_t relu1 = 0;
    if constexpr (WIDE) asm volatile("" : "+v"(w1p_lane));  // (see w1_fetch: its addresses stay loop-variant)
    if constexpr (CHAIN || WIDE) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    // operand reads run ONE step ahead of their MFMAs (explicit double buffer + scheduling fence: without it the
    // compiler hoists every read of the unrolled loop and spills)
    if constexpr (WIDE && !CHAIN) {
      // the loader group's ring does not live across the tile (its registers hold the next x tile from the gather issue to
      // the commit): filled here, one exposed L2 round trip per tile in a group that would otherwise wait at barrier A0
#pragma unroll
      for (int s = 0; s < NW1; ++s) W1f[s] = w1_fetch(S_LO + s);
    }
    Frag xb = read_row_frag(XSI, xs_plane, r * xs_row + 16 * h + 32 * S_LO);
#pragma unroll
    for (int j = 0; j < RL; ++j) {  // this group's steps s = S_LO + j
      const Frag b = xb;
      if (j + 1 < RL) xb = read_row_frag(XSI, xs_plane, r * xs_row + 16 * h + 32 * (S_LO + j + 1));
      acc = mfma3(W1f[j % NW1], b, acc);
      if (WIDE) {
        // refill the slot consumed one step ago with step j - 1 + W1_RING of this tile
        if (j >= 1 && j - 1 + W1_RING < RL) W1f[(j - 1) % NW1] = w1_fetch(S_LO + j - 1 + W1_RING);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if constexpr (WIDE) {
      // the loader group's partial sums of its input range -> the chain group (the dz2 image region is idle here)
      float4* const xch = reinterpret_cast<float4*>(DZ2I) + (w * 4) * 64 + lane;
      if constexpr (LOADER) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xch[g * 64] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      }
      __syncthreads();  // A0
      if constexpr (CHAIN) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = xch[g * 64];
          acc[4 * g] += v.x; acc[4 * g + 1] += v.y; acc[4 * g + 2] += v.z; acc[4 * g + 3] += v.w;
        }
      }
    }
    if constexpr (CHAIN) {
    STAMP(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q] = fmaxf(ACTOR ? acc[q] : acc[q] * WU, 0.0f);
      relu1 |= (acc[q] > 0.0f) ? (1u << q) : 0u;
    }
    {
      half4 ph[4], pl[4];
      sw_write_image(H1I, r, w, h, acc, ph, pl);
    }
    }  // CHAIN
    STAMP(1);
    // next tile's gathers: issued after P1 (their latency hides under P2..P4)
    if constexpr (WIDE && LOADER) {
      // unconditional (row 0 when there is no next tile): the staging registers are then dead from the commit to here,
      // and P1's ring and accumulator live in them
      stage_issue(have_next ? xrow_next : 0u, xr);
    }
    if constexpr (WIDE && CHAIN) {
      // everything the gather block reads is pinned into registers here, before its first load: reloads of spilled values
      // (each followed by s_waitcnt vmcnt(0)) then find no gather of this block in flight
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        asm volatile("" : "+v"(fr_next[q]));
        asm volatile("" : "+v"(cl[q].q), "+v"(cl[q].b), "+v"(cl[q].a));
      }
      asm volatile("" : "+v"(slot_c));
    }
    if (have_next) {
      if (LOADER && !WIDE) stage_issue(xrow_next, xr);
      const bool more = itn + gridDim.x < ntiles;
      if (more) {
        if (LOADER) {
          cursor_gather(cs, ps_next, as_next);
          cursor_advance(cs);
        }
        if (CHAIN) {
          // cursor arithmetic first, loads after it: a cursor value that was spilled comes back with s_waitcnt vmcnt(0),
          // which must not find this block's own gathers in flight (2.4 K cycles per tile in the wide chain role)
          Cursor c_old[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) { c_old[q] = cl[q]; cursor_advance(cl[q]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NP; ++q) cursor_gather(c_old[q], pl_next[q], al_next[q]);
        }
      }
      if (CHAIN) {
#pragma unroll
        for (int q = 0; q < NP; ++q) load_row((long)fr_next[q], n_act[q], n_f0[q], n_f1[q], n_m[q]);
      }
    }
    if constexpr (WIDE && CHAIN) {
      // the chain group's ring for the NEXT tile (steps 0 .. W1_RING - 1: a whole tile of latency to hide under).  Issued
      // here, behind the gather block: a spilled cursor value reloaded there comes with s_waitcnt vmcnt(0), which waited
      // for these fetches when they were issued at the end of the P1 loop (3.0 K cycles per tile)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < NW1; ++s2) W1f[s2] = w1_fetch(S_LO + s2);
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(2);
    __syncthreads();  // A: h1 image complete
    STAMP(3);

    // ---------------------------------------------------------------- P2: z2 = b2 + W2^T h1^T ; head partial logits
    f32x16 h2;
    uint32_t relu2 = 0;
    if constexpr (CHAIN) {
#pragma unroll
    for (int q = 0; q < 16; ++q) h2[q] = B2s[32 * w + (q & 3) + 8 * (q >> 2) + 4 * h] * WS;  // (unscaled with the products below)
    {
      Frag an;
      if constexpr (!W2_RESIDENT) an = sw_read_tr(W2I, W2_PLANE, trS, 0, w);  // W2[16s+8h+e][32w+r]
      Frag bn = sw_read_row(H1I, SW_PLANE, rowB, 0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const Frag a = W2_RESIDENT ? w2r[s] : an, b = bn;
        if (s + 1 < 8) {
          if constexpr (!W2_RESIDENT) an = sw_read_tr(W2I, W2_PLANE, trS, s + 1, w);
          bn = sw_read_row(H1I, SW_PLANE, rowB, s + 1);
        }
        h2 = mfma3(a, b, h2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(4);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      h2[q] = fmaxf(ACTOR ? h2[q] : h2[q] * WU, 0.0f);
      relu2 |= (h2[q] > 0.0f) ? (1u << q) : 0u;
    }
    if (!ACTOR) {
      // critic head on the VALU: this lane's 16 features of row r, the two lane halves added by one exchange; the four
      // waves' partial values meet in YP
      // (this lane's 16 head weights are four aligned groups of four: 16-byte reads, issued together)
      float4 w3g[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) w3g[g] = *reinterpret_cast<const float4*>(W3s + 32 * w + 4 * h + 8 * g);
      float part = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 t = w3g[q >> 2];
        const float w3 = (q & 3) == 0 ? t.x : (q & 3) == 1 ? t.y : (q & 3) == 2 ? t.z : t.w;
        part = fmaf(h2[q], w3, part);
      }
      part += __shfl_xor(part, 32, 64);
      if (h == 0) YP[(w * 32 + r) * (NO + 1)] = part;
    } else {
      half4 ph[4], pl[4];
      sw_write_image(H2I, r, w, h, h2, ph, pl);
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
      // partial logits of this wave: register q of lane (row r, half h) is output (q&3) + 8(q>>2) + 4h
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
        if ((q & 3) + 8 * (q >> 2) < NO) {
          if (o < NO) YP[(w * 32 + r) * (NO + 1) + o] = yacc[q] * W3_UNSCALE;
        }
      }
    }
    }  // CHAIN
    STAMP(5);
    __syncthreads();  // B: partial logits (and the h2 image) complete
    STAMP(6);

    // ---------------------------------------------------------------- P3: loss, d loss / d logits (x R), dW3, dz2
    f32x16 dz;
    float dy0 = 0.0f;  // critic: d loss / d value of row r (x R)
    if constexpr (CHAIN) {
    if (ACTOR) {
      const float lo_c = 1.0f - tk.clip_eps, hi_c = 1.0f + tk.clip_eps;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int row = loss_row(q);
        const bool rvalid = (it * 32 + row) < R;
        const float* yp = YP + row * (NO + 1) + lo;
        const float y = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) + B3s[lo];
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
        const bool inside = (ratio >= lo_c) && (ratio <= hi_c);
        const float g1 = (l1 < l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
        const float g2 = inside ? (1.0f - g1) : 0.0f;
        const float dlp = rvalid ? (-(g1 + g2) * gae * ratio) : 0.0f;  // x R: the 1/R of .mean() is applied at the end
        const float ec = rvalid ? tk.ent_coef : 0.0f;
        const float oh = (lo == act) ? 1.0f : 0.0f;
        const float pl2 = (pr > 0.0f) ? logp : 0.0f;
        float dyo = dlp * (oh - pr) + ec * pr * (pl2 + ent);
        if (z == -FLT_MAX) dyo = 0.0f;
        _Float16 da, db;
        split1(dyo, da, db);
        *reinterpret_cast<_Float16*>(DYI + row * DY_ROW + 2 * lo) = da;
        *reinterpret_cast<_Float16*>(DYI + DY_PLANE + row * DY_ROW + 2 * lo) = db;
        ab3 += dyo;
        if (rvalid && lo == 0) {
          loss_a += pg * invR;
          loss_b += ent * invR;
        }
      }
    } else {
      const float* yp = YP + r * (NO + 1);
      const float v = (((yp[0] + yp[32 * (NO + 1)]) + yp[2 * 32 * (NO + 1)]) + yp[3 * 32 * (NO + 1)]) + B3s[0];
      const float ov = r_f0[0], tg = r_f1[0];
      const float diff = v - ov;
      const float vclip = ov + fminf(fmaxf(diff, -tk.clip_eps), tk.clip_eps);
      const float e1 = v - tg, e2 = vclip - tg;
      const float l1 = e1 * e1, l2 = e2 * e2;
      const bool inside = (diff >= -tk.clip_eps) && (diff <= tk.clip_eps);
      const float g1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
      const float g2 = inside ? (1.0f - g1) : 0.0f;
      dy0 = valid ? (tk.vf_coef * (g1 * e1 + g2 * e2)) : 0.0f;  // x R
      if (tk.agg > 1) {
        // this lane evaluated agent `slot` of row r: the value is shared by the row's agents, so the backward pass
        // runs once on the sum of their loss gradients
        const bool mine = valid && slot < tk.agg;
        dy0 = mine ? dy0 : 0.0f;
        if (mine) {
          loss_a += 0.5f * fmaxf(l1, l2) * invR;
          ab3 += dy0;
        }
        AGG[slot * 33 + r] = dy0;
      } else if (valid && w == 0 && h == 0) {
        loss_a += 0.5f * fmaxf(l1, l2) * invR;
        ab3 += dy0;
      }
    }
    }  // CHAIN
    STAMP(7);
    __syncthreads();  // B2: dy of all 32 rows visible; every reader of the partial logits is done
    STAMP(15);
    if constexpr (CHAIN) {
    if (ACTOR) {
      // gW3^T[o][f = 32w + r] += sum_rows dy[row][o] h2[row][f]   (outputs >= NO of the dy image are zero)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag a = read_tr_frag(DYI + trD + 16 * s * DY_ROW, DY_PLANE, DY_ROW);
        const Frag b = sw_read_tr(H2I, SW_PLANE, trS, s, w);
        gW3 = mfma3(a, b, gW3);
      }
      // dz2^T[f][row] = sum_o W3[f][o] dy[row][o]: one 16-output step
#pragma unroll
      for (int q = 0; q < 16; ++q) dz[q] = 0.0f;
      {
        const Frag b = read_row_frag(DYI, DY_PLANE, r * DY_ROW + 16 * h);
        dz = mfma3(W3d, b, dz);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dz[q] = ((relu2 >> q) & 1u) ? dz[q] * W3_UNSCALE : 0.0f;
        ab2[q] += dz[q];
      }
    } else {
      // every LDS value of this phase is loaded up front and unconditionally: written as a select per element the
      // compiler turned each `relu ? W3s[..] * dy0 : 0` into a branch around one LDS read followed by lgkmcnt(0) -
      // sixteen serial round trips, 3.0 K cycles per tile for this phase (phase stamps) instead of ~1 K
      float4 w3g[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) w3g[g] = *reinterpret_cast<const float4*>(W3s + 32 * w + 4 * h + 8 * g);
      if (tk.agg > 1) {
        float av[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) av[a] = AGG[a * 33 + r];
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < 8; ++a) sum += av[a];  // fixed order: identical in every lane
        dy0 = sum;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 t = w3g[q >> 2];
        const float w3 = (q & 3) == 0 ? t.x : (q & 3) == 1 ? t.y : (q & 3) == 2 ? t.z : t.w;
        const float v = w3 * dy0;
        dz[q] = ((relu2 >> q) & 1u) ? v : 0.0f;
        ab2[q] += dz[q];
        aW3r[q] = fmaf(h2[q], dy0, aW3r[q]);
      }
    }
    {
      half4 ph[4], pl[4];
      sw_write_image(DZ2I, r, w, h, dz, ph, pl);
    }
    }  // CHAIN
    STAMP(8);
    __syncthreads();  // C: dz2 image complete; the h2 image is free (it becomes dz1's)
    STAMP(9);

    // ---------------------------------------------------------------- P4: dh1 = W2 dz2^T -> dz1 ; dW2 ; dW1
    auto do_gw2 = [&]() {
      // gW2[k][n = 32w + r] += sum_rows h1[row][k] dz2[row][n]
      constexpr int D2 = WIDE ? 0 : GW_DEPTH;  // (the wide chain role has no registers left for a deeper prefetch: it spills)
      if constexpr (D2 > 0) {
        // rolling prefetch: the operand of product i + D2 is read while product i runs (pairs of reads followed by pairs
        // of products ran at ~170 cycles per 96-cycle product: the LDS round trip was exposed once per pair)
        const Frag b0 = sw_read_tr(DZ2I, SW_PLANE, trS, 0, w), b1 = sw_read_tr(DZ2I, SW_PLANE, trS, 1, w);
        Frag a[D2 > 0 ? D2 : 1];
#pragma unroll
        for (int i = 0; i < D2; ++i) a[i] = sw_read_tr(H1I, SW_PLANE, trS, i >> 2, i & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const Frag cur = a[i % (D2 > 0 ? D2 : 1)];
          if (i + D2 < 8) a[i % (D2 > 0 ? D2 : 1)] = sw_read_tr(H1I, SW_PLANE, trS, (i + D2) >> 2, (i + D2) & 3);
          gW2[i & 3] = mfma3(cur, (i >> 2) ? b1 : b0, gW2[i & 3]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag b = sw_read_tr(DZ2I, SW_PLANE, trS, s, w);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const Frag a = sw_read_tr(H1I, SW_PLANE, trS, s, t);
            gW2[t] = mfma3(a, b, gW2[t]);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);  // at most two tiles' operand reads in flight
          }
        }
      }
    };
    if constexpr (CHAIN) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
      Frag an;
      if constexpr (!W2_RES4) an = sw_read_row(W2I, W2_PLANE, w2row, 0);
      Frag bn = sw_read_row(DZ2I, SW_PLANE, rowB, 0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const Frag a = W2_RES4 ? w2r[s] : an, b = bn;
        if (s + 1 < 8) {
          if constexpr (!W2_RES4) an = sw_read_row(W2I, W2_PLANE, w2row, s + 1);
          bn = sw_read_row(DZ2I, SW_PLANE, rowB, s + 1);
        }
        acc = mfma3(a, b, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = ((relu1 >> q) & 1u) ? (ACTOR ? acc[q] : acc[q] * WU) : 0.0f;
      half4 ph[4], pl[4];
      sw_write_image(DZ1I, r, w, h, acc, ph, pl);
    }
    if constexpr (!WIDE) {
      // narrow inputs: the next tile's x rows have arrived long ago - split + store them into the other buffer (read
      // from barrier D on)
      if (have_next) stage_commit(buf ^ 1, xr);
    }
    STAMP(10);
    if constexpr (!WIDE) do_gw2();
    STAMP(11);
    __syncthreads();  // D: dz1 image (narrow inputs: and the next x tile) complete
    STAMP(12);
    if constexpr (WIDE && CHAIN) do_gw2();  // beside the loader group's dW1 product
    bool x_lo = true;
    if constexpr (WIDE && LOADER) {
      x_lo = __builtin_amdgcn_readfirstlane((int)xflag[tpar]) != 0 || tk.force_xlo != 0;
      // the next tile's flag: its last readers (the tile before this one) are long past, its writers (the commit between
      // barriers E and F below) come after this
      if (tid == 0) xflag[tpar ^ 1] = 0u;
    }
    if constexpr (LOADER) {
      // gW1[k][n = 32w + r] += sum_rows x[row][k] dz1[row][n]   (row din of gW1 = db1 through the ones column)
      constexpr int NPR = 2 * KT1;                          // products
      constexpr int D1 = WIDE ? MAVA_GW_DEPTH_WIDE : GW_DEPTH;  // operand prefetch depth (see do_gw2)
      if (WIDE && D1 > 0 && !x_lo) {  // x exact in f16: hi plane only, two products per step
        constexpr int DP = (D1 > 0 ? (D1 < NPR ? D1 : NPR) : 1);
        const Frag b0 = sw_read_tr(DZ1I, SW_PLANE, trS, 0, w), b1 = sw_read_tr(DZ1I, SW_PLANE, trS, 1, w);
        half8 a[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) a[i] = read_tr8(XSI + trX + 16 * (i / KT1) * xs_row + 2 * (32 * (i % KT1)), xs_row);
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
          const half8 cur = a[i % DP];
          if (i + DP < NPR) a[i % DP] = read_tr8(XSI + trX + 16 * ((i + DP) / KT1) * xs_row + 2 * (32 * ((i + DP) % KT1)), xs_row);
          const Frag& b = (i / KT1) ? b1 : b0;
          f32x16 c = gW1[i % KT1];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, b.lo, c, 0, 0, 0);
          gW1[i % KT1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, b.hi, c, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (D1 > 0) {
        constexpr int DP = D1 < NPR ? D1 : NPR;
        const Frag b0 = sw_read_tr(DZ1I, SW_PLANE, trS, 0, w), b1 = sw_read_tr(DZ1I, SW_PLANE, trS, 1, w);
        Frag a[DP > 0 ? DP : 1];
#pragma unroll
        for (int i = 0; i < DP; ++i) a[i] = read_tr_frag(XSI + trX + 16 * (i / KT1) * xs_row + 2 * (32 * (i % KT1)), xs_plane, xs_row);
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
          const Frag cur = a[i % (DP > 0 ? DP : 1)];
          if (i + DP < NPR)
            a[i % (DP > 0 ? DP : 1)] = read_tr_frag(XSI + trX + 16 * ((i + DP) / KT1) * xs_row + 2 * (32 * ((i + DP) % KT1)), xs_plane, xs_row);
          gW1[i % KT1] = mfma3(cur, (i / KT1) ? b1 : b0, gW1[i % KT1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag b = sw_read_tr(DZ1I, SW_PLANE, trS, s, w);
#pragma unroll
          for (int t = 0; t < KT1; ++t) {
            const Frag a = read_tr_frag(XSI + trX + 16 * s * xs_row + 2 * (32 * t), xs_plane, xs_row);
            gW1[t] = mfma3(a, b, gW1[t]);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    STAMP(13);
    if constexpr (WIDE) {
      __syncthreads();  // E: every reader of the (single) x tile is done
      if constexpr (LOADER) {
        stage_commit(0, xr, tpar ^ 1);  // (without a next tile: row 0 into a buffer nobody reads again)
      }
      __syncthreads();  // F: next x tile visible
    }
    if constexpr (CHAIN) {
#pragma unroll
      for (int q = 0; q < NP; ++q) { r_act[q] = n_act[q]; r_f0[q] = n_f0[q]; r_f1[q] = n_f1[q]; r_m[q] = n_m[q]; }
    }
    // narrow inputs: no barrier here - the next tile's P1 writes the h1 image, whose last readers (gW2) sit before
    // barrier D, and reads the other x buffer, complete since D; every other image is rewritten only behind barriers A..C
  }
#ifdef MAVA_STAMPS
  if (tk.stamps != nullptr && blockIdx.x == 0 && lane == 0) {
    for (int i = 0; i < 16; ++i) tk.stamps[((ROLE == 2 ? 4 : 0) + w) * 16 + i] = st_acc[i];
  }
#endif

  // ------------------------------------------------------------------ epilogue: one slab per block (x 1/R)
  __syncthreads();
  float* slab = tk.slab + (long)blockIdx.x * tk.slab_stride;
  const int oB2 = mlp_off_b2(din), oB3 = mlp_off_b3(din, no);
  const int Pn = mlp_param_count(din, no);
  if constexpr (LOADER) {
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k = mlp_feat(t, q, h);
        if (k <= din) slab[k * MLP_H + 32 * w + r] = gW1[t][q] * invR;  // row din = db1
      }
  }
  if constexpr (CHAIN) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) slab[oW2 + mlp_feat(t, q, h) * MLP_H + 32 * w + r] = gW2[t][q] * invR;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float v = ab2[q];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
      if (r == 0) slab[oB2 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h] = v * invR;
    }
  }
  float* red = reinterpret_cast<float*>(lds + L.h1);  // epilogue scratch (the tile loop is over)
  if constexpr (!CHAIN) {
    __syncthreads();
  } else if (ACTOR) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (o < no) slab[oW3 + (32 * w + r) * no + o] = gW3[q] * invR;
    }
    float v = ab3;
#pragma unroll
    for (int m = NO; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane < NO) red[w * NO + lane] = v;
    for (int o = 32; o > 0; o >>= 1) {
      loss_a += __shfl_down(loss_a, o, 64);
      loss_b += __shfl_down(loss_b, o, 64);
    }
    if (lane == 0) { red[4 * NO + 2 * w] = loss_a; red[4 * NO + 2 * w + 1] = loss_b; }
    __syncthreads();
    if (tid < no) slab[oB3 + tid] = (((red[tid] + red[NO + tid]) + red[2 * NO + tid]) + red[3 * NO + tid]) * invR;
    if (tid == 0) {
      slab[Pn] = ((red[4 * NO] + red[4 * NO + 2]) + red[4 * NO + 4]) + red[4 * NO + 6];
      slab[Pn + 1] = ((red[4 * NO + 1] + red[4 * NO + 3]) + red[4 * NO + 5]) + red[4 * NO + 7];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float v = aW3r[q];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
      if (r == 0) slab[oW3 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h] = v * invR;
    }
    float v = ab3, l = loss_a;
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_down(v, o, 64);
      l += __shfl_down(l, o, 64);
    }
    if (lane == 0) { red[2 * w] = v; red[2 * w + 1] = l; }
    __syncthreads();
    if (tid == 0) {
      slab[oB3] = (((red[0] + red[2]) + red[4]) + red[6]) * invR;
      slab[Pn] = ((red[1] + red[3]) + red[5]) + red[7];
      slab[Pn + 1] = 0.0f;
    }
  }
}

template <int NO, int S1, bool ACTOR, bool WIDE, int XV>
__global__ __launch_bounds__(WIDE ? 512 : 256, WIDE ? 2 : 1) void ppo_train_h2_kernel(TrainTask tk, H2Layout L,
                                                                                     const uint4* __restrict__ w1p) {
  extern __shared__ __attribute__((aligned(16))) u8 lds[];
  if constexpr (WIDE) {
    if (threadIdx.x < 256) {
      h2_body<NO, S1, ACTOR, WIDE, XV, 1>(tk, L, w1p, lds);
    } else {
      h2_body<NO, S1, ACTOR, WIDE, XV, 2>(tk, L, w1p, lds);
    }
  } else {
    h2_body<NO, S1, ACTOR, WIDE, XV, 0>(tk, L, w1p, lds);
  }
}

// WIDE launches: one pre-split copy of W1 per network kind in the caller's context handle (ctx.h), allocated on first
// use and freed by mava_ctx_destroy (18 steps x 256 lanes x 32 bytes = 147 KB); filled by pack_w1_kernel on the launch
// stream ahead of every launch - launches that share a handle must therefore be on one stream.
template <int NO, int S1, bool ACTOR, bool WIDE, int XV>
int launch_h2(mava_ctx* ctx, const TrainTask& tk, int n_slab, hipStream_t s) {
  const H2Layout L = make_h2_layout<NO, S1, WIDE>(ACTOR);
  MAVA_ARG_CHECK(L.end <= 163840, 8, "ppo_train_h2: %d bytes of LDS exceed the 160 KiB of a CU", L.end);
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)ppo_train_h2_kernel<NO, S1, ACTOR, WIDE, XV>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, L.end));
    attr_set = true;
  }
  const uint4* w1p = nullptr;
  if (WIDE) {
    void*& slot = ctx->w1_split[ACTOR ? 0 : 1];
    if (slot == nullptr) MAVA_HIP_CHECK(hipMalloc(&slot, W1_SPLIT_BYTES));
    uint4* const buf = static_cast<uint4*>(slot);
    int& fresh = ctx->w1_fresh[ACTOR ? 0 : 1];  // set by mava_ppo_finish_f32: its Adam launch has already re-split these weights
    if (!fresh) {
      hipLaunchKernelGGL((pack_w1_kernel<S1>), dim3(1), dim3(256), 0, s, tk.params, tk.din, buf, ACTOR ? 1.0f : W_SCALE_CRITIC);
      MAVA_LAUNCH_CHECK();
    }
    fresh = 0;
    w1p = buf;
  }
  hipLaunchKernelGGL((ppo_train_h2_kernel<NO, S1, ACTOR, WIDE, XV>), dim3(n_slab), dim3(WIDE ? 512 : 256), L.end, s, tk, L,
                     w1p);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, int S1, bool ACTOR, bool WIDE>
int dispatch_xv(mava_ctx* ctx, const TrainTask& tk, int n_slab, hipStream_t s) {
  const uintptr_t a = (uintptr_t)tk.x;
  if constexpr (WIDE) {
    if (tk.din % 4 == 0 && a % 16 == 0) return launch_h2<NO, S1, ACTOR, WIDE, 4>(ctx, tk, n_slab, s);
  } else {
    if (tk.din % 2 == 0 && a % 8 == 0) return launch_h2<NO, S1, ACTOR, WIDE, 2>(ctx, tk, n_slab, s);
  }
  return launch_h2<NO, S1, ACTOR, WIDE, 1>(ctx, tk, n_slab, s);
}

template <int NO, bool ACTOR>
int dispatch_s1(mava_ctx* ctx, const TrainTask& tk, int n_slab, hipStream_t s) {
  const int s1 = (tk.din + 1 + 15) / 16;  // 16-input steps of layer 1, including the ones (bias) column
  switch (s1) {
#ifndef MAVA_FAST_BUILD
    case 1: return dispatch_xv<NO, 1, ACTOR, false>(ctx, tk, n_slab, s);
    case 2: return dispatch_xv<NO, 2, ACTOR, false>(ctx, tk, n_slab, s);
    case 3: return dispatch_xv<NO, 3, ACTOR, false>(ctx, tk, n_slab, s);
    case 4: return dispatch_xv<NO, 4, ACTOR, false>(ctx, tk, n_slab, s);
    case 6: return dispatch_xv<NO, 6, ACTOR, false>(ctx, tk, n_slab, s);
    case 7: case 8: case 9: case 10: case 11: case 12: return dispatch_xv<NO, 12, ACTOR, true>(ctx, tk, n_slab, s);
#endif
    // (the role-split 512-thread form was also measured for this narrow input: actor 0.94 instead of 0.70 ms per launch -
    // the resident W1 fragments and one barrier set per tile beat the second wave group here)
    case 5: return dispatch_xv<NO, 5, ACTOR, false>(ctx, tk, n_slab, s);
    case 13: case 14: case 15: case 16: case 17: case 18: return dispatch_xv<NO, 18, ACTOR, true>(ctx, tk, n_slab, s);
    default: return 1;  // not instantiated (input width > 287): the caller runs the exact-f32 kernel
  }
}

}  // namespace

static int h2_dispatch(mava_ctx* ctx, const TrainTask& tk, int n_slab, bool actor, hipStream_t s) {
  if (actor) {
    if (tk.action_f != nullptr) return 1;  // continuous head: exact-f32 kernel
    if (tk.no <= 8) return dispatch_s1<8, true>(ctx, tk, n_slab, s);
#ifndef MAVA_FAST_BUILD
    if (tk.no <= 16) return dispatch_s1<16, true>(ctx, tk, n_slab, s);
    return dispatch_s1<32, true>(ctx, tk, n_slab, s);
#else
    return 1;
#endif
  }
  return dispatch_s1<1, false>(ctx, tk, n_slab, s);
}

// (ctx is never NULL here: a NULL handle means exact f32 and does not reach this file.)  h2_launches counts the launches
// that really ran on the split-f16 kernel (mava_ctx_get(ctx, MAVA_CTX_H2_LAUNCHES): the parity tests assert it).
int mava_train_h2_launch(mava_ctx* ctx, const TrainTask& tk_in, int n_slab, bool actor, hipStream_t s) {
  TrainTask tk = tk_in;
  tk.force_xlo = (ctx->train_variant & 2) != 0;
  if ((ctx->train_variant & 1) == 0) {  // the eight-wave kernel: the discrete actor and the value network on narrow inputs
    const int rc8 = mava_train_w8_launch(tk, n_slab, actor, s);
    if (rc8 <= 0) {
      if (rc8 == 0) { ++ctx->h2_launches; ++ctx->w8_launches; }
      return rc8;
    }
  }
  const int rc = h2_dispatch(ctx, tk, n_slab, actor, s);
  if (rc == 0) ++ctx->h2_launches;
  return rc;
}
