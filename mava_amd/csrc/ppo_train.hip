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
  unsigned long long* stamps;  // diagnostic builds only (-DMAVA_STAMPS): per-phase cycle sums of block 0
};

#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                   \
  do {                                                             \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                                 \
    st_prev = st_now;                                              \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// LDS carve (floats) computed on the host per launch (din / NO dependent), checked against 160 KiB.
struct TrainLdsLayout {
  int h1t, h2t, dz2t, yp, dy, xs, misc, end;
  int ldx;  // row stride of the staged x tile: 32*KT1 + 4 floats (16-byte rows, an odd number of 16-byte slots =>
            // conflict-free ds_read_b128 row walks in P1 and ds_write_b128 staging)
};

template <int NO>
TrainLdsLayout make_layout(int kt1) {
  TrainLdsLayout L;
  const int tile = MLP_H * LDT;
  L.h1t = MlpLds<NO>::END;
  L.h2t = L.h1t + tile;   // h2^T, later dz1^T (after the P4 sweeps, behind an extra barrier)
  L.dz2t = L.h2t + tile;
  L.yp = L.dz2t + tile;   // [4][NO][32] partial logits
  L.dy = L.yp + 4 * NO * 32;
  L.ldx = 32 * kt1 + 4;
  L.xs = (L.dy + NO * 32 + 3) & ~3;  // [32][ldx] gathered x tile, zero padded to 32*KT1 columns, 16-byte aligned
  L.misc = L.xs + 32 * L.ldx;
  L.end = L.misc + 8;
  return L;
}

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

// flat trajectory row (t*E+e)*A + a of agent-row q of the minibatch
__device__ __forceinline__ long traj_row(const TrainTask& tk, long q, long R) {
  const uint32_t qc = (uint32_t)(q < R ? q : (R - 1));  // R < 2^31 is checked on the host
  const uint32_t b = qc / (uint32_t)tk.A;
  const uint32_t a = qc - b * (uint32_t)tk.A;
  const long p = tk.idx ? (long)tk.idx[b] : tk.idx_base + (long)b;
  return p * tk.A + a;
}

// Register-staged copy of one gathered 32-row x tile: 8 threads per row, thread l8 takes the V-float pieces
// l8, l8+8, ... of its row (V = 4 / 2 / 1 by the alignment of the rows: 128 / 64 / 32 bytes per row per
// instruction).  Immediate offsets only: no per-load address math.  Issued a whole tile ahead of its use
// (global -> registers now, registers -> LDS after the consumers of the previous tile have passed their last
// barrier).  nv = pieces per row (din / V); pieces 8i..8i+7 are all valid when i < nv/8 (uniform scalar
// branch, no exec masking); at most one trailing group is partial.
template <int V> struct StageVec;
template <> struct StageVec<1> { typedef float T; };
template <> struct StageVec<2> { typedef float2 T; };
template <> struct StageVec<4> { typedef float4 T; };

template <int V, int NR>
__device__ __forceinline__ void stage_load(const float* __restrict__ xrow_l8, int nv, int l8, float (&xr)[NR]) {
  typedef typename StageVec<V>::T VT;
  const VT* src = reinterpret_cast<const VT*>(xrow_l8);  // already offset by V*l8 floats
  VT* dst = reinterpret_cast<VT*>(&xr[0]);
  const int nfull = nv >> 3;
#pragma unroll
  for (int i = 0; i < NR / V; ++i) {
    if (i < nfull) dst[i] = src[8 * i];
    else if (i == nfull && l8 + 8 * i < nv) dst[i] = src[8 * i];
  }
}
template <int V, int NR>
__device__ __forceinline__ void stage_write(float* xs_row_l8, int nv, int l8, const float (&xr)[NR]) {
  typedef typename StageVec<V>::T VT;
  VT* dst = reinterpret_cast<VT*>(xs_row_l8);
  const VT* src = reinterpret_cast<const VT*>(&xr[0]);
  const int nfull = nv >> 3;
#pragma unroll
  for (int i = 0; i < NR / V; ++i) {
    if (i < nfull) dst[8 * i] = src[i];
    else if (i == nfull && l8 + 8 * i < nv) dst[8 * i] = src[i];
  }
}

template <int NO, int KT1, bool ACTOR>
__global__ __launch_bounds__(256, 1) void ppo_train_kernel(TrainTask tk, TrainLdsLayout L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const W2s = lds + MlpLds<NO>::W2;
  float* const W3s = lds + MlpLds<NO>::W3;
  float* const H1T = lds + L.h1t;
  float* const H2T = lds + L.h2t;
  float* const DZ1T = H2T;  // alias, see barrier D'
  float* const DZ2T = lds + L.dz2t;
  float* const YP = lds + L.yp;
  float* const DY = lds + L.dy;
  float* const XS = lds + L.xs;
  float* const misc = lds + L.misc;
  constexpr int ldx = 32 * KT1 + 4;  // == L.ldx

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, h = lane >> 5, j = lane & 31;
  const int srow = tid >> 3, l8 = tid & 7;  // staging role: row srow of the tile, 8 threads per row
  const int din = tk.din, no = tk.no;
  const long R = (long)tk.Rb * tk.A;
  const float invR = 1.0f / (float)R;
  constexpr int NR = 4 * KT1;  // staged floats per thread (32*KT1 columns / 8 threads)

  mlp_fill_lds<NO>(lds, tk.params, din, no, 256);
  // Padding columns stay zero for the whole launch except column din, which holds 1.0: "row din" of W1 in the
  // flat parameter vector IS b1, so layer 1 adds its bias and P5 accumulates db1 (row din of gW1) for free.
  for (int i = tid; i < 32 * ldx; i += 256) XS[i] = 0.0f;
  __syncthreads();
  if (tid < 32) XS[tid * ldx + din] = 1.0f;
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
  float ab3 = 0.0f;   // actor: tid < NO: db3[tid]; critic: per-lane partial of db3 (wave 0, half 0)
  // per-lane partial sums over the rows this lane has seen (lane = row j of every tile), reduced across the 32
  // row lanes once in the epilogue: db2 for the wave's 16 (r, h) features and, for the one-output critic, dW3
  float ab2[16], aW3r[ACTOR ? 1 : 16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ab2[r] = 0.0f;
#pragma unroll
  for (int r = 0; r < (ACTOR ? 1 : 16); ++r) aW3r[r] = 0.0f;
  float loss_a = 0.f, loss_b = 0.f;  // wave 0, half 0 lanes: actor (pg, entropy) / critic (value loss)

  const float* const wcol1 = tk.params + 32 * w + j;  // W1[k][32w + j] = wcol1[k*128]
  const int fbase = 32 * w + 4 * h;                   // + (r&3) + 8*(r>>2)

  // per-row inputs of the loss (row j of the tile), prefetched one tile ahead
  // NOTHING in the prefetch path may do arithmetic on a value it has just loaded: the compiler places the
  // vmcnt wait at the first use, and vmcnt retires in order, so one early use exposes the whole HBM gather.
  constexpr int NMB = ACTOR ? (NO + 3) / 4 : 1;  // raw mask bytes, 4 per register
  struct RowIn {
    int act;
    float f0, f1;        // actor: old_logp, advantage ; critic: old_value, target
    uint8_t mraw[4 * NMB];  // raw action-mask bytes (decoded in P3)
  };
  auto load_row = [&](long fr) {
    RowIn ri;
    if (ACTOR) {
      ri.act = tk.action[fr];
      ri.f0 = tk.old_logp[fr];
      ri.f1 = tk.adv[fr];
      const uint8_t* mk = tk.mask ? (tk.mask + fr * no) : nullptr;
#pragma unroll
      for (int o = 0; o < 4 * NMB; ++o) ri.mraw[o] = (mk != nullptr && o < no) ? mk[o] : (uint8_t)1;
    } else {
      ri.act = 0;
      ri.f0 = tk.old_value[fr];
      ri.f1 = tk.targets[fr];
      ri.mraw[0] = 0;
    }
    return ri;
  };
  // minibatch index gather, raw: returns idx[b] (or the identity) without touching it
  auto gather_idx = [&](long q, uint32_t& b_out) -> int32_t {
    const uint32_t qc = (uint32_t)(q < R ? q : (R - 1));
    b_out = qc / (uint32_t)tk.A;
    return tk.idx ? tk.idx[b_out] : (int32_t)(tk.idx_base + (long)b_out);
  };
  auto row_of = [&](long q, int32_t p_raw) -> long {
    const uint32_t qc = (uint32_t)(q < R ? q : (R - 1));
    const uint32_t a = qc % (uint32_t)tk.A;
    return (long)p_raw * tk.A + a;
  };
  const int xv = tk.xv;  // 4 / 2 / 1: widest vector the row alignment allows (uniform)
  auto stage_issue = [&](long fr, float (&xr)[NR]) {
    const uint32_t xrow_idx = (uint32_t)fr / (uint32_t)tk.xshare;  // 32-bit: TE*A < 2^31 is checked on the host
    const float* xrow = tk.x + (long)xrow_idx * din;
    if (xv == 4) stage_load<4, NR>(xrow + 4 * l8, din >> 2, l8, xr);
    else if (xv == 2) stage_load<2, NR>(xrow + 2 * l8, din >> 1, l8, xr);
    else stage_load<1, NR>(xrow + l8, din, l8, xr);
  };
  auto stage_commit = [&](const float (&xr)[NR]) {
    float* row = XS + srow * ldx;
    if (xv == 4) stage_write<4, NR>(row + 4 * l8, din >> 2, l8, xr);
    else if (xv == 2) stage_write<2, NR>(row + 2 * l8, din >> 1, l8, xr);
    else stage_write<1, NR>(row + l8, din, l8, xr);
  };

  // W1 operand ring depth (batches of 8 k-steps); must divide the batch count 2*KT1
  constexpr int RD = ((2 * KT1) % 3 == 0) ? 3 : (((2 * KT1) % 4 == 0) ? 4 : 2);
  float wr[RD][8];
  // k order inside a batch of 16 inputs: MFMA step s multiplies k = 16b + s (lane half 0) and k = 16b + 8 + s
  // (half 1), so a lane's eight x operands of a batch are 32 contiguous bytes of its row (two ds_read_b128)
  const float* const wcol1h = wcol1 + 8 * h * MLP_H;  // W1[k + 8h][32w + j] = wcol1h[k * 128]
#pragma unroll
  for (int d = 0; d < RD; ++d)
#pragma unroll
    for (int s = 0; s < 8; ++s) wr[d][s] = wcol1h[(16 * d + s) * MLP_H];

  const long ntiles = (R + 31) / 32;
  long it = blockIdx.x;
  float xr[NR];
  RowIn rin = {};
  // trajectory rows of the NEXT tile for both thread roles (staging row srow / loss row j): the index
  // gather runs one tile ahead of the loads that depend on it, two tiles ahead of the compute
  int32_t ps_next = 0, pj_next = 0;  // RAW idx values of the next tile (staging row srow / loss row j)
  if (it < ntiles) {
    stage_issue(traj_row(tk, it * 32 + srow, R), xr);
    rin = load_row(traj_row(tk, it * 32 + j, R));
    stage_commit(xr);
    const long itn0 = it + gridDim.x;
    if (itn0 < ntiles) {
      uint32_t bb;
      ps_next = gather_idx(itn0 * 32 + srow, bb);
      pj_next = gather_idx(itn0 * 32 + j, bb);
    }
  }
  __syncthreads();

  STAMP_DECL
  for (; it < ntiles; it += gridDim.x) {
    STAMP(9);
    const bool valid = (it * 32 + j) < R;
    const long itn = it + gridDim.x;
    const bool have_next = itn < ntiles;
    RowIn rnext = rin;

    // ---------------------------------------------------------------- P1: layer 1, tile w
    f32x16 h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = 0.0f;  // b1 enters through the ones column of the x tile
    {
      // W1 operands stream from L2 through a shifting ring of RD batches of 8 k-steps (prefetch distance
      // RD*512 MFMA cycles).  The refill index wraps, so when the loop ends the ring already holds
      // batches 0..RD-1 for the next row tile.  x operands come from the staged LDS tile one batch ahead.
      const float* xb = XS + j * ldx + 8 * h;  // x[row j][k + 8h]
      constexpr int NB = 2 * KT1;          // batches of 8 k-steps (16 inputs)
      static_assert(NB % RD == 0, "ring depth must divide the batch count");
      // Ring slot d holds batch g*RD + d (W1 operands from L2) and xo[d] its x operands (LDS).  Slots are
      // static inside a group of RD batches - no register copies next to the MFMAs - and groups form a
      // runtime loop.  A consumed slot is refilled with batch b+RD, wrapping to the head batches of the NEXT
      // row tile, so the ring is already primed when the next tile starts.
      // Rows k >= din of "W1" are the bias / W2 words that follow it in the flat parameter vector: finite
      // values that meet the zero padding of the x tile, so no clamp and no per-load address math.
      float xo[RD][8];
      {
        const float4 v0 = *reinterpret_cast<const float4*>(xb), v1 = *reinterpret_cast<const float4*>(xb + 4);
        xo[0][0] = v0.x; xo[0][1] = v0.y; xo[0][2] = v0.z; xo[0][3] = v0.w;
        xo[0][4] = v1.x; xo[0][5] = v1.y; xo[0][6] = v1.z; xo[0][7] = v1.w;
      }
#pragma unroll 1
      for (int g = 0; g < NB / RD; ++g) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
          const int b = g * RD + d;
          const int bn = (b + 1 < NB) ? (b + 1) : b;
          {
            const float4 v0 = *reinterpret_cast<const float4*>(xb + 16 * bn);
            const float4 v1 = *reinterpret_cast<const float4*>(xb + 16 * bn + 4);
            float(&xn)[8] = xo[(d + 1) % RD];
            xn[0] = v0.x; xn[1] = v0.y; xn[2] = v0.z; xn[3] = v0.w;
            xn[4] = v1.x; xn[5] = v1.y; xn[6] = v1.z; xn[7] = v1.w;
          }
#pragma unroll
          for (int s = 0; s < 8; ++s) h1 = MFMA32(wr[d][s], xo[d][s], h1);
          // refill the slot consumed ONE batch ago (its MFMAs have retired: no write-after-read wait on
          // operands still being read) with batch b-1+RD
          int bf = b - 1 + RD;
          bf = (bf >= NB) ? (bf - NB) : bf;
          const float* wb = wcol1h + bf * (16 * MLP_H);
#pragma unroll
          for (int s = 0; s < 8; ++s) wr[(d + RD - 1) % RD][s] = wb[s * MLP_H];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // trailing refill of the last slot (batch NB-1+RD wraps to head batch RD-1 of the next tile)
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[RD - 1][s] = wcol1h[(16 * (RD - 1) + s) * MLP_H];
    }
    STAMP(12);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h1[r] = fmaxf(h1[r], 0.0f);
      H1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = h1[r];
    }
    STAMP(0);
    __syncthreads();  // A
    // Next tile's gathers are issued only now: vmcnt retires in order, so issuing them before P1 would
    // put the HBM gather latency in front of every W1 operand wait of P1.
    if (have_next) {
      // idx values were gathered one iteration ago (complete by now); only now are they used
      stage_issue(row_of(itn * 32 + srow, ps_next), xr);  // global -> registers, committed after barrier E
      rnext = load_row(row_of(itn * 32 + j, pj_next));
      const long itnn = itn + gridDim.x;
      if (itnn < ntiles) {  // raw index gather for the tile after next (first used next iteration)
        uint32_t bb;
        ps_next = gather_idx(itnn * 32 + srow, bb);
        pj_next = gather_idx(itnn * 32 + j, bb);
      }
    }
    STAMP(10);

    // ---------------------------------------------------------------- P2: layer 2, tile w
    f32x16 h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = lds[MlpLds<NO>::B2 + fbase + (r & 3) + 8 * (r >> 2)];
    {
      const float* wl = W2s + h * MLP_LDW + 32 * w + j;  // W2[k + h][32w + j]
      const float* hb = H1T + h * LDT + j;               // h1^T[k + h][row j]
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
    STAMP(11);
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
    STAMP(1);
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
        uint32_t mbits = 0;
#pragma unroll
        for (int o = 0; o < NO; ++o)
          if (rin.mraw[o]) mbits |= (1u << o);
        Categorical<NO> cat;
        cat.build_bits(y, mbits, no);
        const int act = rin.act;
        float lp = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o)
          if (o == act) lp = cat.logp[o];
        const float gae = (rin.f1 - adv_mean) * adv_rstd;
        const float ratio = expf(lp - rin.f0);
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
        const float ov = rin.f0, tg = rin.f1;
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
      ab2[r] += dz[r];
      if (!ACTOR) aW3r[r] = fmaf(h2[r], dy[0], aW3r[r]);
    }
    if (!ACTOR && w == 0 && h == 0) ab3 += dy[0];
    STAMP(2);
    __syncthreads();  // C

    // ---------------------------------------------------------------- P4: dh1 tile w, small grads
    f32x16 d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
    {
      const float* wl = W2s + (32 * w + j) * MLP_LDW + h;  // W2[32w + j][n + h]
      const float* db = DZ2T + h * LDT + j;                // dz2^T[n + h][row j]
      float oa[2][8], ob[2][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { oa[0][s] = wl[2 * s]; ob[0][s] = db[(2 * s) * LDT]; }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            oa[(g + 1) & 1][s] = wl[16 * (g + 1) + 2 * s];
            ob[(g + 1) & 1][s] = db[(16 * (g + 1) + 2 * s) * LDT];
          }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) d1 = MFMA32(oa[g & 1][s], ob[g & 1][s], d1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ACTOR) {
      // dW3[f][o] += sum_rows h2[f][row] * dy[o][row];  db3
      // all loads of an 8-row chunk are issued before their first use (one LDS latency per chunk)
      const float* hrow = H2T + sf * LDT;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float hv[8], dv[NOH][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) hv[q] = hrow[8 * c + q];
#pragma unroll
        for (int i = 0; i < NOH; ++i) {
          const int o = 2 * i + og;
#pragma unroll
          for (int q = 0; q < 8; ++q) dv[i][q] = (o < NO) ? DY[o * 32 + 8 * c + q] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NOH; ++i)
#pragma unroll
          for (int q = 0; q < 8; ++q) aW3[i] = fmaf(hv[q], dv[i][q], aW3[i]);
      }
      if (tid < NO) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v[16];
#pragma unroll
          for (int row = 0; row < 16; ++row) v[row] = DY[tid * 32 + 16 * c + row];
#pragma unroll
          for (int row = 0; row < 16; ++row) s += v[row];
        }
        ab3 += s;
      }
    }
    STAMP(3);
    __syncthreads();  // D': every reader of h2^T is done; the tile is reused for dz1^T
    {
      // relu'(z1) from the h1^T tile still in LDS (keeps h1 out of the register file across P2-P4)
      float hm[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hm[r] = H1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d1[r] = (hm[r] > 0.0f) ? d1[r] : 0.0f;
        DZ1T[(fbase + (r & 3) + 8 * (r >> 2)) * LDT + j] = d1[r];
      }
    }
    STAMP(4);
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
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(5);
    {
      // gW1[k_in tile t][n] += sum_rows x[row][k_in] * dz1^T[n][row];  A from the staged x tile
      const float* eb = DZ1T + (32 * w + j) * LDT + h;
      const float* xa = XS + h * ldx + j;  // x[row 2s + h][32t + j]
      float bz[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) bz[s] = eb[2 * s];
#pragma unroll
      for (int t = 0; t < KT1; ++t) {
#pragma unroll
        for (int s = 0; s < 16; ++s) gW1[t] = MFMA32(xa[(2 * s) * ldx + 32 * t], bz[s], gW1[t]);
        __builtin_amdgcn_sched_barrier(0);  // keep the next tile's LDS reads from piling up in VGPRs
      }
    }
    STAMP(6);
    __syncthreads();  // E: every exchange tile and the x tile are free
    STAMP(7);
    if (have_next) stage_commit(xr);
    rin = rnext;
    __syncthreads();  // F: next x tile visible
    STAMP(8);
  }
#ifdef MAVA_STAMPS
  if (tk.stamps != nullptr && blockIdx.x == 0 && lane == 0) {
    for (int i = 0; i < 16; ++i) tk.stamps[w * 16 + i] = st_acc[i];
  }
#endif

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
      if (k <= din) slab[k * MLP_H + 32 * w + j] = gW1[t][r];  // row din = db1 (oB1 == din * 128)
    }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[oW2 + mlp_feat(t, r, h) * MLP_H + 32 * w + j] = gW2[t][r];
  // per-lane partials -> sums over the 32 row lanes of each half (fixed xor tree: reproducible)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = ab2[r];
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if (j == 0) slab[oB2 + fbase + (r & 3) + 8 * (r >> 2)] = v;
  }
  if (ACTOR) {
#pragma unroll
    for (int i = 0; i < NOH; ++i) {
      const int o = 2 * i + og;
      if (o < no) slab[oW3 + sf * no + o] = aW3[i];
    }
    if (tid < no) slab[oB3 + tid] = ab3;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = aW3r[r];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
      if (j == 0) slab[oW3 + fbase + (r & 3) + 8 * (r >> 2)] = v;
    }
    float v = ab3;
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if (tid == 0) slab[oB3] = v;
  }
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
  const TrainLdsLayout L = make_layout<NO>(KT1);
  const size_t lb = (size_t)L.end * sizeof(float);
  MAVA_ARG_CHECK(lb <= 163840, 8,
                 "ppo_train: %zu bytes of LDS needed (n_out pad %d, input width %d) exceed the 160 KiB of a CU",
                 lb, NO, tk.din);
  MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)ppo_train_kernel<NO, KT1, ACTOR>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
  hipLaunchKernelGGL((ppo_train_kernel<NO, KT1, ACTOR>), dim3(n_slab), dim3(256), lb, s, tk, L);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

template <int NO, bool ACTOR>
int dispatch_kt(const TrainTask& tk, int n_slab, hipStream_t s) {
  const int kt = tk.din / 32 + 1;  // 32*kt > din: the x tile always has a spare column for the ones (bias) input
#ifdef MAVA_FAST_BUILD  // developer iteration: only the BASELINE config-2 instantiations
  if (kt == 3) return launch_train<NO, 3, ACTOR>(tk, n_slab, s);
  if (kt == 9) return launch_train<NO, 9, ACTOR>(tk, n_slab, s);
  mava_set_error("fast build: input width %d not instantiated", tk.din);
  return MAVA_EARG(9);
#else
  switch (kt) {
    case 1: return launch_train<NO, 1, ACTOR>(tk, n_slab, s);
    case 2: return launch_train<NO, 2, ACTOR>(tk, n_slab, s);
    case 3: return launch_train<NO, 3, ACTOR>(tk, n_slab, s);
    case 4: return launch_train<NO, 4, ACTOR>(tk, n_slab, s);
    case 5: case 6: return launch_train<NO, 6, ACTOR>(tk, n_slab, s);
    case 7: case 8: case 9: return launch_train<NO, 9, ACTOR>(tk, n_slab, s);
    default:
      mava_set_error("ppo_train: input width %d > 287 is not instantiated", tk.din);
      return MAVA_EARG(9);
  }
#endif
}

}  // namespace

static unsigned long long* g_stamps = nullptr;
// Diagnostic hook (not part of include/mava_hip.h): device buffer of 40 u64 receiving block 0's per-phase
// cycle sums when the library is built with -DMAVA_STAMPS; ignored otherwise.
extern "C" int mava_debug_set_stamps(unsigned long long* p) {
  g_stamps = p;
  return MAVA_OK;
}

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
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024 && (long)Rb * A < (1L << 31), 1,
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
  tk.stamps = g_stamps;
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
  MAVA_ARG_CHECK(Rb >= 1 && A >= 1 && n_slab >= 1 && n_slab <= 1024 && (long)Rb * A < (1L << 31), 1,
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
  tk.stamps = g_stamps;
  return dispatch_kt<1, false>(tk, n_slab, s);
}
