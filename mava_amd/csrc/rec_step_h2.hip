// Fused recurrent ACTING step on split-f16 ("f16x2") operands with PRE-PACKED weights
// (rec_step.hip holds the exact-f32 form, the reference citations and the argument checks shared by both).
//
// What bounded the f32 acting step was not arithmetic but weight traffic: every 32-row tile re-streams the ~0.5 MB of
// one network's weights from L2 (576 tiles per env step at the rec_mappo config-4 shape = 311 MB per step) behind
// eight 4-byte loads per operand.  Here
//  * the weights of both networks are split ONCE per rollout (mava_rec_step_pack_f32: hi + lo f16 with error diffusion
//    along the summation index) and stored in MFMA-FRAGMENT order: an operand fragment of a wave is 1 KB contiguous per
//    plane, two 16-byte loads per lane;
//  * a block multiplies every fragment it loads with RT row tiles (32 RT rows), so the weight traffic per row falls by
//    RT and a launch of <= 256 groups covers the chip in one round;
//  * a product is three v_mfma_f32_32x32x16_f16 (96 instead of 512 matrix-pipe cycles per 16 inputs);
//  * activations cross the waves as LDS images [row][feature] of f16 hi / lo planes, split once by the producing lane.
#include "h2_core.h"
#include "rec_step_task.h"
#include "tanh_normal.h"

namespace {

using h2::Frag;
using h2::half4;
using h2::half8;
using h2::u8;

#ifdef MAVA_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                    \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                                  \
    st_prev = st_now;                                               \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
__device__ unsigned long long* g_step_stamps = nullptr;
#else
#define STAMP_DECL
#define STAMP(i)
#endif

constexpr int G3 = 3 * MLP_H;
constexpr int IROW = h2::IMG_ROW;     // 272: [row][128 f16 + 16]
constexpr int IPLANE = 32 * IROW;
constexpr int IIMG = 2 * IPLANE;      // 17408

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// fragment f of a packed matrix: [f][plane hi, lo][lane][16 bytes]
__device__ __forceinline__ Frag load_frag(const u8* base, int f, int lane) {
  const uint4* p = reinterpret_cast<const uint4*>(base + (long)f * 2048) + lane;
  Frag r;
  r.hi = __builtin_bit_cast(half8, p[0]);
  r.lo = __builtin_bit_cast(half8, p[64]);
  return r;
}

template <int ROWB, int PLANE>
__device__ __forceinline__ void put16(u8* img, int r, int col0, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    half4 ph, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a, b;
      h2::split1(v[4 * g + e], a, b);
      ph[e] = a;
      pl[e] = b;
    }
    u8* p = img + r * ROWB + 2 * (col0 + 8 * g);
    *reinterpret_cast<half4*>(p) = ph;
    *reinterpret_cast<half4*>(p + PLANE) = pl;
  }
}

// One block per (matrix, 32-column tile): lane (j, h) walks the K batches of its column with one carry chain.
__global__ __launch_bounds__(64) void rec_pack_kernel(const float* __restrict__ params, int din, int nb1, u8* __restrict__ out) {
  const int lane = threadIdx.x, h = lane >> 5, j = lane & 31;
  int t = blockIdx.x;  // 0..3 Wpre, 4..15 Wi, 16..27 Wh, 28..31 Wpost
  const float* W;
  int K, N, nb, frag0, nt;
  const float* const Wpre = params;
  const float* const Wi = Wpre + (long)din * MLP_H + MLP_H;
  const float* const Wh = Wi + MLP_H * G3 + G3;
  const float* const Wpost = Wh + MLP_H * G3 + MLP_H;
  if (t < 4) { W = Wpre; K = din; N = MLP_H; nb = nb1; nt = t; frag0 = 0; }
  else if (t < 16) { W = Wi; K = MLP_H; N = G3; nb = 8; nt = t - 4; frag0 = 4 * nb1; }
  else if (t < 28) { W = Wh; K = MLP_H; N = G3; nb = 8; nt = t - 16; frag0 = 4 * nb1 + 96; }
  else { W = Wpost; K = MLP_H; N = MLP_H; nb = 8; nt = t - 28; frag0 = 4 * nb1 + 192; }
  const int col = 32 * nt + j;
  float carry = 0.0f;
  for (int b = 0; b < nb; ++b) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * b + 8 * h + e;
      v[e] = (k < K) ? W[(long)k * N + col] : 0.0f;
    }
    const Frag f = h2::split8_carry(v, carry);
    uint4* p = reinterpret_cast<uint4*>(out + (long)(frag0 + nt * nb + b) * 2048) + lane;
    p[0] = __builtin_bit_cast(uint4, f.hi);
    p[64] = __builtin_bit_cast(uint4, f.lo);
  }
}

#ifdef MAVA_STAMPS
#define STAMP_FLUSH()                                                          \
  do {                                                                         \
    STAMP(10);                                                                 \
    if (g_step_stamps != nullptr && grp == 0 && ACTOR && threadIdx.x == 0)     \
      for (int i = 0; i < 12; ++i) g_step_stamps[i] = st_acc[i];               \
  } while (0)
#else
#define STAMP_FLUSH()
#endif

struct NetH2 {
  RecNet n;
  const u8* pack;
  int nb1, xrow, abytes;  // X image row bytes (32 nb1 + 16), bytes of region A per row tile
};

template <int NO, bool ACTOR, int RT>
__device__ __forceinline__ void rec_step_h2_body(const NetH2& nh, const RecStepOut& out, u8* lds, int grp) {
  const RecNet& nt = nh.n;
  const int tid = threadIdx.x;
  const int lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int din = nt.din, no = nt.no, nb1 = nh.nb1, XROW = nh.xrow, XPLANE = 32 * nh.xrow;
  u8* const RA = lds;                                   // X images, later the E images (RT x abytes)
  u8* const RB = lds + RT * nh.abytes;                  // masked-h images, later the new-h images (RT x IIMG)
  float* const YP = reinterpret_cast<float*>(RB + RT * IIMG);  // [RT][4][NO][32]
  float* const BS = YP + RT * 4 * NO * 32;              // biases: bpre | bi (384) | bhn | bpost | bhead (NO)
  const float* const bpre = nt.params + (long)din * MLP_H;
  const float* const bi = bpre + MLP_H + MLP_H * G3;
  const float* const bhn = bi + G3 + MLP_H * G3;
  const float* const bpost = bhn + MLP_H + MLP_H * MLP_H;
  const float* const Whead = bpost + MLP_H;
  const float* const bhead = Whead + MLP_H * no;
  const u8* const Ppre = nh.pack;
  const u8* const Pi = Ppre + (long)4 * nb1 * 2048;
  const u8* const Ph = Pi + 96 * 2048;
  const u8* const Ppost = Ph + 96 * 2048;
  const int fb = 32 * w + 4 * h;
  const int ntiles = nt.rows / 32;
  const int it0 = grp * RT;

  // Weight fragments come from L2 (~1 us away under load) and a batch of a 128-wide layer is only 9 RT matrix
  // instructions: every layer's fragments run through a ring whose first loads are issued a phase early.
  STAMP_DECL
  constexpr int DP = 6;  // ring depth of the 128-wide layers (pre_torso, post_torso)
  Frag rp[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) rp[d] = load_frag(Ppre, w * nb1 + (d < nb1 ? d : nb1 - 1), lane);
  // ---- every global load of the staging phase is issued first (head fragments, biases, the x rows, the hidden state and its
  // reset flags), then consumed: one memory round trip for the phase instead of one per consumer
  // head fragments (A operand of logits^T[o][row] over this wave's 32 features; the B operand is the post_torso
  // accumulator itself): element e = Whead[32w + 16s + 8(e>>2) + 4h + (e&3)][o = j], scaled into f16's normal range
  float w3v[2][8];
#pragma unroll
  for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 32 * w + 16 * sgm + 8 * (e >> 2) + 4 * h + (e & 3);
      w3v[sgm][e] = Whead[f * no + (j < no ? j : 0)];
    }
  float bv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    const float* src = i < MLP_H ? bpre + i : (i < 4 * MLP_H ? bi + (i - MLP_H) : (i < 5 * MLP_H ? bhn + (i - 4 * MLP_H) : bpost + (i - 5 * MLP_H)));
    bv[k] = *src;
  }
  const float bh = bhead[tid < no ? tid : 0];
  // x tiles: wave w takes rows 8w .. 8w + 7 of every tile, a lane one column per 64-column chunk
  const bool fast_x = nb1 <= 12;
  float xv[RT][8][3];
  if (fast_x) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int row = (it0 + rt) * 32 + 8 * w + q;
        row = row < nt.rows ? row : (nt.rows - 1);
        const float* xrow = nt.x + (long)(row / nt.xshare) * din;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int c = lane + 64 * cc;
          xv[rt][q][cc] = xrow[c < din ? c : 0];
        }
      }
  }
  // masked hidden state entering the step.  The f32 values are NOT kept across the GRU product (48 registers that were
  // spilled and came back one scratch round trip at a time): they are read again before the gate arithmetic.
  uint32_t rs_bits = 0;
  auto load_hp = [&](int rt, float (&v)[16]) {
    const int it = (it0 + rt < ntiles) ? (it0 + rt) : (ntiles - 1);
    const float* hin = nt.h_in + ((long)it * MLP_H + fb) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = hin[((r & 3) + 8 * (r >> 2)) * 32];
  };
  float hv[RT][16];
  uint8_t dn[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int it = (it0 + rt < ntiles) ? (it0 + rt) : (ntiles - 1);
    dn[rt] = nt.done[(long)(it * 32 + j) * nt.done_stride];  // networks.py:253-257
    load_hp(rt, hv[rt]);
  }

  // ---- consume
  Frag W3h[2];
#pragma unroll
  for (int sgm = 0; sgm < 2; ++sgm) {
#pragma unroll
    for (int e = 0; e < 8; ++e) w3v[sgm][e] = (j < no) ? w3v[sgm][e] * h2::W3_SCALE : 0.0f;
    W3h[sgm] = h2::split8(w3v[sgm]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) BS[tid + 256 * k] = bv[k];
  if (tid < NO) BS[6 * MLP_H + tid] = tid < no ? bh : 0.0f;
  if (fast_x) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        u8* const dst = RA + rt * nh.abytes + (8 * w + q) * XROW;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int c = lane + 64 * cc;
          if (c < 16 * nb1) {
            _Float16 a, b;
            h2::split1(c < din ? xv[rt][q][cc] : 0.0f, a, b);
            *reinterpret_cast<_Float16*>(dst + 2 * c) = a;
            *reinterpret_cast<_Float16*>(dst + XPLANE + 2 * c) = b;
          }
        }
      }
  } else {
    for (int rr = w; rr < 32 * RT; rr += 4) {
      int row = it0 * 32 + rr;
      row = row < nt.rows ? row : (nt.rows - 1);
      const float* xrow = nt.x + (long)(row / nt.xshare) * din;
      u8* const dst = RA + (rr >> 5) * nh.abytes + (rr & 31) * XROW;
      for (int c = lane; c < 16 * nb1; c += 64) {
        const float v = c < din ? xrow[c] : 0.0f;
        _Float16 a, b;
        h2::split1(v, a, b);
        *reinterpret_cast<_Float16*>(dst + 2 * c) = a;
        *reinterpret_cast<_Float16*>(dst + XPLANE + 2 * c) = b;
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const bool rs = dn[rt] != 0;
    rs_bits |= rs ? (1u << rt) : 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) hv[rt][r] = rs ? 0.0f : hv[rt][r];
    put16<IROW, IPLANE>(RB + rt * IIMG, j, fb, hv[rt]);
  }
  STAMP(0);
  __syncthreads();
  STAMP(1);

  // the GRU's first two batches of fragments (gate column tiles of this wave: r -> tile w, z -> 4 + w, n -> 8 + w)
  constexpr int DG = 2;
  Frag wi[DG][3], wh[DG][3];
#pragma unroll
  for (int d = 0; d < DG; ++d)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      wi[d][g] = load_frag(Pi, (4 * g + w) * 8 + d, lane);
      wh[d][g] = load_frag(Ph, (4 * g + w) * 8 + d, lane);
    }
  // ---- pre_torso: e = relu(x Wpre + bpre)
  {
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = BS[fb + (r & 3) + 8 * (r >> 2)];
#pragma unroll 1
    for (int b0 = 0; b0 < nb1; b0 += DP) {
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const int b = b0 + d;
        if (b < nb1) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const Frag xf = h2::read_row_frag(RA + rt * nh.abytes, XPLANE, j * XROW + 32 * b + 16 * h);
            acc[rt] = h2::mfma3(rp[d], xf, acc[rt]);
          }
        }
        const int bn = b + DP;
        rp[d] = load_frag(Ppre, w * nb1 + (bn < nb1 ? bn : nb1 - 1), lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(2);
    __syncthreads();  // every wave has read the x images: region A becomes the E images
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) e[r] = fmaxf(acc[rt][r], 0.0f);
      put16<IROW, IPLANE>(RA + rt * nh.abytes, j, fb, e);
    }
  }
  STAMP(3);
  __syncthreads();
  STAMP(4);

  // ---- GRU cell (flax GRUCell): r = s(W_ir e + b_ir + W_hr h), z likewise, n = tanh(W_in e + b_in + r (W_hn h + b_hn))
  float hn[RT][16];
  {
    f32x16 ga[RT][4];  // {W_in e + b_in, r, z, W_hn h + b_hn}
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fb + (r & 3) + 8 * (r >> 2);
        ga[rt][0][r] = BS[MLP_H + 2 * MLP_H + f];
        ga[rt][1][r] = BS[MLP_H + f];
        ga[rt][2][r] = BS[MLP_H + MLP_H + f];
        ga[rt][3][r] = BS[4 * MLP_H + f];
      }
#pragma unroll 1
    for (int b0 = 0; b0 < 8; b0 += DG) {
#pragma unroll
      for (int d = 0; d < DG; ++d) {
        const int b = b0 + d;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int ro = j * IROW + 32 * b + 16 * h;
          const Frag ef = h2::read_row_frag(RA + rt * nh.abytes, IPLANE, ro);
          const Frag hf = h2::read_row_frag(RB + rt * IIMG, IPLANE, ro);
          ga[rt][1] = h2::mfma3(wi[d][0], ef, ga[rt][1]);
          ga[rt][2] = h2::mfma3(wi[d][1], ef, ga[rt][2]);
          ga[rt][0] = h2::mfma3(wi[d][2], ef, ga[rt][0]);
          ga[rt][1] = h2::mfma3(wh[d][0], hf, ga[rt][1]);
          ga[rt][2] = h2::mfma3(wh[d][1], hf, ga[rt][2]);
          ga[rt][3] = h2::mfma3(wh[d][2], hf, ga[rt][3]);
        }
        const int bn = (b + DG < 8) ? b + DG : 7;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          wi[d][g] = load_frag(Pi, (4 * g + w) * 8 + bn, lane);
          wh[d][g] = load_frag(Ph, (4 * g + w) * 8 + bn, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // post_torso's fragments travel during the gate arithmetic
#pragma unroll
    for (int d = 0; d < DP; ++d) rp[d] = load_frag(Ppost, w * 8 + d, lane);
    float hp[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) load_hp(rt, hp[rt]);
    STAMP(5);
    __syncthreads();  // every wave has read the masked-h images: region B becomes the new-h images
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rs = (rs_bits >> rt) & 1u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rr = sigm(ga[rt][1][r]);
        const float zz = sigm(ga[rt][2][r]);
        const float nn = tanh_(ga[rt][0][r] + rr * ga[rt][3][r]);
        hn[rt][r] = (1.0f - zz) * nn + zz * (rs ? 0.0f : hp[rt][r]);
      }
      if (it0 + rt < ntiles) {  // (one uniform branch per tile, straight-line stores)
        float* hout = nt.h_out + ((long)(it0 + rt) * MLP_H + fb) * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) hout[((r & 3) + 8 * (r >> 2)) * 32] = hn[rt][r];
      }
      put16<IROW, IPLANE>(RB + rt * IIMG, j, fb, hn[rt]);
    }
  }
  STAMP(6);
  __syncthreads();
  STAMP(7);

  // ---- post_torso + partial head
  {
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = BS[5 * MLP_H + fb + (r & 3) + 8 * (r >> 2)];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const Frag hf = h2::read_row_frag(RB + rt * IIMG, IPLANE, j * IROW + 32 * b + 16 * h);
        acc[rt] = h2::mfma3(rp[b % DP], hf, acc[rt]);
      }
      if (b + DP < 8) rp[b % DP] = load_frag(Ppost, w * 8 + b + DP, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // relu, split, and the packed groups ARE the operand of the head product (groups (2s, 2s+1) = k-step s)
      half4 ph[4], pl[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 a, b;
          h2::split1(fmaxf(acc[rt][4 * g + e], 0.0f), a, b);
          ph[g][e] = a;
          pl[g][e] = b;
        }
      f32x16 yacc;
#pragma unroll
      for (int q = 0; q < 16; ++q) yacc[q] = 0.0f;
#pragma unroll
      for (int sgm = 0; sgm < 2; ++sgm) {
        Frag b;
        b.hi = __builtin_shufflevector(ph[2 * sgm], ph[2 * sgm + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        b.lo = __builtin_shufflevector(pl[2 * sgm], pl[2 * sgm + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        yacc = h2::mfma3(W3h[sgm], b, yacc);
      }
      // partial logits of this wave: register q of lane (row j, half h) is output (q&3) + 8(q>>2) + 4h
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = (q & 3) + 8 * (q >> 2) + 4 * h;
        if ((q & 3) + 8 * (q >> 2) < NO) {
          if (o < NO) YP[((rt * 4 + w) * NO + o) * 32 + j] = yacc[q] * h2::W3_UNSCALE;
        }
      }
    }
  }
  STAMP(8);
  __syncthreads();
  STAMP(9);

  // ---- epilogue: eight lanes per row (all 32 rows of a tile at once), lane l8 owns outputs l8, l8 + 8, ...
  constexpr int OPL = NO / 8;
  const int l8 = lane & 7, erow = tid >> 3;
  auto add_op = [](float a, float b) { return a + b; };
  auto max_op = [](float a, float b) { return fmaxf(a, b); };
  auto min_op = [](float a, float b) { return fminf(a, b); };
  const float* const B3 = BS + 6 * MLP_H;
#pragma unroll 1
  for (int rt = 0; rt < RT; ++rt) {
    if (it0 + rt >= ntiles) break;
    const int row = (it0 + rt) * 32 + erow;
    float y[OPL];
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
      const int o = l8 + 8 * k;
      const float* yp = YP + ((rt * 4) * NO + o) * 32 + erow;
      y[k] = (((yp[0] + yp[NO * 32]) + yp[2 * NO * 32]) + yp[3 * NO * 32]) + B3[o];
    }
    if (!ACTOR) {
      if (l8 == 0)
        for (int b = 0; b < out.vbroadcast; ++b) out.value[(long)row * out.vbroadcast + b] = y[0];
    } else if (out.action_f != nullptr) {
      // ContinuousActionHead (networks.py:127-169): same noise stream as mava_seq_sample_continuous_f32
      const float* const log_std = bhead + no;
      const uint32_t gid = out.row_offset + (uint32_t)row;
      float lp = 0.0f;
#pragma unroll
      for (int k = 0; k < OPL; ++k) {
        const int o = l8 + 8 * k;
        if (o < no) {
          const float sc = tn::scale_of(log_std[o], out.min_scale);
          const float eps = out.greedy ? 0.0f : tn::noise(gid, out.step, o, tn::STREAM_SAMPLE, out.seed_lo, out.seed_hi);
          const float a = tanhf(fmaf(sc, eps, y[k]));
          lp += tn::log_prob(a, y[k], sc).lp;
          out.action_f[(long)row * no + o] = a;
        }
      }
      lp = h2::group_allreduce<8>(lp, add_op);
      if (l8 == 0) out.log_prob[row] = lp;
    } else {
      // masked Categorical (networks.py:116-124, distributions.py:146-165) + Gumbel-max: argmax_o z[o] - log(-log(u_o)),
      // first index wins ties (same stream as mava_seq_sample_f32)
      float z[OPL], sc[OPL];
      float mx = -FLT_MAX;
      const uint32_t gid = out.row_offset + (uint32_t)row;
#pragma unroll
      for (int k = 0; k < OPL; ++k) {
        const int o = l8 + 8 * k;
        const bool legal = (o < no) && (out.mask == nullptr || out.mask[(long)row * no + o] != 0);
        z[k] = legal ? y[k] : -FLT_MAX;
        mx = fmaxf(mx, z[k]);
      }
      mx = h2::group_allreduce<8>(mx, max_op);
      float se = 0.0f;
#pragma unroll
      for (int k = 0; k < OPL; ++k) se += expf(z[k] - mx);
      se = h2::group_allreduce<8>(se, add_op);
      const float lse = mx + logf(se);
      float best = -FLT_MAX;
#pragma unroll
      for (int k = 0; k < OPL; ++k) {
        const int o = l8 + 8 * k;
        sc[k] = -FLT_MAX;
        if (o < no) {
          if (out.greedy) {
            sc[k] = z[k];
          } else {
            Philox4 rnd = philox4x32_10(gid, out.step, (uint32_t)(o >> 2), 0x504f4c49u /*"POLI"*/, out.seed_lo, out.seed_hi);
            const uint32_t wds[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
            sc[k] = z[k] - logf(-logf(u01_open(wds[o & 3])));
          }
        }
        best = fmaxf(best, sc[k]);
      }
      best = h2::group_allreduce<8>(best, max_op);
      float ai = 1e9f;  // lowest output index that reaches the best score
#pragma unroll
      for (int k = OPL - 1; k >= 0; --k)
        if (sc[k] == best && l8 + 8 * k < no) ai = (float)(l8 + 8 * k);
      ai = h2::group_allreduce<8>(ai, min_op);
      const int a = ai < 1e8f ? (int)ai : 0;
      float lp = 0.0f;
#pragma unroll
      for (int k = 0; k < OPL; ++k)
        if (l8 + 8 * k == a) lp = z[k] - lse;
      lp = h2::group_allreduce<8>(lp, add_op);
      if (l8 == 0) {
        out.action[row] = a;
        out.log_prob[row] = lp;
      }
    }
  }
  STAMP_FLUSH();
}

template <int NOA, int RT>
__global__ __launch_bounds__(256, 1) void rec_step_h2_kernel(NetH2 actor, NetH2 critic, int ngrp_actor, RecStepOut out) {
  extern __shared__ __attribute__((aligned(16))) u8 lds_h2[];
  if ((int)blockIdx.x < ngrp_actor) rec_step_h2_body<NOA, true, RT>(actor, out, lds_h2, (int)blockIdx.x);
  else rec_step_h2_body<8, false, RT>(critic, out, lds_h2, (int)blockIdx.x - ngrp_actor);  // (one output of the 8-wide head tile)
}

long pack_bytes(int din) { return (long)(4 * ((din + 15) / 16) + 96 + 96 + 32) * 2048; }

void setup(NetH2& n) {
  n.nb1 = (n.n.din + 15) / 16;
  n.xrow = 32 * n.nb1 + 16;
  const int xb = 2 * 32 * n.xrow;
  n.abytes = xb > IIMG ? xb : IIMG;
}

template <int NOA, int RT>
int launch_step(const NetH2& a, const NetH2& c, int ga, int gc, const RecStepOut& so, hipStream_t s) {
  const size_t la = (size_t)RT * (a.abytes + IIMG) + (size_t)(RT * 4 * NOA * 32 + 6 * MLP_H + NOA) * 4;
  const size_t lc = (size_t)RT * (c.abytes + IIMG) + (size_t)(RT * 4 * 8 * 32 + 6 * MLP_H + 8) * 4;
  const size_t lb = la > lc ? la : lc;
  if (lb > 163840) return 1;
  static bool attr_set = false;
  if (!attr_set) {
    MAVA_HIP_CHECK(hipFuncSetAttribute((const void*)rec_step_h2_kernel<NOA, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((rec_step_h2_kernel<NOA, RT>), dim3(ga + gc), dim3(256), lb, s, a, c, ga, so);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

#ifdef MAVA_STAMPS
extern "C" int mava_debug_set_step_stamps(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_step_stamps), &p, sizeof(p));
}
#endif

extern "C" long mava_rec_step_pack_bytes(int din) { return pack_bytes(din); }

extern "C" int mava_rec_step_pack_f32(const float* params, int din, void* pack, hipStream_t s) {
  MAVA_ARG_CHECK(params && pack && din >= 1, 0, "mava_rec_step_pack_f32: bad arguments (din=%d)", din);
  hipLaunchKernelGGL(rec_pack_kernel, dim3(32), dim3(64), 0, s, params, din, (din + 15) / 16, reinterpret_cast<u8*>(pack));
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

int mava_rec_step_h2_launch(const RecNet& a_, const RecNet& c_, const void* pack_a, const void* pack_c, const RecStepOut& so,
                            hipStream_t s) {
  NetH2 a = {}, c = {};
  a.n = a_; c.n = c_;
  a.pack = reinterpret_cast<const u8*>(pack_a);
  c.pack = reinterpret_cast<const u8*>(pack_c);
  setup(a);
  setup(c);
  const int ta = a.n.rows / 32, tc = c.n.rows / 32;
  const int noa = a.n.no <= 8 ? 8 : (a.n.no <= 16 ? 16 : 32);
  // the smallest RT (row tiles per weight pass) whose groups cover the chip in one round
  int rt = 1;
  while (rt < 3 && (ta + rt - 1) / rt + (tc + rt - 1) / rt > 256) ++rt;
  const int ga = (ta + rt - 1) / rt, gc = (tc + rt - 1) / rt;
#define STEP(NOAv, RTv) \
  if (noa == NOAv && rt == RTv) return launch_step<NOAv, RTv>(a, c, ga, gc, so, s)
  STEP(8, 1); STEP(8, 2); STEP(8, 3); STEP(16, 1); STEP(16, 2); STEP(16, 3);
#undef STEP
  return 1;  // 17..32 outputs: the f32 kernel
}
