// Split-f16 ("f16x2") matrix building blocks shared by the fused PPO gradient kernel (ppo_train_h2.hip) and the fused
// rollout kernel (rollout_h2.hip): operand fragments of v_mfma_f32_32x32x16_f16 as hi + lo f16 terms, the three-MFMA
// product, LDS images ([32 rows][128 features] f16, hi and lo plane) with row reads (ds_read_b128) and hardware-
// transposed reads (ds_read_b64_tr_b16), and the producer-side split of an accumulator tile.
#pragma once
#include "mlp_core.h"

namespace h2 {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned char u8;

constexpr int IMG_ROW = 272;       // bytes per image row: 128 f16 + 16 (an odd number of 16-byte slots)
constexpr int IMG_PLANE = 32 * IMG_ROW;
constexpr int IMG_BYTES = 2 * IMG_PLANE;  // hi plane, lo plane
constexpr int DY_ROW = 80;         // [32 rows][32 outputs] f16 + 16
constexpr int DY_PLANE = 32 * DY_ROW;
constexpr float W3_SCALE = 128.0f, W3_UNSCALE = 1.0f / 128.0f;


struct Frag {  // one split MFMA operand: 8 k-values per lane as hi + lo
  half8 hi, lo;
};

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
// WEIGHT split with error diffusion along the product's summation index: what the two f16 terms of one weight fail
// to represent (|.| <= 2^-22 |w|) is carried into the low term of the NEXT weight of the same output feature, so the
// representation errors of a column sum to one final carry instead of adding up.  Why: a weight's representation
// error is the same for every batch row - with non-negative (post-ReLU) inputs it shifts an output feature
// coherently in all rows, and in a gradient that cancels to 1/sqrt(rows) of its terms (the value loss at 10^6 rows)
// that shift was the one visible difference to exact f32 (2.6e-4 of the gradient's rms on 3 of 50 561 entries).
__device__ __forceinline__ void split1_carry(float v, float& carry, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  const float rr = (v - (float)hi) + carry;  // exact residual of hi, plus what earlier weights could not represent
  lo = (_Float16)rr;
  carry = rr - (float)lo;
}
__device__ __forceinline__ Frag split8_carry(const float (&v)[8], float& carry) {
  Frag f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 a, b;
    split1_carry(v[i], carry, a, b);
    f.hi[i] = a;
    f.lo[i] = b;
  }
  return f;
}
__device__ __forceinline__ Frag split8(const float (&v)[8]) {
  Frag f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 a, b;
    split1(v[i], a, b);
    f.hi[i] = a;
    f.lo[i] = b;
  }
  return f;
}
__device__ __forceinline__ f32x16 mfma3(const Frag& a, const Frag& b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);  // small terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
}

// B-operand fragment where the product sums over the image's COLUMN (feature) index: 8 consecutive features of row r
__device__ __forceinline__ Frag read_row_frag(const u8* img, int plane_bytes, int byte_off) {
  Frag f;
  f.hi = *reinterpret_cast<const half8*>(img + byte_off);
  f.lo = *reinterpret_cast<const half8*>(img + plane_bytes + byte_off);
  return f;
}
// Operand fragment where the product sums over the image's ROW (batch row) index: lane (r, h) receives column
// (C0 + r) of rows R0 + 8h + 0..7 - two hardware-transposed reads per plane.  `a0` is this lane's byte address of
// block row q = (lane & 15) >> 2, columns 4 * (lane & 3) .. + 3 of the FIRST 4-row block; the second block is 4 rows on.
#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))
__device__ __forceinline__ half8 read_tr8(const u8* a0, int row_bytes) {
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(a0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(a0 + 4 * row_bytes));
  return __builtin_bit_cast(half8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ Frag read_tr_frag(const u8* a0, int plane_bytes, int row_bytes) {
  Frag f;
  f.hi = read_tr8(a0, row_bytes);
  f.lo = read_tr8(a0 + plane_bytes, row_bytes);
  return f;
}

// accumulator tile (feature fo(q,h) of the wave's slice on register q, row r on the lane) -> split -> image rows.
// Registers 4g..4g+3 of a lane are 4 consecutive features: one 8-byte store per plane and group.  The packed groups
// are returned: groups (2s, 2s+1) of a lane are also the k-step-s fragment of a product that sums over these features.
__device__ __forceinline__ void write_image(u8* img, int r, int col0 /* 32w + 4h */, const f32x16& acc, half4 (&ph)[4],
                                            half4 (&pl)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a, b;
      split1(acc[4 * g + e], a, b);
      ph[g][e] = a;
      pl[g][e] = b;
    }
    u8* p = img + r * IMG_ROW + 2 * (col0 + 8 * g);
    *reinterpret_cast<half4*>(p) = ph[g];
    *reinterpret_cast<half4*>(p + IMG_PLANE) = pl[g];
  }
}


// ---- Swizzled images: [rows][128 features] f16 in 256-byte rows, 16-byte chunk c of row `row` stored at chunk
// c ^ sw(row), sw(row) = ((row & 3) << 2) | ((row >> 2) & 3).  The padded images above (272-byte rows) serve row reads
// without conflicts, but the four block rows a transposed read's 32-lane half touches land 4 banks apart there: every
// ds_read_b64_tr_b16 is 4-way conflicted (8 LDS cycles instead of 2).  With the XOR both kinds of read are conflict-free
// (bank = (a / 4) % 64 for ds_read_b128 / _b64_tr_b16; lane groups as in MI355X_MICROARCH.md, LDS).  Rows are addressed
// modulo 16 by sw, so an image may have any multiple of 16 rows (W2: 128).
constexpr int SW_ROW = 256;
constexpr int SW_PLANE = 32 * SW_ROW;
__device__ __forceinline__ int sw_of(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int sw_off(int row, int chunk) { return SW_ROW * row + 16 * (chunk ^ sw_of(row)); }
// row read by lane (r, h): features 16 s + 8 h .. + 7 (chunk 2 s + h) of image row `row`
struct SwRow {
  int base, hx;
};
__device__ __forceinline__ SwRow sw_row(int row, int h) {
  SwRow a;
  a.base = SW_ROW * row;
  a.hx = (16 * h) ^ (16 * sw_of(row));
  return a;
}
__device__ __forceinline__ Frag sw_read_row(const u8* img, int plane_bytes, const SwRow& a, int s) {
  return read_row_frag(img, plane_bytes, a.base + ((32 * s) ^ a.hx));
}
// transposed read: rows 16 s + 8 h + 0..7 of column tile t (columns 32 t + r).  Lane 4q + p of a 16-lane group supplies
// row q (second read: q + 4) of the block, chunk 4 t + 2 g1 + (p >> 1), byte 8 (p & 1) of it.
struct SwTr {
  int b0, b1, q64;
};
__device__ __forceinline__ SwTr sw_tr(int lane) {
  const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int cx = 2 * g1 + (p >> 1);
  const int trb = SW_ROW * (8 * h + q) + 8 * (p & 1);
  SwTr a;
  a.b0 = trb + 16 * (cx ^ (2 * h));                    // rows 8h + q:     (row >> 2) & 3 = 2h
  a.b1 = trb + 16 * (cx ^ (2 * h + 1)) + 4 * SW_ROW;   // rows 8h + 4 + q: (row >> 2) & 3 = 2h + 1
  a.q64 = 64 * q;
  return a;
}
__device__ __forceinline__ Frag sw_read_tr(const u8* img, int plane_bytes, const SwTr& a, int s, int t) {
  const int x = ((64 * t) ^ a.q64) + 16 * s * SW_ROW;
  const u8* p0 = img + a.b0 + x;
  const u8* p1 = img + a.b1 + x;
  Frag f;
  {
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1));
    f.hi = __builtin_bit_cast(half8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
  }
  {
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0 + plane_bytes));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1 + plane_bytes));
    f.lo = __builtin_bit_cast(half8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
  }
  return f;
}
// write_image for a swizzled image: the wave's features 32 w + 4 h + 8 g .. + 3 of row r = chunk 4 w + g, byte 8 h
__device__ __forceinline__ void sw_write_image(u8* img, int r, int w, int h, const f32x16& acc, half4 (&ph)[4],
                                               half4 (&pl)[4]) {
  const int sw16 = 16 * sw_of(r);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a, b;
      split1(acc[4 * g + e], a, b);
      ph[g][e] = a;
      pl[g][e] = b;
    }
    u8* p = img + SW_ROW * r + 8 * h + ((64 * w + 16 * g) ^ sw16);
    *reinterpret_cast<half4*>(p) = ph[g];
    *reinterpret_cast<half4*>(p + SW_PLANE) = pl[g];
  }
}

// Pre-split copy of a WIDE layer-1 weight matrix in fragment order for ppo_train_h2.hip's register ring:
// out[step s][wave w][lane] = {8 x hi, 8 x lo} (32 bytes per lane); lane (r, h) of wave w holds W1[k = 16 s + 8 h + e][f = 32 w + r];
// k == din is b1, k > din zero.  One 256-thread group (tid = 0..255): the error-diffusion carry of split1_carry runs
// along k over each lane's own inputs, so the work is a chain of memory round trips - every load of the thread
// (8 x STEPS <= 144 floats) is issued before the first is used: one round trip instead of one per few steps.
// `scale` (a power of two): the weights are split as scale * w (the consumer unscales its f32 accumulator), see W_SCALE_CRITIC.
template <int STEPS>
__device__ __forceinline__ void pack_w1_body(const float* __restrict__ P, int din, uint4* __restrict__ out, int tid, float scale) {
  const int w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  float v[STEPS][8];
  // pass 1: nothing but loads (rows past din clamped to a valid address); pass 2: select, scale, split.  A select or a
  // multiplication next to its load made the compiler wait for every load in turn - 144 round trips: 62 instead of 9 us
  // for the stand-alone launch, 58 instead of 25 us for the Adam launch that carries this body in its last block.
#pragma unroll
  for (int i = 0; i < STEPS; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * i + 8 * h + e;
      v[i][e] = P[(k <= din ? k : din) * MLP_H + 32 * w + r];
    }
  __builtin_amdgcn_sched_barrier(0);
  float carry = 0.0f;
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = (16 * i + 8 * h + e <= din) ? v[i][e] * scale : 0.0f;
    const Frag f = split8_carry(v[i], carry);
    const int gid = i * 256 + tid;
    out[2 * gid] = __builtin_bit_cast(uint4, f.hi);
    out[2 * gid + 1] = __builtin_bit_cast(uint4, f.lo);
  }
}
// The critic's W1 / W2 are split as 16 w.  A weight of magnitude ~0.1 has its low term (|w - hi| <= 2^-11 |w| ~ 4e-5) in
// f16's SUBNORMAL range, whose spacing is an absolute 2^-24 = 6e-8: what hi + lo then fails to represent is ~2e-7 |w| - and it
// is the same in every batch row, so with non-negative (post-ReLU) inputs it shifts the value coherently by ~1e-8, which
// the value-loss gradient (entries that cancel to 1 / sqrt(rows) of their terms) shows at the full launch shape: dW3 entries at
// 1.1 - 1.6 x the 1e-4 tolerance over five seeds where exact f32 sits at 0.2 - 0.9 (tools/debug_fullshape.py; the matrix pipe
// keeps subnormal inputs and rounds a 16-product group once, without bias: tools/microbench/mfma_round_probe.hip).  Scaled by
// 2^4 the low terms are normal numbers with their full 11 bits; the f32 accumulator is unscaled by an exact multiplication.
constexpr float W_SCALE_CRITIC = 16.0f;
constexpr size_t W1_SPLIT_BYTES = (size_t)18 * 256 * 32;  // the largest instantiated layer (18 steps = 287 inputs + bias)

// All-reduce over aligned groups of G consecutive lanes (G = 8, 16, 32) on the VALU's DPP path (as ppo_train.hip)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G, typename Op>
__device__ __forceinline__ float group_allreduce(float v, Op op) {
  v = op(v, dpp_f<0xB1>(v));
  v = op(v, dpp_f<0x4E>(v));
  if (G >= 8) v = op(v, dpp_f<0x141>(v));
  if (G >= 16) v = op(v, dpp_f<0x140>(v));
  if (G >= 32) v = op(v, __shfl_xor(v, 16, 64));
  return v;
}


}  // namespace h2
