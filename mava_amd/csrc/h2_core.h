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
