// Fused GAE reverse scan (K6 / SURVEY §8 row A4).
//
// Reference semantics: mava/systems/ppo/ff_mappo.py:112-139 (_calculate_gae) and
// mava/systems/ppo/rec_mappo.py:177-199 (the recurrent variant, which masks with the *next*
// step's stored done flag and seeds the carry with last_done).
//
//   delta_t = r_t + gamma * V_{t+1} * (1 - dn_t) - V_t
//   A_t     = delta_t + gamma * lambda * (1 - dn_t) * A_{t+1},   A_T = 0, V_T = last_val
//   target_t = A_t + V_t
//   dn_t = done[t]               (feed-forward systems)
//   dn_t = done[t+1], done[T] = last_done   (recurrent systems: done[] holds the flag entering step t)
//
// Layout: time-major (T, N) with N = envs*agents contiguous, so one wave reads 64*VEC adjacent
// columns of one time row per instruction (fully coalesced).
//
// MI355X mapping: the scan is HBM-bound (17 B per element).  A column-sequential scan has
// only N/64 waves; instead the time axis is cut into NC chunks of L steps, one wave per
// (64*VEC columns, chunk).  Every wave issues all of its loads up front (no dependence on other
// chunks), runs the affine recurrence A_t = delta_t + c_t * A_in locally with A_in = 0 while
// tracking the running product P_t = prod c, publishes its chunk-head pair (P, S) to LDS, and
// after ONE barrier composes the later chunks' pairs (<= NC-1 FMAs) to get its true A_in and
// fixes up A_t = S_t + P_t * A_in from registers.
#include "common.h"

namespace {

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
  typedef float f;
  typedef uint8_t b;
};
template <>
struct VecT<2> {
  typedef float2 f;
  typedef uchar2 b;
};
template <>
struct VecT<4> {
  typedef float4 f;
  typedef uchar4 b;
};

template <int VEC>
__device__ inline void ldf(const float* p, float (&o)[VEC]) {
  typename VecT<VEC>::f v = *reinterpret_cast<const typename VecT<VEC>::f*>(p);
  const float* s = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = s[i];
}
template <int VEC>
__device__ inline void ldb(const uint8_t* p, float (&o)[VEC]) {
  typename VecT<VEC>::b v = *reinterpret_cast<const typename VecT<VEC>::b*>(p);
  const uint8_t* s = reinterpret_cast<const uint8_t*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = s[i] ? 0.0f : 1.0f;  // 1 - done
}
template <int VEC>
__device__ inline void stf(float* p, const float (&o)[VEC]) {
  typename VecT<VEC>::f v;
  float* s = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = o[i];
  *reinterpret_cast<typename VecT<VEC>::f*>(p) = v;
}

// One block = NC waves = one strip of 64*VEC columns.  Slabs of L*NC time steps are walked from
// the end of the rollout to the start; T <= L*NC (the BASELINE shapes) is a single slab.
template <int VEC, int L, int NC>
__global__ __launch_bounds__(64 * NC) void gae_kernel(
    const float* __restrict__ reward, const float* __restrict__ value,
    const uint8_t* __restrict__ done, const float* __restrict__ last_val,
    const uint8_t* __restrict__ last_done, int T, int N, float gamma, float lambda,
    float* __restrict__ adv, float* __restrict__ tgt) {
  __shared__ float sP[NC][64 * VEC];
  __shared__ float sS[NC][64 * VEC];
  __shared__ float sAin[64 * VEC];  // advantage entering the current slab from later time

  const int lane = threadIdx.x & 63;
  const int chunk = threadIdx.x >> 6;
  const long col = ((long)blockIdx.x * 64 + lane) * VEC;
  const bool live = col < N;
  const bool shifted = last_done != nullptr;  // recurrent masking
  const float gl = gamma * lambda;

  const int slab_len = L * NC;
  const int n_slab = (T + slab_len - 1) / slab_len;

  if (chunk == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sAin[lane * VEC + i] = 0.0f;
  }

  for (int slab = n_slab - 1; slab >= 0; --slab) {
    const int t0 = slab * slab_len + chunk * L;  // first step of this wave's chunk
    float r[L][VEC], v[L + 1][VEC], nd[L][VEC];

    // ---- issue every load of the chunk before any arithmetic
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const int t = t0 + s;
      if (live && t < T) {
        ldf<VEC>(reward + (long)t * N + col, r[s]);
        ldf<VEC>(value + (long)t * N + col, v[s]);
        if (!shifted) {
          ldb<VEC>(done + (long)t * N + col, nd[s]);
        } else if (t + 1 < T) {
          ldb<VEC>(done + (long)(t + 1) * N + col, nd[s]);
        } else {
          ldb<VEC>(last_done + col, nd[s]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { r[s][i] = 0.f; v[s][i] = 0.f; nd[s][i] = 0.f; }
      }
    }
    {
      const int t = t0 + L;  // V_{t+1} of the chunk's last step
      if (live && t < T) {
        ldf<VEC>(value + (long)t * N + col, v[L]);
      } else if (live && t0 < T) {
        ldf<VEC>(last_val + col, v[L]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[L][i] = 0.f;
      }
    }
    // Steps past T inside the chunk (ragged T) must behave as identity: with r=v=0, nd=0 they
    // produce S=0 and P=0, but the bootstrap value must then sit at the last *valid* step.
    // Move last_val to v[s+1] of the last valid step.
    if (live && t0 < T && t0 + L > T) {
      float lv[VEC];
      ldf<VEC>(last_val + col, lv);
#pragma unroll
      for (int s = 0; s < L; ++s) {
        if (t0 + s + 1 == T) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[s + 1][i] = lv[i];
        }
      }
    }

    // ---- local reverse scan with zero incoming advantage
    float S[L][VEC], P[L][VEC];
    float a[VEC], p[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a[i] = 0.f; p[i] = 1.f; }
#pragma unroll
    for (int s = L - 1; s >= 0; --s) {
      const bool valid = (t0 + s) < T;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (valid) {
          const float c = gl * nd[s][i];
          const float delta = r[s][i] + gamma * v[s + 1][i] * nd[s][i] - v[s][i];
          a[i] = delta + c * a[i];
          p[i] = c * p[i];
        }
        S[s][i] = a[i];
        P[s][i] = p[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sP[chunk][lane * VEC + i] = p[i];
      sS[chunk][lane * VEC + i] = a[i];
    }
    __syncthreads();

    // ---- compose the later chunks of this slab (and the carry from later slabs)
    float ain[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ain[i] = sAin[lane * VEC + i];
    for (int k = NC - 1; k > chunk; --k) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) ain[i] = sS[k][lane * VEC + i] + sP[k][lane * VEC + i] * ain[i];
    }

    // ---- fix up and store
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const int t = t0 + s;
      if (live && t < T) {
        float oa[VEC], ot[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          oa[i] = S[s][i] + P[s][i] * ain[i];
          ot[i] = oa[i] + v[s][i];
        }
        stf<VEC>(adv + (long)t * N + col, oa);
        stf<VEC>(tgt + (long)t * N + col, ot);
      }
    }

    if (slab > 0) {
      __syncthreads();  // everyone has read sAin / sP / sS of this slab
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sAin[lane * VEC + i] = a[i] + p[i] * ain[i];
      }
      __syncthreads();
    }
  }
}

template <int VEC, int L, int NC>
int launch_gae(const float* reward, const float* value, const uint8_t* done, const float* last_val,
               const uint8_t* last_done, int T, int N, float gamma, float lambda, float* adv,
               float* tgt, hipStream_t s) {
  const int cols_per_block = 64 * VEC;
  dim3 grid(mava_cdiv(N, cols_per_block)), block(64 * NC);
  hipLaunchKernelGGL((gae_kernel<VEC, L, NC>), grid, block, 0, s, reward, value, done, last_val,
                     last_done, T, N, gamma, lambda, adv, tgt);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

// Tuning knob exposed for bench sweeps: 0 = default.
static int g_gae_variant = 0;

extern "C" int mava_gae_set_variant(int variant) {
  g_gae_variant = variant;
  return MAVA_OK;
}

extern "C" int mava_gae_f32(const float* reward, const float* value, const uint8_t* done,
                            const float* last_val, const uint8_t* last_done, int T, int N,
                            float gamma, float lambda, float* adv, float* tgt, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 0 && N >= 0, 0, "mava_gae_f32: negative shape T=%d N=%d", T, N);
  if (T == 0 || N == 0) return MAVA_OK;
  MAVA_ARG_CHECK(reward && value && done && last_val && adv && tgt, 1,
                 "mava_gae_f32: null pointer argument");
  // vector width: columns must stay VEC-aligned in every time row
  int vec = 1;
  if (N % 4 == 0) vec = 4;
  else if (N % 2 == 0) vec = 2;
  // Small problems: prefer more, narrower strips so the grid still covers the chip.
  if (vec == 4 && N < 256 * 64 * 4) vec = (N >= 256 * 64 * 2) ? 2 : 1;
  int variant = g_gae_variant;
  if (variant == 1) vec = 1;
  if (variant == 2 && N % 2 == 0) vec = 2;
  if (variant == 3 && N % 4 == 0) vec = 4;
  const bool deep = (variant >= 10);  // 16 chunks of 8 steps instead of 8 chunks of 16
  if (deep) {
    int vv = variant - 10;
    if (vv == 1) vec = 1;
    if (vv == 2 && N % 2 == 0) vec = 2;
    if (vv == 4 && N % 4 == 0) vec = 4;
    switch (vec) {
      case 1: return launch_gae<1, 8, 16>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
      case 2: return launch_gae<2, 8, 16>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
      default: return launch_gae<4, 8, 16>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
    }
  }
  switch (vec) {
    case 1: return launch_gae<1, 16, 8>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
    case 2: return launch_gae<2, 16, 8>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
    default: return launch_gae<4, 16, 8>(reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s);
  }
}
