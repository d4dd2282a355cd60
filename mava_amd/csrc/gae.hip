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
#include "ctx.h"

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

// One block = one strip of CB = LPC*VEC columns; the NC time chunks of a slab are spread over
// NC groups of LPC lanes (LPC = 64: one wave per chunk; LPC = 16 with VEC = 4: a wave covers four chunks
// and still reads 256 contiguous bytes per chunk row).  Slabs of L*NC time steps are walked from the end
// of the rollout to the start; T <= L*NC (the BASELINE shapes) is a single slab.
template <int VEC, int L, int NC, int LPC>
__global__ __launch_bounds__(LPC * NC) void gae_kernel_ragged(
    const float* __restrict__ reward, const float* __restrict__ value,
    const uint8_t* __restrict__ done, const float* __restrict__ last_val,
    const uint8_t* __restrict__ last_done, int T, int N, float gamma, float lambda,
    float* __restrict__ adv, float* __restrict__ tgt) {
  constexpr int CB = LPC * VEC;
  __shared__ float sP[NC][CB];
  __shared__ float sS[NC][CB];
  __shared__ float sAin[CB];  // advantage entering the current slab from later time

  const int chunk = threadIdx.x / LPC;
  const int lc = threadIdx.x - chunk * LPC;
  const long col = ((long)blockIdx.x * LPC + lc) * VEC;
  const bool live = col < N;
  const bool shifted = last_done != nullptr;  // recurrent masking
  const float gl = gamma * lambda;

  const int slab_len = L * NC;
  const int n_slab = (T + slab_len - 1) / slab_len;

  if (chunk == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sAin[lc * VEC + i] = 0.0f;
  }

  for (int slab = n_slab - 1; slab >= 0; --slab) {
    const int t0 = slab * slab_len + chunk * L;  // first step of this thread's chunk
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
      sP[chunk][lc * VEC + i] = p[i];
      sS[chunk][lc * VEC + i] = a[i];
    }
    __syncthreads();

    // ---- compose the later chunks of this slab (and the carry from later slabs)
    float ain[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ain[i] = sAin[lc * VEC + i];
    for (int k = NC - 1; k > chunk; --k) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) ain[i] = sS[k][lc * VEC + i] + sP[k][lc * VEC + i] * ain[i];
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
        for (int i = 0; i < VEC; ++i) sAin[lc * VEC + i] = a[i] + p[i] * ain[i];
      }
      __syncthreads();
    }
  }
}


// Fast path (T % L == 0, so a chunk is either entirely inside the rollout or entirely past its end).
// The load phase is branch-free and only moves raw bits into registers: out-of-range rows / columns are
// clamped to valid addresses and masked later, the done bytes are converted after every load of the
// chunk has been issued (a conversion next to its load makes the compiler wait for memory once per step,
// i.e. L serial round trips instead of one).  Loads are issued last step first, the order the reverse
// scan consumes them.  The cross-chunk composition is a fixed-trip predicated loop so its LDS reads are
// issued ahead of the dependent FMA chain.
template <int VEC, int L, int NC, int LPC>
__global__ __launch_bounds__(LPC * NC) void gae_kernel(
    const float* __restrict__ reward, const float* __restrict__ value,
    const uint8_t* __restrict__ done, const float* __restrict__ last_val,
    const uint8_t* __restrict__ last_done, int T, int N, float gamma, float lambda,
    float* __restrict__ adv, float* __restrict__ tgt) {
  typedef typename VecT<VEC>::f VF;
  typedef typename VecT<VEC>::b VB;
  constexpr int CB = LPC * VEC;
  __shared__ float sP[NC][CB];
  __shared__ float sS[NC][CB];
  __shared__ float sAin[CB];

  const int chunk = threadIdx.x / LPC;
  const int lc = threadIdx.x - chunk * LPC;
  // XCD-aware strip order: workgroups go to the 8 XCDs round-robin, so consecutive strips are given to
  // workgroups of the SAME XCD - neighbouring strips share 64-byte sectors of the done bytes (and, for narrow
  // strips, 128-byte lines of reward / value), which then hit that XCD's L2 instead of being fetched twice.
  const int nblk = gridDim.x;
  const int strip = (nblk % 8 == 0) ? ((int)(blockIdx.x & 7) * (nblk >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
  const long col = ((long)strip * LPC + lc) * VEC;
  const bool live = col < N;
  const long colc = live ? col : 0;
  const bool shifted = last_done != nullptr;
  const float gl = gamma * lambda;
  constexpr int slab_len = L * NC;
  const int n_slab = (T + slab_len - 1) / slab_len;

  if (chunk == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sAin[lc * VEC + i] = 0.0f;
  }

  for (int slab = n_slab - 1; slab >= 0; --slab) {
    const int t0 = slab * slab_len + chunk * L;
    const bool valid = t0 < T;            // whole chunk (T % L == 0)
    const int tb = valid ? t0 : T - L;    // clamped base row for the loads
    VF rr[L], vv[L + 1];
    VB dd[L];
    {
      const int t = tb + L;
      const float* pv = (t < T) ? value + (long)t * N + colc : last_val + colc;
      vv[L] = *reinterpret_cast<const VF*>(pv);
    }
#pragma unroll
    for (int s = L - 1; s >= 0; --s) {
      const int t = tb + s;
      const long off = (long)t * N + colc;
      const uint8_t* pd = done + off;
      if (shifted) pd = (t + 1 < T) ? pd + N : last_done + colc;
      vv[s] = *reinterpret_cast<const VF*>(value + off);
      rr[s] = *reinterpret_cast<const VF*>(reward + off);
      dd[s] = *reinterpret_cast<const VB*>(pd);
    }

    float S[L][VEC], P[L][VEC];
    float a[VEC], p[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a[i] = 0.f; p[i] = 1.f; }
#pragma unroll
    for (int s = L - 1; s >= 0; --s) {
      const float* r_ = reinterpret_cast<const float*>(&rr[s]);
      const float* v_ = reinterpret_cast<const float*>(&vv[s]);
      const float* vn = reinterpret_cast<const float*>(&vv[s + 1]);
      const uint8_t* d_ = reinterpret_cast<const uint8_t*>(&dd[s]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float nd = d_[i] ? 0.0f : 1.0f;
        const float c = gl * nd;
        const float delta = r_[i] + gamma * vn[i] * nd - v_[i];
        a[i] = delta + c * a[i];
        p[i] = c * p[i];
        S[s][i] = a[i];
        P[s][i] = p[i];
      }
    }
    // a chunk past the end of the rollout is the identity map
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sP[chunk][lc * VEC + i] = valid ? p[i] : 1.0f;
      sS[chunk][lc * VEC + i] = valid ? a[i] : 0.0f;
    }
    __syncthreads();

    float ain[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ain[i] = sAin[lc * VEC + i];
  constexpr int CU_ = NC <= 16 ? NC : 8;
#pragma unroll CU_
    for (int k = NC - 1; k >= 1; --k) {
      VF s4 = *reinterpret_cast<const VF*>(&sS[k][lc * VEC]);
      VF p4 = *reinterpret_cast<const VF*>(&sP[k][lc * VEC]);
      const float* s_ = reinterpret_cast<const float*>(&s4);
      const float* p_ = reinterpret_cast<const float*>(&p4);
#pragma unroll
      for (int i = 0; i < VEC; ++i) ain[i] = (k > chunk) ? s_[i] + p_[i] * ain[i] : ain[i];
    }

    if (live && valid) {
#pragma unroll
      for (int s = 0; s < L; ++s) {
        const float* v_ = reinterpret_cast<const float*>(&vv[s]);
        float oa[VEC], ot[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          oa[i] = S[s][i] + P[s][i] * ain[i];
          ot[i] = oa[i] + v_[i];
        }
        stf<VEC>(adv + (long)(t0 + s) * N + col, oa);
        stf<VEC>(tgt + (long)(t0 + s) * N + col, ot);
      }
    }

    if (slab > 0) {
      __syncthreads();
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sAin[lc * VEC + i] = (valid ? a[i] : 0.0f) + (valid ? p[i] : 1.0f) * ain[i];
      }
      __syncthreads();
    }
  }
}

template <int VEC, int L, int NC, int LPC>
int launch_gae(const float* reward, const float* value, const uint8_t* done, const float* last_val,
               const uint8_t* last_done, int T, int N, float gamma, float lambda, float* adv,
               float* tgt, hipStream_t s) {
  const int cols_per_block = LPC * VEC;
  dim3 grid(mava_cdiv(N, cols_per_block)), block(LPC * NC);
  if (T % L == 0)
    hipLaunchKernelGGL((gae_kernel<VEC, L, NC, LPC>), grid, block, 0, s, reward, value, done, last_val,
                       last_done, T, N, gamma, lambda, adv, tgt);
  else
    hipLaunchKernelGGL((gae_kernel_ragged<VEC, L, NC, LPC>), grid, block, 0, s, reward, value, done, last_val,
                       last_done, T, N, gamma, lambda, adv, tgt);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

}  // namespace

extern "C" int mava_gae_f32(const mava_ctx* ctx, const float* reward, const float* value, const uint8_t* done,
                            const float* last_val, const uint8_t* last_done, int T, int N,
                            float gamma, float lambda, float* adv, float* tgt, hipStream_t s) {
  MAVA_ARG_CHECK(T >= 0 && N >= 0, 0, "mava_gae_f32: negative shape T=%d N=%d", T, N);
  if (T == 0 || N == 0) return MAVA_OK;
  MAVA_ARG_CHECK(reward && value && done && last_val && adv && tgt, 1,
                 "mava_gae_f32: null pointer argument");
#define GAE_ARGS reward, value, done, last_val, last_done, T, N, gamma, lambda, adv, tgt, s
  // columns must stay VEC-aligned in every time row
  const int align = (N % 4 == 0) ? 4 : ((N % 2 == 0) ? 2 : 1);
  int variant = mava_ctx_gae_variant(ctx);  // tuning knob of bench sweeps (tools/gae_sweep.py), 0 = default
  if (variant == 0) {
    // default (tools/gae_sweep.py on MI355X, graph-timed): scalar columns in 32-column strips (one 128-byte
    // line per time row), 8 chunks of 16 steps - 10.4 us from HBM / 7.2 us from cache at (128, 16384),
    // next to 9.1 / 5.1 us for a device copy of the same byte count
    variant = (N >= 32 * 64) ? 13 : 1;
  }
  switch (variant) {
    case 41: if (align == 4) return launch_gae<4, 4, 32, 16>(GAE_ARGS); break;   // CB 64, 512 threads
    case 42: if (align == 4) return launch_gae<4, 8, 16, 16>(GAE_ARGS); break;   // CB 64, 256 threads
    case 43: if (align == 4) return launch_gae<4, 4, 32, 32>(GAE_ARGS); break;   // CB 128, 1024 threads
    case 44: if (align == 4) return launch_gae<4, 16, 8, 16>(GAE_ARGS); break;   // CB 64, 128 threads
    case 45: if (align == 4) return launch_gae<4, 4, 32, 8>(GAE_ARGS); break;    // CB 32, 256 threads
    case 46: if (align == 4) return launch_gae<4, 8, 16, 8>(GAE_ARGS); break;    // CB 32, 128 threads
    case 47: if (align == 4) return launch_gae<4, 2, 64, 16>(GAE_ARGS); break;   // CB 64, 1024 threads
    case 24: if (align >= 2) return launch_gae<2, 4, 32, 16>(GAE_ARGS); break;   // CB 32, 512 threads
    case 21: if (align >= 2) return launch_gae<2, 8, 16, 32>(GAE_ARGS); break;   // CB 64, 512 threads
    case 22: if (align >= 2) return launch_gae<2, 4, 32, 32>(GAE_ARGS); break;   // CB 64, 1024 threads
    case 23: if (align >= 2) return launch_gae<2, 16, 8, 64>(GAE_ARGS); break;   // CB 128, 512 threads
    case 11: return launch_gae<1, 8, 16, 64>(GAE_ARGS);                            // CB 64, 1024 threads
    case 12: return launch_gae<1, 8, 16, 32>(GAE_ARGS);                            // CB 32, 512 threads
    case 13: return launch_gae<1, 16, 8, 32>(GAE_ARGS);                            // CB 32, 256 threads
    case 14: return launch_gae<1, 32, 4, 64>(GAE_ARGS);                            // CB 64, 256 threads
    case 15: return launch_gae<1, 4, 32, 32>(GAE_ARGS);                            // CB 32, 1024 threads
    case 16: return launch_gae<1, 16, 8, 16>(GAE_ARGS);                            // CB 16, 128 threads
    default: break;
  }
  return launch_gae<1, 16, 8, 64>(GAE_ARGS);  // CB 64, 512 threads (any N)
#undef GAE_ARGS
}
