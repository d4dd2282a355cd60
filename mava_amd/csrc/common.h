// Shared host/device helpers for libmavahip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MAVA_OK 0
#define MAVA_EARG(k) (-1000 - (k))

// Thread-local last-error text, read through mava_last_error().
void mava_set_error(const char* fmt, ...);

#define MAVA_ARG_CHECK(cond, k, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      mava_set_error(__VA_ARGS__);     \
      return MAVA_EARG(k);             \
    }                                  \
  } while (0)

#define MAVA_HIP_CHECK(expr)                                                   \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      mava_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                     __FILE__, __LINE__);                                      \
      return -(int)_e;                                                         \
    }                                                                          \
  } while (0)

#define MAVA_LAUNCH_CHECK() MAVA_HIP_CHECK(hipGetLastError())

static inline int mava_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- Philox4x32-10 (Salmon et al. 2011), counter-based RNG shared by the policy sampler
// and the synthetic environment.  Restated bit-for-bit in oracle/philox.py.
struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ static inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                        uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 r = {c0, c1, c2, c3};
  return r;
}

// 24-bit uniform in the open interval (0,1): (top24 + 0.5) * 2^-24 (exact in f32).
__host__ __device__ static inline float u01_open(uint32_t x) {
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
#endif
