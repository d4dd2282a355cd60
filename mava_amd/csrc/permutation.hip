// Epoch permutation of the minibatch shuffle (mava/systems/ppo/ff_mappo.py:272-273 `jax.random.permutation(key,
// batch_size)`, rec_mappo.py:277-279 over the env axis), generated in place: no keys to sort.
//
// torch.randperm / a key sort costs ~0.25 ms of merge-sort passes per 524 288-entry permutation (1 ms per update at
// K = 4, 9 % of the headline update).  A keyed bijection of [0, 2^b) (b = ceil(log2 n), alternating unbalanced Feistel
// network, 16 rounds) evaluated per index, with cycle walking for the part of the domain >= n, writes the same array
// in one pass of pure integer work: out[i] = P^m(i), m = the first power with P^m(i) < n (a bijection of [0, n): the
// walk follows i's cycle of P and i itself is < n, so it terminates; 2^b < 2n makes the expected walk < 2 steps).
// The reference's threefry stream cannot be reproduced without JAX, so like every random draw of this library the
// stream differs from Mava's; what is pinned bit-for-bit is oracle/permutation.py (same rounds, same keys).
#include "common.h"

namespace {

constexpr int PERM_ROUNDS = 16;

struct PermKeys {
  uint32_t k[PERM_ROUNDS];
};

// lowbias32 (C. Wellons' 32-bit integer hash), keyed by addition
__device__ __host__ inline uint32_t perm_mix(uint32_t x, uint32_t k) {
  uint32_t h = x + k;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void permutation_kernel(uint32_t n, int lb, int rb, PermKeys keys,
                                                          int32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask_r = (1u << rb) - 1u;
  uint32_t v = i;
  do {
    uint32_t L = v >> rb, R = v & mask_r;
#pragma unroll
    for (int r = 0; r < PERM_ROUNDS; r += 2) {
      L ^= perm_mix(R, keys.k[r]) >> (32 - lb);
      R ^= perm_mix(L, keys.k[r + 1]) >> (32 - rb);
    }
    v = (L << rb) | R;
  } while (v >= n);
  out[i] = (int32_t)v;
}

}  // namespace

extern "C" int mava_permutation_i32(long n, uint64_t seed, uint64_t counter, int32_t* out, hipStream_t s) {
  MAVA_ARG_CHECK(n >= 1 && n < (1L << 31), 0, "mava_permutation_i32: n=%ld (1 <= n < 2^31)", n);
  MAVA_ARG_CHECK(out, 1, "mava_permutation_i32: null pointer argument");
  int b = 2;
  while ((1L << b) < n) ++b;
  const int lb = b >> 1, rb = b - lb;
  PermKeys keys;
  uint64_t st = seed + 0x9E3779B97F4A7C15ull * (counter + 1);  // splitmix64 stream of this (seed, counter)
  for (int r = 0; r < PERM_ROUNDS; ++r) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    keys.k[r] = (uint32_t)(z >> 32);
  }
  hipLaunchKernelGGL(permutation_kernel, dim3(mava_cdiv(n, 256)), dim3(256), 0, s, (uint32_t)n, lb, rb, keys, out);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
