// Ground-truth probe for the operand maps the f16x2 train kernel relies on (run once on a gfx950):
//   1. ds_read_b64_tr_b16: lane (r = l&31, h = l>>5) must receive T[R0 + 8h + j][C0 + r], j = 0..7, from two reads whose
//      per-lane addresses follow "lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3" (cdna_hip_programming.md T10)
//   2. v_mfma_f32_32x32x16_f16: A[r][8h+j], B[8h+j][r] -> C[(q&3)+8(q>>2)+4h][r]
//   3. accumulator tile as the next product's B operand: element j of half h of k-step s is row 16s+8(j>>2)+4h+(j&3)
// hipcc --offload-arch=gfx950 -O2 tools/microbench/f16_probe.hip -o tools/microbench/f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS3(p) ((__attribute__((address_space(3))) s16x4*)(p))
constexpr int STRIDE = 136;  // halves per image row (272 bytes)

__global__ void probe(float* out_tr, float* out_c, float* out_y, const float* Ah, const float* Bh, const float* Wh) {
  __shared__ __attribute__((aligned(16))) _Float16 img[32 * STRIDE];
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int i = l; i < 32 * STRIDE; i += 64) { const int row = i / STRIDE, col = i % STRIDE; img[i] = (_Float16)(float)(row * 64 + (col & 63)); }
  __syncthreads();
  // ---- 1. transposed read of rows [8h, 8h+8) x columns [C0, C0+32), C0 = 32
  const int i16 = l & 15, q = i16 >> 2, p = i16 & 3, G = l >> 4;
  const int C0 = 32 + 16 * (G & 1), R0 = 8 * h;
  s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(img + (R0 + q) * STRIDE + C0 + 4 * p));
  s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(img + (R0 + 4 + q) * STRIDE + C0 + 4 * p));
  half8 t = __builtin_bit_cast(half8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
  for (int j = 0; j < 8; ++j) out_tr[l * 8 + j] = (float)t[j];
  // ---- 2. MFMA maps with exact small integers: A (32 x 16), B (16 x 32) row-major on the host
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)Ah[r * 16 + 8 * h + j]; b[j] = (_Float16)Bh[(8 * h + j) * 32 + r]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int qq = 0; qq < 16; ++qq) out_c[l * 16 + qq] = c[qq];
  // ---- 3. Y = W (32 x 32) . X, X = C above (32 x 32, column on the lane): two k-steps fed from the accumulator
  f32x16 y = {};
  for (int s = 0; s < 2; ++s) {
    half8 xb, wa;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (_Float16)c[8 * s + j];
      wa[j] = (_Float16)Wh[r * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, y, 0, 0, 0);
  }
  for (int qq = 0; qq < 16; ++qq) out_y[l * 16 + qq] = y[qq];
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), W(32 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (float)((k * 2 + n * 7) % 5 - 2);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) W[i * 32 + k] = (float)((i + 2 * k) % 3 - 1);
  float *dA, *dB, *dW, *d1, *d2, *d3;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dW, W.size() * 4);
  hipMalloc(&d1, 64 * 8 * 4); hipMalloc(&d2, 64 * 16 * 4); hipMalloc(&d3, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d1, d2, d3, dA, dB, dW);
  std::vector<float> o1(64 * 8), o2(64 * 16), o3(64 * 16);
  hipMemcpy(o1.data(), d1, o1.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(o2.data(), d2, o2.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(o3.data(), d3, o3.size() * 4, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0, bad3 = 0;
  for (int l = 0; l < 64; ++l) {
    const int r = l & 31, h = l >> 5;
    for (int j = 0; j < 8; ++j) {
      const float want = (float)((8 * h + j) * 64 + ((32 + r) & 63));
      if (o1[l * 8 + j] != want) { if (bad1 < 6) printf("tr lane %d j %d got %g want %g\n", l, j, o1[l * 8 + j], want); ++bad1; }
    }
  }
  std::vector<float> C(32 * 32, 0.f), Y(32 * 32, 0.f);
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) C[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 32; ++k) Y[i * 32 + n] += W[i * 32 + k] * C[k * 32 + n];
  for (int l = 0; l < 64; ++l) {
    const int r = l & 31, h = l >> 5;
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (o2[l * 16 + q] != C[row * 32 + r]) { if (bad2 < 6) printf("mfma lane %d q %d got %g want %g\n", l, q, o2[l * 16 + q], C[row * 32 + r]); ++bad2; }
      if (o3[l * 16 + q] != Y[row * 32 + r]) { if (bad3 < 6) printf("chain lane %d q %d got %g want %g\n", l, q, o3[l * 16 + q], Y[row * 32 + r]); ++bad3; }
    }
  }
  printf("tr_read mismatches %d / 512, mfma %d / 1024, acc-as-operand %d / 1024\n", bad1, bad2, bad3);
  printf(bad1 + bad2 + bad3 == 0 ? "F16_PROBE_OK\n" : "F16_PROBE_FAIL\n");
  return bad1 + bad2 + bad3 == 0 ? 0 : 1;
}
