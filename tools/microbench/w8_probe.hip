// Ground-truth probe for the operand maps of the 8-wave f16x2 gradient kernel (csrc/ppo_train_w8.hip); run once on a gfx950:
//   1. v_mfma_f32_16x16x32_f16: lane (i = l & 15, kg = l >> 4) holds A[i][8 kg + j], B[8 kg + j][i], j = 0..7; C[4 kg + r][i], r = 0..3
//   2. v_mfma_f32_16x16x16_f16: A[i][4 kg + j], B[4 kg + j][i], j = 0..3; same C map
//   3. an accumulator tile of (1) as the B operand of (2): B[k = 4 kg + r][n = i] IS register r of lane (i, kg)
//   4. the swizzled image of the kernel ([rows][128 halves], 256-byte rows, 16-byte chunk c of row r stored at chunk
//      c ^ sw(r), sw(r) = ((r & 3) << 1) | (9 * ((r >> 3) & 1))): the row read of the 16x16x32 B operand (lane (n, kg), step s:
//      row 16 nt + n, halves 32 s + 8 kg .. + 7 = chunk 4 s + kg) and the hardware-transposed read of the 16x16x32 A operand
//      (lane (m, kg): column 16 t + m of rows 8 kg .. 8 kg + 7: two ds_read_b64_tr_b16, lane 4q + p of a 16-lane group
//      supplies row 8 kg + q (+ 4), halves 16 t + 4 p .. + 3 = chunk 2 t + (p >> 1), byte 8 (p & 1))
// hipcc --offload-arch=gfx950 -O2 tools/microbench/w8_probe.hip -o /tmp/w8_probe && /tmp/w8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS3(p) ((__attribute__((address_space(3))) s16x4*)(p))
__device__ __host__ inline int sw(int r) { return ((r & 3) << 1) | (((r >> 3) & 1) * 9); }
__device__ __host__ inline int off(int r, int c) { return 256 * r + 16 * (c ^ sw(r)); }  // bytes

__global__ void probe(const float* A32, const float* B32, const float* A16, float* c32, float* c16, float* y16, float* rowr, float* trr) {
  __shared__ __attribute__((aligned(16))) unsigned char img[32 * 256];
  const int l = threadIdx.x, i = l & 15, kg = l >> 4;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A32[i * 32 + 8 * kg + j]; b[j] = (_Float16)B32[(8 * kg + j) * 16 + i]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c32[l * 4 + r] = c[r];
  half4 a4, b4;
  for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)A16[i * 16 + 4 * kg + j]; b4[j] = (_Float16)B32[(4 * kg + j) * 16 + i]; }
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c16[l * 4 + r] = d[r];
  // 3. Y = A16 (16 x 16) . C (16 x 16 from step 1, small integers: exact in f16)
  half4 cb;
  for (int r = 0; r < 4; ++r) cb[r] = (_Float16)c[r];
  f32x4 y = {0, 0, 0, 0};
  y = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, cb, y, 0, 0, 0);
  for (int r = 0; r < 4; ++r) y16[l * 4 + r] = y[r];
  // 4. image: value(row, col) = (row * 5 + col * 3) % 251
  for (int e = l; e < 32 * 128; e += 64) {
    const int row = e >> 7, col = e & 127;
    *reinterpret_cast<_Float16*>(img + off(row, col >> 3) + 2 * (col & 7)) = (_Float16)(float)((row * 5 + col * 3) % 251);
  }
  __syncthreads();
  for (int nt = 0; nt < 2; ++nt)
    for (int s = 0; s < 4; ++s) {
      const half8 v = *reinterpret_cast<const half8*>(img + off(16 * nt + i, 4 * s + kg));
      for (int j = 0; j < 8; ++j) rowr[((nt * 4 + s) * 64 + l) * 8 + j] = (float)v[j];
    }
  const int q = i >> 2, p = i & 3;
  for (int t = 0; t < 8; ++t) {
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(img + off(8 * kg + q, 2 * t + (p >> 1)) + 8 * (p & 1)));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(img + off(8 * kg + 4 + q, 2 * t + (p >> 1)) + 8 * (p & 1)));
    const half8 v = __builtin_bit_cast(half8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    for (int j = 0; j < 8; ++j) trr[(t * 64 + l) * 8 + j] = (float)v[j];
  }
}

int main() {
  std::vector<float> A32(16 * 32), B32(32 * 16), A16(16 * 16);
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) A32[m * 32 + k] = (float)((m * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B32[k * 16 + n] = (float)((k * 2 + n * 7) % 5 - 2);
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 16; ++k) A16[m * 16 + k] = (float)((m + 2 * k) % 3 - 1);
  float *dA, *dB, *dA16, *o1, *o2, *o3, *o4, *o5;
  hipMalloc(&dA, A32.size() * 4); hipMalloc(&dB, B32.size() * 4); hipMalloc(&dA16, A16.size() * 4);
  hipMalloc(&o1, 256 * 4); hipMalloc(&o2, 256 * 4); hipMalloc(&o3, 256 * 4); hipMalloc(&o4, 8 * 64 * 8 * 4); hipMalloc(&o5, 8 * 64 * 8 * 4);
  hipMemcpy(dA, A32.data(), A32.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B32.data(), B32.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dA16, A16.data(), A16.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dA16, o1, o2, o3, o4, o5);
  std::vector<float> c32(256), c16(256), y16(256), rowr(8 * 64 * 8), trr(8 * 64 * 8);
  hipMemcpy(c32.data(), o1, 256 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c16.data(), o2, 256 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(y16.data(), o3, 256 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(rowr.data(), o4, rowr.size() * 4, hipMemcpyDeviceToHost);
  if (hipMemcpy(trr.data(), o5, trr.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("W8_PROBE_FAIL (hip error)\n"); return 1; }
  std::vector<float> C32(256, 0.f), C16(256, 0.f), Y(256, 0.f);
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    for (int k = 0; k < 32; ++k) C32[m * 16 + n] += A32[m * 32 + k] * B32[k * 16 + n];
    for (int k = 0; k < 16; ++k) C16[m * 16 + n] += A16[m * 16 + k] * B32[k * 16 + n];
  }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 16; ++k) Y[m * 16 + n] += A16[m * 16 + k] * C32[k * 16 + n];
  int b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
  for (int l = 0; l < 64; ++l) {
    const int i = l & 15, kg = l >> 4;
    for (int r = 0; r < 4; ++r) {
      b1 += c32[l * 4 + r] != C32[(4 * kg + r) * 16 + i];
      b2 += c16[l * 4 + r] != C16[(4 * kg + r) * 16 + i];
      b3 += y16[l * 4 + r] != Y[(4 * kg + r) * 16 + i];
    }
    for (int nt = 0; nt < 2; ++nt) for (int s = 0; s < 4; ++s) for (int j = 0; j < 8; ++j) {
      const int row = 16 * nt + i, col = 32 * s + 8 * kg + j;
      b4 += rowr[((nt * 4 + s) * 64 + l) * 8 + j] != (float)((row * 5 + col * 3) % 251);
    }
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 8; ++j) {
      const int row = 8 * kg + j, col = 16 * t + i;
      const float want = (float)((row * 5 + col * 3) % 251);
      if (trr[(t * 64 + l) * 8 + j] != want) { if (b5 < 6) printf("tr t %d lane %d j %d got %g want %g\n", t, l, j, trr[(t * 64 + l) * 8 + j], want); ++b5; }
    }
  }
  printf("mfma16x16x32 %d, mfma16x16x16 %d, acc-as-operand %d, row read %d, transposed read %d mismatches\n", b1, b2, b3, b4, b5);
  printf(b1 + b2 + b3 + b4 + b5 == 0 ? "W8_PROBE_OK\n" : "W8_PROBE_FAIL\n");
  return b1 + b2 + b3 + b4 + b5 == 0 ? 0 : 1;
}
