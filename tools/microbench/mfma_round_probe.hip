// How does v_mfma_f32_32x32x16_f16 round?  (1) 16 products that are each below half an ulp of the accumulator but sum to 8 ulps:
// are they added exactly before the one rounding, or aligned to the accumulator and truncated one by one?  (2) signed error
// statistics of one MFMA (16 positive products + a positive accumulator) against the exactly rounded float64 sum.
// (3) f16 subnormal inputs (|x| < 2^-14: the low terms of weights of magnitude ~0.1 all are): kept or flushed to zero?
// hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_round_probe.hip -o /tmp/mfma_round_probe && /tmp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A, const float* B, const float* C, float* out, int n_case) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int c = 0; c < n_case; ++c) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)A[(c * 32 + r) * 16 + 8 * h + j];
      b[j] = (_Float16)B[(c * 16 + 8 * h + j) * 32 + r];
    }
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = C[(c * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 16; ++q) out[(c * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
  }
}

static float h(float x) { return (float)(_Float16)x; }
int main() {
  const int NC = 64;
  std::vector<float> A(NC * 32 * 16), B(NC * 16 * 32), C(NC * 32 * 32), O(NC * 32 * 32);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  // case 0: tiny products against a big accumulator
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = 1.0f;
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = ldexpf(1.0f, -14);  // each product 2^-14
  for (int i = 0; i < 32 * 32; ++i) C[i] = 1024.0f;                                                 // ulp = 2^-13
  // case 1: subnormal f16 A operand (2^-20 and 3 * 2^-24) times 1, zero accumulator
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[512 + i * 16 + k] = (k == 0) ? ldexpf(1.0f, -20) : (k == 1 ? 3.0f * ldexpf(1.0f, -24) : 0.0f);
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[512 + k * 32 + n] = 1.0f;
  for (int i = 0; i < 32 * 32; ++i) C[1024 + i] = 0.0f;
  // cases 2..: random positive operands, accumulator of the size of a running sum
  for (int c = 2; c < NC; ++c) {
    for (int i = 0; i < 32 * 16; ++i) A[c * 512 + i] = h(0.05f + 0.2f * rnd());
    for (int i = 0; i < 16 * 32; ++i) B[c * 512 + i] = h(0.1f + rnd());
    for (int i = 0; i < 32 * 32; ++i) C[c * 1024 + i] = 3.0f + 5.0f * rnd();
  }
  float *dA, *dB, *dC, *dO;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dO, O.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO, NC);
  if (hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
  printf("case 0: 1024 + 16 x 2^-14 (exact: 1024 + 2^-10 = 8 ulps above): got 1024 + %g ulps\n", (O[0] - 1024.0f) / ldexpf(1.0f, -13));
  printf("case 1: (2^-20 + 3 * 2^-24) * 1 with f16 subnormal A operands: got %g x 2^-24 (kept: 19, flushed: 0)\n", O[1024] / ldexpf(1.0f, -24));
  double sum_err = 0.0, sum_abs = 0.0, sum_err_seq = 0.0;
  long n = 0, exact_rne = 0;
  for (int c = 2; c < NC; ++c)
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ex = C[c * 1024 + i * 32 + j];
        float seq = C[c * 1024 + i * 32 + j];
        for (int k = 0; k < 16; ++k) {
          ex += (double)A[c * 512 + i * 16 + k] * (double)B[c * 512 + k * 32 + j];
          seq = fmaf(A[c * 512 + i * 16 + k], B[c * 512 + k * 32 + j], seq);
        }
        const float got = O[c * 1024 + i * 32 + j];
        const double ulp = ldexp(1.0, ilogb(ex) - 23);
        sum_err += (got - ex) / ulp;
        sum_err_seq += (seq - ex) / ulp;
        sum_abs += fabs(got - ex) / ulp;
        exact_rne += ((float)ex == got);
        ++n;
      }
  printf("random positive operands, %ld outputs: mean signed error %+.4f ulp (an fmaf chain: %+.4f), mean |error| %.4f ulp, equal to the correctly rounded sum in %.1f %%\n",
         n, sum_err / n, sum_err_seq / n, sum_abs / n, 100.0 * exact_rne / n);
  return 0;
}
