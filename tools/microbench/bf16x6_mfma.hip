// Micro-benchmark for DESIGN §9 item 2 (not part of libmavahip.so): matrix-pipe throughput of an f32-accurate product
// built from 6 bf16 MFMAs (v_mfma_f32_32x32x16_bf16, operands split hi/mid/lo on the VALU) against the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32) the train kernels use today.  One wave per SIMD (256 threads per block, one block per CU),
// like ppo_train_kernel; B operands (activations) come from LDS as f32 and are split on the fly, A operands (weights) are
// split once and stay in registers.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x6_mfma tools/microbench/bf16x6_mfma.hip && ./bf16x6_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
  }
}

constexpr int KB = 8;      // 16-input blocks per tile pass (K = 128, one hidden layer)
constexpr int LDB = 132;   // row stride of the B tile in LDS (floats)

// MODE 0: exact f32 MFMA; MODE 1: 6 bf16 MFMAs with on-the-fly split of B; MODE 2: 6 bf16 MFMAs, B pre-split (MFMA rate only);
// MODE 3: like 1 with the six products alternating between two accumulators (no back-to-back dependent MFMAs)
template <int MODE>
__global__ __launch_bounds__(256, 1) void bench_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                       int iters, unsigned long long* __restrict__ cyc) {
  __shared__ __attribute__((aligned(16))) float BT[32 * LDB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
  for (int i = threadIdx.x; i < 32 * 128; i += 256) BT[(i >> 7) * LDB + (i & 127)] = b[i];
  __syncthreads();
  // A: column block w of a (128 x 128) weight matrix; lane (i = j, half h) holds A[k = 16 kb + 8h + s][32w + j]
  float af[KB][8];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int s = 0; s < 8; ++s) af[kb][s] = a[(16 * kb + 8 * h + s) * 128 + 32 * w + j];
  bf16x8 ah[KB], am[KB], al[KB];
  if (MODE != 0) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) split3(af[kb], ah[kb], am[kb], al[kb]);
  }
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
  const float* xb = BT + j * LDB + 8 * h;
  const unsigned long long c0 = __builtin_readcyclecounter();   // shader-clock cycles (s_memtime)
  const unsigned long long w0 = wall_clock64();                  // constant 100 MHz
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float x[8];
      const float4 v0 = *reinterpret_cast<const float4*>(xb + 16 * kb), v1 = *reinterpret_cast<const float4*>(xb + 16 * kb + 4);
      x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
      if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][s], x[s], acc, 0, 0, 0);
      } else {
        bf16x8 xh, xm, xl;
        if (MODE != 2 || it == 0) split3(x, xh, xm, xl);
        if (MODE == 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kb], xh, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xl, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], xm, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], xh, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xm, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xh, acc2, 0, 0, 0);
          continue;
        }
        // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kb], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], xm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], xh, acc, 0, 0, 0);
      }
    }
  }
  // C layout: col = lane & 31 (row j of B), row = (r & 3) + 8 (r >> 2) + 4 h (feature within the block)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cyc[0] = __builtin_readcyclecounter() - c0;
    cyc[1] = wall_clock64() - w0;
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * w + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
  }
}

template <int MODE>
double run(const float* a, const float* b, float* out, int iters, const char* name, const std::vector<double>& ref, int blocks) {
  static unsigned long long* cyc = nullptr;
  if (!cyc) CHECK(hipMalloc(&cyc, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, a, b, out, 1, cyc);
  CHECK(hipDeviceSynchronize());
  std::vector<float> got(128 * 32);
  CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
  double err = 0, rms = 0;
  for (size_t i = 0; i < got.size(); ++i) { err = fmax(err, fabs(got[i] - ref[i])); rms += ref[i] * ref[i]; }
  rms = sqrt(rms / got.size());
  hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, a, b, out, iters, cyc);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, a, b, out, iters, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 128 * 128 * 32 * (double)iters * blocks;  // algorithmic f32 FLOPs of the product
  unsigned long long hc[2];
  CHECK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
  const double n_mfma = (double)iters * KB * (MODE == 0 ? 8 : 6);
  printf("%-34s %8.3f ms  %7.1f TFLOP/s (f32-equivalent)  max err / rms vs f64: %.2e | %.1f shader cycles per MFMA, shader clock %.2f GHz\n",
         name, ms, flop / ms / 1e9, err / rms, hc[0] / n_mfma, hc[0] / (hc[1] * 10.0));
  return ms;
}

int main() {
  const int blocks = 256, iters = 20000;
  std::vector<float> ha(128 * 128), hb(32 * 128);
  srand(1);
  for (auto& v : ha) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
  std::vector<double> ref(128 * 32);
  for (int f = 0; f < 128; ++f)
    for (int r = 0; r < 32; ++r) {
      double s = 0;
      for (int k = 0; k < 128; ++k) s += (double)ha[k * 128 + f] * (double)hb[r * 128 + k];
      ref[f * 32 + r] = s;
    }
  float *a, *b, *out;
  CHECK(hipMalloc(&a, ha.size() * 4)); CHECK(hipMalloc(&b, hb.size() * 4)); CHECK(hipMalloc(&out, 128 * 32 * 4));
  CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  const double t0 = run<0>(a, b, out, iters, "f32 MFMA 32x32x2", ref, blocks);
  const double t1 = run<1>(a, b, out, iters, "6 x bf16 MFMA, B split on the fly", ref, blocks);
  const double t2 = run<2>(a, b, out, iters, "6 x bf16 MFMA, operands pre-split", ref, blocks);
  const double t3 = run<3>(a, b, out, iters, "6 x bf16 MFMA, split, 2 accumulators", ref, blocks);
  printf("speed-up over f32 MFMA: %.2fx with the on-the-fly split, %.2fx matrix pipe only, %.2fx with two accumulators\n", t0 / t1,
         t0 / t2, t0 / t3);
  return 0;
}
