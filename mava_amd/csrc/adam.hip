// Fused per-network global-norm clip + Adam, in place, one launch (K11 / SURVEY §8 row A10),
// and the deterministic slab reducer that produces the flat gradient it consumes.
//
// Reference wiring: mava/systems/ppo/ff_mappo.py:359-366 builds, per network,
//   optax.chain(optax.clip_by_global_norm(max_grad_norm), optax.adam(lr, eps=1e-5))
// and ff_mappo.py:241-250 applies it; the learning rate is constant or the linear schedule of
// mava/utils/training.py:20-64 evaluated at the optimiser's pre-increment step count.
// optax is not vendored in the reference; its published semantics are restated here
// (parity unpinned, see oracle/ppo_oracle.py):
//   n = sqrt(sum g^2) over the whole network;  g <- g if n < c else (g / n) * c
//   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; t = count + 1
//   p <- p - lr(count) * (m / (1-b1^t)) / (sqrt(v / (1-b2^t)) + eps) ; count <- t
//
// The incoming gradient is the SUM over (update-batch replicas x ranks); grad_scale = 1/(U*D)
// turns it into the pmean of ff_mappo.py:224-238.
//
// MI355X mapping: 77 K parameters = 2.1 MB of traffic, i.e. launch/latency bound.  One launch:
// every block first reduces the whole (L2-resident) gradient of each network in the same fixed
// order, so all blocks hold a bit-identical norm with no grid barrier and no atomics, then
// updates its own slice of p/m/v with 16-byte accesses where alignment allows.
#include "h2_core.h"
#include "ctx.h"

namespace {

constexpr int ADAM_THREADS = 256;
constexpr int MAX_SEG = 8;

struct AdamSegs {
  int off[MAX_SEG + 1];
  float lr[MAX_SEG];
};

__device__ inline double block_sum(double x, double* sh) {
  // wave reduce (64 lanes) then across the 4 waves, fixed order
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = x;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < ADAM_THREADS / 64; ++i) t += sh[i];
  return t;
}

// Fused tail (mava_ppo_finish_f32): `norm_partials` != null - the squared norms come from the per-block partial sums the slab
// reducer left behind (n_partial x 2 doubles: segment 0, segment 1) instead of a pass over g; `ticket` != null - the block
// that finishes LAST (arrival ticket, no grid barrier: a barrier costs more than the kernel boundary it would replace,
// MI355X_MICROARCH.md price list) increments the step counts and, for PACK > 0, re-splits the wide critic's W1 (PACK =
// its 16-input steps) from the parameters this launch has just written into `w1_split` for the next gradient launch.
struct AdamTail {
  const double* norm_partials;
  int n_partial;
  unsigned int* ticket;
  const float* pack_params;  // start of the network whose W1 is re-split (inside p)
  int pack_din;
  uint4* w1_split;
};

template <int PACK>
__global__ __launch_bounds__(ADAM_THREADS) void clip_adam_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, int32_t* __restrict__ count, AdamSegs segs, int n_seg,
    float grad_scale, float max_norm, int decay, int steps_per_update, int num_updates, float b1,
    float b2, float eps, const float* __restrict__ loss_sums, float vf_coef, float ent_coef,
    float* __restrict__ metrics_out, AdamTail tail) {
  __shared__ double sh[ADAM_THREADS / 64];
  __shared__ float s_clip[MAX_SEG];
  __shared__ float s_bc1[MAX_SEG], s_bc2[MAX_SEG];  // 1 - b^t, evaluated in f64 (no f32 cancellation)
  __shared__ int s_last;

  // ---- phase 1: every block computes every segment's norm identically
  for (int sgi = 0; sgi < n_seg; ++sgi) {
    const int lo = segs.off[sgi], hi = segs.off[sgi + 1];
    if (tail.norm_partials != nullptr) {
      // fixed order: thread t adds partials t, t + 256, ...; then the block sum - bit-identical in every block
      double a = 0.0;
      for (int b = threadIdx.x; b < tail.n_partial; b += ADAM_THREADS) a += tail.norm_partials[2 * b + sgi];
      const double tot = block_sum(a, sh);
      if (threadIdx.x == 0) {
        const float nrm = (float)sqrt(tot);
        s_clip[sgi] = (nrm < max_norm) ? -1.0f : nrm;
        const double t = (double)(count[sgi] + 1);
        s_bc1[sgi] = (float)(1.0 - pow((double)b1, t));
        s_bc2[sgi] = (float)(1.0 - pow((double)b2, t));
      }
      continue;
    }
    // 16-byte loads, 8 of them in flight per thread: this phase is a chain of L2 round trips (every block reads the
    // whole gradient), ~10 of them at 77 K parameters instead of the ~40 of one float per load
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int al = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? min(hi, (lo + 3) & ~3) : hi;  // first aligned index
    if (lo + (int)threadIdx.x < al) {
      const float gi = g[lo + threadIdx.x] * grad_scale;  // (al - lo <= 3 unless g itself is unaligned)
      a0 += (double)gi * (double)gi;
    }
    for (int i = lo + ADAM_THREADS + threadIdx.x; i < al; i += ADAM_THREADS) {
      const float gi = g[i] * grad_scale;
      a0 += (double)gi * (double)gi;
    }
    const int n4 = (hi - al) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g + al);
    int i = threadIdx.x;
    for (; i + 7 * ADAM_THREADS < n4; i += 8 * ADAM_THREADS) {
      float4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = g4[i + u * ADAM_THREADS];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float x = q[u].x * grad_scale, y = q[u].y * grad_scale, z = q[u].z * grad_scale, w = q[u].w * grad_scale;
        a0 += (double)x * x;
        a1 += (double)y * y;
        a2 += (double)z * z;
        a3 += (double)w * w;
      }
    }
    for (; i < n4; i += ADAM_THREADS) {
      const float4 q = g4[i];
      const float x = q.x * grad_scale, y = q.y * grad_scale, z = q.z * grad_scale, w = q.w * grad_scale;
      a0 += (double)x * x;
      a1 += (double)y * y;
      a2 += (double)z * z;
      a3 += (double)w * w;
    }
    if (al + 4 * n4 + (int)threadIdx.x < hi) {
      const float gi = g[al + 4 * n4 + threadIdx.x] * grad_scale;
      a0 += (double)gi * (double)gi;
    }
    const double tot = block_sum((a0 + a1) + (a2 + a3), sh);
    if (threadIdx.x == 0) {
      const float nrm = (float)sqrt(tot);
      // optax.clip_by_global_norm: trigger = n < c ; else (g / n) * c
      s_clip[sgi] = (nrm < max_norm) ? -1.0f : nrm;
      const double t = (double)(count[sgi] + 1);
      s_bc1[sgi] = (float)(1.0 - pow((double)b1, t));
      s_bc2[sgi] = (float)(1.0 - pow((double)b2, t));
    }
  }
  __syncthreads();

  // ---- phase 2: slice update
  const int total = segs.off[n_seg];
  const int gid = blockIdx.x * ADAM_THREADS + threadIdx.x;
  const int stride = gridDim.x * ADAM_THREADS;
  for (int i = gid; i < total; i += stride) {
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
      if (k < n_seg && i >= segs.off[k]) sgi = k;
    const int cnt = count[sgi];
    float lr = segs.lr[sgi];
    if (decay) {
      // mava/utils/training.py:36-42: frac = 1 - (count // (epochs*minibatches)) / num_updates
      const float frac = 1.0f - (float)(cnt / steps_per_update) / (float)num_updates;
      lr = lr * frac;
    }
    float gi = g[i] * grad_scale;
    const float nrm = s_clip[sgi];
    if (nrm >= 0.0f) gi = (gi / nrm) * max_norm;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float mhat = mi / s_bc1[sgi];
    const float vhat = vi / s_bc2[sgi];
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr * (mhat / (sqrtf(vhat) + eps));
  }

  // ---- metrics (ff_mappo.py:255-265): loss_sums = [actor_loss, entropy, value_loss] summed
  // over replicas/ranks of per-replica means.
  if (metrics_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    const float actor_loss = loss_sums[0] * grad_scale;
    const float entropy = loss_sums[1] * grad_scale;
    const float value_loss = loss_sums[2] * grad_scale;
    metrics_out[0] = (actor_loss - ent_coef * entropy) + vf_coef * value_loss;  // total_loss
    metrics_out[1] = value_loss;
    metrics_out[2] = actor_loss;
    metrics_out[3] = entropy;
  }

  // ---- last arriver: count increment (every block has read the counts by the time it takes its ticket) + W1 re-split
  if (tail.ticket != nullptr) {
    __threadfence();  // this thread's parameter stores are visible device-wide before the block's arrival
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int t = __hip_atomic_fetch_add(tail.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == gridDim.x - 1) ? 1 : 0;
      if (s_last) {
        __hip_atomic_store(tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        for (int k = 0; k < n_seg; ++k) count[k] += 1;
      }
    }
    __syncthreads();
    if constexpr (PACK > 0) {
      if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other blocks' parameter stores, not this CU's cached lines
        h2::pack_w1_body<PACK>(tail.pack_params, tail.pack_din, tail.w1_split, threadIdx.x, h2::W_SCALE_CRITIC);
      }
    }
  }
}

// Without a ticket the counter increment cannot race with readers because counts are bumped by
// a separate 1-thread kernel on the same stream after the update kernel.
__global__ void bump_counts_kernel(int32_t* count, int n_seg) {
  if (threadIdx.x < n_seg) count[threadIdx.x] += 1;
}

// Column sum over the slabs, shared by both reducers.  A block owns 64 columns; its 4 waves each sum a quarter of
// the slabs (ascending, 8 loads in flight), and the four partial sums are added in a fixed order through LDS - the
// result is a fixed function of the inputs (bitwise reproducible run to run), with 4x the waves and twice the loads
// in flight of one thread per column, which spent its time waiting on 64 dependent groups of loads.
__device__ __forceinline__ float slab_column_sum(const float* __restrict__ slab, int n_slab, long slab_stride,
                                                 int col, bool live, float (&part)[4][64]) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int per = (n_slab + 3) >> 2;
  const int b0 = g * per, b1 = min(n_slab, b0 + per);
  float acc = 0.0f;
  if (live) {
    const float* p = slab + (long)b0 * slab_stride + col;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float a[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = p[(long)q * slab_stride];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += a[q];
      p += 8 * slab_stride;
    }
    for (; b < b1; ++b) { acc += *p; p += slab_stride; }
  }
  part[g][c] = acc;
  __syncthreads();
  return ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}

// out[i] = sum_b slab[b][i] for i < n (fixed order).  Optionally accumulates onto the existing out (used when
// several update-batch replicas add into one flat gradient).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab,
                                                          int n_slab, long slab_stride, int n,
                                                          int accumulate,
                                                          float* __restrict__ out) {
  __shared__ float part[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const float acc = slab_column_sum(slab, n_slab, slab_stride, i, i < n, part);
  if (threadIdx.x < 64 && i < n) out[i] = accumulate ? (out[i] + acc) : acc;
}

}  // namespace

extern "C" int mava_clip_adam(float* p, const float* g, float* m, float* v, int32_t* count,
                              const int* seg_off, const float* seg_lr, int n_seg,
                              float grad_scale, float max_norm, int decay, int steps_per_update,
                              int num_updates, float b1, float b2, float eps,
                              const float* loss_sums, float vf_coef, float ent_coef,
                              float* metrics_out, hipStream_t s) {
  MAVA_ARG_CHECK(n_seg >= 1 && n_seg <= MAX_SEG, 0, "mava_clip_adam: n_seg=%d out of [1,%d]",
                 n_seg, MAX_SEG);
  MAVA_ARG_CHECK(p && g && m && v && count && seg_off && seg_lr, 1,
                 "mava_clip_adam: null pointer argument");
  MAVA_ARG_CHECK(metrics_out == nullptr || loss_sums != nullptr, 2,
                 "mava_clip_adam: metrics_out needs loss_sums");
  MAVA_ARG_CHECK(!decay || (steps_per_update > 0 && num_updates > 0), 3,
                 "mava_clip_adam: decay needs steps_per_update>0 and num_updates>0");
  AdamSegs segs;
  for (int i = 0; i <= n_seg; ++i) {
    segs.off[i] = seg_off[i];
    MAVA_ARG_CHECK(i == 0 || seg_off[i] >= seg_off[i - 1], 4,
                   "mava_clip_adam: seg_off must be non-decreasing");
  }
  MAVA_ARG_CHECK(seg_off[0] == 0, 5, "mava_clip_adam: seg_off[0] must be 0");
  for (int i = 0; i < n_seg; ++i) segs.lr[i] = seg_lr[i];
  const int total = seg_off[n_seg];
  if (total == 0) return MAVA_OK;
  int blocks = mava_cdiv(total, ADAM_THREADS * 4);
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(clip_adam_kernel<0>, dim3(blocks), dim3(ADAM_THREADS), 0, s, p, g, m, v, count,
                     segs, n_seg, grad_scale, max_norm, decay, steps_per_update, num_updates, b1,
                     b2, eps, loss_sums, vf_coef, ent_coef, metrics_out, AdamTail{});
  MAVA_LAUNCH_CHECK();
  hipLaunchKernelGGL(bump_counts_kernel, dim3(1), dim3(64), 0, s, count, n_seg);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

namespace {
// Same reduction with the slab row split in two destinations: columns [0, n_main) go to out_main
// and columns [n_main, n_main + n_tail) to out_tail (the loss sums that follow the gradient).
__global__ __launch_bounds__(256) void slab_reduce2_kernel(const float* __restrict__ slab, int n_slab,
                                                           long slab_stride, int n_main, int n_tail,
                                                           int accumulate, float* __restrict__ out_main,
                                                           float* __restrict__ out_tail) {
  __shared__ float part[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = i < n_main + n_tail;
  const float acc = slab_column_sum(slab, n_slab, slab_stride, i, live, part);
  if (threadIdx.x < 64 && live) {
    float* dst = (i < n_main) ? (out_main + i) : (out_tail + (i - n_main));
    *dst = accumulate ? (*dst + acc) : acc;
  }
}

}  // namespace

extern "C" int mava_slab_reduce2_f32(const float* slab, int n_slab, long slab_stride, int n_main,
                                     float* out_main, int n_tail, float* out_tail, int accumulate,
                                     hipStream_t s) {
  MAVA_ARG_CHECK(n_main >= 0 && n_tail >= 0 && n_slab >= 0 && slab_stride >= n_main + n_tail, 0,
                 "mava_slab_reduce2_f32: bad shape");
  if (n_main + n_tail == 0) return MAVA_OK;
  MAVA_ARG_CHECK(slab && (n_main == 0 || out_main) && (n_tail == 0 || out_tail), 1,
                 "mava_slab_reduce2_f32: null pointer argument");
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(mava_cdiv(n_main + n_tail, 64)), dim3(256), 0, s, slab,
                     n_slab, slab_stride, n_main, n_tail, accumulate, out_main, out_tail);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_slab_reduce_f32(const float* slab, int n_slab, long slab_stride, int n,
                                    int accumulate, float* out, hipStream_t s) {
  MAVA_ARG_CHECK(n >= 0 && n_slab >= 0 && slab_stride >= n, 0,
                 "mava_slab_reduce_f32: bad shape n=%d n_slab=%d stride=%ld", n, n_slab,
                 slab_stride);
  if (n == 0) return MAVA_OK;
  MAVA_ARG_CHECK(slab && out, 1, "mava_slab_reduce_f32: null pointer argument");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(mava_cdiv(n, 64)), dim3(256), 0, s, slab, n_slab,
                     slab_stride, n, accumulate, out);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

namespace {
// Both networks' slabs in ONE launch (fused tail): virtual column c of [actor slab (Pa + 2 columns) | critic slab (Pc + 1)]
// goes to g laid out [actor grad Pa | critic grad Pc | actor_loss, entropy, value_loss]; the column sums are the ones
// slab_reduce2_kernel forms (same partition of the slabs, same order: bit-identical), and every block also leaves the
// squared norm of its 64 columns per network - (sum * grad_scale)^2 in double, fixed order - for the Adam launch.
__global__ __launch_bounds__(256) void slab_reduce_pair_kernel(const float* __restrict__ slab_a, long stride_a,
                                                               const float* __restrict__ slab_c, long stride_c, int n_slab, int Pa,
                                                               int Pc, float grad_scale, float* __restrict__ g,
                                                               double* __restrict__ norm_partials) {
  __shared__ float part[4][64];
  __shared__ double sq[2][64];
  const int na = Pa + 2;                  // actor columns (gradient + 2 loss sums)
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = c < na + Pc + 1;
  const bool actor = c < na;
  const float* slab = actor ? slab_a : slab_c;
  const long stride = actor ? stride_a : stride_c;
  const int col = actor ? c : c - na;
  const float acc = slab_column_sum(slab, n_slab, stride, col, live, part);
  if (threadIdx.x < 64) {
    int dst = -1, seg = -1;
    if (live) {
      if (actor) { dst = (col < Pa) ? col : (Pa + Pc + (col - Pa)); seg = (col < Pa) ? 0 : -1; }
      else { dst = (col < Pc) ? (Pa + col) : (Pa + Pc + 2); seg = (col < Pc) ? 1 : -1; }
      g[dst] = acc;
    }
    const float gi = acc * grad_scale;
    const double q = (double)gi * (double)gi;
    sq[0][threadIdx.x] = (seg == 0) ? q : 0.0;
    sq[1][threadIdx.x] = (seg == 1) ? q : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int k = 0; k < 64; ++k) t += sq[threadIdx.x][k];  // fixed order
    norm_partials[2 * blockIdx.x + threadIdx.x] = t;
  }
}

inline int finish_blocks(int Pa, int Pc) { return mava_cdiv(Pa + 2 + Pc + 1, 64); }
}  // namespace

extern "C" size_t mava_ppo_finish_workspace_bytes(int Pa, int Pc) {
  return (size_t)finish_blocks(Pa, Pc) * 2 * sizeof(double) + 64;  // norm partials | arrival ticket
}

extern "C" int mava_ppo_finish_f32(mava_ctx* ctx, const float* slab_a, long stride_a, const float* slab_c, long stride_c,
                                   int n_slab, int Pa, int Pc, float* g, float* p, float* m, float* v, int32_t* count,
                                   float lr_a, float lr_c, float grad_scale, float max_norm, int decay, int steps_per_update,
                                   int num_updates, float b1, float b2, float eps, float vf_coef, float ent_coef,
                                   float* metrics_out, int critic_din, void* workspace, size_t workspace_bytes,
                                   hipStream_t s) {
  // n_slab == 0: `g` already holds [actor grad | critic grad | 3 loss sums] (summed over slabs, replicas and ranks by the
  // caller - the multi-rank path, whose all-reduce sits between the slab sums and Adam): only the Adam launch runs, with the
  // norms taken from g, and still carries the count increment and the W1 re-split (one launch instead of three)
  const bool reduced = n_slab == 0;
  MAVA_ARG_CHECK(Pa >= 1 && Pc >= 1 && n_slab >= 0 && (reduced || (stride_a >= Pa + 2 && stride_c >= Pc + 1)), 0,
                 "mava_ppo_finish_f32: Pa=%d Pc=%d n_slab=%d strides %ld %ld", Pa, Pc, n_slab, stride_a, stride_c);
  MAVA_ARG_CHECK((reduced || (slab_a && slab_c)) && g && p && m && v && count && workspace, 1,
                 "mava_ppo_finish_f32: null pointer argument");
  MAVA_ARG_CHECK(workspace_bytes >= mava_ppo_finish_workspace_bytes(Pa, Pc) && ((uintptr_t)workspace & 7) == 0, 2,
                 "mava_ppo_finish_f32: workspace of %zu bytes (8-byte aligned, zeroed once) required",
                 mava_ppo_finish_workspace_bytes(Pa, Pc));
  MAVA_ARG_CHECK(!decay || (steps_per_update > 0 && num_updates > 0), 3,
                 "mava_ppo_finish_f32: decay needs steps_per_update>0 and num_updates>0");
  const int nblk = finish_blocks(Pa, Pc);
  double* partials = static_cast<double*>(workspace);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(partials + 2 * nblk);
  if (!reduced) {
    hipLaunchKernelGGL(slab_reduce_pair_kernel, dim3(nblk), dim3(256), 0, s, slab_a, stride_a, slab_c, stride_c, n_slab, Pa, Pc,
                       grad_scale, g, partials);
    MAVA_LAUNCH_CHECK();
  }
  AdamSegs segs = {};
  segs.off[0] = 0; segs.off[1] = Pa; segs.off[2] = Pa + Pc;
  segs.lr[0] = lr_a; segs.lr[1] = lr_c;
  AdamTail tail = {reduced ? nullptr : partials, nblk, ticket, nullptr, 0, nullptr};
  // the wide f16x2 critic (96 .. 287 inputs) reads W1 through a pre-split copy: re-split it here, from the parameters this
  // launch writes, and mark the handle's copy fresh - the next critic launch of the handle then skips its own pack launch
  const int steps = (critic_din + 1 + 15) / 16;
  const bool pack = ctx != nullptr && ctx->matmul_mode == 1 && critic_din > 0 && steps >= 7 && steps <= 18 &&
                    mlp_param_count(critic_din, 1) == Pc;
  if (pack) {
    void*& slot = ctx->w1_split[1];
    if (slot == nullptr) MAVA_HIP_CHECK(hipMalloc(&slot, h2::W1_SPLIT_BYTES));
    tail.pack_params = p + Pa;
    tail.pack_din = critic_din;
    tail.w1_split = static_cast<uint4*>(slot);
  }
  int blocks = mava_cdiv(Pa + Pc, ADAM_THREADS * 4);
  if (blocks > 256) blocks = 256;
#define FINISH_ADAM(PK)                                                                                                \
  hipLaunchKernelGGL(clip_adam_kernel<PK>, dim3(blocks), dim3(ADAM_THREADS), 0, s, p, g, m, v, count, segs, 2, grad_scale, \
                     max_norm, decay, steps_per_update, num_updates, b1, b2, eps, g + Pa + Pc, vf_coef, ent_coef,      \
                     metrics_out, tail)
  if (pack && steps <= 12) FINISH_ADAM(12);
  else if (pack) FINISH_ADAM(18);
  else FINISH_ADAM(0);
#undef FINISH_ADAM
  MAVA_LAUNCH_CHECK();
  if (pack) ctx->w1_fresh[1] = 1;
  return MAVA_OK;
}
