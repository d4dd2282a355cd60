// Layer kernels of the GENERAL network path: torsos the fused kernels do not instantiate - MLPTorso with any layer sizes,
// activation relu | tanh, use_layer_norm (mava/networks.py:39-58) and CNNTorso (mava/networks.py:61-85: nn.Conv with
// 'SAME' padding -> [LayerNorm(use_scale=False)] -> activation, then the last three axes collapsed).  The matrix products
// of that path are the recurrent path's T32 dense / X^T Y kernels (rec_dense.hip, rec_dense_h2.hip); this file adds what
// sits between them.  All matrices are T32 (rec_dense.hip): element (row, f) at ((row / 32) * N + f) * 32 + row % 32.
//
//  * norm_act:      y = act(LayerNorm(x) + bias) or act(x); flax LayerNorm over the LAST axis, eps 1e-6, no scale, learned
//                   bias; saves xhat and 1/sigma for the backward pass.  A lane owns a row: every access is a coalesced
//                   128-byte run of 32 rows of one feature.
//  * norm_act_bwd:  dz = dy * act'(y);  dx = rstd (dz - mean(dz) - xhat mean(dz xhat)); dz is also written (the bias
//                   gradient is its column sum).
//  * colsum:        per-block partial column sums (fixed-order slabs, like the X^T Y kernels).
//  * im2col / col2im, (rows*P x C) <-> (rows x P*C) reshapes: one thread per destination element, gather form (col2im
//    sums the <= k*k patch entries an input pixel appears in: no atomics, fixed order).
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-6f;  // flax.linen.LayerNorm default epsilon

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == 1) return fmaxf(v, 0.0f);
  if (act == 2) return tanhf(v);
  return v;
}
__device__ __forceinline__ float act_bwd(float y, int act) {  // derivative in terms of the OUTPUT
  if (act == 1) return y > 0.0f ? 1.0f : 0.0f;
  if (act == 2) return 1.0f - y * y;
  return 1.0f;
}

__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ x, int N, long rows, int use_ln,
                                                       const float* __restrict__ ln_bias, int act, float* __restrict__ y,
                                                       float* __restrict__ xhat, float* __restrict__ rstd) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const long base = (row >> 5) * N * 32 + (row & 31);
  if (!use_ln) {
    for (int f = 0; f < N; ++f) y[base + (long)f * 32] = act_fwd(x[base + (long)f * 32], act);
    return;
  }
  float s = 0.0f;
  for (int f = 0; f < N; ++f) s += x[base + (long)f * 32];
  const float mean = s / (float)N;
  float v = 0.0f;
  for (int f = 0; f < N; ++f) {
    const float d = x[base + (long)f * 32] - mean;
    v = fmaf(d, d, v);
  }
  const float r = rsqrtf(v / (float)N + LN_EPS);
  rstd[row] = r;
  for (int f = 0; f < N; ++f) {
    const float xh = (x[base + (long)f * 32] - mean) * r;
    xhat[base + (long)f * 32] = xh;
    y[base + (long)f * 32] = act_fwd(xh + ln_bias[f], act);
  }
}

__global__ __launch_bounds__(256) void norm_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int N,
                                                           long rows, int use_ln, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, int act, float* __restrict__ dz,
                                                           float* __restrict__ dx) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const long base = (row >> 5) * N * 32 + (row & 31);
  float s1 = 0.0f, s2 = 0.0f;
  for (int f = 0; f < N; ++f) {
    const long i = base + (long)f * 32;
    const float g = dy[i] * act_bwd(y[i], act);
    dz[i] = g;
    if (use_ln) {
      s1 += g;
      s2 = fmaf(g, xhat[i], s2);
    }
  }
  if (!use_ln) {
    if (dx != dz)
      for (int f = 0; f < N; ++f) dx[base + (long)f * 32] = dz[base + (long)f * 32];
    return;
  }
  const float m1 = s1 / (float)N, m2 = s2 / (float)N, r = rstd[row];
  for (int f = 0; f < N; ++f) {
    const long i = base + (long)f * 32;
    dx[i] = r * (dz[i] - m1 - xhat[i] * m2);
  }
}

// slab[b][f] = sum over the tiles of block b of the 32 rows of feature f
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ y, int N, long rows, float scale,
                                                     float* __restrict__ slab, long slab_stride) {
  const long ntiles = rows / 32;
  for (int f = threadIdx.x; f < N; f += 256) {
    float s = 0.0f;
    for (long it = blockIdx.x; it < ntiles; it += gridDim.x) {
      const float4* p = reinterpret_cast<const float4*>(y + (it * N + f) * 32);
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = p[q];
        t += (v.x + v.y) + (v.z + v.w);
      }
      s += t;
    }
    slab[(long)blockIdx.x * slab_stride + f] = s * scale;
  }
}

struct ConvGeo {
  int Hin, Win, C, k, stride, Hout, Wout, pad_h, pad_w;
  int src_flat;  // 1: source is (samples x Hin*Win*C) with feature (iy*Win + ix)*C + c; 0: (samples*Hin*Win x C)
};

__device__ __forceinline__ long t32(long row, long f, long N) { return ((row >> 5) * N + f) * 32 + (row & 31); }

__device__ __forceinline__ long src_index(const ConvGeo& g, long s, int iy, int ix, int c) {
  if (g.src_flat) return t32(s, ((long)iy * g.Win + ix) * g.C + c, (long)g.Hin * g.Win * g.C);
  return t32((s * g.Hin + iy) * g.Win + ix, c, g.C);
}

// dst: (samples*Hout*Wout x k*k*C), feature (ky*k + kx)*C + c
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ src, ConvGeo g, long samples, float* __restrict__ dst) {
  const long KK = (long)g.k * g.k * g.C, P = (long)g.Hout * g.Wout;
  const long total = samples * P * KK;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // destination in T32 order: i -> (tile, feature, lane)
    const long tile = i / (KK * 32);
    const long rem = i - tile * KK * 32;
    const long f = rem >> 5;
    const long row = tile * 32 + (rem & 31);
    const long s = row / P, pos = row - s * P;
    const int oy = (int)(pos / g.Wout), ox = (int)(pos - (long)oy * g.Wout);
    const int c = (int)(f % g.C), kk = (int)(f / g.C), ky = kk / g.k, kx = kk - ky * g.k;
    const int iy = oy * g.stride + ky - g.pad_h, ix = ox * g.stride + kx - g.pad_w;
    float v = 0.0f;
    if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) v = src[src_index(g, s, iy, ix, c)];
    dst[i] = v;
  }
}

// dsrc (layout of the im2col SOURCE) = sum of the patch entries each pixel appears in
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, ConvGeo g, long samples, float* __restrict__ dsrc) {
  const long KK = (long)g.k * g.k * g.C, P = (long)g.Hout * g.Wout;
  const long total = samples * g.Hin * g.Win * g.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long s;
    int iy, ix, c;
    if (g.src_flat) {
      const long Nf = (long)g.Hin * g.Win * g.C;
      const long tile = i / (Nf * 32), rem = i - tile * Nf * 32;
      const long f = rem >> 5;
      s = tile * 32 + (rem & 31);
      c = (int)(f % g.C);
      const long p = f / g.C;
      iy = (int)(p / g.Win);
      ix = (int)(p - (long)iy * g.Win);
    } else {
      const long tile = i / ((long)g.C * 32), rem = i - tile * g.C * 32;
      c = (int)(rem >> 5);
      const long row = tile * 32 + (rem & 31);
      s = row / ((long)g.Hin * g.Win);
      const long p = row - s * g.Hin * g.Win;
      iy = (int)(p / g.Win);
      ix = (int)(p - (long)iy * g.Win);
    }
    float acc = 0.0f;
    for (int ky = 0; ky < g.k; ++ky) {
      const int ty = iy + g.pad_h - ky;
      if (ty < 0 || ty % g.stride) continue;
      const int oy = ty / g.stride;
      if (oy >= g.Hout) continue;
      for (int kx = 0; kx < g.k; ++kx) {
        const int tx = ix + g.pad_w - kx;
        if (tx < 0 || tx % g.stride) continue;
        const int ox = tx / g.stride;
        if (ox >= g.Wout) continue;
        acc += dcol[t32(s * P + (long)oy * g.Wout + ox, ((long)ky * g.k + kx) * g.C + c, KK)];
      }
    }
    dsrc[i] = acc;
  }
}

// to_flat: (samples*P x C) -> (samples x P*C), feature p*C + c; else the inverse
__global__ __launch_bounds__(256) void flatten_kernel(const float* __restrict__ src, long samples, int P, int C, int to_flat,
                                                      float* __restrict__ dst) {
  const long total = samples * P * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    if (to_flat) {
      const long Nf = (long)P * C;
      const long tile = i / (Nf * 32), rem = i - tile * Nf * 32;
      const long f = rem >> 5, s = tile * 32 + (rem & 31);
      dst[i] = src[t32(s * P + f / C, f % C, C)];
    } else {
      const long tile = i / ((long)C * 32), rem = i - tile * C * 32;
      const long c = rem >> 5, row = tile * 32 + (rem & 31);
      const long s = row / P, p = row - s * P;
      dst[i] = src[t32(s, p * C + c, (long)P * C)];
    }
  }
}

int blocks_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}


// ---- GRU cell, one time step at a time, for hidden widths the register-resident scans (rec_gru.hip: 128) do not serve:
// network.hidden_state_dim != 128 (mava/networks.py:222-266 ScannedRNN over flax GRUCell).  The recurrent product h W_h is a
// T32 dense launch per step (rec_dense.hip); these kernels are what sits around it.  A lane owns a row (sequence) of the
// step; feature f of the step's T32 matrices at ((row / 32) * N + f) * 32 + row % 32.
__device__ __forceinline__ bool step_done(const uint8_t* done_t, const int32_t* idx, int E, int A, long row) {
  const long e_local = row / A, a = row - e_local * A;
  const long env = idx ? idx[e_local] : e_local;
  (void)E;
  return done_t[env * A + a] != 0;
}

// hprev = done ? 0 : h   (networks.py:253-257: the flag entering the step resets the carried state)
__global__ __launch_bounds__(256) void gru_mask_kernel(const float* __restrict__ h, const uint8_t* __restrict__ done_t,
                                                       const int32_t* __restrict__ idx, int E, int A, int Hd, long rows,
                                                       float* __restrict__ hprev) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const bool d = step_done(done_t, idx, E, A, row);
  const long base = (row >> 5) * Hd * 32 + (row & 31);
  for (int f = 0; f < Hd; ++f) hprev[base + (long)f * 32] = d ? 0.0f : h[base + (long)f * 32];
}

// flax GRUCell: r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r (gh_n + b_hn)), h' = (1 - z) n + z h.
// saved = [r | z | n | gh_n + b_hn]; hprev_next (optional) = the state entering the next step, masked by its done flag.
__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                        const float* __restrict__ bhn, const float* __restrict__ hprev, int Hd,
                                                        long rows, float* __restrict__ hs, float* __restrict__ saved,
                                                        float* __restrict__ hprev_next, const uint8_t* __restrict__ done_next,
                                                        const int32_t* __restrict__ idx, int E, int A) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const long t3 = (row >> 5) * (3L * Hd) * 32 + (row & 31), t1 = (row >> 5) * (long)Hd * 32 + (row & 31);
  const long t4 = (row >> 5) * (4L * Hd) * 32 + (row & 31);
  const bool dn = (hprev_next != nullptr) && step_done(done_next, idx, E, A, row);
  for (int f = 0; f < Hd; ++f) {
    const float r = 1.0f / (1.0f + expf(-(gi[t3 + (long)f * 32] + gh[t3 + (long)f * 32])));
    const float z = 1.0f / (1.0f + expf(-(gi[t3 + (long)(Hd + f) * 32] + gh[t3 + (long)(Hd + f) * 32])));
    const float hl = gh[t3 + (long)(2 * Hd + f) * 32] + bhn[f];
    const float n = tanhf(gi[t3 + (long)(2 * Hd + f) * 32] + r * hl);
    const float hp = hprev[t1 + (long)f * 32];
    const float hn = (1.0f - z) * n + z * hp;
    hs[t1 + (long)f * 32] = hn;
    if (saved != nullptr) {
      saved[t4 + (long)f * 32] = r;
      saved[t4 + (long)(Hd + f) * 32] = z;
      saved[t4 + (long)(2 * Hd + f) * 32] = n;
      saved[t4 + (long)(3 * Hd + f) * 32] = hl;
    }
    if (hprev_next != nullptr) hprev_next[t1 + (long)f * 32] = dn ? 0.0f : hn;
  }
}

// One step of BPTT through the cell.  dh = dh_out + (the gradient carried from step t + 1: acc_next + dhp_next unless that
// step's flag cut the chain); writes dgi (3 Hd), dgh (3 Hd: r and z thirds equal dgi's) and dhp = dh z (the direct path).
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ saved, const float* __restrict__ hprev,
                                                            const float* __restrict__ dh_out, const float* __restrict__ acc_next,
                                                            const float* __restrict__ dhp_next,
                                                            const uint8_t* __restrict__ done_next, const int32_t* __restrict__ idx,
                                                            int E, int A, int Hd, long rows, float* __restrict__ dgi,
                                                            float* __restrict__ dgh, float* __restrict__ dhp) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const long t3 = (row >> 5) * (3L * Hd) * 32 + (row & 31), t1 = (row >> 5) * (long)Hd * 32 + (row & 31);
  const long t4 = (row >> 5) * (4L * Hd) * 32 + (row & 31);
  const bool carry = (acc_next != nullptr) && !step_done(done_next, idx, E, A, row);
  for (int f = 0; f < Hd; ++f) {
    const float rr = saved[t4 + (long)f * 32], zz = saved[t4 + (long)(Hd + f) * 32];
    const float nn = saved[t4 + (long)(2 * Hd + f) * 32], hl = saved[t4 + (long)(3 * Hd + f) * 32];
    const float hp = hprev[t1 + (long)f * 32];
    const float dh = dh_out[t1 + (long)f * 32] + (carry ? (acc_next[t1 + (long)f * 32] + dhp_next[t1 + (long)f * 32]) : 0.0f);
    const float dn = dh * (1.0f - zz);
    const float dz = dh * (hp - nn);
    const float dn_pre = dn * (1.0f - nn * nn);
    const float dr = dn_pre * hl;
    const float dghn = dn_pre * rr;
    const float dz_pre = dz * zz * (1.0f - zz);
    const float dr_pre = dr * rr * (1.0f - rr);
    dgi[t3 + (long)f * 32] = dr_pre;
    dgi[t3 + (long)(Hd + f) * 32] = dz_pre;
    dgi[t3 + (long)(2 * Hd + f) * 32] = dn_pre;
    dgh[t3 + (long)f * 32] = dr_pre;
    dgh[t3 + (long)(Hd + f) * 32] = dz_pre;
    dgh[t3 + (long)(2 * Hd + f) * 32] = dghn;
    dhp[t1 + (long)f * 32] = dh * zz;
  }
}

}  // namespace

extern "C" int mava_t32_norm_act_f32(const float* x, int N, long rows, int use_layer_norm, const float* ln_bias, int act,
                                     float* y, float* xhat, float* rstd, hipStream_t s) {
  MAVA_ARG_CHECK(N >= 1 && rows >= 0 && rows % 32 == 0 && act >= 0 && act <= 2, 0, "mava_t32_norm_act_f32: N=%d rows=%ld act=%d", N,
                 rows, act);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(x && y && (!use_layer_norm || (ln_bias && xhat && rstd)), 1, "mava_t32_norm_act_f32: null pointer argument");
  hipLaunchKernelGGL(norm_act_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, N, rows, use_layer_norm, ln_bias, act, y,
                     xhat, rstd);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_norm_act_bwd_f32(const float* dy, const float* y, int N, long rows, int use_layer_norm, const float* xhat,
                                         const float* rstd, int act, float* dz, float* dx, hipStream_t s) {
  MAVA_ARG_CHECK(N >= 1 && rows >= 0 && rows % 32 == 0 && act >= 0 && act <= 2, 0, "mava_t32_norm_act_bwd_f32: N=%d rows=%ld", N, rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(dy && y && dz && dx && (!use_layer_norm || (xhat && rstd)), 1, "mava_t32_norm_act_bwd_f32: null pointer argument");
  hipLaunchKernelGGL(norm_act_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, dy, y, N, rows, use_layer_norm, xhat,
                     rstd, act, dz, dx);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_colsum_f32(const float* y, int N, long rows, float scale, float* slab, long slab_stride, int n_slab,
                                   hipStream_t s) {
  MAVA_ARG_CHECK(N >= 1 && rows >= 32 && rows % 32 == 0 && n_slab >= 1 && slab_stride >= N, 0, "mava_t32_colsum_f32: bad shape");
  MAVA_ARG_CHECK(y && slab, 1, "mava_t32_colsum_f32: null pointer argument");
  hipLaunchKernelGGL(colsum_kernel, dim3(n_slab), dim3(256), 0, s, y, N, rows, scale, slab, slab_stride);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

static int conv_geo(ConvGeo& g, int Hin, int Win, int C, int k, int stride, int src_flat) {
  MAVA_ARG_CHECK(Hin >= 1 && Win >= 1 && C >= 1 && k >= 1 && stride >= 1, 0, "conv: bad geometry");
  g.Hin = Hin; g.Win = Win; g.C = C; g.k = k; g.stride = stride; g.src_flat = src_flat;
  // flax nn.Conv padding='SAME': out = ceil(in / stride), total padding split low = total / 2
  g.Hout = (Hin + stride - 1) / stride;
  g.Wout = (Win + stride - 1) / stride;
  const int ph = (g.Hout - 1) * stride + k - Hin, pw = (g.Wout - 1) * stride + k - Win;
  g.pad_h = (ph > 0 ? ph : 0) / 2;
  g.pad_w = (pw > 0 ? pw : 0) / 2;
  return MAVA_OK;
}

extern "C" int mava_t32_im2col_f32(const float* src, int src_flat, long samples, int Hin, int Win, int C, int k, int stride,
                                   float* dst, hipStream_t s) {
  ConvGeo g;
  const int rc = conv_geo(g, Hin, Win, C, k, stride, src_flat);
  if (rc != MAVA_OK) return rc;
  const long rows_out = samples * g.Hout * g.Wout;
  MAVA_ARG_CHECK(samples >= 0 && samples % 32 == 0 && rows_out % 32 == 0, 1, "mava_t32_im2col_f32: samples=%ld must be a multiple of 32", samples);
  if (samples == 0) return MAVA_OK;
  MAVA_ARG_CHECK(src && dst, 2, "mava_t32_im2col_f32: null pointer argument");
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks_for(rows_out * k * k * C)), dim3(256), 0, s, src, g, samples, dst);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_col2im_f32(const float* dcol, int src_flat, long samples, int Hin, int Win, int C, int k, int stride,
                                   float* dsrc, hipStream_t s) {
  ConvGeo g;
  const int rc = conv_geo(g, Hin, Win, C, k, stride, src_flat);
  if (rc != MAVA_OK) return rc;
  MAVA_ARG_CHECK(samples >= 0 && samples % 32 == 0, 1, "mava_t32_col2im_f32: samples=%ld must be a multiple of 32", samples);
  if (samples == 0) return MAVA_OK;
  MAVA_ARG_CHECK(dcol && dsrc, 2, "mava_t32_col2im_f32: null pointer argument");
  hipLaunchKernelGGL(col2im_kernel, dim3(blocks_for(samples * Hin * Win * C)), dim3(256), 0, s, dcol, g, samples, dsrc);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_flatten_f32(const float* src, long samples, int P, int C, int to_flat, float* dst, hipStream_t s) {
  MAVA_ARG_CHECK(samples >= 0 && samples % 32 == 0 && P >= 1 && C >= 1, 0, "mava_t32_flatten_f32: samples=%ld P=%d C=%d", samples, P, C);
  if (samples == 0) return MAVA_OK;
  MAVA_ARG_CHECK(src && dst, 1, "mava_t32_flatten_f32: null pointer argument");
  hipLaunchKernelGGL(flatten_kernel, dim3(blocks_for(samples * P * C)), dim3(256), 0, s, src, samples, P, C, to_flat, dst);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_gru_mask_f32(const float* h, const uint8_t* done_t, const int32_t* idx, int E, int A, int Hd, long rows,
                                     float* hprev, hipStream_t s) {
  MAVA_ARG_CHECK(Hd >= 1 && rows >= 0 && rows % 32 == 0 && E >= 1 && A >= 1, 0, "mava_t32_gru_mask_f32: Hd=%d rows=%ld", Hd, rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(h && done_t && hprev, 1, "mava_t32_gru_mask_f32: null pointer argument");
  hipLaunchKernelGGL(gru_mask_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, h, done_t, idx, E, A, Hd, rows, hprev);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_gru_gates_f32(const float* gi, const float* gh, const float* bhn, const float* hprev, int Hd, long rows,
                                      float* hs, float* saved, float* hprev_next, const uint8_t* done_next, const int32_t* idx,
                                      int E, int A, hipStream_t s) {
  MAVA_ARG_CHECK(Hd >= 1 && rows >= 0 && rows % 32 == 0 && E >= 1 && A >= 1, 0, "mava_t32_gru_gates_f32: Hd=%d rows=%ld", Hd, rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(gi && gh && bhn && hprev && hs && (hprev_next == nullptr || done_next != nullptr), 1,
                 "mava_t32_gru_gates_f32: null pointer argument");
  hipLaunchKernelGGL(gru_gates_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, gi, gh, bhn, hprev, Hd, rows, hs, saved,
                     hprev_next, done_next, idx, E, A);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}

extern "C" int mava_t32_gru_gates_bwd_f32(const float* saved, const float* hprev, const float* dh_out, const float* acc_next,
                                          const float* dhp_next, const uint8_t* done_next, const int32_t* idx, int E, int A,
                                          int Hd, long rows, float* dgi, float* dgh, float* dhp, hipStream_t s) {
  MAVA_ARG_CHECK(Hd >= 1 && rows >= 0 && rows % 32 == 0 && E >= 1 && A >= 1, 0, "mava_t32_gru_gates_bwd_f32: Hd=%d rows=%ld", Hd, rows);
  if (rows == 0) return MAVA_OK;
  MAVA_ARG_CHECK(saved && hprev && dh_out && dgi && dgh && dhp && (acc_next == nullptr || (dhp_next && done_next)), 1,
                 "mava_t32_gru_gates_bwd_f32: null pointer argument");
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, saved, hprev, dh_out, acc_next,
                     dhp_next, done_next, idx, E, A, Hd, rows, dgi, dgh, dhp);
  MAVA_LAUNCH_CHECK();
  return MAVA_OK;
}
