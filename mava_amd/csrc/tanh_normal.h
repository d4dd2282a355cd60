// Device math of the continuous action head: Independent(TanhTransformed(Normal(loc, softplus(log_std) + min_scale)))
// as defined by mava/networks.py:127-169 (ContinuousActionHead) and mava/distributions.py:24-91
// (TanhTransformedDistribution: log_prob with the event clipped to +-0.999 and the tail mass beyond the clip averaged
// over the clipped interval; entropy = Normal entropy + forward log-det-Jacobian at a fresh sample).
// tensorflow_probability is a third-party dependency that is not in /root/reference; its published formulas are
// restated here (Normal.log_prob / log_cdf / log_survival_function / entropy, Tanh.forward_log_det_jacobian,
// special.log_ndtr with its float32 branch points -10 and 5 and a 3-term asymptotic series).  Parity is against
// oracle/tanh_normal.py (float64), which is "parity unpinned" with respect to tfp itself.
#pragma once
#include "common.h"

namespace tn {

constexpr float THRESH = 0.999f;
constexpr float ATANH_THRESH = 3.8002011672502f;   // atanh(0.999)
constexpr float LOG_EPS = -6.907755278982136f;     // log(1 - 0.999)
constexpr float MIN_SCALE = 1e-3f;                 // ContinuousActionHead.min_scale default (networks.py:134); passed per call
constexpr float HALF_LOG_2PI = 0.9189385332046727f;
constexpr float LOG2 = 0.6931471805599453f;
constexpr float RSQRT2 = 0.7071067811865476f;
constexpr uint32_t STREAM_SAMPLE = 0x544e5341u;   // "TNSA": acting noise
constexpr uint32_t STREAM_ENTROPY = 0x544e454eu;  // "TNEN": entropy sample of the loss

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float scale_of(float raw, float min_scale) { return softplus(raw) + min_scale; }

// log of the standard normal CDF
__device__ __forceinline__ float log_ndtr(float z) {
  if (z > 5.0f) return -0.5f * erfcf(z * RSQRT2);  // log(1 - e) ~ -e, e = ndtr(-z)
  if (z > -10.0f) return logf(0.5f * erfcf(-z * RSQRT2));
  const float r2 = 1.0f / (z * z);
  return -0.5f * z * z - logf(-z) - HALF_LOG_2PI + logf(1.0f + r2 * (-1.0f + r2 * (3.0f - 15.0f * r2)));
}

// Tanh.forward_log_det_jacobian(x) = log(1 - tanh(x)^2) = 2 (log 2 - x - softplus(-2x))
__device__ __forceinline__ float tanh_fldj(float x) { return 2.0f * (LOG2 - x - softplus(-2.0f * x)); }

// standard normal from two Philox words (Box-Muller, cosine branch)
__device__ __forceinline__ float normal_of(uint32_t a, uint32_t b) {
  return sqrtf(-2.0f * logf(u01_open(a))) * cosf(6.283185307179586f * u01_open(b));
}
// noise of dimension d of global row gid: words (2(d&1), 2(d&1)+1) of the Philox block with counter d/2
__device__ __forceinline__ float noise(uint32_t gid, uint32_t step, int d, uint32_t stream, uint32_t k0, uint32_t k1) {
  const Philox4 r = philox4x32_10(gid, step, (uint32_t)(d >> 1), stream, k0, k1);
  return (d & 1) ? normal_of(r.z, r.w) : normal_of(r.x, r.y);
}

struct LogProb {
  float lp, dmean, dscale;  // log-density of one action dimension and its derivatives
};

// TanhTransformedDistribution.log_prob of action component y under Normal(mean, scale)
__device__ __forceinline__ LogProb log_prob(float y, float mean, float scale) {
  const float yc = fminf(fmaxf(y, -THRESH), THRESH);
  const float inv = 1.0f / scale;
  LogProb o;
  if (yc <= -THRESH || yc >= THRESH) {
    // left: log_cdf(-atanh(th)) ; right: log_survival_function(atanh(th)) = log_ndtr((mean - atanh(th)) / scale)
    const bool left = yc <= -THRESH;
    const float z = left ? (-ATANH_THRESH - mean) * inv : (mean - ATANH_THRESH) * inv;
    const float l = log_ndtr(z);
    const float g = expf(-0.5f * z * z - HALF_LOG_2PI - l);  // pdf(z) / cdf(z)
    o.lp = l - LOG_EPS;
    o.dmean = left ? -g * inv : g * inv;
    o.dscale = -g * z * inv;
  } else {
    const float x = atanhf(yc);
    const float d = (x - mean) * inv;
    o.lp = -0.5f * d * d - logf(scale) - HALF_LOG_2PI - tanh_fldj(x);
    o.dmean = d * inv;
    o.dscale = (d * d - 1.0f) * inv;
  }
  return o;
}

}  // namespace tn
