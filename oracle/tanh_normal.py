"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

NumPy (float64 by default) restatement of the continuous action head of the reference:

  * mava/networks.py:127-169 ContinuousActionHead: loc = Dense(action_dim)(embedding), scale = softplus(log_std) +
    min_scale with an observation-independent log_std parameter (independent_std=True, the default),
    Independent(TanhTransformedDistribution(Normal(loc, scale)), 1);
  * mava/distributions.py:24-91 TanhTransformedDistribution: log_prob with the event clipped to +-0.999 and the
    Normal mass beyond atanh(0.999) averaged over the clipped interval, mode = tanh(loc), entropy = Normal entropy +
    Tanh.forward_log_det_jacobian(fresh sample);
  * the continuous branch of the PPO actor loss, mava/systems/ppo/ff_mappo.py:160-180 (entropy with a key).

PARITY UNPINNED: the distribution arithmetic lives in tensorflow_probability (a pinned dependency of the reference,
requirements: tensorflow_probability, not vendored in /root/reference and not installable here).  Its published
formulas are restated: Normal.log_prob/log_cdf/log_survival_function/entropy, Tanh fldj = 2 (log 2 - x - softplus(-2x)),
special.log_ndtr.  tests/test_oracle.py checks them against scipy.special.log_ndtr / scipy.stats.norm and the
gradients against central finite differences.  The noise (JAX threefry in the reference) is an input: Philox words
turned into normals by Box-Muller, exactly as in mava_amd/csrc/tanh_normal.h.
"""
from __future__ import annotations

import numpy as np

from . import philox
from . import ppo_oracle as po

THRESH = 0.999
ATANH_THRESH = float(np.arctanh(0.999))
LOG_EPS = float(np.log(1.0 - 0.999))
MIN_SCALE = 1e-3
HALF_LOG_2PI = 0.5 * float(np.log(2.0 * np.pi))
STREAM_SAMPLE = 0x544E5341  # "TNSA"
STREAM_ENTROPY = 0x544E454E  # "TNEN"


def softplus(x):
    x = np.asarray(x)
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x)))


def scale_of(log_std):
    return softplus(log_std) + MIN_SCALE


def log_ndtr(z):
    """log Phi(z) by erfc on the central range and the 3-term asymptotic series below -10 (tfp float32 branch points)."""
    from scipy.special import erfc

    z = np.asarray(z, np.float64)
    out = np.empty_like(z)
    hi = z > 5.0
    lo = z <= -10.0
    mid = ~hi & ~lo
    out[hi] = -0.5 * erfc(z[hi] / np.sqrt(2.0))
    out[mid] = np.log(0.5 * erfc(-z[mid] / np.sqrt(2.0)))
    zl = z[lo]
    r2 = 1.0 / (zl * zl)
    out[lo] = -0.5 * zl * zl - np.log(-zl) - HALF_LOG_2PI + np.log(1.0 + r2 * (-1.0 + r2 * (3.0 - 15.0 * r2)))
    return out


def tanh_fldj(x):
    return 2.0 * (np.log(2.0) - x - softplus(-2.0 * x))


def normal_noise(seed: int, step: int, rows: int, dim: int, stream: int, row_offset: int = 0, gid=None):
    """(rows, dim) standard normals: dimension d of global row g uses words (2(d&1), 2(d&1)+1) of the Philox block
    with counter (g, step, d // 2, stream) - float32 Box-Muller like the kernel."""
    slo, shi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    if gid is None:
        gid = np.arange(rows, dtype=np.uint64)
    gid = ((np.asarray(gid, np.uint64) + np.uint64(row_offset)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out = np.empty((len(gid), dim), np.float32)
    for c in range((dim + 1) // 2):
        w = philox.philox4x32_10(gid, step & 0xFFFFFFFF, c, stream, slo, shi)
        for q in range(2):
            d = 2 * c + q
            if d < dim:
                u1, u2 = philox.u01_open(w[2 * q]), philox.u01_open(w[2 * q + 1])
                out[:, d] = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return out


def log_prob_terms(action, mean, scale):
    """Per-dimension log density and its derivatives with respect to mean and scale."""
    action, mean = np.asarray(action, np.float64), np.asarray(mean, np.float64)
    scale = np.broadcast_to(np.asarray(scale, np.float64), mean.shape)
    yc = np.clip(action, -THRESH, THRESH)
    left, right = yc <= -THRESH, yc >= THRESH
    inv = 1.0 / scale
    z = np.where(left, (-ATANH_THRESH - mean) * inv, (mean - ATANH_THRESH) * inv)
    l = log_ndtr(z)
    g = np.exp(-0.5 * z * z - HALF_LOG_2PI - l)
    x = np.arctanh(np.where(left | right, 0.0, yc))
    d = (x - mean) * inv
    lp_in = -0.5 * d * d - np.log(scale) - HALF_LOG_2PI - tanh_fldj(x)
    edge = left | right
    lp = np.where(edge, l - LOG_EPS, lp_in)
    dmean = np.where(left, -g * inv, np.where(right, g * inv, d * inv))
    dscale = np.where(edge, -g * z * inv, (d * d - 1.0) * inv)
    return lp, dmean, dscale


def log_prob(action, mean, log_std):
    return log_prob_terms(action, mean, scale_of(log_std))[0].sum(-1)


def sample(mean, log_std, eps):
    """(action, log_prob) of Independent(TanhTransformed(Normal)): action = tanh(mean + scale * eps)."""
    a = np.tanh(np.asarray(mean, np.float64) + scale_of(np.asarray(log_std, np.float64)) * eps)
    return a, log_prob(a, mean, log_std)


def split_params(flat, din: int, dim: int):
    n = po.mlp_param_count(din, dim)
    return flat[:n], flat[n : n + dim]


def actor_loss_and_grad(flat, din, dim, obs, action, old_log_prob, gae_mb, clip_eps, ent_coef, eps):
    """ff_mappo.py:160-180 with the continuous head.  flat = [MLP | log_std]; action (R, dim); eps (R, dim) the
    entropy noise.  Returns (total, actor_loss, entropy, flat_grad)."""
    fm, ls = split_params(flat, din, dim)
    p = po.mlp_unflatten(fm, din, dim)
    obs = np.asarray(obs, flat.dtype)
    R = obs.shape[0]
    mean, cache = po.mlp_forward(p, obs, keep=True)
    scale = scale_of(ls)
    lpd, dmean_lp, dscale_lp = log_prob_terms(action, mean, scale)
    lp = lpd.sum(-1)
    ratio = np.exp(lp - old_log_prob)
    adv = po.normalise_advantages(np.asarray(gae_mb, flat.dtype))
    l1 = ratio * adv
    rc = np.clip(ratio, 1.0 - clip_eps, 1.0 + clip_eps)
    l2 = rc * adv
    loss_actor = -np.minimum(l1, l2).mean()
    xs = mean + scale * eps
    ent_rows = (0.5 + HALF_LOG_2PI + np.log(scale) + tanh_fldj(xs)).sum(-1)
    entropy = ent_rows.mean()
    total = loss_actor - ent_coef * entropy

    inside = (ratio >= 1.0 - clip_eps) & (ratio <= 1.0 + clip_eps)
    g1 = np.where(l1 < l2, 1.0, np.where(l1 == l2, 0.5, 0.0))
    g2 = 1.0 - g1
    dlp = (-(g1 * adv + g2 * adv * inside) / R * ratio)[:, None]
    th = np.tanh(xs)
    ec = ent_coef / R
    dmean = dlp * dmean_lp + ec * 2.0 * th
    dscale = dlp * dscale_lp - ec * (1.0 / scale - 2.0 * th * eps)
    grads = po.mlp_backward(p, cache, dmean)
    dls = dscale.sum(0) * sigmoid(ls)
    return total, loss_actor, entropy, np.concatenate([po.mlp_flatten(grads), dls.astype(flat.dtype)])
