"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

CPU restatement (NumPy, float64 by default) of Mava's PPO hot path, written from the reference
source text; every function cites the lines it follows (paths relative to the Mava repo).

PARITY UNPINNED.  The reference cannot be executed in the build container (its first import
fails: `No module named 'chex'`; jax/flax/optax/tfp are absent and there is no network) and its
own tests hold no numeric vectors (test/integration_test.py:35-46 only checks that a float comes
back).  Third-party arithmetic is restated from the libraries' published semantics:
  * flax.linen.Dense:  y = x @ kernel + bias, kernel (in, out)        (mava/networks.py:54,114,205)
  * jax.nn.relu:       max(x, 0), derivative 0 at x <= 0
  * tfd.Categorical:   log_prob = log_softmax(logits)[a];  entropy = -sum p log p with 0*log0 = 0
                                                                   (mava/networks.py:116-124)
  * optax.clip_by_global_norm / optax.adam(eps=1e-5, eps_root=0)      (ff_mappo.py:359-366)
  * jnp.std: population standard deviation (ddof = 0)                 (ff_mappo.py:164)
  * jnp.minimum / jnp.maximum / jnp.clip gradients: ties split evenly (lax.min/max JVP)
A second, independent restatement on torch autograd (oracle/torch_ref.py) cross-checks every
gradient produced here; tests/golden/*.npz are generated from this file
(tests/golden/make_golden.py) and are the only pins this path has.
"""
from __future__ import annotations

from typing import Dict, NamedTuple, Optional, Tuple

import numpy as np

H = 128  # configs/network/mlp.yaml: layer_sizes [128, 128]
F32_MIN = float(np.finfo(np.float32).min)  # jnp.finfo(jnp.float32).min, networks.py:119


# --------------------------------------------------------------------------------------------
# GAE - mava/systems/ppo/ff_mappo.py:112-139 ; recurrent: mava/systems/ppo/rec_mappo.py:177-199
# --------------------------------------------------------------------------------------------
def gae(reward, value, done, last_val, gamma, gae_lambda, last_done=None, dtype=np.float64):
    """reward/value/done: (T, ...) time-major; returns (advantages, targets).

    Feed-forward (last_done is None), ff_mappo.py:117-128:
        delta = reward + gamma * next_value * (1 - done) - value
        gae   = delta + gamma * gae_lambda * (1 - done) * gae
    Recurrent, rec_mappo.py:180-188: the carry also holds next_done, which replaces `done` in
    both lines, is seeded with last_done (:192) and is replaced by the transition's stored done
    (the flag entering the step) after each step.
    Sequential evaluation in `dtype` with separate multiply/add roundings (no FMA).
    """
    reward = np.asarray(reward, dtype)
    value = np.asarray(value, dtype)
    d = np.asarray(done).astype(dtype)
    T = reward.shape[0]
    g = np.zeros_like(np.asarray(last_val, dtype))
    next_value = np.asarray(last_val, dtype)
    gamma = dtype(gamma)
    lam = dtype(gae_lambda)
    one = dtype(1)
    adv = np.empty_like(reward)
    next_done = None if last_done is None else np.asarray(last_done).astype(dtype)
    for t in range(T - 1, -1, -1):
        mask_src = d[t] if next_done is None else next_done
        nd = one - mask_src
        delta = reward[t] + gamma * next_value * nd - value[t]
        g = delta + gamma * lam * nd * g
        adv[t] = g
        next_value = value[t]
        if next_done is not None:
            next_done = d[t]
    return adv, adv + value


# --------------------------------------------------------------------------------------------
# Networks - mava/networks.py:39-58 (MLPTorso), :88-124 (DiscreteActionHead),
#            :172-183 (FeedForwardActor), :186-207 (FeedForwardValueNet)
# flat layout [W1 (din,128) | b1 | W2 (128,128) | b2 | W3 (128,no) | b3]
# --------------------------------------------------------------------------------------------
class MlpParams(NamedTuple):
    W1: np.ndarray
    b1: np.ndarray
    W2: np.ndarray
    b2: np.ndarray
    W3: np.ndarray
    b3: np.ndarray


def mlp_param_count(din: int, no: int) -> int:
    return din * H + H + H * H + H + H * no + no


def mlp_unflatten(flat, din: int, no: int) -> MlpParams:
    flat = np.asarray(flat)
    o = 0
    out = []
    for shape in [(din, H), (H,), (H, H), (H,), (H, no), (no,)]:
        n = int(np.prod(shape))
        out.append(flat[o : o + n].reshape(shape))
        o += n
    assert o == flat.size, (o, flat.size)
    return MlpParams(*out)


def mlp_flatten(p: MlpParams):
    return np.concatenate([np.asarray(a).reshape(-1) for a in p])


def mlp_forward(p: MlpParams, x, keep=False):
    """Dense-ReLU-Dense-ReLU-Dense (networks.py:52-57 + head Dense :114 / :205)."""
    z1 = x @ p.W1 + p.b1
    h1 = np.maximum(z1, 0)
    z2 = h1 @ p.W2 + p.b2
    h2 = np.maximum(z2, 0)
    y = h2 @ p.W3 + p.b3
    if keep:
        return y, (x, z1, h1, z2, h2)
    return y


def mlp_backward(p: MlpParams, cache, dy) -> MlpParams:
    """Gradients of sum(y * dy) w.r.t. the parameters."""
    x, z1, h1, z2, h2 = cache
    dW3 = h2.T @ dy
    db3 = dy.sum(0)
    dh2 = dy @ p.W3.T
    dz2 = dh2 * (z2 > 0)
    dW2 = h1.T @ dz2
    db2 = dz2.sum(0)
    dh1 = dz2 @ p.W2.T
    dz1 = dh1 * (z1 > 0)
    dW1 = x.T @ dz1
    db1 = dz1.sum(0)
    return MlpParams(dW1, db1, dW2, db2, dW3, db3)


def masked_logits(logits, mask):
    """networks.py:116-120: where(action_mask, logits, finfo(f32).min)."""
    if mask is None:
        return logits
    return np.where(np.asarray(mask).astype(bool), logits, np.asarray(F32_MIN, logits.dtype))


def log_softmax(z):
    m = z.max(-1, keepdims=True)
    return z - (m + np.log(np.exp(z - m).sum(-1, keepdims=True)))


def categorical_entropy(logp):
    p = np.exp(logp)
    return -np.where(p > 0, p * logp, 0.0).sum(-1)


def gumbel_argmax(z, u):
    """jax.random.categorical == argmax(logits + Gumbel); u uniform in (0,1); first max wins."""
    g = -np.log(-np.log(u.astype(z.dtype)))
    return np.argmax(z + g, axis=-1).astype(np.int32)


# --------------------------------------------------------------------------------------------
# Losses - mava/systems/ppo/ff_mappo.py:150-180 (_actor_loss_fn), :182-201 (_critic_loss_fn)
# --------------------------------------------------------------------------------------------
def normalise_advantages(gae_mb):
    """ff_mappo.py:164: (gae - gae.mean()) / (gae.std() + 1e-8), over the whole minibatch."""
    return (gae_mb - gae_mb.mean()) / (gae_mb.std() + 1e-8)


def actor_loss_and_grad(flat, din, no, obs, mask, action, old_log_prob, gae_mb, clip_eps, ent_coef, part_of=None):
    """Returns (total_loss, actor_loss, entropy, flat_grad).  Shapes: obs (R, din), mask (R, no),
    action/old_log_prob/gae_mb (R,) with R = minibatch rows * agents, flattened.
    part_of = (R_total, adv_mean, adv_std): the rows are one CHUNK of a minibatch of R_total rows whose advantage
    statistics are given; the returned losses and gradient are this chunk's share of the minibatch means (summing
    the chunks gives the whole minibatch - how the full-launch-shape tests evaluate 10^6 rows in bounded memory)."""
    p = mlp_unflatten(flat, din, no)
    obs = np.asarray(obs, flat.dtype)
    n_rows = obs.shape[0]
    y, cache = mlp_forward(p, obs, keep=True)
    z = masked_logits(y, mask)
    logp_all = log_softmax(z)
    probs = np.exp(logp_all)
    lp = logp_all[np.arange(n_rows), action]
    ratio = np.exp(lp - old_log_prob)
    if part_of is None:
        R = n_rows
        adv = normalise_advantages(np.asarray(gae_mb, flat.dtype))
    else:
        R, a_mean, a_std = part_of
        adv = (np.asarray(gae_mb, flat.dtype) - a_mean) / (a_std + 1e-8)
    l1 = ratio * adv
    rc = np.clip(ratio, 1.0 - clip_eps, 1.0 + clip_eps)
    l2 = rc * adv
    loss_actor = -np.minimum(l1, l2).sum() / R
    ent_rows = categorical_entropy(logp_all)
    entropy = ent_rows.sum() / R
    total = loss_actor - ent_coef * entropy

    # d(-min(l1,l2))/d ratio with even tie split (lax.min) and clip's pass-through inside the range
    inside = (ratio >= 1.0 - clip_eps) & (ratio <= 1.0 + clip_eps)
    g1 = np.where(l1 < l2, 1.0, np.where(l1 == l2, 0.5, 0.0))
    g2 = 1.0 - g1
    dratio = -(g1 * adv + g2 * adv * inside) / R
    dlp = dratio * ratio
    onehot = np.zeros_like(y)
    onehot[np.arange(n_rows), action] = 1.0
    dz = dlp[:, None] * (onehot - probs)
    # entropy: dH/dz_o = -p_o (log p_o + H); total has -ent_coef * mean(H)
    dH = -probs * (np.where(probs > 0, logp_all, 0.0) + ent_rows[:, None])
    dz += (-ent_coef / R) * dH
    if mask is not None:
        dz = np.where(np.asarray(mask).astype(bool), dz, 0.0)
    grads = mlp_backward(p, cache, dz)
    return total, loss_actor, entropy, mlp_flatten(grads)


def critic_loss_and_grad(flat, din, x, old_value, targets, clip_eps, vf_coef, R_total=None):
    """Returns (critic_total_loss, value_loss, flat_grad); x (R, din), old_value/targets (R,).
    R_total: the rows are one chunk of a minibatch of R_total rows (see actor_loss_and_grad.part_of)."""
    p = mlp_unflatten(flat, din, 1)
    x = np.asarray(x, flat.dtype)
    R = x.shape[0] if R_total is None else R_total
    y, cache = mlp_forward(p, x, keep=True)
    v = y[:, 0]
    diff = v - old_value
    vclip = old_value + np.clip(diff, -clip_eps, clip_eps)
    l1 = (v - targets) ** 2
    l2 = (vclip - targets) ** 2
    value_loss = 0.5 * np.maximum(l1, l2).sum() / R
    total = vf_coef * value_loss
    inside = (diff >= -clip_eps) & (diff <= clip_eps)
    g1 = np.where(l1 > l2, 1.0, np.where(l1 == l2, 0.5, 0.0))
    g2 = 1.0 - g1
    dv = vf_coef * 0.5 * (g1 * 2.0 * (v - targets) + g2 * 2.0 * (vclip - targets) * inside) / R
    grads = mlp_backward(p, cache, dv[:, None])
    return total, value_loss, mlp_flatten(grads)


# --------------------------------------------------------------------------------------------
# Optimiser - ff_mappo.py:359-366, :241-250 ; schedule mava/utils/training.py:20-64
# --------------------------------------------------------------------------------------------
def learning_rate(init_lr, count, decay, ppo_epochs, num_minibatches, num_updates):
    if not decay:
        return init_lr
    frac = 1.0 - (count // (ppo_epochs * num_minibatches)) / num_updates  # training.py:36-42
    return init_lr * frac


def clip_adam(p, g, m, v, count, lr, max_norm, b1=0.9, b2=0.999, eps=1e-5, dtype=np.float64):
    """One optax.chain(clip_by_global_norm(max_norm), adam(lr, eps=1e-5)) step on one network.
    `lr` is already evaluated at the pre-increment count.  Returns (p, m, v, count + 1)."""
    p, g, m, v = (np.asarray(a, dtype) for a in (p, g, m, v))
    n = np.sqrt((g * g).sum())
    if not (n < max_norm):
        g = (g / n) * max_norm
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    t = count + 1
    mhat = m / (1 - dtype(b1) ** t)
    vhat = v / (1 - dtype(b2) ** t)
    p = p - lr * (mhat / (np.sqrt(vhat) + eps))
    return p, m, v, t


# --------------------------------------------------------------------------------------------
# Minibatching - ff_mappo.py:268-285 + mava/utils/jax_utils.py:33-49
# --------------------------------------------------------------------------------------------
def minibatch_rows(permutation, num_minibatches: int, mb: int):
    """(T,E,...) -> merge_leading_dims(.,2) -> take(perm) -> reshape(M, -1, ...): rows of the
    flattened (T*E) batch that form minibatch `mb`."""
    B = permutation.shape[0] // num_minibatches
    return permutation[mb * B : (mb + 1) * B]


def orthogonal(rng: np.random.Generator, shape, scale):
    """flax.linen.initializers.orthogonal(scale) restated: QR of a normal matrix, sign-fixed.
    Used only to give test networks reference-like magnitudes (init parity with JAX's PRNG is
    not a goal)."""
    rows, cols = shape
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return scale * q[:rows, :cols]


def init_mlp(rng, din, no, head_scale):
    """networks.py:54 (orthogonal(sqrt 2), zero bias), :114 (0.01) / :205 (1.0)."""
    return MlpParams(
        orthogonal(rng, (din, H), np.sqrt(2.0)),
        np.zeros(H),
        orthogonal(rng, (H, H), np.sqrt(2.0)),
        np.zeros(H),
        orthogonal(rng, (H, no), head_scale),
        np.zeros(no),
    )
