"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

Restatement of the reference's configurable networks in torch float64 (autograd for the gradients), written from
mava/networks.py:39-58 (MLPTorso: Dense -> [LayerNorm(use_scale=False)] -> relu | tanh per layer), :61-85 (CNNTorso:
nn.Conv(channel, (k, k), (s, s)) with flax's default 'SAME' padding -> [LayerNorm] -> activation, then
jax.lax.collapse(x, -3)), :88-169 (DiscreteActionHead; ContinuousActionHead with independent_std True / False) and the
PPO losses of mava/systems/ppo/ff_mappo.py:150-218.  The flat parameter layout is mava_amd/generic_networks.py's: per
torso layer [kernel | bias | layer-norm bias], per head [kernel | bias], then the raw log_std vector.
flax LayerNorm: over the last axis, epsilon 1e-6, learned bias, no scale.  flax 'SAME': out = ceil(in / stride),
pad_total = max((out - 1) * stride + k - in, 0), pad_low = pad_total // 2.  torch's conv2d is an INDEPENDENT implementation
of the convolution (the product path uses im2col + matrix products).  Parity unpinned against Mava itself (see
ppo_oracle.py header); the third-party pieces are pinned through torch.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .rec_oracle import F32_MIN, t_tanh_normal_log_prob

LN_EPS = 1e-6


def spec_mlp(din, layer_sizes, heads, activation="relu", layer_norm=False, raw_tail=0):
    return dict(kind="mlp", din=din, layer_sizes=list(layer_sizes), heads=list(heads), act=activation, ln=layer_norm, raw_tail=raw_tail)


def spec_cnn(obs_shape, channels, kernels, strides, heads, activation="relu", layer_norm=False, raw_tail=0):
    return dict(kind="cnn", obs_shape=tuple(obs_shape), din=int(np.prod(obs_shape)), channels=list(channels), kernels=list(kernels),
                strides=list(strides), heads=list(heads), act=activation, ln=layer_norm, raw_tail=raw_tail)


def _act(x, name):
    return torch.relu(x) if name == "relu" else torch.tanh(x)


def _ln(x, bias):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + LN_EPS) + bias


def param_count(spec) -> int:
    n, feat = 0, None
    if spec["kind"] == "mlp":
        k = spec["din"]
        for s in spec["layer_sizes"]:
            n += k * s + s + (s if spec["ln"] else 0)
            k = s
        feat = k
    else:
        H, W, C = spec["obs_shape"]
        for co, kk, st in zip(spec["channels"], spec["kernels"], spec["strides"]):
            n += kk * kk * C * co + co + (co if spec["ln"] else 0)
            H, W, C = -(-H // st), -(-W // st), co
        feat = H * W * C
    for no in spec["heads"]:
        n += feat * no + no
    return n + spec["raw_tail"]


def forward(flat: torch.Tensor, spec, x: torch.Tensor, features: bool = False):
    """x: (rows, din).  Returns the heads' outputs [(rows, n_out)], or the torso's features when `features`."""
    off = 0

    def take(shape):
        nonlocal off
        n = int(np.prod(shape))
        v = flat[off : off + n].reshape(shape)
        off += n
        return v

    if spec["kind"] == "mlp":
        h = x
        k = spec["din"]
        for s in spec["layer_sizes"]:
            w, b = take((k, s)), take((s,))
            h = h @ w + b
            if spec["ln"]:
                h = _ln(h, take((s,)))
            h = _act(h, spec["act"])
            k = s
        feat = h
    else:
        H, W, C = spec["obs_shape"]
        h = x.reshape(-1, H, W, C)
        for co, kk, st in zip(spec["channels"], spec["kernels"], spec["strides"]):
            w, b = take((kk, kk, C, co)), take((co,))
            Ho, Wo = -(-H // st), -(-W // st)
            ph, pw = max((Ho - 1) * st + kk - H, 0), max((Wo - 1) * st + kk - W, 0)
            hp = F.pad(h.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))  # NCHW, (left, right, top, bottom)
            h = F.conv2d(hp, w.permute(3, 2, 0, 1), b, stride=st).permute(0, 2, 3, 1)
            if spec["ln"]:
                h = _ln(h, take((co,)))
            h = _act(h, spec["act"])
            H, W, C = Ho, Wo, co
        feat = h.reshape(h.shape[0], -1)  # jax.lax.collapse(x, -3): (H, W, C) row-major
    if features:
        return feat
    outs = []
    for no in spec["heads"]:
        w, b = take((feat.shape[1], no)), take((no,))
        outs.append(feat @ w + b)
    return outs


def init(rng: np.random.Generator, spec, head_scale=1.0) -> np.ndarray:
    """Random non-degenerate parameters for tests (not the reference's initialisers)."""
    n = param_count(spec)
    flat = rng.standard_normal(n) * 0.1
    return flat


def np_forward(flat, spec, x):
    with torch.no_grad():
        return [o.numpy() for o in forward(torch.tensor(np.asarray(flat, np.float64)), spec, torch.tensor(np.asarray(x, np.float64)))]


def _norm_adv(g):
    return (g - g.mean()) / (g.std(unbiased=False) + 1e-8)


def actor_loss_grad(flat, spec, obs, mask, action, old_log_prob, gae, clip_eps, ent_coef):
    """ff_mappo.py:150-187 on one minibatch (rows = agent rows): returns (total, loss_actor, entropy, flat grad)."""
    f = torch.tensor(np.asarray(flat, np.float64), requires_grad=True)
    tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt)
    logits = forward(f, spec, tt(obs))[0]
    if mask is not None:
        logits = torch.where(tt(mask, torch.bool), logits, torch.full_like(logits, F32_MIN))
    lsm = torch.log_softmax(logits, -1)
    lp = lsm.gather(-1, tt(action, torch.int64)[..., None])[..., 0]
    ratio = torch.exp(lp - tt(old_log_prob))
    g = _norm_adv(tt(gae))
    loss_actor = -torch.minimum(ratio * g, torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps) * g).mean()
    pr = lsm.exp()
    entropy = -(torch.where(pr > 0, pr * lsm, torch.zeros_like(pr))).sum(-1).mean()
    total = loss_actor - ent_coef * entropy
    total.backward()
    return float(total.detach()), float(loss_actor.detach()), float(entropy.detach()), f.grad.numpy()


def actor_loss_grad_continuous(flat, spec, obs, action, old_log_prob, gae, clip_eps, ent_coef, eps, independent_std=True):
    """The same with ContinuousActionHead (networks.py:127-169): heads = [mean] (+ raw_tail = dim) or [mean, log_std]."""
    f = torch.tensor(np.asarray(flat, np.float64), requires_grad=True)
    tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt)
    outs = forward(f, spec, tt(obs))
    mean = outs[0]
    raw = f[-spec["raw_tail"] :] if independent_std else outs[1]
    scale = F.softplus(raw) + 1e-3
    lp = t_tanh_normal_log_prob(tt(action), mean, scale)
    ratio = torch.exp(lp - tt(old_log_prob))
    g = _norm_adv(tt(gae))
    loss_actor = -torch.minimum(ratio * g, torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps) * g).mean()
    x = mean + scale * tt(eps)
    fldj = 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(scale) + fldj).sum(-1).mean()
    total = loss_actor - ent_coef * entropy
    total.backward()
    return float(total.detach()), float(loss_actor.detach()), float(entropy.detach()), f.grad.numpy()


def critic_loss_grad(flat, spec, x, old_value, targets, clip_eps, vf_coef):
    """ff_mappo.py:189-218: returns (total, value_loss, flat grad)."""
    f = torch.tensor(np.asarray(flat, np.float64), requires_grad=True)
    tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt)
    v = forward(f, spec, tt(x))[0][..., 0]
    ov, tg = tt(old_value), tt(targets)
    vc = ov + (v - ov).clamp(-clip_eps, clip_eps)
    value_loss = 0.5 * torch.maximum((v - tg) ** 2, (vc - tg) ** 2).mean()
    total = vf_coef * value_loss
    total.backward()
    return float(total.detach()), float(value_loss.detach()), f.grad.numpy()
