"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

The CPU baseline that bench.py times beside the GPU path (`cpu_baseline`, kind "port"):
Mava's feed-forward PPO update loop (mava/systems/ppo/ff_mappo.py:56-300: rollout -> GAE ->
K epochs x M minibatches of actor/critic SGD with clip + Adam) restated on torch-CPU float32 with
autograd, multi-threaded over the host cores, on the same synthetic RWARE-shaped inputs
(distributions of SURVEY.md §8d, generated with torch.rand).  It is a SUBSTITUTE for "Mava's own
JAX path on the host CPU", which cannot run here (jax/flax/optax are not installed and the
reference does not travel to the GPU box) - BASELINE.md §4.
"""
from __future__ import annotations

import math
import time
from typing import Dict

import torch

from .torch_ref import actor_loss, critic_loss, mlp

H = 128


def _init(din, no, head_scale, gen):
    def orth(r, c, s):
        a = torch.randn(max(r, c), min(r, c), generator=gen)
        q, rr = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(rr))
        q = q.T if r < c else q
        return (s * q[:r, :c]).reshape(-1)

    return torch.cat([orth(din, H, math.sqrt(2)), torch.zeros(H), orth(H, H, math.sqrt(2)), torch.zeros(H),
                      orth(H, no, head_scale), torch.zeros(no)]).requires_grad_(True)


def _observe(E, A, O, nA, gen):
    raw = (torch.rand(E, A, O, generator=gen) < 0.2).float()
    raw[:, :, :2] = torch.randint(0, 10, (E, A, 2), generator=gen).float()
    av = torch.cat([torch.eye(A).expand(E, A, A), raw], -1)
    gs = raw.reshape(E, 1, A * O).expand(E, A, A * O)
    mask = torch.ones(E, A, nA, dtype=torch.bool)
    mask[:, :, 1] = torch.rand(E, A, generator=gen) >= 0.2
    return av, gs, mask


def run(E: int = 256, A: int = 4, O: int = 66, nA: int = 5, T: int = 128, K: int = 4, M: int = 2, updates: int = 2,
        warmup: int = 1, threads: int = 0, seed: int = 42, max_seconds: float = 30.0) -> Dict[str, float]:
    """Times `updates` full PPO updates (after `warmup`) and returns env-steps/sec and the core count."""
    if threads > 0:
        torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(seed)
    Oa, Oc = A + O, A * O
    pa, pc = _init(Oa, nA, 0.01, gen), _init(Oc, 1, 1.0, gen)
    opt_a = torch.optim.Adam([pa], lr=2.5e-4, eps=1e-5)
    opt_c = torch.optim.Adam([pc], lr=2.5e-4, eps=1e-5)
    gamma, lam, clip, ent_c, vf_c, mgn = 0.99, 0.95, 0.2, 0.01, 0.5, 0.5
    av, gs, mask = _observe(E, A, O, nA, gen)
    step_count = torch.zeros(E, dtype=torch.int32)
    done_steps = 0
    t_start = None
    for upd in range(warmup + updates):
        if upd == warmup:
            t_start = time.perf_counter()
        tr = {k: [] for k in ("av", "gs", "mask", "action", "value", "reward", "lp", "done")}
        with torch.no_grad():
            for _t in range(T):
                logits = mlp(pa, Oa, nA, av.reshape(E * A, Oa))
                logits = torch.where(mask.reshape(E * A, nA), logits, torch.full_like(logits, torch.finfo(torch.float32).min))
                u = torch.rand(E * A, nA, generator=gen).clamp_(1e-7, 1 - 1e-7)
                action = (logits - torch.log(-torch.log(u))).argmax(-1)
                lp = torch.log_softmax(logits, -1).gather(-1, action[:, None])[:, 0]
                value = mlp(pc, Oc, 1, gs.reshape(E * A, Oc))[:, 0]
                reward = (torch.rand(E, generator=gen) < 0.02).float()[:, None].expand(E, A)
                step_count += 1
                done = (step_count >= 500) | (torch.rand(E, generator=gen) < 0.002)
                step_count[done] = 0
                for k, v in (("av", av), ("gs", gs), ("mask", mask), ("action", action.reshape(E, A)), ("value", value.reshape(E, A)),
                             ("reward", reward), ("lp", lp.reshape(E, A)), ("done", done[:, None].expand(E, A))):
                    tr[k].append(v)
                av, gs, mask = _observe(E, A, O, nA, gen)
            tr = {k: torch.stack(v, 0) for k, v in tr.items()}
            last_val = mlp(pc, Oc, 1, gs.reshape(E * A, Oc))[:, 0].reshape(E, A)
            adv = torch.zeros(T, E, A)
            g = torch.zeros(E, A)
            nv = last_val
            for t in range(T - 1, -1, -1):
                nd = 1.0 - tr["done"][t].float()
                delta = tr["reward"][t] + gamma * nv * nd - tr["value"][t]
                g = delta + gamma * lam * nd * g
                adv[t] = g
                nv = tr["value"][t]
            tgt = adv + tr["value"]
        flat = {k: v.reshape((T * E,) + v.shape[2:]) for k, v in tr.items()}
        fadv, ftgt = adv.reshape(T * E, A), tgt.reshape(T * E, A)
        B = T * E // M
        for _k in range(K):
            perm = torch.randperm(T * E, generator=gen)
            for mb in range(M):
                rows = perm[mb * B : (mb + 1) * B]
                R = B * A
                tot_a, _, _ = actor_loss(pa, Oa, nA, flat["av"][rows].reshape(R, Oa), flat["mask"][rows].reshape(R, nA),
                                         flat["action"][rows].reshape(R), flat["lp"][rows].reshape(R), fadv[rows].reshape(R),
                                         clip, ent_c)
                tot_c, _ = critic_loss(pc, Oc, flat["gs"][rows].reshape(R, Oc), flat["value"][rows].reshape(R),
                                       ftgt[rows].reshape(R), clip, vf_c)
                opt_a.zero_grad(set_to_none=True)
                opt_c.zero_grad(set_to_none=True)
                tot_a.backward()
                tot_c.backward()
                torch.nn.utils.clip_grad_norm_([pa], mgn)
                torch.nn.utils.clip_grad_norm_([pc], mgn)
                opt_a.step()
                opt_c.step()
        if upd >= warmup:
            done_steps += T * E
            if time.perf_counter() - t_start > max_seconds:
                break
    elapsed = time.perf_counter() - t_start
    return {"env_steps_per_sec": done_steps / elapsed, "seconds": elapsed, "env_steps": done_steps,
            "threads": torch.get_num_threads()}


if __name__ == "__main__":
    print(run())
