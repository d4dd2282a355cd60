"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

The CPU baseline that bench.py times beside the GPU path for the RECURRENT systems (`cpu_baseline`, kind
"port"): Mava's recurrent PPO update loop (mava/systems/ppo/rec_mappo.py:68-423: rollout carrying GRU hidden
states and last_done -> bootstrap -> GAE with next_done masking -> K epochs x M env-permutation minibatches, each
re-unrolling the whole sequence from hstates[0] for both networks, clip + Adam) restated on torch-CPU float32
with autograd, multi-threaded over the host cores, on SMAX-shaped synthetic inputs (torch.rand).  A SUBSTITUTE
for "Mava's own JAX path on the host CPU", which cannot run here.
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch

from . import rec_oracle as ro

H = 128


def _loss_actor(fa, Oa, nA, obs, done, h0, mask, action, old_lp, adv, clip, ent_c):
    logits, _ = ro.t_rec_forward(fa, Oa, nA, obs, done, h0)
    logits = torch.where(mask, logits, torch.full_like(logits, torch.finfo(torch.float32).min))
    lsm = torch.log_softmax(logits, -1)
    lp = lsm.gather(-1, action[..., None])[..., 0]
    ratio = torch.exp(lp - old_lp)
    g = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
    la = -torch.minimum(ratio * g, torch.clamp(ratio, 1 - clip, 1 + clip) * g).mean()
    pr = lsm.exp()
    ent = -(torch.where(pr > 0, pr * lsm, torch.zeros_like(pr))).sum(-1).mean()
    return la - ent_c * ent


def _loss_critic(fc, Oc, x, done, h0, old_v, tgt, clip, vf_c):
    v, _ = ro.t_rec_forward(fc, Oc, 1, x, done, h0)
    v = v[..., 0]
    vc = old_v + (v - old_v).clamp(-clip, clip)
    return vf_c * 0.5 * torch.maximum((v - tgt) ** 2, (vc - tgt) ** 2).mean()


def run(E: int = 32, A: int = 8, Oa: int = 155, Oc: int = 188, nA: int = 13, T: int = 128, K: int = 4, M: int = 2,
        updates: int = 2, warmup: int = 0, threads: int = 0, seed: int = 42, max_seconds: float = 30.0,
        shared_state: bool = True) -> Dict[str, float]:
    """Times `updates` full recurrent PPO updates (after `warmup`) and returns env-steps/sec and the core count."""
    if threads > 0:
        torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    fa = torch.tensor(ro.init_rec(rng, Oa, nA, 0.01), dtype=torch.float32, requires_grad=True)
    fc = torch.tensor(ro.init_rec(rng, Oc, 1, 1.0), dtype=torch.float32, requires_grad=True)
    opt_a = torch.optim.Adam([fa], lr=2.5e-4, eps=1e-5)
    opt_c = torch.optim.Adam([fc], lr=2.5e-4, eps=1e-5)
    gamma, lam, clip, ent_c, vf_c, mgn = 0.99, 0.95, 0.2, 0.01, 0.5, 0.5
    R = E * A

    def observe():
        av = (torch.rand(E, A, Oa, generator=gen) < 0.2).float()
        st = (torch.rand(E, 1, Oc, generator=gen) < 0.2).float()
        st = st.expand(E, A, Oc) if shared_state else (torch.rand(E, A, Oc, generator=gen) < 0.2).float()
        mask = torch.ones(E, A, nA, dtype=torch.bool)
        mask[:, :, 1] = torch.rand(E, A, generator=gen) >= 0.2
        return av, st, mask

    av, st, mask = observe()
    ha, hc = torch.zeros(R, H), torch.zeros(R, H)
    dones = torch.zeros(E, A, dtype=torch.bool)
    step_count = torch.zeros(E, dtype=torch.int32)
    done_steps, t_start = 0, None
    for upd in range(warmup + updates):
        if upd == warmup:
            t_start = time.perf_counter()
        tr = {k: [] for k in ("av", "st", "mask", "action", "value", "reward", "lp", "done_in")}
        h0a, h0c = ha.clone(), hc.clone()
        with torch.no_grad():
            for _t in range(T):
                d_in = dones.reshape(1, R)
                logits, ha = ro.t_rec_forward(fa, Oa, nA, av.reshape(1, R, Oa), d_in, ha)
                logits = torch.where(mask.reshape(R, nA), logits[0], torch.full_like(logits[0], torch.finfo(torch.float32).min))
                u = torch.rand(R, nA, generator=gen).clamp_(1e-7, 1 - 1e-7)
                action = (logits - torch.log(-torch.log(u))).argmax(-1)
                lp = torch.log_softmax(logits, -1).gather(-1, action[:, None])[:, 0]
                value, hc = ro.t_rec_forward(fc, Oc, 1, st.reshape(1, R, Oc), d_in, hc)
                reward = (torch.rand(E, generator=gen) < 0.02).float()[:, None].expand(E, A)
                step_count += 1
                done = (step_count >= 500) | (torch.rand(E, generator=gen) < 0.002)
                step_count[done] = 0
                for k, v in (("av", av), ("st", st), ("mask", mask), ("action", action.reshape(E, A)),
                             ("value", value[0, :, 0].reshape(E, A)), ("reward", reward), ("lp", lp.reshape(E, A)),
                             ("done_in", dones)):
                    tr[k].append(v)
                dones = done[:, None].expand(E, A).clone()
                av, st, mask = observe()
            tr = {k: torch.stack(v, 0) for k, v in tr.items()}
            lv, _ = ro.t_rec_forward(fc, Oc, 1, st.reshape(1, R, Oc), dones.reshape(1, R), hc)
            nv, g, nd_flag = lv[0, :, 0].reshape(E, A), torch.zeros(E, A), dones
            adv = torch.zeros(T, E, A)
            for t in range(T - 1, -1, -1):  # rec_mappo.py:180-188: masks with the NEXT step's stored flag
                nd = 1.0 - nd_flag.float()
                delta = tr["reward"][t] + gamma * nv * nd - tr["value"][t]
                g = delta + gamma * lam * nd * g
                adv[t] = g
                nv, nd_flag = tr["value"][t], tr["done_in"][t]
            tgt = adv + tr["value"]
        Em = E // M
        for _k in range(K):
            perm = torch.randperm(E, generator=gen)
            for mb in range(M):
                envs = perm[mb * Em : (mb + 1) * Em]
                sel = lambda x: x[:, envs].reshape((T, Em * A) + x.shape[3:])
                h0a_m = h0a.reshape(E, A, H)[envs].reshape(Em * A, H)
                h0c_m = h0c.reshape(E, A, H)[envs].reshape(Em * A, H)
                tot_a = _loss_actor(fa, Oa, nA, sel(tr["av"]), sel(tr["done_in"]), h0a_m, sel(tr["mask"]), sel(tr["action"]),
                                    sel(tr["lp"]), sel(adv), clip, ent_c)
                tot_c = _loss_critic(fc, Oc, sel(tr["st"]), sel(tr["done_in"]), h0c_m, sel(tr["value"]), sel(tgt), clip, vf_c)
                opt_a.zero_grad(set_to_none=True)
                opt_c.zero_grad(set_to_none=True)
                tot_a.backward()
                tot_c.backward()
                torch.nn.utils.clip_grad_norm_([fa], mgn)
                torch.nn.utils.clip_grad_norm_([fc], mgn)
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
    print(run(updates=1))
