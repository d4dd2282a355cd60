"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

Whole-update restatement of Mava's feed-forward PPO `_update_step`
(mava/systems/ppo/ff_mappo.py:56-300) in NumPy float64, composed from oracle/ppo_oracle.py and the
synthetic environment restatement oracle/synth_env.py.  Randomness is an INPUT: action noise comes
from the library's Philox stream (oracle/philox.py) and the epoch permutations are passed in, per
north_star "identical trajectory inputs".  U update-batch replicas and D "virtual ranks" are
looped; their gradients and loss infos are averaged like the two pmeans of ff_mappo.py:224-238.
Parity unpinned (see ppo_oracle.py header).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import philox
from . import ppo_oracle as po
from . import tanh_normal as tn
from .synth_env import SynthRware


class OracleLearner:
    def __init__(self, *, E, A, O, nA, T, K, M, U=1, D=1, centralised=True, seed=42, gamma=0.99, gae_lambda=0.95,
                 clip_eps=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, actor_lr=2.5e-4, critic_lr=2.5e-4,
                 decay=False, num_updates=1, time_limit=500, shared_gs=True, continuous=False, reward_mode="random",
                 actor_spec=None, critic_spec=None, independent_std=True):
        # actor_spec / critic_spec (oracle/generic_oracle.py): configurable torsos instead of the default [128, 128] MLP
        self.actor_spec, self.critic_spec, self.independent_std = actor_spec, critic_spec, independent_std
        self.continuous = continuous  # ContinuousActionHead (oracle/tanh_normal.py): nA = action dimensions
        self.ent_step = 0
        self.E, self.A, self.O, self.nA, self.T, self.K, self.M, self.U, self.D = E, A, O, nA, T, K, M, U, D
        self.centralised, self.seed = centralised, seed
        self.h = dict(gamma=gamma, lam=gae_lambda, clip=clip_eps, ent=ent_coef, vf=vf_coef, mgn=max_grad_norm,
                      lrs=(actor_lr, critic_lr), decay=decay, num_updates=num_updates)
        self.Oa = A + O
        self.Oc = A * O if centralised else self.Oa
        self.envs = [[SynthRware(E, A, O, nA, time_limit, seed, env_offset=(d * U + u) * E, reward_mode=reward_mode) for u in range(U)] for d in range(D)]
        self.obs = [[e.reset(0) for e in row] for row in self.envs]
        self.t_global = 0
        self.counts = [0, 0]

    def set_params(self, actor_flat, critic_flat):
        self.pa = np.asarray(actor_flat, np.float64).copy()
        self.pc = np.asarray(critic_flat, np.float64).copy()
        self.ma, self.va = np.zeros_like(self.pa), np.zeros_like(self.pa)
        self.mc, self.vc = np.zeros_like(self.pc), np.zeros_like(self.pc)

    def _critic_in(self, obs):
        if self.centralised:  # global_state tiled per agent (mava/wrappers/jumanji.py:57-58)
            return np.repeat(obs["global_state"][:, :1, :], self.A, 1).astype(np.float64)
        return obs["agents_view"].astype(np.float64)

    def _rollout(self, d, u, forced_action=None):
        """ff_mappo.py:76-106 for one (rank, replica).  `forced_action` (T, E, A) int: "identical trajectory inputs" in
        the strict sense - the sampled action is an INPUT (the oracle still draws its own and records it as
        tr["own_action"], with the Gumbel-score margin between the two in tr["action_margin"], so that a test can show
        that every disagreement is a near-tie of the argmax and nothing else)."""
        E, A, T, nA = self.E, self.A, self.T, self.nA
        env, obs = self.envs[d][u], self.obs[d][u]
        if self.actor_spec is not None:
            from . import generic_oracle as go

            pa = pc = None
            a_fwd = lambda x: go.np_forward(self.pa, self.actor_spec, x)
            c_fwd = lambda x: go.np_forward(self.pc, self.critic_spec, x)[0]
        else:
            pa, pc = po.mlp_unflatten(self.pa[: po.mlp_param_count(self.Oa, nA)], self.Oa, nA), po.mlp_unflatten(self.pc, self.Oc, 1)
            a_fwd = lambda x: [po.mlp_forward(pa, x)]
            c_fwd = lambda x: po.mlp_forward(pc, x)
        tr = {k: [] for k in ("av", "cx", "mask", "action", "value", "reward", "log_prob", "done", "ret", "len", "term")}
        own_action, margin = [], []
        for t in range(T):
            step = self.t_global + t
            av = obs["agents_view"].astype(np.float64)
            cx = self._critic_in(obs)
            mask = obs["action_mask"]
            outs = a_fwd(av.reshape(E * A, -1))
            y = outs[0]
            log_std = self.pa[-nA:] if self.independent_std else outs[1]
            if self.continuous:
                eps = tn.normal_noise(self.seed, step, E * A, nA, tn.STREAM_SAMPLE, row_offset=(d * self.U + u) * E * A)
                # the action is stored in float32 (the trajectory dtype) and scored as stored
                action = tn.sample(y, log_std, eps.astype(np.float64))[0].astype(np.float32).astype(np.float64)
                lp = tn.log_prob(action, y, log_std)
                action = action.reshape(E, A, nA)
            else:
                z = po.masked_logits(y, mask.reshape(E * A, nA))
                uni = philox.policy_uniforms(self.seed, step, E * A, nA, row_offset=(d * self.U + u) * E * A)
                action = po.gumbel_argmax(z, uni).reshape(E, A)
                if forced_action is not None:
                    fa_ = np.asarray(forced_action[t]).reshape(-1).astype(np.int64)
                    with np.errstate(divide="ignore"):
                        score = z - np.log(-np.log(uni))
                    rows_ = np.arange(E * A)
                    own_action.append(action.copy())
                    margin.append((score[rows_, action.reshape(-1)] - score[rows_, fa_]).reshape(E, A))
                    action = fa_.reshape(E, A).astype(action.dtype)
                lp = po.log_softmax(z)[np.arange(E * A), action.reshape(-1)]
            value = c_fwd(cx.reshape(E * A, -1))[:, 0]
            obs, reward, done, info = env.step(step + 1, action=None if self.continuous else action)
            for k, v in (("av", av), ("cx", cx), ("mask", mask), ("action", action), ("value", value.reshape(E, A)),
                         ("reward", reward.astype(np.float64)), ("log_prob", lp.reshape(E, A)), ("done", done),
                         ("ret", info["episode_return"]), ("len", info["episode_length"]), ("term", info["is_terminal_step"])):
                tr[k].append(v)
        self.obs[d][u] = obs
        tr = {k: np.stack(v, 0) for k, v in tr.items()}
        if own_action:
            tr["own_action"], tr["action_margin"] = np.stack(own_action, 0), np.stack(margin, 0)
        last_val = c_fwd(self._critic_in(obs).reshape(E * A, -1))[:, 0].reshape(E, A)  # ff_mappo.py:110
        tr["adv"], tr["tgt"] = po.gae(tr["reward"], tr["value"], tr["done"], last_val, self.h["gamma"], self.h["lam"])
        tr["last_val"] = last_val
        return tr

    def update(self, permutations: List[np.ndarray], forced_actions=None) -> Dict[str, np.ndarray]:
        """One `_update_step` on every (rank, replica); returns train metrics (K, M, 4) and keeps the
        trajectories of the call in self.last_traj[d][u].  forced_actions[d][u]: see _rollout."""
        E, A, T, K, M, nA = self.E, self.A, self.T, self.K, self.M, self.nA
        h = self.h
        trajs = [[self._rollout(d, u, None if forced_actions is None else forced_actions[d][u]) for u in range(self.U)]
                 for d in range(self.D)]
        self.t_global += T
        self.last_traj = trajs
        metrics = np.zeros((K, M, 4))
        flat = lambda x: x.reshape((T * E,) + x.shape[2:])  # merge_leading_dims(x, 2)
        for k in range(K):
            perm = permutations[k]
            for mb in range(M):
                rows = po.minibatch_rows(perm, M, mb)
                ga = np.zeros_like(self.pa)
                gc = np.zeros_like(self.pc)
                info = np.zeros(3)
                for d in range(self.D):
                    for u in range(self.U):
                        tr = trajs[d][u]
                        sel = lambda x: flat(x)[rows]
                        R = rows.size * A
                        if self.actor_spec is not None:
                            from . import generic_oracle as go

                            if self.continuous:
                                gid = (rows[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
                                eps = tn.normal_noise(self.seed, self.ent_step, 0, nA, tn.STREAM_ENTROPY,
                                                      row_offset=(d * self.U + u) * T * E * A, gid=gid).astype(np.float64)
                                _, la, ent, g1 = go.actor_loss_grad_continuous(
                                    self.pa, self.actor_spec, sel(tr["av"]).reshape(R, -1), sel(tr["action"]).reshape(R, nA),
                                    sel(tr["log_prob"]).reshape(R), sel(tr["adv"]).reshape(R), h["clip"], h["ent"], eps,
                                    self.independent_std)
                            else:
                                _, la, ent, g1 = go.actor_loss_grad(
                                    self.pa, self.actor_spec, sel(tr["av"]).reshape(R, -1), sel(tr["mask"]).reshape(R, nA),
                                    sel(tr["action"]).reshape(R), sel(tr["log_prob"]).reshape(R), sel(tr["adv"]).reshape(R),
                                    h["clip"], h["ent"])
                        elif self.continuous:
                            gid = (rows[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)  # trajectory rows
                            eps = tn.normal_noise(self.seed, self.ent_step, 0, nA, tn.STREAM_ENTROPY,
                                                  row_offset=(d * self.U + u) * T * E * A, gid=gid).astype(np.float64)
                            _, la, ent, g1 = tn.actor_loss_and_grad(
                                self.pa, self.Oa, nA, sel(tr["av"]).reshape(R, -1), sel(tr["action"]).reshape(R, nA),
                                sel(tr["log_prob"]).reshape(R), sel(tr["adv"]).reshape(R), h["clip"], h["ent"], eps)
                        else:
                            _, la, ent, g1 = po.actor_loss_and_grad(
                                self.pa, self.Oa, nA, sel(tr["av"]).reshape(R, -1), sel(tr["mask"]).reshape(R, nA),
                                sel(tr["action"]).reshape(R), sel(tr["log_prob"]).reshape(R), sel(tr["adv"]).reshape(R),
                                h["clip"], h["ent"])
                        if self.critic_spec is not None:
                            _, vl, g2 = go.critic_loss_grad(self.pc, self.critic_spec, sel(tr["cx"]).reshape(R, -1),
                                                            sel(tr["value"]).reshape(R), sel(tr["tgt"]).reshape(R), h["clip"], h["vf"])
                        else:
                            _, vl, g2 = po.critic_loss_and_grad(
                                self.pc, self.Oc, sel(tr["cx"]).reshape(R, -1), sel(tr["value"]).reshape(R),
                                sel(tr["tgt"]).reshape(R), h["clip"], h["vf"])
                        ga += g1
                        gc += g2
                        info += np.array([la, ent, vl])
                n = self.U * self.D  # pmean over "batch" then "device" (ff_mappo.py:224-238)
                ga, gc, info = ga / n, gc / n, info / n
                lra = po.learning_rate(h["lrs"][0], self.counts[0], h["decay"], K, M, h["num_updates"])
                lrc = po.learning_rate(h["lrs"][1], self.counts[1], h["decay"], K, M, h["num_updates"])
                self.pa, self.ma, self.va, self.counts[0] = po.clip_adam(self.pa, ga, self.ma, self.va, self.counts[0], lra, h["mgn"])
                self.pc, self.mc, self.vc, self.counts[1] = po.clip_adam(self.pc, gc, self.mc, self.vc, self.counts[1], lrc, h["mgn"])
                la, ent, vl = info
                metrics[k, mb] = [(la - h["ent"] * ent) + h["vf"] * vl, vl, la, ent]  # ff_mappo.py:255-265
                self.ent_step += 1
        return {"train_metrics": metrics}
