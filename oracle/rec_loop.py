"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

Whole-update restatement of Mava's RECURRENT PPO `_update_step` (mava/systems/ppo/rec_mappo.py:68-423)
in float64: rollout with GRU hidden states and last_done bookkeeping (:91-153), bootstrap (:155-175), GAE
with next_done masking (:177-199), epochs over env-permutation minibatches with a full-sequence re-unroll
from hstates[0] (:201-365).  Randomness is an input (Philox action noise of the library, permutations
passed in).  Parity unpinned (see rec_oracle.py header).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import philox
from . import ppo_oracle as po
from . import rec_oracle as ro
from . import tanh_normal as tn
from .synth_env import SynthRware


class OracleRecLearner:
    def __init__(self, *, E, A, O, nA, T, K, M, U=1, centralised=True, seed=42, gamma=0.99, gae_lambda=0.95, clip_eps=0.2,
                 ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, actor_lr=2.5e-4, critic_lr=2.5e-4, time_limit=500, state_dim=0,
                 continuous=False, reward_mode="random", actor_net=None, critic_net=None):
        self.continuous = continuous  # ContinuousActionHead (oracle/tanh_normal.py): nA = action dimensions
        self.ent_step = 0
        self.E, self.A, self.O, self.nA, self.T, self.K, self.M, self.U = E, A, O, nA, T, K, M, U
        self.centralised, self.seed = centralised, seed
        self.h = dict(gamma=gamma, lam=gae_lambda, clip=clip_eps, ent=ent_coef, vf=vf_coef, mgn=max_grad_norm, lrs=(actor_lr, critic_lr))
        self.Oa = A + O
        self.Oc = (state_dim if state_dim > 0 else A * O) if centralised else self.Oa
        # network descriptions handed to oracle/rec_oracle.py: the input width (network/rnn.yaml's default torsos) or a
        # rec_oracle.rec_spec(...) dict (other layer sizes / tanh / layer norm)
        self.Na = actor_net if actor_net is not None else self.Oa
        self.Nc = critic_net if critic_net is not None else self.Oc
        self.envs = [SynthRware(E, A, O, nA, time_limit, seed, env_offset=u * E, state_dim=state_dim, reward_mode=reward_mode)
                     for u in range(U)]
        self.obs = [e.reset(0) for e in self.envs]
        self.dones = [np.zeros((E, A), bool) for _ in range(U)]
        self.Ha, self.Hc = ro.hidden_of(self.Na), ro.hidden_of(self.Nc)  # network.hidden_state_dim
        self.ha = [np.zeros((E * A, self.Ha)) for _ in range(U)]
        self.hc = [np.zeros((E * A, self.Hc)) for _ in range(U)]
        self.t_global = 0
        self.counts = [0, 0]

    def set_params(self, actor_flat, critic_flat):
        self.pa, self.pc = np.asarray(actor_flat, np.float64).copy(), np.asarray(critic_flat, np.float64).copy()
        self.ma, self.va = np.zeros_like(self.pa), np.zeros_like(self.pa)
        self.mc, self.vc = np.zeros_like(self.pc), np.zeros_like(self.pc)

    def _cx(self, obs):
        if self.centralised:
            return np.repeat(obs["global_state"][:, :1, :], self.A, 1).astype(np.float64)
        return obs["agents_view"].astype(np.float64)

    def _rollout(self, u):
        E, A, T, nA = self.E, self.A, self.T, self.nA
        env, obs = self.envs[u], self.obs[u]
        tr = {k: [] for k in ("av", "cx", "mask", "action", "value", "reward", "log_prob", "done_in", "ret", "len", "term")}
        tr["h0a"], tr["h0c"] = self.ha[u].copy(), self.hc[u].copy()
        for t in range(T):
            step = self.t_global + t
            av, cx, mask = obs["agents_view"].astype(np.float64), self._cx(obs), obs["action_mask"]
            d_in = self.dones[u].reshape(E * A)
            n_net = ro.rec_param_count(self.Na, nA)
            y, _, self.ha[u] = ro.rec_forward(self.pa[:n_net], self.Na, nA, av.reshape(1, E * A, -1), d_in[None], self.ha[u])
            if self.continuous:
                eps = tn.normal_noise(self.seed, step, E * A, nA, tn.STREAM_SAMPLE, row_offset=u * E * A)
                two = isinstance(self.Na, dict) and self.Na.get("two_heads")  # ContinuousActionHead(independent_std=False)
                mean, ls = (y[0][:, :nA], y[0][:, nA:]) if two else (y[0], self.pa[n_net:])
                action = tn.sample(mean, ls, eps.astype(np.float64))[0].astype(np.float32).astype(np.float64)
                lp = tn.log_prob(action, mean, ls)
                action = action.reshape(E, A, nA)
            else:
                z = po.masked_logits(y[0], mask.reshape(E * A, nA))
                action = po.gumbel_argmax(z, philox.policy_uniforms(self.seed, step, E * A, nA, row_offset=u * E * A))
                lp = po.log_softmax(z)[np.arange(E * A), action]
                action = action.reshape(E, A)
            v, _, self.hc[u] = ro.rec_forward(self.pc, self.Nc, 1, cx.reshape(1, E * A, -1), d_in[None], self.hc[u])
            obs, reward, done, info = env.step(step + 1, action=None if self.continuous else action)
            for k, val in (("av", av), ("cx", cx), ("mask", mask), ("action", action), ("value", v[0, :, 0].reshape(E, A)),
                           ("reward", reward.astype(np.float64)), ("log_prob", lp.reshape(E, A)), ("done_in", self.dones[u].copy()),
                           ("ret", info["episode_return"]), ("len", info["episode_length"]), ("term", info["is_terminal_step"])):
                tr[k].append(val)
            self.dones[u] = done.copy()
        self.obs[u] = obs
        for k in ("av", "cx", "mask", "action", "value", "reward", "log_prob", "done_in", "ret", "len", "term"):
            tr[k] = np.stack(tr[k], 0)
        lv, _, _ = ro.rec_forward(self.pc, self.Nc, 1, self._cx(obs).reshape(1, E * A, -1), self.dones[u].reshape(1, E * A), self.hc[u])
        tr["last_val"] = lv[0, :, 0].reshape(E, A)
        tr["adv"], tr["tgt"] = po.gae(tr["reward"], tr["value"], tr["done_in"], tr["last_val"], self.h["gamma"], self.h["lam"],
                                      last_done=self.dones[u])
        return tr

    def update(self, permutations: List[np.ndarray]) -> Dict[str, np.ndarray]:
        E, A, T, K, M, nA, h = self.E, self.A, self.T, self.K, self.M, self.nA, self.h
        trajs = [self._rollout(u) for u in range(self.U)]
        self.t_global += T
        self.last_traj = trajs
        metrics = np.zeros((K, M, 4))
        Em = E // M
        for k in range(K):
            for mb in range(M):
                envs = permutations[k][mb * Em : (mb + 1) * Em]
                ga, gc, info = np.zeros_like(self.pa), np.zeros_like(self.pc), np.zeros(3)
                for u, tr in enumerate(trajs):
                    sel = lambda x: x[:, envs].reshape((T, Em * A) + x.shape[3:])
                    h0a = tr["h0a"].reshape(E, A, self.Ha)[envs].reshape(Em * A, self.Ha)
                    h0c = tr["h0c"].reshape(E, A, self.Hc)[envs].reshape(Em * A, self.Hc)
                    if self.continuous:
                        # trajectory rows (t*E + env)*A + a of the minibatch, time-major like sel()
                        gid = ((np.arange(T)[:, None, None] * E + np.asarray(envs)[None, :, None]) * A + np.arange(A)[None, None, :]).reshape(-1)
                        eps = tn.normal_noise(self.seed, self.ent_step, 0, nA, tn.STREAM_ENTROPY, row_offset=u * T * E * A,
                                              gid=gid).astype(np.float64).reshape(T, Em * A, nA)
                        _, la, ent, g1 = ro.rec_actor_loss_grad_continuous(
                            self.pa, self.Na, nA, sel(tr["av"]), sel(tr["done_in"]), h0a, sel(tr["action"]), sel(tr["log_prob"]),
                            sel(tr["adv"]), h["clip"], h["ent"], eps)
                    else:
                        _, la, ent, g1 = ro.rec_actor_loss_grad(self.pa, self.Na, nA, sel(tr["av"]), sel(tr["done_in"]), h0a,
                                                                sel(tr["mask"]), sel(tr["action"]), sel(tr["log_prob"]),
                                                                sel(tr["adv"]), h["clip"], h["ent"])
                    _, vl, g2 = ro.rec_critic_loss_grad(self.pc, self.Nc, sel(tr["cx"]), sel(tr["done_in"]), h0c, sel(tr["value"]),
                                                        sel(tr["tgt"]), h["clip"], h["vf"])
                    ga += g1
                    gc += g2
                    info += np.array([la, ent, vl])
                ga, gc, info = ga / self.U, gc / self.U, info / self.U
                self.pa, self.ma, self.va, self.counts[0] = po.clip_adam(self.pa, ga, self.ma, self.va, self.counts[0], h["lrs"][0], h["mgn"])
                self.pc, self.mc, self.vc, self.counts[1] = po.clip_adam(self.pc, gc, self.mc, self.vc, self.counts[1], h["lrs"][1], h["mgn"])
                la, ent, vl = info
                metrics[k, mb] = [(la - h["ent"] * ent) + h["vf"] * vl, vl, la, ent]
                self.ent_step += 1
        return {"train_metrics": metrics}
