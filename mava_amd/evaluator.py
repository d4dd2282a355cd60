"""Evaluator seam (mava/evaluator.py:80-209): `get_eval_fn(env, act_fn, config, absolute_metric)`
runs whole episodes with `act_fn(params, timestep, key, actor_state) -> (action, actor_state)` and
returns per-episode metrics; `make_ff_eval_act_fn(actor_apply_fn, config)` (:175-186) and
`make_rec_eval_act_fn` (:189-207) build the act functions from `actor_network.apply`.  Off the timed path: the policy forward runs
on the HIP kernel behind `apply`, the episode loop is host Python.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Tuple

import torch


def make_ff_eval_act_fn(actor_apply_fn: Callable, config) -> Callable:
    greedy = bool(config.arch.evaluation_greedy)

    def eval_act_fn(params, timestep, key: torch.Generator, actor_state: Dict) -> Tuple[torch.Tensor, Dict]:
        pi = actor_apply_fn(params, timestep.observation)
        action = pi.mode() if greedy else pi.sample(seed=key)
        return action, actor_state

    return eval_act_fn


def make_rec_eval_act_fn(actor_apply_fn: Callable, config) -> Callable:
    """mava/evaluator.py:189-207: the recurrent act function.  `actor_state["hidden_state"]` carries the policy
    hidden state (E, A, 128); the done flag entering the step resets it inside the network (networks.py:253-257)."""
    greedy = bool(config.arch.evaluation_greedy)
    _hidden_state = "hidden_state"

    def eval_act_fn(params, timestep, key: torch.Generator, actor_state: Dict) -> Tuple[torch.Tensor, Dict]:
        hidden_state = actor_state[_hidden_state]
        obs = timestep.observation
        n_agents = obs.agents_view.shape[1]
        last = timestep.last()  # (E,)
        last_done = last.unsqueeze(-1).expand(last.shape[0], n_agents)
        # add the time axis to the observation and the done flags (mava/evaluator.py:202-203)
        ac_obs = type(obs)(*[None if f is None else f.unsqueeze(0) for f in obs])
        hidden_state, pi = actor_apply_fn(params, hidden_state, (ac_obs, last_done.unsqueeze(0)))
        action = pi.mode() if greedy else pi.sample(seed=key)
        return action.squeeze(0), {_hidden_state: hidden_state}

    return eval_act_fn


def get_eval_fn(env, act_fn: Callable, config, absolute_metric: bool) -> Callable:
    """Every env of `env` plays one episode per loop; metrics are taken at the first terminal step of
    each env (mava/evaluator.py:139-148)."""
    n_episodes = int(config.arch.num_absolute_metric_eval_episodes if absolute_metric else config.arch.num_eval_episodes)
    loops = max(1, -(-n_episodes // env.num_envs))

    def eval_fn(params: Any, seed: int, init_act_state: Any = None) -> Dict[str, torch.Tensor]:
        gen = torch.Generator(device=env.device).manual_seed(int(seed))
        rets, lens = [], []
        for _ in range(loops):
            state, ts = env.reset()
            E = env.num_envs
            finished = torch.zeros(E, dtype=torch.bool, device=env.device)
            ep_ret = torch.zeros(E, device=env.device)
            ep_len = torch.zeros(E, dtype=torch.int32, device=env.device)
            act_state = init_act_state or {}
            for _t in range(int(env.time_limit)):
                action, act_state = act_fn(params, ts, gen, act_state)
                state, ts = env.step(state, action)
                m = ts.extras["episode_metrics"]
                newly = m["is_terminal_step"] & ~finished
                ep_ret = torch.where(newly, m["episode_return"], ep_ret)
                ep_len = torch.where(newly, m["episode_length"], ep_len)
                finished |= newly
                if bool(finished.all()):
                    break
            rets.append(ep_ret)
            lens.append(ep_len)
        return {"episode_return": torch.cat(rets), "episode_length": torch.cat(lens)}

    return eval_fn
