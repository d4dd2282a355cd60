"""Synthetic RWARE-shaped vectorised environment.

Stands in for `environments.make(config, add_global_state=...)` (mava/utils/make_env.py:215-240):
Jumanji's RobotWarehouse is third-party JAX code that is not available, so observations, masks,
rewards and dones are generated with RWARE's shapes and statistics by mava_synth_rware_step while
the wrapper semantics the learner records (agent ids, global state, auto-reset, episode metrics)
are reproduced.  The object follows the MarlEnv protocol (mava/types.py:34-108) in a natively
BATCHED form: `reset`/`step` act on all `num_envs` environments of one (device, update-batch)
replica at once, and `step_into` writes straight into trajectory slots owned by the learner.
"""
from __future__ import annotations

from typing import Any, Dict, NamedTuple, Optional, Tuple

import torch

from .._lib import check, lib, ptr, stream_ptr
from ..types import Observation, ObservationGlobalState, TimeStep


EVAL_KEY_TAG = 0x4556414C4556414C  # "EVALEVAL": xor-ed into the Philox key of the evaluation environments


class SynthState(NamedTuple):
    step_count: torch.Tensor  # (E, A) i32
    run_return: torch.Tensor  # (E,) f32   running_count_episode_return
    run_length: torch.Tensor  # (E,) i32
    ep_return: torch.Tensor  # (E,) f32   episode_return (last finished)
    ep_length: torch.Tensor  # (E,) i32
    t: torch.Tensor  # () i64 host-side step counter of this replica's stream


class ObsSpec(NamedTuple):
    agents_view: Tuple[int, ...]
    action_mask: Tuple[int, ...]
    global_state: Optional[Tuple[int, ...]]
    step_count: Tuple[int, ...]


class SyntheticRware:
    # all agents of an env receive the same global_state row (mava/wrappers/jumanji.py:53-59)
    global_state_shared = True
    # mava_rollout_ff_f32 (csrc/rollout_h2.hip) carries THIS generator's env phase: the feed-forward learner routes an
    # env to the one-launch rollout only when it declares so (any other MarlEnv steps through step_into per time step)
    supports_fused_rollout = True

    def __init__(self, num_envs: int, num_agents: int, obs_dim: int = 66, num_actions: int = 5, time_limit: int = 500,
                 add_global_state: bool = False, add_agent_id: bool = True, seed: int = 42, env_offset: int = 0,
                 tile_global_state: bool = False, device: Optional[torch.device] = None, state_dim: int = 0,
                 reward_mode: str = "random"):
        if not add_agent_id:
            raise NotImplementedError("the synthetic generator always prepends the agent one-hot id (add_agent_id=True)")
        self.num_envs, self.num_agents = int(num_envs), int(num_agents)
        self.raw_obs_dim, self.action_dim, self.time_limit = int(obs_dim), int(num_actions), int(time_limit)
        self.add_global_state = add_global_state
        self.seed, self.env_offset = int(seed), int(env_offset)
        self.gs_tiles = self.num_agents if tile_global_state else 1
        self.synth_state_dim = int(state_dim)  # 0: global_state = concatenated raw views; > 0: own state vector
        self.global_state_shared = not tile_global_state
        # "random": Bernoulli(0.02) team reward, independent of the actions (the measurement workload of SURVEY 8d);
        # "match": team reward = fraction of agents whose action equals (first grid coordinate they observed) mod
        # n_actions - an action-dependent task, so that runs can show the PPO stack LEARNS (tests/test_gpu_learning.py)
        if reward_mode not in ("random", "match"):
            raise ValueError(f"reward_mode must be 'random' or 'match', got {reward_mode!r}")
        self.reward_mode = reward_mode
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        # image view of the observation / state vectors for CNN torsos: (H, W, C) with H*W*C = obs_dim / state_dim
        # (env.synthetic.obs_shape / state_shape; Mava's CNN environments emit such observations natively)
        self.obs_shape: Optional[tuple] = None
        self.state_shape: Optional[tuple] = None

    def clone(self, env_offset: int, num_envs: Optional[int] = None) -> "SyntheticRware":
        """Same environment family on a disjoint range of global env ids (one per replica / rank)."""
        c = SyntheticRware(num_envs or self.num_envs, self.num_agents, self.raw_obs_dim, self.action_dim, self.time_limit,
                           self.add_global_state, True, self.seed, env_offset, self.gs_tiles != 1, self.device,
                           self.synth_state_dim, self.reward_mode)
        c.obs_shape, c.state_shape = self.obs_shape, self.state_shape
        return c

    # ---- specs ----------------------------------------------------------------------------
    @property
    def obs_dim(self) -> int:
        return self.num_agents + self.raw_obs_dim

    @property
    def state_dim(self) -> int:
        return self.synth_state_dim if self.synth_state_dim > 0 else self.num_agents * self.raw_obs_dim

    def observation_spec(self) -> ObsSpec:
        A = self.num_agents
        return ObsSpec((A, self.obs_dim), (A, self.action_dim), (A, self.state_dim) if self.add_global_state else None, (A,))

    def alloc_state(self) -> SynthState:
        E, A, d = self.num_envs, self.num_agents, self.device
        return SynthState(torch.zeros((E, A), dtype=torch.int32, device=d), torch.zeros(E, device=d),
                          torch.zeros(E, dtype=torch.int32, device=d), torch.zeros(E, device=d),
                          torch.zeros(E, dtype=torch.int32, device=d), torch.zeros((), dtype=torch.int64))

    def alloc_obs(self) -> Dict[str, torch.Tensor]:
        E, A, d = self.num_envs, self.num_agents, self.device
        return {
            "agents_view": torch.empty((E, A, self.obs_dim), device=d),
            "global_state": torch.empty((E, self.gs_tiles, self.state_dim), device=d),
            "action_mask": torch.empty((E, A, self.action_dim), dtype=torch.uint8, device=d),
            "step_count": torch.empty((E, A), dtype=torch.int32, device=d),
        }

    # ---- kernel call ----------------------------------------------------------------------
    def step_into(self, state: SynthState, t: int, obs: Dict[str, torch.Tensor], reward=None, done=None, info_return=None,
                  info_length=None, info_terminal=None, is_reset: bool = False, env_offset: Optional[int] = None,
                  t_base: Optional[torch.Tensor] = None, action: Optional[torch.Tensor] = None) -> None:
        """One vectorised step (or reset) writing the next observation into `obs` and the transition
        into the given (E, A) / (E,) slots.  `t` is the replica's global step index (Philox counter); `t_base`
        (a device int32 word) is added to it on the device, for rollouts replayed from a captured graph."""
        off = self.env_offset if env_offset is None else env_offset
        match = self.reward_mode == "match"
        if match and not is_reset:
            if action is None or action.dtype != torch.int32 or action.numel() != self.num_envs * self.num_agents:
                raise ValueError("reward_mode='match' needs the (E, A) int32 actions of the step")
        check(
            lib().mava_synth_rware_step(self.num_envs, self.num_agents, self.raw_obs_dim, self.action_dim, self.gs_tiles,
                                        self.synth_state_dim, self.time_limit, self.seed & 0xFFFFFFFFFFFFFFFF, t & 0xFFFFFFFF, ptr(t_base),
                                        off & 0xFFFFFFFF,
                                        int(is_reset), ptr(state.step_count), ptr(state.run_return), ptr(state.run_length),
                                        ptr(state.ep_return), ptr(state.ep_length), ptr(obs["agents_view"]),
                                        ptr(obs["global_state"]), ptr(obs["action_mask"]), ptr(obs["step_count"]),
                                        ptr(reward), ptr(done), ptr(info_return), ptr(info_length), ptr(info_terminal),
                                        ptr(action) if match else None, int(match), stream_ptr()),
            "mava_synth_rware_step",
        )

    # ---- MarlEnv-style batched API (allocating; the learner uses step_into) -----------------
    def _observation(self, obs: Dict[str, torch.Tensor]):
        mask = obs["action_mask"].bool()
        if self.add_global_state:
            gs = obs["global_state"]
            gs = gs.expand(-1, self.num_agents, -1) if self.gs_tiles == 1 else gs
            return ObservationGlobalState(obs["agents_view"], mask, gs, obs["step_count"])
        return Observation(obs["agents_view"], mask, obs["step_count"])

    def reset(self, key: Any = None) -> Tuple[SynthState, TimeStep]:
        state, obs = self.alloc_state(), self.alloc_obs()
        self.step_into(state, 0, obs, is_reset=True)
        E, A, d = self.num_envs, self.num_agents, self.device
        extras = {"episode_metrics": {"episode_return": torch.zeros(E, device=d),
                                      "episode_length": torch.zeros(E, dtype=torch.int32, device=d),
                                      "is_terminal_step": torch.zeros(E, dtype=torch.bool, device=d)}}
        ts = TimeStep(torch.zeros(E, dtype=torch.int8, device=d), torch.zeros((E, A), device=d),
                      torch.ones((E, A), device=d), self._observation(obs), extras)
        return state, ts

    def step(self, state: SynthState, action: torch.Tensor) -> Tuple[SynthState, TimeStep]:
        E, A, d = self.num_envs, self.num_agents, self.device
        obs = self.alloc_obs()
        reward = torch.empty((E, A), device=d)
        done = torch.empty((E, A), dtype=torch.uint8, device=d)
        ir = torch.empty(E, device=d)
        il = torch.empty(E, dtype=torch.int32, device=d)
        it = torch.empty(E, dtype=torch.uint8, device=d)
        t = int(state.t) + 1
        self.step_into(state, t, obs, reward, done, ir, il, it,
                       action=action.to(torch.int32).contiguous() if self.reward_mode == "match" else None)
        state = state._replace(t=torch.tensor(t, dtype=torch.int64))
        last = it.bool()
        extras = {"episode_metrics": {"episode_return": ir, "episode_length": il, "is_terminal_step": last}}
        step_type = torch.where(last, 2, 1).to(torch.int8)
        ts = TimeStep(step_type, reward, 1.0 - done.float(), self._observation(obs), extras)
        return state, ts


def make(config, add_global_state: bool = False, device=None, env_offset: int = 0):
    """Counterpart of mava/utils/make_env.py:215-240 for the synthetic stand-in: returns
    (train_env, eval_env) sized by config.arch.num_envs / config.arch.num_eval_episodes."""
    tc = config.env.scenario.task_config
    syn = config.env.get("synthetic", {"obs_dim": 66, "num_actions": 5})
    kw = dict(num_agents=int(tc.num_agents), obs_dim=int(syn["obs_dim"]), num_actions=int(syn["num_actions"]),
              time_limit=int(config.env.kwargs.get("time_limit", 500)), add_global_state=add_global_state,
              add_agent_id=bool(config.system.add_agent_id) and not bool(config.env.implicit_agent_id),
              device=device, state_dim=int(syn.get("state_dim", 0) or 0), reward_mode=str(syn.get("reward_mode", "random")))
    seed = int(config.system.seed)
    train = SyntheticRware(num_envs=int(config.arch.num_envs), env_offset=env_offset, seed=seed, **kw)
    # The evaluation envs draw from their own Philox KEY (not an env-id offset: the kernel forms the per-agent counter
    # (env_offset + e) * A + agent in 32 bits, where an offset of 2^30 wraps back onto the training envs for A >= 4).
    evale = SyntheticRware(num_envs=int(config.arch.num_eval_episodes), env_offset=env_offset, seed=seed ^ EVAL_KEY_TAG, **kw)
    for e in (train, evale):
        e.obs_shape = tuple(int(v) for v in syn["obs_shape"]) if syn.get("obs_shape", None) else None
        e.state_shape = tuple(int(v) for v in syn["state_shape"]) if syn.get("state_shape", None) else None
    return train, evale
