"""State / transition containers with the reference's field names.

Mirrors mava/systems/ppo/types.py:26-91 (Params, OptStates, HiddenStates, LearnerState,
RNNLearnerState, PPOTransition, RNNPPOTransition) and mava/types.py:111-160 (Observation,
ObservationGlobalState, ExperimentOutput, LearnerFn).  Leaves are torch tensors that ALIAS the
device buffers the HIP kernels update in place; every non-env leaf carries the reference's leading
(device, update_batch) dims - with one process per GPU the device dim is 1 locally, so the host
idioms `x[:, 0]` (unreplicate_batch_dim) and `x[0, 0]` (unreplicate_n_dims) keep working.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Generic, NamedTuple, Optional, TypeVar

import torch


class Params(NamedTuple):
    actor_params: Any  # {"params": {"torso": {"Dense_0": {kernel,bias}, "Dense_1": ...}, "action_head": {...}}}
    critic_params: Any


class AdamState(NamedTuple):
    """optax.ScaleByAdamState fields (count, mu, nu) for one network."""

    count: torch.Tensor
    mu: Any
    nu: Any


class OptStates(NamedTuple):
    actor_opt_state: AdamState
    critic_opt_state: AdamState


class HiddenStates(NamedTuple):
    policy_hidden_state: torch.Tensor
    critic_hidden_state: torch.Tensor


class Observation(NamedTuple):
    agents_view: torch.Tensor  # (..., num_agents, num_obs_features)
    action_mask: torch.Tensor  # (..., num_agents, num_actions)
    step_count: torch.Tensor  # (..., num_agents)


class ObservationGlobalState(NamedTuple):
    agents_view: torch.Tensor
    action_mask: torch.Tensor
    global_state: torch.Tensor  # (..., num_agents, num_agents * num_obs_features)
    step_count: torch.Tensor


class TimeStep(NamedTuple):
    """jumanji.types.TimeStep fields used by the learner (mava/types.py:34-108)."""

    step_type: torch.Tensor  # 0 FIRST, 1 MID, 2 LAST
    reward: torch.Tensor
    discount: torch.Tensor
    observation: Any
    extras: Dict[str, Any]

    def last(self) -> torch.Tensor:
        return self.step_type == 2


class LearnerState(NamedTuple):
    params: Params
    opt_states: OptStates
    key: torch.Tensor
    env_state: Any
    timestep: TimeStep


class RNNLearnerState(NamedTuple):
    params: Params
    opt_states: OptStates
    key: torch.Tensor
    env_state: Any
    timestep: TimeStep
    dones: torch.Tensor
    hstates: HiddenStates


class PPOTransition(NamedTuple):
    done: torch.Tensor
    action: torch.Tensor
    value: torch.Tensor
    reward: torch.Tensor
    log_prob: torch.Tensor
    obs: Any
    info: Dict[str, torch.Tensor]


class RNNPPOTransition(NamedTuple):
    done: torch.Tensor
    action: torch.Tensor
    value: torch.Tensor
    reward: torch.Tensor
    log_prob: torch.Tensor
    obs: Any
    hstates: HiddenStates
    info: Dict[str, torch.Tensor]


MavaState = TypeVar("MavaState")


class ExperimentOutput(NamedTuple):
    learner_state: Any
    episode_metrics: Dict[str, torch.Tensor]
    train_metrics: Dict[str, torch.Tensor]


LearnerFn = Callable[[Any], ExperimentOutput]
