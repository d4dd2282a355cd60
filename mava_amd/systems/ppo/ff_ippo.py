"""ff_ippo: feed-forward PPO with a decentralised critic (each agent's own view).

Same entry points as mava/systems/ppo/ff_ippo.py (`learner_setup`, `run_experiment`); differs from
ff_mappo only by `centralised_critic=False` and `add_global_state=False`.
"""
from __future__ import annotations

from ... import envs as environments
from ... import learner as _learner
from . import anakin

CENTRALISED_CRITIC = False


def learner_setup(env, keys, config, device=None):
    return _learner.learner_setup(env, keys, config, CENTRALISED_CRITIC, device)


def run_experiment(config, log=None) -> float:
    return anakin.run_experiment(config, learner_setup, environments.make, add_global_state=False, log=log)


if __name__ == "__main__":
    import sys

    from ...config import compose

    print(run_experiment(compose("default_ff_ippo", sys.argv[1:])))
