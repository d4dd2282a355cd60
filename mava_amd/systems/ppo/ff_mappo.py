"""ff_mappo: feed-forward PPO with a centralised critic (global_state input).

Same entry points as mava/systems/ppo/ff_mappo.py: `learner_setup(env, keys, config)` (:333-432)
and `run_experiment(config)` (:435-553); the only differences from ff_ippo are
`centralised_critic=True` (:354) and `add_global_state=True` (:442).
"""
from __future__ import annotations

from ... import envs as environments
from ... import learner as _learner
from . import anakin

CENTRALISED_CRITIC = True


def learner_setup(env, keys, config, device=None):
    return _learner.learner_setup(env, keys, config, CENTRALISED_CRITIC, device)


def run_experiment(config, log=None) -> float:
    return anakin.run_experiment(config, learner_setup, environments.make, add_global_state=True, log=log)


if __name__ == "__main__":
    import sys

    from ...config import compose

    print(run_experiment(compose("default_ff_mappo", sys.argv[1:])))
