"""rec_mappo: recurrent PPO (GRU actor/critic) with a centralised critic.

Same entry points as mava/systems/ppo/rec_mappo.py: `learner_setup(env, keys, config)` (:437-576) and
`run_experiment(config)` (:579-731); differs from rec_ippo only by `centralised_critic=True` (:470) and
`add_global_state=True` (:594).
"""
from __future__ import annotations

from ... import envs as environments
from ... import rec_learner as _learner
from . import anakin

CENTRALISED_CRITIC = True


def learner_setup(env, keys, config, device=None):
    return _learner.learner_setup(env, keys, config, CENTRALISED_CRITIC, device)


def run_experiment(config, log=None) -> float:
    if config.system.get("recurrent_chunk_size", None) is None:  # rec_mappo.py:586-587
        config.system.recurrent_chunk_size = config.system.rollout_length
    else:
        assert config.system.rollout_length % config.system.recurrent_chunk_size == 0, (
            "Rollout length must be divisible by recurrent chunk size."  # rec_mappo.py:589-591
        )
    return anakin.run_experiment(config, learner_setup, environments.make, add_global_state=True, log=log, recurrent=True)


if __name__ == "__main__":
    import sys

    from ...config import compose

    print(run_experiment(compose("default_rec_mappo", sys.argv[1:])))
