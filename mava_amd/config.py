"""Config composition with Mava's Hydra surface.

`compose("default_ff_mappo", ["env/scenario=tiny-4ag", "arch.num_envs=4096"])` returns a tree
with the same keys the reference reads (config.arch.*, config.system.*, config.network.*,
config.env.*, config.logger.*).  Supported override forms, as on the reference's command line
(mava/systems/ppo/ff_mappo.py:556-565): `group=choice` (e.g. `env=rware`, `network=mlp`,
`env/scenario=tiny-4ag`) and dotted `a.b.c=value` (YAML-typed values, `~`/`null` for None).
Like OmegaConf with struct mode off (ff_mappo.py:560) new keys may be added at run time.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Iterable, List, Optional

import yaml

_GROUPS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "groups.yaml")


class Config(dict):
    """dict with attribute access, nested (the subset of DictConfig behaviour the learner uses)."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = _wrap(v)

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_container(self) -> dict:
        return {k: (v.to_container() if isinstance(v, Config) else copy.deepcopy(v)) for k, v in self.items()}


def _wrap(v: Any) -> Any:
    if isinstance(v, dict) and not isinstance(v, Config):
        return Config({k: _wrap(x) for k, x in v.items()})
    return v


def _load_groups() -> dict:
    with open(_GROUPS_FILE) as f:
        return yaml.safe_load(f)


def _set_dotted(cfg: Config, dotted: str, value: Any) -> None:
    parts = dotted.split(".")
    node = cfg
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            node[p] = Config()
        node = node[p]
    node[parts[-1]] = _wrap(value)


def compose(config_name: str = "default_ff_mappo", overrides: Optional[Iterable[str]] = None) -> Config:
    groups = _load_groups()
    name = config_name[:-5] if config_name.endswith(".yaml") else config_name
    if name not in groups["roots"]:
        raise ValueError(f"unknown config '{config_name}' (have {sorted(groups['roots'])})")
    choice = dict(groups["roots"][name])
    scenario: Optional[str] = None
    dotted: List[str] = []
    for ov in list(overrides or []):
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        if k in ("env/scenario", "env.scenario") and "." not in v and v in groups["scenario"]:
            scenario = v
        elif k in choice and k in groups and (v in groups[k] or f"ppo/{v}" in groups[k]):
            choice[k] = v if v in groups[k] else f"ppo/{v}"
        else:
            dotted.append(ov.lstrip("+"))
    cfg = Config()
    for g in ("logger", "arch", "system", "network", "env"):
        if choice[g] not in groups[g]:
            raise ValueError(f"unknown {g} choice '{choice[g]}'")
        cfg[g] = _wrap(copy.deepcopy(groups[g][choice[g]]))
    cfg.env.pop("_", None)
    scen = scenario or cfg.env.pop("default_scenario")
    cfg.env.pop("default_scenario", None)
    cfg.env["scenario"] = _wrap(copy.deepcopy(groups["scenario"][scen]))
    for ov in dotted:
        k, v = ov.split("=", 1)
        _set_dotted(cfg, k, yaml.safe_load(v) if v != "" else "")
    return cfg


def check_total_timesteps(config: Config, n_devices: int) -> Config:
    """mava/utils/total_timestep_checker.py:21-49: derive total_timesteps <-> num_updates."""
    s = config.system
    if s.total_timesteps is None:
        s.num_updates = int(s.num_updates)
        s.total_timesteps = int(n_devices * s.num_updates * s.rollout_length * s.update_batch_size * config.arch.num_envs)
    else:
        s.total_timesteps = int(s.total_timesteps)
        s.num_updates = int(
            s.total_timesteps // s.rollout_length // s.update_batch_size // config.arch.num_envs // n_devices
        )
    return config
