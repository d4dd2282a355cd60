"""Metrics logging behind the reference's names (mava/utils/logger.py): LogEvent, MavaLogger (TRAIN metrics are
mean-reduced, everything else is described as mean/std/min/max, :72-79), MultiLogger, ConsoleLogger and the
marl-eval JsonLogger (:211-251).  Neptune and TensorBoard are third-party services / packages that are not in the
container; asking for them raises.

JSON wire format: the reference delegates to `marl_eval.json_tools.JsonLogger` (third-party, not vendored).  Its
published layout is restated here - parity unpinned, no fixture of it exists in the reference:

    {environment_name: {task_name: {algorithm_name: {"seed_<seed>": {
        "step_<eval_step>": {"step_count": <timestep>, "<metric>": [values...]},
        "absolute_metrics": {"<metric>": [values...]}}}}}}

written to <base_exp_path>/json/<system_name>/<unique_token>/metrics.json (or json/<json_path>/metrics.json), only the
metrics marl-eval plots (episode_return/mean -> "mean_episode_return", win_rate, steps_per_second), only for EVAL /
ABSOLUTE events.
"""
from __future__ import annotations

import json
import logging
import os
from datetime import datetime
from enum import Enum
from typing import Any, ClassVar, Dict, List

import numpy as np
import torch


class LogEvent(Enum):
    ACT = "actor"
    TRAIN = "trainer"
    EVAL = "evaluator"
    ABSOLUTE = "absolute"
    MISC = "misc"


def _to_numpy(x: Any) -> Any:
    return x.detach().float().cpu().numpy() if isinstance(x, torch.Tensor) else x


def describe(x: Any) -> Any:
    """Summary statistics of an array of metrics (mean, std, min, max); scalars pass through (:353-360)."""
    x = _to_numpy(x)
    if not isinstance(x, np.ndarray) or x.size <= 1:
        return x
    return {"mean": float(np.mean(x)), "std": float(np.std(x)), "min": float(np.min(x)), "max": float(np.max(x))}


def _flatten(data: Dict, sep: str, prefix: str = "") -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for k, v in data.items():
        key = f"{prefix}{sep}{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(_flatten(v, sep, key))
        else:
            out[key] = v
    return out


class BaseLogger:
    def log_stat(self, key: str, value: float, step: int, eval_step: int, event: LogEvent) -> None:
        raise NotImplementedError

    def log_dict(self, data: Dict, step: int, eval_step: int, event: LogEvent) -> None:
        for key, value in _flatten(data, "/").items():
            self.log_stat(key, value, step, eval_step, event)

    def stop(self) -> None:
        return None


class MultiLogger(BaseLogger):
    def __init__(self, loggers: List[BaseLogger]) -> None:
        self.loggers = loggers

    def log_stat(self, key, value, step, eval_step, event) -> None:
        for lg in self.loggers:
            lg.log_stat(key, value, step, eval_step, event)

    def log_dict(self, data, step, eval_step, event) -> None:
        for lg in self.loggers:
            lg.log_dict(data, step, eval_step, event)

    def stop(self) -> None:
        for lg in self.loggers:
            lg.stop()


class JsonLogger(BaseLogger):
    """Json logger for marl-eval (mava/utils/logger.py:211-251)."""

    _METRICS_TO_LOG: ClassVar[List[str]] = ["episode_return/mean", "win_rate", "steps_per_second"]

    def __init__(self, cfg, unique_token: str) -> None:
        path = os.path.join(cfg.logger.base_exp_path, f"{get_logger_path(cfg, 'json')}/{unique_token}")
        if cfg.logger.kwargs.json_path is not None:
            path = os.path.join(cfg.logger.base_exp_path, "json", cfg.logger.kwargs.json_path)
        os.makedirs(path, exist_ok=True)
        self.file = os.path.join(path, "metrics.json")
        self.env, self.task = str(cfg.env.env_name), str(cfg.env.scenario.task_name)
        self.algo, self.run = str(cfg.logger.system_name), f"seed_{int(cfg.system.seed)}"
        self.data: Dict[str, Any] = {}
        if os.path.isfile(self.file):  # several experiments may share one file (json_path)
            with open(self.file) as f:
                self.data = json.load(f)
        self.run_data = (self.data.setdefault(self.env, {}).setdefault(self.task, {}).setdefault(self.algo, {})
                         .setdefault(self.run, {"absolute_metrics": {}}))

    def log_stat(self, key: str, value: float, step: int, eval_step: int, event: LogEvent) -> None:
        if key not in self._METRICS_TO_LOG:
            return
        if "/" in key:  # <metric>/<agg> -> <agg>_<metric>
            key = "_".join(reversed(key.split("/")))
        value = _to_numpy(value)
        value = value.tolist() if isinstance(value, np.ndarray) else value
        values = value if isinstance(value, list) else [float(value)]
        if event == LogEvent.ABSOLUTE:
            self.run_data["absolute_metrics"][key] = values
        elif event == LogEvent.EVAL:
            step_rec = self.run_data.setdefault(f"step_{int(eval_step)}", {"step_count": int(step)})
            step_rec["step_count"] = int(step)
            step_rec[key] = values
        else:
            return
        with open(self.file, "w") as f:
            json.dump(self.data, f, indent=4)


class ConsoleLogger(BaseLogger):
    """Logger for writing to stdout (:254-302), without the colour codes."""

    def __init__(self, cfg, unique_token: str) -> None:
        self.logger = logging.getLogger("mava_amd")
        if not self.logger.handlers:
            ch = logging.StreamHandler()
            ch.setFormatter(logging.Formatter("%(message)s"))
            self.logger.addHandler(ch)
        self.logger.setLevel("INFO")

    def log_stat(self, key, value, step, eval_step, event) -> None:
        self.logger.info(f"{event.value.upper()} - {key.replace('_', ' ').capitalize()}: {float(value):.3f}")

    def log_dict(self, data, step, eval_step, event) -> None:
        flat = _flatten(data, " ")
        parts = []
        for k, v in flat.items():
            v = _to_numpy(v)
            v = v.item() if isinstance(v, np.ndarray) and v.size == 1 else v
            parts.append(f"{k.replace('_', ' ').capitalize()}: {v:.3f}" if isinstance(v, float) else f"{k.replace('_', ' ').capitalize()}: {v}")
        self.logger.info(f"{event.value.upper()} - " + " | ".join(parts))


def get_logger_path(config, logger_type: str) -> str:
    return f"{logger_type}/{config.logger.system_name}"  # :347-349


def _make_multi_logger(cfg) -> BaseLogger:
    loggers: List[BaseLogger] = []
    unique_token = datetime.now().strftime("%Y%m%d%H%M%S")
    if cfg.logger.use_neptune or cfg.logger.use_tb:
        raise NotImplementedError("Neptune / TensorBoard logging needs third-party packages that are not available here")
    if cfg.logger.use_json:
        loggers.append(JsonLogger(cfg, unique_token))
    if cfg.logger.use_console:
        loggers.append(ConsoleLogger(cfg, unique_token))
    return MultiLogger(loggers)


class MavaLogger:
    """The main logger for Mava systems (:44-107)."""

    def __init__(self, config) -> None:
        self.logger = _make_multi_logger(config)
        self.cfg = config

    def log(self, metrics: Dict, t: int, t_eval: int, event: LogEvent) -> None:
        metrics = dict(metrics)
        if "won_episode" in metrics:
            metrics = self.calc_winrate(metrics, event)
        if event == LogEvent.TRAIN:
            metrics = {k: float(np.mean(_to_numpy(v))) for k, v in metrics.items()}  # only mean losses matter
        else:
            metrics = {k: describe(v) for k, v in metrics.items()}
        self.logger.log_dict(metrics, t, t_eval, event)

    def calc_winrate(self, episode_metrics: Dict, event: LogEvent) -> Dict:
        n_episodes = self.cfg.arch.num_eval_episodes * (10 if event == LogEvent.ABSOLUTE else 1)
        n_won = float(np.sum(_to_numpy(episode_metrics["won_episode"])))
        episode_metrics["win_rate"] = (n_won / n_episodes) * 100
        episode_metrics.pop("won_episode")
        return episode_metrics

    def stop(self) -> None:
        self.logger.stop()
