"""Checkpointer with the interface of mava/utils/checkpointing.py:38-207 (save / restore_params / get_cfg) and
the retention policy the reference asks of Orbax's CheckpointManager (:79-86: best = max episode_return,
save_interval_steps, max_to_keep, keep_period).  The on-disk container is this build's own - Orbax is a
third-party format that is not in the container - but the directory scheme
`<cwd>/<rel_dir>/<model_name>/<checkpoint_uid>/<step>/` and the item / field names ("learner_state", "params",
"hstates", metadata["checkpointer_version"]) are the reference's:

    <uid>/metadata.json                      {"checkpointer_version": 1.0, **config}
    <uid>/<step>/learner_state.safetensors   every tensor leaf, keyed by its "/"-joined path in the state
    <uid>/<step>/learner_state.json          tree structure: non-tensor leaves and the kind of every container
    <uid>/<step>/metrics.json                {"episode_return": ...}
"""
from __future__ import annotations

import json
import os
import shutil
from datetime import datetime
from typing import Any, Dict, Optional, Tuple, Type

import torch
from safetensors.torch import load_file, save_file

CHECKPOINTER_VERSION = 1.0  # mava/utils/checkpointing.py:35


def _json_ready(obj: Any) -> Any:
    if isinstance(obj, dict):
        return {str(k): _json_ready(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_json_ready(v) for v in obj]
    if isinstance(obj, (bool, str, int, float, type(None))):
        return obj
    return str(obj)  # checkpointing.py:88-92 (get_json_ready)


def _flatten(node: Any, path: str, tensors: Dict[str, torch.Tensor]) -> Any:
    """Returns the JSON skeleton of `node`; tensor leaves go to `tensors` under their path."""
    if isinstance(node, torch.Tensor):
        tensors[path] = node.detach().to("cpu").contiguous().clone()
        return {"__tensor__": path}
    if hasattr(node, "_asdict"):  # NamedTuple state types
        return {"__fields__": {k: _flatten(v, f"{path}/{k}" if path else k, tensors) for k, v in node._asdict().items()}}
    if isinstance(node, dict):
        return {"__dict__": {str(k): _flatten(v, f"{path}/{k}" if path else str(k), tensors) for k, v in node.items()}}
    if isinstance(node, (list, tuple)):
        return {"__list__": [_flatten(v, f"{path}/{i}" if path else str(i), tensors) for i, v in enumerate(node)]}
    if isinstance(node, (bool, int, float, str, type(None))):
        return {"__value__": node}
    return {"__value__": str(node)}


def _unflatten(skel: Any, tensors: Dict[str, torch.Tensor]) -> Any:
    if "__tensor__" in skel:
        return tensors[skel["__tensor__"]]
    if "__fields__" in skel:
        return {k: _unflatten(v, tensors) for k, v in skel["__fields__"].items()}
    if "__dict__" in skel:
        return {k: _unflatten(v, tensors) for k, v in skel["__dict__"].items()}
    if "__list__" in skel:
        return [_unflatten(v, tensors) for v in skel["__list__"]]
    return skel["__value__"]


class Checkpointer:
    """Model checkpointer for saving and restoring the `learner_state` (mava/utils/checkpointing.py:38)."""

    def __init__(self, model_name: str, metadata: Optional[Dict] = None, rel_dir: str = "checkpoints",
                 checkpoint_uid: Optional[str] = None, save_interval_steps: int = 1, max_to_keep: Optional[int] = 1,
                 keep_period: Optional[int] = None):
        uid = checkpoint_uid if checkpoint_uid else datetime.now().strftime("%Y%m%d%H%M%S")
        self.directory = os.path.join(os.getcwd(), rel_dir, model_name, uid)
        self.save_interval_steps, self.max_to_keep, self.keep_period = int(save_interval_steps), max_to_keep, keep_period
        os.makedirs(self.directory, exist_ok=True)
        meta_path = os.path.join(self.directory, "metadata.json")
        if metadata is not None and hasattr(metadata, "to_container"):
            metadata = metadata.to_container()
        if not os.path.exists(meta_path) or metadata is not None:
            meta = {"checkpointer_version": CHECKPOINTER_VERSION, **(_json_ready(metadata) if metadata is not None else {})}
            with open(meta_path, "w") as f:
                json.dump(meta, f)
        self._last_saved: Optional[int] = None

    # ------------------------------------------------------------------------------------- bookkeeping
    def all_steps(self):
        return sorted(int(d) for d in os.listdir(self.directory) if d.isdigit())

    def latest_step(self) -> Optional[int]:
        steps = self.all_steps()
        return steps[-1] if steps else None

    def _metric(self, step: int) -> float:
        with open(os.path.join(self.directory, str(step), "metrics.json")) as f:
            return float(json.load(f)["episode_return"])

    def best_step(self) -> Optional[int]:
        steps = self.all_steps()
        return max(steps, key=lambda s: (self._metric(s), s)) if steps else None

    def _prune(self) -> None:
        """Keep the `max_to_keep` best checkpoints by episode_return (best_mode="max"), never deleting steps with
        step % keep_period == 0."""
        if self.max_to_keep is None:
            return
        steps = self.all_steps()
        ranked = sorted(steps, key=lambda s: (self._metric(s), s), reverse=True)
        for s in ranked[self.max_to_keep:]:
            if self.keep_period and s % self.keep_period == 0:
                continue
            shutil.rmtree(os.path.join(self.directory, str(s)), ignore_errors=True)

    # -------------------------------------------------------------------------------------------- API
    def save(self, timestep: int, unreplicated_learner_state: Any, episode_return: float = 0.0) -> bool:
        """Save the (unreplicated) learner state; returns whether a checkpoint was written (:117-146)."""
        timestep = int(timestep)
        if self._last_saved is not None and timestep - self._last_saved < self.save_interval_steps:
            return False
        tensors: Dict[str, torch.Tensor] = {}
        skeleton = _flatten(unreplicated_learner_state, "", tensors)
        tmp = os.path.join(self.directory, f".tmp_{timestep}")
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        save_file(tensors, os.path.join(tmp, "learner_state.safetensors"))
        with open(os.path.join(tmp, "learner_state.json"), "w") as f:
            json.dump(skeleton, f)
        with open(os.path.join(tmp, "metrics.json"), "w") as f:
            json.dump({"episode_return": float(episode_return)}, f)
        final = os.path.join(self.directory, str(timestep))
        shutil.rmtree(final, ignore_errors=True)
        os.replace(tmp, final)  # a step directory appears whole or not at all
        self._last_saved = timestep
        self._prune()
        return True

    def restore_learner_state_raw(self, timestep: Optional[int] = None) -> Dict[str, Any]:
        step = int(timestep) if timestep else self.latest_step()
        if step is None:
            raise FileNotFoundError(f"no checkpoint under {self.directory}")
        d = os.path.join(self.directory, str(step))
        tensors = load_file(os.path.join(d, "learner_state.safetensors"))
        with open(os.path.join(d, "learner_state.json")) as f:
            return _unflatten(json.load(f), tensors)

    def restore_params(self, input_params: Any, timestep: Optional[int] = None, restore_hstates: bool = False,
                       THiddenState: Optional[Type] = None) -> Tuple[Any, Any]:  # noqa: N803
        """Restore the params (and the hidden states of recurrent systems) as the types of the inputs (:148-207)."""
        with open(os.path.join(self.directory, "metadata.json")) as f:
            version = json.load(f)["checkpointer_version"]
        assert (version // 1) == (CHECKPOINTER_VERSION // 1), (
            "Loaded checkpoint was created with a different major version of the checkpointer.")
        raw = self.restore_learner_state_raw(timestep)
        TParams = type(input_params)  # noqa: N806
        restored_params = TParams(**raw["params"])
        restored_hstates = None
        if restore_hstates and THiddenState is not None:
            restored_hstates = THiddenState(**raw["hstates"])
        return restored_params, restored_hstates

    def get_cfg(self) -> Dict[str, Any]:
        """The metadata of the checkpoint (:209-212)."""
        with open(os.path.join(self.directory, "metadata.json")) as f:
            return json.load(f)
