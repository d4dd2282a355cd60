"""Host utilities: checkpoint container + retention policy (mava/utils/checkpointing.py) and the MavaLogger /
marl-eval JSON writer (mava/utils/logger.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from mava_amd.config import compose
from mava_amd.types import HiddenStates, LearnerState, OptStates, Params, AdamState, TimeStep


def _state(scale: float) -> LearnerState:
    tree = lambda n: {"params": {"torso": {"Dense_0": {"kernel": torch.full((3, 4), scale * n), "bias": torch.zeros(4)}}}}
    opt = AdamState(torch.tensor(7, dtype=torch.int32), tree(3), tree(4))
    ts = TimeStep(torch.zeros(2, dtype=torch.int8), torch.zeros(2, 2), torch.ones(2, 2), None, {"episode_metrics": {"episode_return": torch.zeros(2)}})
    return LearnerState(Params(tree(1), tree(2)), OptStates(opt, opt), torch.tensor([42, 0]), {"step_count": torch.arange(4)}, ts)


def test_checkpointer_roundtrip_and_retention(tmp_path, monkeypatch):
    from mava_amd.utils.checkpointing import CHECKPOINTER_VERSION, Checkpointer

    monkeypatch.chdir(tmp_path)
    ck = Checkpointer(model_name="ff_mappo", metadata={"system": {"seed": 42}, "obj": object()}, checkpoint_uid="run1", max_to_keep=2,
                      keep_period=1000)
    assert os.path.isdir(tmp_path / "checkpoints" / "ff_mappo" / "run1")  # <cwd>/<rel_dir>/<model_name>/<uid>
    assert ck.get_cfg()["checkpointer_version"] == CHECKPOINTER_VERSION and ck.get_cfg()["system"]["seed"] == 42
    for step, ret in ((1000, 5.0), (1500, 1.0), (2000, 3.0), (2500, 4.0)):
        assert ck.save(step, _state(float(step)), episode_return=ret)
    # best two by episode_return are kept (1000: 5.0, 2500: 4.0); 2000 survives through keep_period; 1500 is gone
    assert ck.all_steps() == [1000, 2000, 2500]
    assert ck.best_step() == 1000 and ck.latest_step() == 2500

    reader = Checkpointer(model_name="ff_mappo", checkpoint_uid="run1")
    template = _state(0.0).params
    params, hs = reader.restore_params(template)
    assert isinstance(params, Params) and hs is None
    k = params.actor_params["params"]["torso"]["Dense_0"]["kernel"]
    assert torch.equal(k, torch.full((3, 4), 2500.0))
    params, _ = reader.restore_params(template, timestep=1000)
    assert torch.equal(params.critic_params["params"]["torso"]["Dense_0"]["kernel"], torch.full((3, 4), 2000.0))
    raw = reader.restore_learner_state_raw(2000)
    assert set(raw) == {"params", "opt_states", "key", "env_state", "timestep"}
    assert int(raw["opt_states"]["actor_opt_state"]["count"]) == 7 and raw["timestep"]["observation"] is None
    assert torch.equal(raw["env_state"]["step_count"], torch.arange(4))


def test_checkpointer_hidden_states_and_interval(tmp_path, monkeypatch):
    from mava_amd.utils.checkpointing import Checkpointer

    monkeypatch.chdir(tmp_path)
    ck = Checkpointer(model_name="rec_mappo", checkpoint_uid="u", save_interval_steps=100, max_to_keep=None)
    st = {"params": _state(1.0).params, "hstates": HiddenStates(torch.ones(2, 3, 128), torch.zeros(2, 3, 128))}
    assert ck.save(100, st)
    assert not ck.save(150, st)  # inside the save interval
    assert ck.save(200, st)
    params, hs = ck.restore_params(_state(0.0).params, restore_hstates=True, THiddenState=HiddenStates)
    assert isinstance(hs, HiddenStates) and torch.equal(hs.policy_hidden_state, torch.ones(2, 3, 128))
    with open(tmp_path / "checkpoints" / "rec_mappo" / "u" / "metadata.json", "w") as f:
        json.dump({"checkpointer_version": 2.0}, f)
    with pytest.raises(AssertionError):
        ck.restore_params(_state(0.0).params)


def test_mava_logger_json_wire_format(tmp_path):
    from mava_amd.utils.logger import LogEvent, MavaLogger, describe

    cfg = compose("default_ff_mappo", ["env/scenario=tiny-4ag"])
    cfg.logger.base_exp_path = str(tmp_path)
    cfg.logger.use_json, cfg.logger.use_console = True, False
    cfg.logger.kwargs.json_path = "shared"
    lg = MavaLogger(cfg)
    rets = torch.tensor([1.0, 2.0, 3.0, 6.0])
    lg.log({"episode_return": rets, "episode_length": torch.tensor([5, 5, 5, 5]), "steps_per_second": 1234.5}, 4096, 0, LogEvent.EVAL)
    lg.log({"total_loss": torch.tensor([[1.0, 3.0]]), "value_loss": np.array([2.0])}, 4096, 0, LogEvent.TRAIN)  # not written to json
    lg.log({"episode_return": rets * 2, "won_episode": np.array([1, 0, 1, 1])}, 8192, 1, LogEvent.EVAL)
    lg.log({"episode_return": rets * 3}, 8192, 1, LogEvent.ABSOLUTE)
    lg.stop()
    with open(tmp_path / "json" / "shared" / "metrics.json") as f:
        data = json.load(f)
    run = data[str(cfg.env.env_name)][str(cfg.env.scenario.task_name)]["ff_mappo"]["seed_42"]
    assert run["step_0"] == {"step_count": 4096, "mean_episode_return": [3.0], "steps_per_second": [1234.5]}
    assert run["step_1"]["mean_episode_return"] == [6.0]
    assert run["step_1"]["win_rate"] == [pytest.approx(3 / cfg.arch.num_eval_episodes * 100)]
    assert run["absolute_metrics"] == {"mean_episode_return": [9.0]}
    d = describe(rets)
    assert d == {"mean": 3.0, "std": pytest.approx(float(np.std([1, 2, 3, 6]))), "min": 1.0, "max": 6.0}
    assert describe(torch.tensor(2.0)) == pytest.approx(2.0)
    cfg.logger.use_tb = True
    with pytest.raises(NotImplementedError):
        MavaLogger(cfg)
