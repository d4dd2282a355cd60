"""Experiment driver shared by the four PPO systems: the host loop of
mava/systems/ppo/ff_mappo.py:435-553 (run_experiment) - train for num_updates_per_eval updates,
synchronise, report steps_per_second with the reference's own accounting, evaluate, repeat.
Logging is reduced to the metric names of SURVEY.md §5.5 on stdout/JSON (mava_amd/utils/logger.py holds the
MavaLogger / marl-eval JSON writer); checkpointing uses mava_amd/utils/checkpointing.py.
"""
from __future__ import annotations

import copy
import json
import time
from typing import Any, Callable, Dict, Optional

import torch

from ...config import Config, check_total_timesteps
from ...learner import get_final_step_metrics


def _unreplicate_n_dims(tree: Any, n: int = 2) -> Any:
    """mava/utils/jax_utils.py:52-59: x[0, 0] on every array leaf with at least n leading dims."""
    if isinstance(tree, torch.Tensor):
        return tree[(0,) * n] if tree.dim() >= n else tree
    if hasattr(tree, "_asdict"):
        return type(tree)(**{k: _unreplicate_n_dims(v, n) for k, v in tree._asdict().items()})
    if isinstance(tree, dict):
        return {k: _unreplicate_n_dims(v, n) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_unreplicate_n_dims(v, n) for v in tree)
    return tree


def _tree_to(tree: Any, device) -> Any:
    if isinstance(tree, torch.Tensor):
        return tree.to(device)
    if hasattr(tree, "_asdict"):
        return type(tree)(**{k: _tree_to(v, device) for k, v in tree._asdict().items()})
    if isinstance(tree, dict):
        return {k: _tree_to(v, device) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_tree_to(v, device) for v in tree)
    return tree


def _tree_clone(tree: Any) -> Any:
    if isinstance(tree, torch.Tensor):
        return tree.clone()
    if isinstance(tree, dict):
        return {k: _tree_clone(v) for k, v in tree.items()}
    return copy.deepcopy(tree)


def run_experiment(_config: Config, learner_setup: Callable, make_env: Callable, add_global_state: bool,
                   log: Optional[Callable[[Dict[str, Any]], None]] = None, recurrent: bool = False) -> float:
    config = copy.deepcopy(_config)
    import torch.distributed as dist

    n_devices = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if n_devices > 1 else 0
    env, eval_env = make_env(config, add_global_state=add_global_state)

    seed = int(config.system.seed)
    key, key_e, actor_key, critic_key = seed, seed + 1, seed + 2, seed + 3  # stands in for random.split(PRNGKey(seed), 4)

    config = check_total_timesteps(config, n_devices)
    assert config.system.num_updates > config.arch.num_evaluation, (
        "Number of updates per evaluation must be less than total number of updates."  # ff_mappo.py:462-464
    )
    config.system.num_updates_per_eval = config.system.num_updates // config.arch.num_evaluation
    steps_per_rollout = (n_devices * config.system.num_updates_per_eval * config.system.rollout_length
                         * config.system.update_batch_size * config.arch.num_envs)  # ff_mappo.py:468-474

    learn, actor_network, learner_state = learner_setup(env, (key, actor_key, critic_key), config)

    from ...evaluator import get_eval_fn, make_ff_eval_act_fn, make_rec_eval_act_fn

    # evaluator.py:175-207: feed-forward or recurrent act function over actor_network.apply
    if recurrent:
        act_fn = make_rec_eval_act_fn(actor_network.apply, config)
        init_act_state = {"hidden_state": torch.zeros((eval_env.num_envs, eval_env.num_agents, int(config.network.get("hidden_state_dim", 128))),
                                                      device=eval_env.device)}  # rec_mappo.py:623-629: ScannedRNN.initialize_carry(hidden_state_dim)
    else:
        act_fn, init_act_state = make_ff_eval_act_fn(actor_network.apply, config), None
    evaluator = get_eval_fn(eval_env, act_fn, config, absolute_metric=False)

    # checkpointing (ff_mappo.py:440-459 load, :520-531 save)
    ck = config.logger.checkpointing
    checkpointer = None
    if bool(ck.save_model) and rank == 0:
        from ...utils.checkpointing import Checkpointer

        checkpointer = Checkpointer(metadata=config.to_container() if hasattr(config, "to_container") else None,
                                    model_name=str(config.logger.system_name),
                                    **{k: (v if v != "" else None) for k, v in dict(ck.save_args).items()})
    if bool(ck.load_model):
        from ...utils.checkpointing import Checkpointer

        loaded = Checkpointer(model_name=str(config.logger.system_name), **{k: v for k, v in dict(ck.load_args).items()})
        restored_params, _ = loaded.restore_params(input_params=learner_state.params)
        restored_params = _tree_to(restored_params, eval_env.device)  # checkpoints hold host tensors
        learner_state = learner_state._replace(params=restored_params)  # learn() adopts trees that do not alias its buffers

    # Logging (ff_mappo.py:476-480): the MavaLogger of the configuration (console / marl-eval JSON) on rank 0, unless
    # the caller takes the per-evaluation records itself through `log`.
    from ...utils.logger import LogEvent, MavaLogger

    logger = MavaLogger(config) if (log is None and rank == 0) else None

    def emit(rec: Dict[str, Any]) -> None:
        if rank == 0 and log is not None:
            log(rec)

    eval_return = 0.0
    max_episode_return = -float("inf")
    best_params = None
    t = 0
    for eval_step in range(int(config.arch.num_evaluation)):
        # The reference evaluates `learner_state.params` of the state that went INTO learn() (ff_mappo.py:513-518; the
        # state is only advanced at :535 - SURVEY Q3).  Here learn() updates the parameter buffers in place and the
        # state's leaves are views of them, so the pre-update parameters are snapshotted first (outside the timed window).
        trained_params = _tree_clone(learner_state.params.actor_params)
        torch.cuda.synchronize()
        start = time.time()
        out = learn(learner_state)
        torch.cuda.synchronize()  # jax.block_until_ready, ff_mappo.py:498
        elapsed = time.time() - start
        t = int(steps_per_rollout * (eval_step + 1))
        ep_metrics, ep_completed = get_final_step_metrics(out.episode_metrics)
        ep_metrics = dict(ep_metrics)
        ep_metrics["steps_per_second"] = steps_per_rollout / elapsed
        rec: Dict[str, Any] = {"timestep": t, "steps_per_second": ep_metrics["steps_per_second"]}
        if ep_completed:
            rec["episode_return"] = float(ep_metrics["episode_return"].float().mean())
            rec["episode_length"] = float(ep_metrics["episode_length"].float().mean())
        for k, v in out.train_metrics.items():
            rec[k] = float(v.float().mean())  # TRAIN metrics are mean-reduced (mava/utils/logger.py:72-74)
        if logger is not None:  # separately: timestep, acting metrics, training metrics (ff_mappo.py:509-513)
            logger.log({"timestep": t}, t, eval_step, LogEvent.MISC)
            if ep_completed:
                logger.log(ep_metrics, t, eval_step, LogEvent.ACT)
            logger.log(out.train_metrics, t, eval_step, LogEvent.TRAIN)
        # evaluation uses the PRE-update parameters, exactly like the reference (ff_mappo.py:513-518 vs :535)
        eval_metrics = evaluator(trained_params, key_e + eval_step, init_act_state)
        if logger is not None:
            logger.log(eval_metrics, t, eval_step, LogEvent.EVAL)
        eval_return = float(eval_metrics["episode_return"].float().mean())
        rec["eval_episode_return"] = eval_return
        emit(rec)
        if bool(config.arch.absolute_metric) and max_episode_return <= eval_return:
            best_params = trained_params  # copy.deepcopy(trained_params), ff_mappo.py:537-539: already a snapshot
            max_episode_return = eval_return
        learner_state = out.learner_state
        if checkpointer is not None:
            # unreplicate_n_dims (jax_utils.py:52-59): drop the (device, update_batch) leading dims of the leaves
            unrep = _unreplicate_n_dims(learner_state)
            checkpointer.save(timestep=t, unreplicated_learner_state=unrep, episode_return=eval_return)

    # Measure the absolute metric (ff_mappo.py:546-553): the best parameters, 10x the evaluation episodes
    if bool(config.arch.absolute_metric) and best_params is not None:
        abs_evaluator = get_eval_fn(eval_env, act_fn, config, absolute_metric=True)
        abs_metrics = abs_evaluator(best_params, key, init_act_state)
        if logger is not None:
            logger.log(abs_metrics, t, int(config.arch.num_evaluation) - 1, LogEvent.ABSOLUTE)
        emit({"timestep": t, "absolute_episode_return": float(abs_metrics["episode_return"].float().mean())})
    if logger is not None:
        logger.stop()
    return eval_return
