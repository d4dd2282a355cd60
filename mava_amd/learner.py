"""Feed-forward PPO learner (ff_ippo / ff_mappo) behind Mava's LearnerFn contract.

Reference: mava/systems/ppo/ff_mappo.py:45-330 (get_learner_fn: _env_step, _calculate_gae,
_update_epoch, _update_minibatch, learner_fn) and :333-432 (learner_setup); ff_ippo.py is the same
file with a decentralised critic.  The XLA program that `jax.pmap(learn)` compiles
is replaced by hand-written gfx950 kernels (libmavahip.so) launched from this module:

    per update:  T x { policy_step -> env.step }         (rollout, time-major trajectory in HBM)
                 critic forward on the last observation    (bootstrap value)
                 GAE reverse scan                          (mava_gae_f32)
                 K epochs x M minibatches x {adv stats, actor grad, critic grad, slab reduce,
                                             [RCCL all-reduce], clip+Adam}

Data layout (per GPU; U = update_batch_size replicas share parameters, ff_mappo.py:417-426):
    params / grads / Adam moments : flat f32 [actor | critic] (+4 loss scalars after the grads)
    trajectory, per replica       : time-major, (T+1, E, A, .) for observations (slot T is the
                                    bootstrap observation and becomes slot 0 of the next rollout),
                                    (T, E, A) for action / value / reward / log_prob / done
    shuffle                       : an int32 permutation of T*E row ids; minibatches are slices of
                                    it and rows are gathered inside the gradient kernels.
One process drives one GPU; with world_size > 1 the env axis is sharded over ranks and the flat
gradient is summed with one all-reduce per minibatch (ff_mappo.py:224-238 pmean "device").
"""
from __future__ import annotations

import math
import contextlib
import os
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import ops, parallel
from .guards import check_f16_range
from .networks import FeedForwardActor, FeedForwardValueNet, MLPTorso, make_action_head
from .types import (AdamState, ExperimentOutput, LearnerState, Observation, ObservationGlobalState, OptStates, Params,
                    TimeStep)

NUM_CU = 256  # MI355X compute units: one persistent gradient block per CU


def _dist_info() -> Tuple[int, int]:
    return parallel.rank_world()


class _Replica:
    """Trajectory + env buffers of one update-batch replica (the vmapped axis of ff_mappo.py:319)."""

    def __init__(self, env, T: int, n_upd: int, critic_uses_state: bool, device, continuous: bool = False):
        E, A = env.num_envs, env.num_agents
        self.env = env
        self.state = env.alloc_state()
        d = device
        self.agents_view = torch.empty((T + 1, E, A, env.obs_dim), device=d)
        self.global_state = torch.empty((T + 1, E, env.gs_tiles, env.state_dim), device=d)
        self.action_mask = torch.empty((T + 1, E, A, env.action_dim), dtype=torch.uint8, device=d)
        self.step_count = torch.empty((T + 1, E, A), dtype=torch.int32, device=d)
        # discrete: action index per agent; continuous head: action vector in (-1, 1) per agent
        self.action = (torch.empty((T, E, A, env.action_dim), device=d) if continuous
                       else torch.empty((T, E, A), dtype=torch.int32, device=d))
        self.value = torch.empty((T, E, A), device=d)
        self.reward = torch.empty((T, E, A), device=d)
        self.log_prob = torch.empty((T, E, A), device=d)
        self.done = torch.empty((T, E, A), dtype=torch.uint8, device=d)
        self.last_val = torch.empty((E, A), device=d)
        self.adv = torch.empty((T, E, A), device=d)
        self.tgt = torch.empty((T, E, A), device=d)
        # episode metrics of every update of one learn() call: (N_upd, T, E)
        self.info_return = torch.zeros((n_upd, T, E), device=d)
        self.info_length = torch.zeros((n_upd, T, E), dtype=torch.int32, device=d)
        self.info_terminal = torch.zeros((n_upd, T, E), dtype=torch.uint8, device=d)
        # the rollout writes the metrics of the CURRENT update here (fixed addresses: one captured HIP graph serves
        # every update index n); FFLearner._rollout copies them to slot n afterwards
        self.cur_return = torch.zeros((T, E), device=d)
        self.cur_length = torch.zeros((T, E), dtype=torch.int32, device=d)
        self.cur_terminal = torch.zeros((T, E), dtype=torch.uint8, device=d)
        self.last_reward = torch.zeros((E, A), device=d)
        self.last_done = torch.zeros((E, A), dtype=torch.uint8, device=d)

    def obs_slot(self, t: int) -> Dict[str, torch.Tensor]:
        return {"agents_view": self.agents_view[t], "global_state": self.global_state[t],
                "action_mask": self.action_mask[t], "step_count": self.step_count[t]}


class FFLearner:
    def __init__(self, env, config, centralised_critic: bool, device: Optional[torch.device] = None):
        self.config = config
        self.centralised = centralised_critic
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world = _dist_info()
        s, arch = config.system, config.arch
        self.E, self.U, self.T = int(arch.num_envs), int(s.update_batch_size), int(s.rollout_length)
        self.K, self.M = int(s.ppo_epochs), int(s.num_minibatches)
        if (self.T * self.E) % self.M:
            raise ValueError("rollout_length * num_envs must be divisible by num_minibatches (ff_mappo.py:277-279)")
        self.n_upd = int(s.get("num_updates_per_eval", 1))
        # one env object per replica; global env ids are disjoint over (rank, replica)
        self.reps: List[_Replica] = []
        if env.num_envs != self.E:
            raise ValueError(f"env.num_envs={env.num_envs} != arch.num_envs={self.E}")
        if centralised_critic and not getattr(env, "add_global_state", False):
            raise ValueError("Global state must be provided to the centralised critic.")  # networks.py:196-197
        # ff_mappo.py:348-350: the action head of the configuration, sized by the env's action dimension
        action_head = make_action_head(config.network.get("action_head", None), env.action_dim)
        self.continuous = type(action_head).__name__ == "ContinuousActionHead"
        self.min_scale = float(getattr(action_head, "min_scale", 1e-3))
        for u in range(self.U):
            rep_env = env.clone(env_offset=getattr(env, "env_offset", 0) + (self.rank * self.U + u) * self.E)
            self.reps.append(_Replica(rep_env, self.T, self.n_upd, centralised_critic, self.device, self.continuous))
        env0 = self.reps[0].env
        self.A, self.nA = env0.num_agents, env0.action_dim
        config.system.num_agents = self.A  # ff_mappo.py:341
        self.Oa = env0.obs_dim
        if centralised_critic:
            self.Oc = env0.state_dim
            self.critic_share = self.A if env0.global_state_shared else 1
        else:
            self.Oc = self.Oa
            self.critic_share = 1

        # networks (ff_mappo.py:347-354) - hydra.utils.instantiate is replaced by direct construction
        net = config.network
        # The default configuration (network/mlp.yaml: [128, 128] relu torsos, observation-independent log_std) runs on the
        # fused kernels; any other torso (layer sizes, tanh, layer norm, CNNTorso) or ContinuousActionHead(
        # independent_std=False) runs on the general layer-wise path (mava_amd/generic_networks.py).
        from . import generic_networks as gn

        self.generic = (not gn.is_default_mlp(net.actor_network.pre_torso) or not gn.is_default_mlp(net.critic_network.pre_torso)
                        or (self.continuous and not action_head.independent_std))
        if self.generic:
            if (self.E * env0.num_agents) % 32 or ((self.T * self.E // self.M) * env0.num_agents) % 32:
                raise ValueError("the general network path needs num_envs * num_agents and the agent rows of a minibatch to be "
                                 "multiples of 32")
            obs_shape = getattr(env0, "obs_shape", None)
            state_shape = getattr(env0, "state_shape", None) if centralised_critic else obs_shape
            self.actor_network = gn.GenericActor(gn.torso_from_config(net.actor_network.pre_torso), action_head, self.Oa, obs_shape)
            self.critic_network = gn.GenericCritic(gn.torso_from_config(net.critic_network.pre_torso), centralised_critic, self.Oc,
                                                   state_shape)
        else:
            a_torso = MLPTorso(**{k: v for k, v in net.actor_network.pre_torso.items() if k != "_target_"})
            c_torso = MLPTorso(**{k: v for k, v in net.critic_network.pre_torso.items() if k != "_target_"})
            self.actor_network = FeedForwardActor(a_torso, action_head, self.Oa)
            self.critic_network = FeedForwardValueNet(c_torso, centralised_critic, self.Oc)
        self.Pa, self.Pc = self.actor_network.num_params, self.critic_network.num_params
        self.P = self.Pa + self.Pc

        d = self.device
        self.p = torch.zeros(self.P, device=d)
        self.m = torch.zeros(self.P, device=d)
        self.v = torch.zeros(self.P, device=d)
        self.count = torch.zeros(2, dtype=torch.int32, device=d)
        self.g = torch.zeros(self.P + 4, device=d)  # [actor grad | critic grad | actor_loss, entropy, value_loss, pad]
        self.seg_off = [0, self.Pa, self.P]
        self.seg_lr = [float(s.actor_lr), float(s.critic_lr)]

        self.Rb = self.T * self.E // self.M  # env rows per minibatch
        ntiles = (self.Rb * self.A + 31) // 32
        # The gradient kernels are persistent blocks that own a CU each (all of its registers and LDS): with several ranks
        # a few CUs stay free of them, so that RCCL's all-reduce kernel of the actor's slice can really run WHILE the
        # critic's gradient kernel does (parallel.allreduce_sum_async below); system.rccl_cus / MAVA_RCCL_CUS, default 8
        # with a process group (a ~0.3 MB message uses a handful of channels), 0 on a single rank.
        self.rccl_cus = int(s.get("rccl_cus", None) if s.get("rccl_cus", None) is not None
                            else os.environ.get("MAVA_RCCL_CUS", "8" if self.world > 1 else "0"))
        self.n_slab = max(1, min(NUM_CU - max(0, min(self.rccl_cus, NUM_CU - 1)), ntiles))
        self.slab_a = torch.zeros((self.n_slab, self.Pa + 2), device=d)
        self.slab_c = torch.zeros((self.n_slab, self.Pc + 2), device=d)
        self.stats = torch.zeros((ops.lib().mava_adv_stats_blocks(), 2), dtype=torch.float64, device=d)
        self.train_metrics = torch.zeros((self.n_upd, self.K, self.M, 4), device=d)
        self.perm_count = 0  # epoch permutations drawn so far (counter of mava_permutation_i32)
        self.t_global = 0  # env steps taken per env so far (Philox step counter)
        self.ent_step = 0  # minibatches trained so far: counter of the continuous head's entropy sample
        # the same counter on the device: the kernels add it to their relative step, so a rollout captured in a HIP
        # graph replays with the next counters
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=d)
        # Single-rank jobs only: with a process group the replay measured SLOWER (RCCL, one rank, forced exchange:
        # 22.5 vs 22.1 ms per update; two gloo ranks on one card: 100x slower), see tools/graph_pg_rehearsal.py.
        # MAVA_GRAPH_ROLLOUT=0 turns it off, =force keeps it on with several ranks.
        mode = os.environ.get("MAVA_GRAPH_ROLLOUT", "1")
        self.graph_rollout = (mode == "force" or (mode != "0" and self.world == 1)) and not self.generic
        self._graphs: Dict[int, Any] = {}  # keyed by the seed: ONE graph for every update index n
        self._graph_seen: set = set()
        self.seed = int(s.seed)
        self.timers: Optional[Dict[str, list]] = None  # bench.py: name -> [(start_event, end_event)]
        # Arithmetic of the matrix products (not a Mava key): "f16x2" (default) = every operand split into two f16
        # terms, three f16 MFMAs per product with f32 accumulation (ppo_train_h2.hip, rollout_h2.hip: ~22 mantissa bits
        # per operand, the 1e-4 gradient parity tests run on it) for the shapes those kernels instantiate; "f32" = the
        # exact-f32 MFMA kernels everywhere.  system.matmul_mode / MAVA_MATMUL select it.
        self.matmul_mode = str(s.get("matmul_mode", None) or os.environ.get("MAVA_MATMUL", "f16x2"))
        if self.matmul_mode not in ("f16x2", "f32"):
            raise ValueError(f"system.matmul_mode must be 'f16x2' or 'f32', got {self.matmul_mode!r}")
        # the library keeps no process-wide settings: arithmetic mode, critic aggregation and the kernels' own workspaces
        # belong to this learner's context handle (include/mava_hip.h mava_ctx_*)
        self.ctx = ops.Ctx(self.matmul_mode, critic_aggregation=os.environ.get("MAVA_CRITIC_AGGREGATION", "1") != "0")
        # MAVA_TRAIN_VARIANT=1: the four-wave gradient kernels only (A/B measurements against the eight-wave actor kernel)
        self.ctx.set(self.ctx.TRAIN_VARIANT, int(os.environ.get("MAVA_TRAIN_VARIANT", "0")))
        for net_ in (self.actor_network, self.critic_network):
            net_.ctx = self.ctx
        # the whole rollout in one launch (rollout_h2.hip) when the shape is instantiated; MAVA_FUSED_ROLLOUT=0 keeps
        # the per-step kernels
        # (mava_rollout_ff_f32 hard-codes the synthetic generator's env phase: only an env that declares the capability -
        # SyntheticRware.supports_fused_rollout - is routed there; any other MarlEnv keeps the per-step kernels)
        self.fused_rollout = (self.matmul_mode == "f16x2" and not self.continuous and not self.generic
                              and os.environ.get("MAVA_FUSED_ROLLOUT", "1") != "0"
                              and bool(getattr(env0, "supports_fused_rollout", False))
                              and int(getattr(env0, "synth_state_dim", 0)) == 0
                              and (not centralised_critic or (env0.gs_tiles == 1 and env0.global_state_shared)))
        self._learn_calls = 0  # guards.check_f16_range: the previous call's metrics are checked from the second call on
        # One rank: nothing sits between the gradient kernels and Adam, so both slab sums, clip + Adam, the count increment and
        # the wide critic's W1 re-split run as TWO launches (ops.ppo_finish) instead of six.  With several replicas every
        # replica's gradient kernels write their own rows of the slab matrices and the one sum runs over all U * n_slab rows
        # (fixed order: replica 0's slabs, then replica 1's, ...; the mean over replicas of ff_mappo.py:232-238 is grad_scale).
        self.fused_tail = (self.world == 1 and not self.generic and os.environ.get("MAVA_FUSED_TAIL", "1") != "0")
        self._finish_ws = ops.ppo_finish_workspace(self.Pa, self.Pc, d) if self.fused_tail else None
        # (several ranks: the slab sums stay separate launches around the split all-reduce; the Adam launch still carries the
        # count increment and the W1 re-split: mava_ppo_finish_f32 with n_slab = 0)
        self._finish_ws_mr = (ops.ppo_finish_workspace(self.Pa, self.Pc, d)
                              if (not self.fused_tail and not self.generic and os.environ.get("MAVA_FUSED_TAIL", "1") != "0") else None)
        if self.fused_tail and self.U > 1:
            self.slab_a = torch.zeros((self.U * self.n_slab, self.Pa + 2), device=d)
            self.slab_c = torch.zeros((self.U * self.n_slab, self.Pc + 2), device=d)

    def _timed(self, name: str, fn, *args, **kwargs):
        """Run one kernel launch, optionally bracketed by HIP events on the launch stream."""
        if self.timers is None:
            return fn(*args, **kwargs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn(*args, **kwargs)
        b.record()
        self.timers.setdefault(name, []).append((a, b))
        return out

    # ------------------------------------------------------------------------------------ setup
    def init_params(self, actor_seed: int, critic_seed: int) -> None:
        self.p[: self.Pa].copy_(self.actor_network.init_flat(actor_seed))
        self.p[self.Pa :].copy_(self.critic_network.init_flat(critic_seed))
        self.m.zero_()
        self.v.zero_()
        self.count.zero_()

    def reset_envs(self) -> None:
        for rep in self.reps:
            rep.env.step_into(rep.state, 0, rep.obs_slot(0), is_reset=True)
        self.t_global = 0
        self.ent_step = 0
        self.step_dev.zero_()
        self.perm_count = 0

    # ---------------------------------------------------------------------------- state <-> views
    def _params_tree(self) -> Params:
        lead = (1, self.U)
        return Params(self.actor_network.tree(self.p[: self.Pa], lead), self.critic_network.tree(self.p[self.Pa :], lead))

    def _opt_tree(self) -> OptStates:
        lead = (1, self.U)
        out = []
        for i, (net, sl) in enumerate(((self.actor_network, slice(0, self.Pa)), (self.critic_network, slice(self.Pa, self.P)))):
            out.append(AdamState(self.count[i].expand(1, self.U), net.tree(self.m[sl], lead), net.tree(self.v[sl], lead)))
        return OptStates(*out)

    def _timestep(self, slot: int) -> TimeStep:
        def stack(f):
            return torch.stack([f(r) for r in self.reps], 0).unsqueeze(0)  # (1, U, E, ...)

        av = stack(lambda r: r.agents_view[slot])
        mask = stack(lambda r: r.action_mask[slot]).bool()
        sc = stack(lambda r: r.step_count[slot])
        if self.centralised:
            gs = stack(lambda r: r.global_state[slot].expand(-1, self.A, -1) if r.env.gs_tiles == 1 else r.global_state[slot])
            obs: Any = ObservationGlobalState(av, mask, gs, sc)
        else:
            obs = Observation(av, mask, sc)
        done = stack(lambda r: r.last_done[:, 0]).bool()
        step_type = torch.where(done, 2, 1).to(torch.int8)
        reward = stack(lambda r: r.last_reward)
        n = max(self.n_upd - 1, 0)
        extras = {"episode_metrics": {
            "episode_return": stack(lambda r: r.info_return[n, -1]),
            "episode_length": stack(lambda r: r.info_length[n, -1]),
            "is_terminal_step": stack(lambda r: r.info_terminal[n, -1]).bool()}}
        return TimeStep(step_type, reward, 1.0 - stack(lambda r: r.last_done).float(), obs, extras)

    def learner_state(self) -> LearnerState:
        key = torch.tensor([[[self.seed, self.t_global]] * self.U], dtype=torch.int64)  # (1, U, 2) host tensor
        env_state = {
            "step_count": torch.stack([r.state.step_count for r in self.reps], 0).unsqueeze(0),
            "running_count_episode_return": torch.stack([r.state.run_return for r in self.reps], 0).unsqueeze(0),
            "running_count_episode_length": torch.stack([r.state.run_length for r in self.reps], 0).unsqueeze(0),
            "episode_return": torch.stack([r.state.ep_return for r in self.reps], 0).unsqueeze(0),
            "episode_length": torch.stack([r.state.ep_length for r in self.reps], 0).unsqueeze(0),
        }
        return LearnerState(self._params_tree(), self._opt_tree(), key, env_state, self._timestep(0))

    def adopt(self, state: LearnerState) -> None:
        """Make the device buffers equal to `state` (no-op for leaves that already alias them), so that
        learn() is a function of its argument like the reference's pure learner_fn."""
        pa = self.actor_network.first_leaf(state.params.actor_params)
        if pa.data_ptr() != self.p.data_ptr():
            self.actor_network.flat_from_tree(state.params.actor_params, self.p[: self.Pa])
            self.critic_network.flat_from_tree(state.params.critic_params, self.p[self.Pa :])
            for i, (net, sl, st) in enumerate(((self.actor_network, slice(0, self.Pa), state.opt_states.actor_opt_state),
                                               (self.critic_network, slice(self.Pa, self.P), state.opt_states.critic_opt_state))):
                net.flat_from_tree(st.mu, self.m[sl])
                net.flat_from_tree(st.nu, self.v[sl])
                self.count[i] = int(st.count.reshape(-1)[0])

    # ------------------------------------------------------------------------------------ update
    def _rollout(self, n: int) -> None:
        """The launch-bound part of an update (2*T + 3 small kernels per replica): rollout, bootstrap value and GAE.
        The first call runs eagerly, the second is captured into a HIP graph, later ones replay it - the kernels
        read the moving step counter from step_dev, everything else they touch (incl. the metrics staging slots)
        is persistent, so the one graph serves every update index."""
        if self.fused_rollout and self._rollout_fused(n):
            self.t_global += self.T
            return
        if not self.graph_rollout or self.timers is not None:
            self._rollout_body()
            self._bootstrap_and_gae()
        else:
            key = self.seed
            graph = self._graphs.get(key)
            if graph is None and key in self._graph_seen:
                # thread_local: every launch of the rollout comes from this thread; RCCL's watchdog thread may query
                # its events meanwhile
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._rollout_body()
                    self._bootstrap_and_gae()
                self._graphs[key] = graph
            if graph is None:
                self._graph_seen.add(key)
                self._rollout_body()
                self._bootstrap_and_gae()
            else:
                graph.replay()
        for rep in self.reps:  # episode metrics of update n (ff_mappo.py:95,299); outside the graph: n moves
            rep.info_return[n].copy_(rep.cur_return)
            rep.info_length[n].copy_(rep.cur_length)
            rep.info_terminal[n].copy_(rep.cur_terminal)
        self.t_global += self.T

    def _rollout_fused(self, n: int) -> bool:
        """ff_mappo.py:76-139: one launch per replica for all T acting + env steps, the bootstrap value and GAE
        (mava_rollout_ff_f32).  False when the library does not instantiate the shape."""
        s = self.config.system
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        EA = self.E * self.A
        # The replicas' rollouts are independent (ff_mappo.py:389 vmaps over update_batch_size) and one replica of the reference's
        # default shape (2 x 2048 envs) fills half the CUs: replicas 1.. are launched on side streams, under replica 0's.
        main = torch.cuda.current_stream() if self.device.type == "cuda" else None
        side = self.U > 1 and self.timers is None and main is not None and os.environ.get("MAVA_ROLLOUT_STREAMS", "1") != "0"
        if side and not hasattr(self, "_roll_streams"):
            from .streams import overlapping_stream

            self._roll_streams = [overlapping_stream(self.device) for _ in range(self.U - 1)]  # (probed for real concurrency)
            if any(st is None for st in self._roll_streams):
                self._roll_streams = []
        side = side and len(self._roll_streams) == self.U - 1
        if side:
            for st in self._roll_streams:  # (before replica 0's launch is queued: the side streams wait for the parameters only)
                st.wait_stream(main)
        for u, rep in enumerate(self.reps):
            env = rep.env
            with (torch.cuda.stream(self._roll_streams[u - 1]) if (side and u > 0) else contextlib.nullcontext()):
              ok = self._timed(
                "rollout_fused", ops.rollout_ff, pa, pc, n_actions=self.nA, critic_shared=self.centralised, E=self.E,
                A=self.A, O=env.raw_obs_dim, T=self.T, time_limit=env.time_limit, policy_seed=self.seed,
                env_seed=env.seed, t0=self.t_global, row_offset=(self.rank * self.U + u) * EA, env_offset=env.env_offset,
                reward_mode=1 if env.reward_mode == "match" else 0, env_state=rep.state, agents_view=rep.agents_view,
                global_state=rep.global_state, action_mask=rep.action_mask, obs_step_count=rep.step_count,
                action=rep.action, value=rep.value, reward=rep.reward, log_prob=rep.log_prob, done=rep.done,
                last_val=rep.last_val, info_return=rep.info_return[n], info_length=rep.info_length[n],
                info_terminal=rep.info_terminal[n], adv=rep.adv, tgt=rep.tgt, gamma=float(s.gamma),
                gae_lambda=float(s.gae_lambda))
              if not ok:
                assert u == 0
                self.fused_rollout = False
                return False
              rep.last_reward.copy_(rep.reward[self.T - 1])
              rep.last_done.copy_(rep.done[self.T - 1])
        if side:
            for st in self._roll_streams:
                main.wait_stream(st)
        self.step_dev.add_(self.T)
        return True

    def _rollout_body(self) -> None:
        """ff_mappo.py:76-106: T acting steps, recording the time-major trajectory in place."""
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        EA = self.E * self.A
        for t in range(self.T):
            step = t  # relative; the kernels add step_dev (= t_global at the start of this rollout)
            for u, rep in enumerate(self.reps):
                av = rep.agents_view[t].view(EA, self.Oa)
                if self.centralised:
                    cx = rep.global_state[t].view(-1, self.Oc)
                else:
                    cx = av
                # a critic input row shared by the A agents of an env (critic_share == A) is evaluated once and its
                # value written to all A agent slots - the same numbers as A identical forward passes
                shared = self.critic_share == self.A and self.A > 1
                common = dict(critic_share=1 if shared else self.critic_share, critic_rows=self.E if shared else EA,
                              value_broadcast=self.A if shared else 1, seed=self.seed, step=step,
                              step_base=self.step_dev, row_offset=(self.rank * self.U + u) * EA, ctx=self.ctx)
                if self.generic:
                    self._timed("policy_step", self._generic_act, u, rep, t, step)
                elif self.continuous:
                    self._timed("policy_step", ops.policy_step_continuous, pa, pc, av, cx, action_dim=self.nA, min_scale=self.min_scale,
                                out=(rep.action[t].view(EA, self.nA), rep.log_prob[t].view(EA), rep.value[t].view(EA)),
                                **common)
                else:
                    self._timed("policy_step", ops.policy_step, pa, pc, av, rep.action_mask[t].view(EA, self.nA), cx,
                                n_actions=self.nA,
                                out=(rep.action[t].view(EA), rep.log_prob[t].view(EA), rep.value[t].view(EA)), **common)
                last = t == self.T - 1
                self._timed("env_step", rep.env.step_into, rep.state, step + 1, rep.obs_slot(t + 1), rep.reward[t], rep.done[t],
                            rep.cur_return[t], rep.cur_length[t], rep.cur_terminal[t], t_base=self.step_dev,
                            action=None if self.continuous else rep.action[t])
                if last:
                    rep.last_reward.copy_(rep.reward[t])
                    rep.last_done.copy_(rep.done[t])
        self.step_dev.add_(self.T)

    def _bootstrap_and_gae(self) -> None:
        """ff_mappo.py:109-139."""
        s = self.config.system
        pc = self.p[self.Pa :]
        EA = self.E * self.A
        for rep in self.reps:
            cx = rep.global_state[self.T].view(-1, self.Oc) if self.centralised else rep.agents_view[self.T].view(EA, self.Oa)
            if self.generic:
                v = self.critic_network.net.forward(pc, self._gws_roll[1], cx.view(1, self.E, -1, self.Oc), self.critic_share, None, EA,
                                                    self.E, self.A)[0]
                rep.last_val.view(EA).copy_(v)
            else:
                ops.mlp_forward(pc, self.Oc, 1, cx, rows=EA, x_share=self.critic_share, out=rep.last_val.view(EA, 1), ctx=self.ctx)
            self._timed("gae", ops.gae, rep.reward.view(self.T, EA), rep.value.view(self.T, EA), rep.done.view(self.T, EA),
                        rep.last_val.view(EA), float(s.gamma), float(s.gae_lambda),
                        out=(rep.adv.view(self.T, EA), rep.tgt.view(self.T, EA)), ctx=self.ctx)

    def _minibatch(self, n: int, k: int, mb: int, perm: Optional[torch.Tensor]) -> None:
        """ff_mappo.py:144-266 for minibatch `mb` of epoch `k`."""
        s = self.config.system
        T, E, A = self.T, self.E, self.A
        TEA = T * E * A
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        idx = None if perm is None else perm[mb * self.Rb : (mb + 1) * self.Rb]
        base = mb * self.Rb
        if self.generic:
            return self._minibatch_generic(n, k, mb, idx if idx is not None else torch.arange(base, base + self.Rb, dtype=torch.int32,
                                                                                             device=self.device))
        # actor: gradient kernels of every replica, fixed-order slab sum into g[:Pa] (+ actor_loss, entropy)
        ns = self.n_slab
        per_rep = self.fused_tail and self.U > 1  # (each replica's kernels own n_slab rows of the slab matrices)
        for u, rep in enumerate(self.reps):
            slab_a = self.slab_a[u * ns : (u + 1) * ns] if per_rep else self.slab_a
            av = rep.agents_view[:T].view(TEA, self.Oa)
            if getattr(self, "_stats_batched", False) and perm is self._perm_bufs[k]:
                stats = self._stats_all[u, k * self.M + mb]
            else:
                stats = ops.adv_stats(rep.adv.view(TEA), idx, base, self.Rb, A, out=self.stats)
            if self.continuous:
                self._timed("actor_grad", ops.ppo_actor_grad_continuous, pa, av, rep.action.view(TEA, self.nA),
                            rep.log_prob.view(TEA), rep.adv.view(TEA), stats, idx, base, self.Rb, A, self.nA,
                            float(s.clip_eps), float(s.ent_coef), self.seed, self.ent_step,
                            (self.rank * self.U + u) * TEA, slab_a, min_scale=self.min_scale)
            else:
                self._timed("actor_grad", ops.ppo_actor_grad, pa, av, rep.action_mask[:T].view(TEA, self.nA),
                            rep.action.view(TEA), rep.log_prob.view(TEA), rep.adv.view(TEA), stats, idx, base,
                            self.Rb, A, self.nA, float(s.clip_eps), float(s.ent_coef), slab_a, ctx=self.ctx)
            if not self.fused_tail:
                ops.slab_reduce2(self.slab_a, self.Pa, self.g[: self.Pa], 2, self.g[self.P : self.P + 2], accumulate=u > 0)
        # pmean "device" of ff_mappo.py:228-238, RCCL over xGMI: the actor's slice travels on RCCL's stream while
        # the critic's backward kernels run on this one
        w_actor = parallel.allreduce_sum_async(self.g[: self.Pa])
        for u, rep in enumerate(self.reps):
            av = rep.agents_view[:T].view(TEA, self.Oa)
            cx = rep.global_state[:T].view(-1, self.Oc) if self.centralised else av
            self._timed("critic_grad", ops.ppo_critic_grad, pc, cx, self.critic_share, rep.value.view(TEA), rep.tgt.view(TEA),
                        idx, base, self.Rb, A, float(s.clip_eps), float(s.vf_coef),
                        self.slab_c[u * ns : (u + 1) * ns] if per_rep else self.slab_c, ctx=self.ctx)
            if not self.fused_tail:
                ops.slab_reduce2(self.slab_c, self.Pc, self.g[self.Pa : self.P], 1, self.g[self.P + 2 : self.P + 3], accumulate=u > 0)
        if self.fused_tail:
            self._timed("finish", ops.ppo_finish, self.ctx, self.slab_a, self.slab_c, self.Pa, self.Pc, self.g, self.p, self.m, self.v,
                        self.count, self.seg_lr[0], self.seg_lr[1], grad_scale=1.0 / self.U, max_norm=float(s.max_grad_norm),
                        decay=bool(s.decay_learning_rates), steps_per_update=self.K * self.M,
                        num_updates=int(s.get("num_updates", 1) or 1), vf_coef=float(s.vf_coef), ent_coef=float(s.ent_coef),
                        metrics_out=self.train_metrics[n, k, mb], critic_din=self.Oc, workspace=self._finish_ws)
            self.ent_step += 1
            return
        w_rest = parallel.allreduce_sum_async(self.g[self.Pa :])  # critic gradient + the loss scalars
        for wk in (w_actor, w_rest):
            if wk is not None:
                wk.wait()
        if self._finish_ws_mr is not None:  # several ranks: Adam + count increment + W1 re-split as one launch on the all-reduced g
            self._timed("clip_adam", ops.ppo_finish, self.ctx, None, None, self.Pa, self.Pc, self.g, self.p, self.m, self.v, self.count,
                        self.seg_lr[0], self.seg_lr[1], grad_scale=1.0 / (self.U * self.world), max_norm=float(s.max_grad_norm),
                        decay=bool(s.decay_learning_rates), steps_per_update=self.K * self.M,
                        num_updates=int(s.get("num_updates", 1) or 1), vf_coef=float(s.vf_coef), ent_coef=float(s.ent_coef),
                        metrics_out=self.train_metrics[n, k, mb], critic_din=self.Oc, workspace=self._finish_ws_mr)
            self.ent_step += 1
            return
        self._timed("clip_adam", ops.clip_adam, self.p, self.g, self.m, self.v, self.count, self.seg_off, self.seg_lr,
                      grad_scale=1.0 / (self.U * self.world), max_norm=float(s.max_grad_norm),
                      decay=bool(s.decay_learning_rates), steps_per_update=self.K * self.M,
                      num_updates=int(s.get("num_updates", 1) or 1), loss_sums=self.g[self.P :], vf_coef=float(s.vf_coef),
                      ent_coef=float(s.ent_coef), metrics_out=self.train_metrics[n, k, mb])
        self.ent_step += 1

    # ---------------------------------------------------------------- general network path (mava_amd/generic_networks.py)
    def _generic_setup(self) -> None:
        d = self.device
        EA, Rm = self.E * self.A, self.Rb * self.A
        a, c = self.actor_network.net, self.critic_network.net
        self._gws_roll = (a.workspace(EA, d, False), c.workspace(EA, d, False))
        self._gws = (a.workspace(Rm, d, True), c.workspace(Rm, d, True))
        # the backward chain runs in units of a power of two near the row count (mava_seq_actor_loss_f32: f16 range)
        self._g_scale = float(2 ** math.ceil(math.log2(Rm)))
        self._g_dscale = torch.zeros((self._gws[0].loss_partials.shape[0], max(self.nA, 1)), device=d)

    def _generic_act(self, u: int, rep, t: int, step: int) -> None:
        """ff_mappo.py:76-94 for one replica and step: actor forward -> sample / log-prob, critic forward -> value."""
        if not hasattr(self, "_gws_roll"):
            self._generic_setup()
        L, st = ops.lib(), ops.stream_ptr()
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        E, A, EA = self.E, self.A, self.E * self.A
        an = self.actor_network
        outs = an.net.forward(pa, self._gws_roll[0], rep.agents_view[t : t + 1], 1, None, EA, E, A)
        seed, row_off = self.seed & (2**64 - 1), ((self.rank * self.U + u) * EA) & 0xFFFFFFFF
        gstep = (self.t_global + step) & 0xFFFFFFFF  # (the general path is not graph-captured: host-side step counter)
        if self.continuous:
            ind = an.independent_std
            ops.check(L.mava_seq_sample_continuous_f32(EA, self.nA, self.min_scale, ops.ptr(outs[0]), ops.ptr(an.log_std(pa)) if ind else None,
                                                       None if ind else ops.ptr(outs[1]), seed, gstep, row_off, 0, ops.ptr(rep.action[t]),
                                                       ops.ptr(rep.log_prob[t]), st), "mava_seq_sample_continuous_f32")
        else:
            ops.check(L.mava_seq_sample_f32(EA, self.nA, ops.ptr(outs[0]), ops.ptr(rep.action_mask[t]), seed, gstep, row_off, 0,
                                            ops.ptr(rep.action[t]), ops.ptr(rep.log_prob[t]), st), "mava_seq_sample_f32")
        cx = rep.global_state[t : t + 1] if self.centralised else rep.agents_view[t : t + 1]
        v = self.critic_network.net.forward(pc, self._gws_roll[1], cx, self.critic_share, None, EA, E, A)[0]
        rep.value[t].view(EA).copy_(v)  # a (rows x 1) T32 matrix is row-major

    def _minibatch_generic(self, n: int, k: int, mb: int, idx: torch.Tensor) -> None:
        """ff_mappo.py:144-266 on the general path: the (t, e) rows of the minibatch are presented to the sequence kernels
        as T = 1 'sequences' over a flattened env axis of T*E entries (idx = their indices)."""
        if not hasattr(self, "_gws"):
            self._generic_setup()
        s = self.config.system
        T, E, A = self.T, self.E, self.A
        TE, Rm = T * E, self.Rb * A
        L, st = ops.lib(), ops.stream_ptr()
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        idx = idx.contiguous()
        wa, wc = self._gws
        an, cn, gs = self.actor_network, self.critic_network, self._g_scale
        nblk = wa.loss_partials.shape[0]
        for u, rep in enumerate(self.reps):
            acc = u > 0
            outs = an.net.forward(pa, wa, rep.agents_view[:T].view(1, TE, A, self.Oa), 1, idx, Rm, TE, A)
            ops.adv_stats(rep.adv.view(-1), idx, 0, self.Rb, A, out=self.stats)
            if self.continuous:
                ind = an.independent_std
                ops.check(L.mava_seq_actor_loss_continuous_f32(
                    1, Rm, TE, A, self.nA, self.min_scale, ops.ptr(idx), ops.ptr(outs[0]), ops.ptr(an.log_std(pa)) if ind else None,
                    None if ind else ops.ptr(outs[1]), ops.ptr(rep.action), ops.ptr(rep.log_prob), ops.ptr(rep.adv), ops.ptr(self.stats),
                    self.stats.shape[0], float(s.clip_eps), float(s.ent_coef), self.seed & (2**64 - 1), self.ent_step & 0xFFFFFFFF,
                    ((self.rank * self.U + u) * TE * A) & 0xFFFFFFFF, gs, ops.ptr(wa.dout[0]), None if ind else ops.ptr(wa.dout[1]),
                    ops.ptr(wa.loss_partials), ops.ptr(self._g_dscale), nblk, st), "mava_seq_actor_loss_continuous_f32")
                if ind:
                    ops.slab_reduce(self._g_dscale, self.nA, an.log_std(self.g[: self.Pa]), accumulate=acc)
            else:
                ops.check(L.mava_seq_actor_loss_f32(1, Rm, TE, A, self.nA, ops.ptr(idx), ops.ptr(outs[0]), ops.ptr(rep.action_mask[:T]),
                                                    ops.ptr(rep.action), ops.ptr(rep.log_prob), ops.ptr(rep.adv), ops.ptr(self.stats),
                                                    self.stats.shape[0], float(s.clip_eps), float(s.ent_coef), gs, ops.ptr(wa.dout[0]),
                                                    ops.ptr(wa.loss_partials), nblk, st), "mava_seq_actor_loss_f32")
            ops.slab_reduce(wa.loss_partials, 2, self.g[self.P : self.P + 2], accumulate=acc)
            an.net.backward(pa, wa, wa.dout[: len(an.net.heads)], self.g[: self.Pa], accumulate=acc, grad_scale=gs)
        w_actor = parallel.allreduce_sum_async(self.g[: self.Pa])
        for u, rep in enumerate(self.reps):
            acc = u > 0
            cx = (rep.global_state[:T].view(1, TE, -1, self.Oc) if self.centralised else rep.agents_view[:T].view(1, TE, A, self.Oa))
            v = cn.net.forward(pc, wc, cx, self.critic_share, idx, Rm, TE, A)[0]
            ops.check(L.mava_seq_critic_loss_f32(1, Rm, TE, A, 1, ops.ptr(idx), ops.ptr(v), ops.ptr(rep.value), ops.ptr(rep.tgt),
                                                 float(s.clip_eps), float(s.vf_coef), gs, ops.ptr(wc.dout[0]), ops.ptr(wc.loss_partials),
                                                 nblk, st), "mava_seq_critic_loss_f32")
            ops.slab_reduce(wc.loss_partials, 1, self.g[self.P + 2 : self.P + 3], accumulate=acc)
            cn.net.backward(pc, wc, [wc.dout[0]], self.g[self.Pa : self.P], accumulate=acc, grad_scale=gs)
        w_rest = parallel.allreduce_sum_async(self.g[self.Pa :])
        for wk in (w_actor, w_rest):
            if wk is not None:
                wk.wait()
        ops.clip_adam(self.p, self.g, self.m, self.v, self.count, self.seg_off, self.seg_lr,
                      grad_scale=1.0 / (self.U * self.world), max_norm=float(s.max_grad_norm),
                      decay=bool(s.decay_learning_rates), steps_per_update=self.K * self.M,
                      num_updates=int(s.get("num_updates", 1) or 1), loss_sums=self.g[self.P :], vf_coef=float(s.vf_coef),
                      ent_coef=float(s.ent_coef), metrics_out=self.train_metrics[n, k, mb])
        self.ent_step += 1

    def _permutations(self) -> List[torch.Tensor]:
        """ff_mappo.py:272-273: one permutation of the T*E rows per epoch, identical on every replica and rank (the
        reference hands the same PRNG key to all of them, :417-426): mava_permutation_i32 keyed by (seed, a running
        epoch counter) - one 5 us launch each, where torch.randperm's sort passes were ~1 ms of device time per update
        at T*E = 524 288.  Written into persistent buffers (read by the kernels of this update only)."""
        if not hasattr(self, "_perm_bufs"):
            self._perm_all = torch.empty((self.K, self.T * self.E), dtype=torch.int32, device=self.device)
            self._perm_bufs = [self._perm_all[k] for k in range(self.K)]
            self._stats_all = torch.empty((self.U, self.K * self.M, self.stats.shape[0], 2), dtype=torch.float64, device=self.device)
        for buf in self._perm_bufs:
            ops.permutation(self.T * self.E, self.seed, self.perm_count, out=buf)
            self.perm_count += 1
        return self._perm_bufs

    def update(self, n: int, permutations: Optional[List[torch.Tensor]] = None) -> None:
        own = permutations is None
        if own:
            permutations = self._permutations()
        # parameters may have been written from outside since the last minibatch (adopt(), a test): the first wide critic
        # launch of an update always re-splits W1 itself (MAVA_CTX_W1_SPLIT_FRESH)
        self.ctx.set(self.ctx.W1_SPLIT_FRESH, 0)
        self._rollout(n)
        # advantage statistics of ALL K x M minibatches in one launch per replica (the permutations are this learner's
        # own contiguous buffer: minibatch (k, mb) = slice k * M + mb of it); otherwise one launch per minibatch
        self._stats_batched = own and not self.generic and self.T * self.E == self.M * self.Rb  # (slices tile the buffer)
        if self._stats_batched:
            for u, rep in enumerate(self.reps):
                ops.adv_stats_batched(rep.adv.view(-1), self._perm_all.view(-1), self.Rb, self.A, self.K * self.M,
                                      out=self._stats_all[u])
        for k in range(self.K):
            perm = permutations[k]
            for mb in range(self.M):
                self._minibatch(n, k, mb, perm)
        # the bootstrap observation becomes the first observation of the next rollout
        for rep in self.reps:
            rep.agents_view[0].copy_(rep.agents_view[self.T])
            rep.global_state[0].copy_(rep.global_state[self.T])
            rep.action_mask[0].copy_(rep.action_mask[self.T])
            rep.step_count[0].copy_(rep.step_count[self.T])

    def learn(self, learner_state: LearnerState) -> ExperimentOutput:
        """LearnerFn (mava/types.py:154): num_updates_per_eval updates, asynchronous on the current
        stream - the caller synchronises before reading the clock (ff_mappo.py:497-498)."""
        self.adopt(learner_state)
        if self.matmul_mode == "f16x2":
            obs = [t for r in self.reps for t in (r.agents_view[0], r.global_state[0] if self.centralised else None)]
            check_f16_range(self.p, obs, self.train_metrics if self._learn_calls else None, type(self).__name__)
        self._learn_calls += 1
        for n in range(self.n_upd):
            self.update(n)
        U = self.U
        episode_metrics = {
            "episode_return": torch.stack([r.info_return for r in self.reps], 1).unsqueeze(0),  # (1,N,U,T,E)
            "episode_length": torch.stack([r.info_length for r in self.reps], 1).unsqueeze(0),
            "is_terminal_step": torch.stack([r.info_terminal for r in self.reps], 1).unsqueeze(0).bool(),
        }
        tm = self.train_metrics.unsqueeze(1).expand(self.n_upd, U, self.K, self.M, 4).unsqueeze(0)  # (1,N,U,K,M,4)
        train_metrics = {"total_loss": tm[..., 0], "value_loss": tm[..., 1], "actor_loss": tm[..., 2], "entropy": tm[..., 3]}
        return ExperimentOutput(self.learner_state(), episode_metrics, train_metrics)


def learner_setup(env, keys, config, centralised_critic: bool, device=None):
    """Counterpart of learner_setup (ff_mappo.py:333-432): returns (learn, actor_network,
    init_learner_state).  `keys` = (key, actor_net_key, critic_net_key) integer seeds."""
    key, actor_key, critic_key = (int(k) for k in keys)
    learner = FFLearner(env, config, centralised_critic, device)
    learner.seed = key
    learner.init_params(actor_key, critic_key)
    parallel.broadcast_(learner.p, src=0)
    learner.reset_envs()
    def learn(learner_state: LearnerState) -> ExperimentOutput:
        return learner.learn(learner_state)

    learn.learner = learner  # type: ignore[attr-defined]  (handle for tests / bench)
    return learn, learner.actor_network, learner.learner_state()


def get_final_step_metrics(metrics: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], bool]:
    """mava/wrappers/episode_metrics.py:114-132."""
    metrics = dict(metrics)
    is_final = metrics.pop("is_terminal_step").bool()
    has_final = bool(is_final.any())
    if not has_final:
        return {k: torch.zeros_like(v) for k, v in metrics.items()}, False
    return {k: v[is_final] for k, v in metrics.items()}, True
