"""Recurrent PPO learner (rec_ippo / rec_mappo) behind Mava's LearnerFn contract.

Reference: mava/systems/ppo/rec_mappo.py:45-434 (get_learner_fn) and :437-576 (learner_setup);
rec_ippo.py is the same file with a decentralised critic.  Differences from the feed-forward learner
(mava_amd/learner.py), all visible in the cited lines:
  * the carry holds `dones` (E, A) and the GRU hidden states of actor and critic (:91-105, RNNLearnerState);
  * a transition stores the done flag ENTERING the step and the loss re-unrolls the whole sequence from
    the hidden state at the start of the rollout, hstates[0] (:136-145, :219-222, :252-255);
  * GAE masks with the NEXT step's stored flag and is seeded with the post-rollout `dones` (:177-199);
  * minibatches are slices of a permutation over ENVS and keep all T steps (:334-365); only
    recurrent_chunk_size == rollout_length is meaningful and anything else is rejected.

Kernel chain per network and minibatch (mava_amd/rec_networks.py): dense(pre) -> dense(gi) -> GRU scan ->
dense(post) -> dense(head) -> sequence loss -> dense(dpost) -> dense(dh) -> GRU BPTT scan -> dense(dxpre)
-> five X^T Y weight-gradient products; then [all-reduce] and the same fused clip+Adam as the ff systems.
"""
from __future__ import annotations

import math
import os

from typing import Any, Dict, List, Optional

import torch

from . import ops, parallel
from ._lib import check, launch, lib, ptr, stream_ptr
from .guards import check_f16_range
from .learner import NUM_CU, _Replica
from .networks import MLPTorso, make_action_head
from .rec_networks import H, RecurrentActor, RecurrentValueNet, RecWorkspace, rows_to_t32, t32_to_rows
from .types import (AdamState, ExperimentOutput, HiddenStates, Observation, ObservationGlobalState, OptStates, Params,
                    RNNLearnerState, TimeStep)


class _RecReplica(_Replica):
    def __init__(self, env, T, n_upd, central, device, continuous: bool = False, hidden: int = H):
        super().__init__(env, T, n_upd, central, device, continuous)
        self.Hd = int(hidden)  # network.hidden_state_dim
        E, A = env.num_envs, env.num_agents
        EA = E * A
        if EA % 32:
            raise ValueError(f"num_envs * num_agents = {EA} must be a multiple of 32 for the recurrent kernels")
        self.dones = torch.zeros((E, A), dtype=torch.uint8, device=device)       # flag entering the next step
        self.done_in = torch.zeros((T, E, A), dtype=torch.uint8, device=device)  # transition.done = last_done (:136-137)
        # current hidden states in the kernels' T32 layout, and the rollout-initial copies (hstates[0], row-major)
        self.h_actor = torch.zeros(EA * self.Hd, device=device)
        self.h_actor_next = torch.zeros_like(self.h_actor)
        self.h0_actor = torch.zeros((E, A, self.Hd), device=device)
        self.set_critic_rows(A)

    def set_critic_rows(self, ac: int) -> None:
        """Critic sequences per env: A (one per agent), or 1 when the agents share the critic input and the critic
        runs once per env (RecLearner.critic_agg)."""
        E, dev = self.env.num_envs, self.dones.device
        self.Ac = ac
        # full E*A size even when only E rows are used: the buffer is swapped with the shared rollout workspace's
        # hidden-state output every step, which the actor fills with E*A rows
        self.h_critic = torch.zeros(E * self.env.num_agents * self.Hd, device=dev)
        self.h_critic_next = torch.zeros_like(self.h_critic)  # ping-pong partner for the fused acting step
        self.h0_critic = torch.zeros((E, ac, self.Hd), device=dev)
        # per-env copies of the entering done flags and a value scratch, used by the once-per-env critic
        self.done_env = torch.zeros((E, 1), dtype=torch.uint8, device=dev)
        self.done_env_in = torch.zeros((self.done_in.shape[0], E, 1), dtype=torch.uint8, device=dev)
        self.value_env = torch.zeros(E, device=dev)


class RecLearner:
    def __init__(self, env, config, centralised_critic: bool, device: Optional[torch.device] = None):
        self.config = config
        self.centralised = centralised_critic
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world = parallel.rank_world()
        s, arch = config.system, config.arch
        self.E, self.U, self.T = int(arch.num_envs), int(s.update_batch_size), int(s.rollout_length)
        self.K, self.M = int(s.ppo_epochs), int(s.num_minibatches)
        chunk = s.get("recurrent_chunk_size", None)
        if chunk is not None and int(chunk) != self.T:
            # rec_mappo.py:342-349 reshapes (T,E,..)->(chunk, E*n,..) row-major, which interleaves time and env
            # unless chunk == T; the default (null -> T, :586-587) is the only meaningful setting.
            raise ValueError("recurrent_chunk_size must be null or equal to rollout_length")
        if self.E % self.M:
            raise ValueError("num_envs must be divisible by num_minibatches (rec_mappo.py:354-357)")
        self.n_upd = int(s.get("num_updates_per_eval", 1))
        # arithmetic of the dense / X^T Y products on T32 operands: "f16x2" (default, rec_dense_h2.hip) or "f32"
        self.matmul_mode = str(s.get("matmul_mode", None) or os.environ.get("MAVA_MATMUL", "f16x2"))
        if self.matmul_mode not in ("f16x2", "f32"):
            raise ValueError(f"system.matmul_mode must be 'f16x2' or 'f32', got {self.matmul_mode!r}")
        self.ctx = ops.Ctx(self.matmul_mode)  # this learner's context handle (include/mava_hip.h mava_ctx_*): no process-wide mode
        if env.num_envs != self.E:
            raise ValueError(f"env.num_envs={env.num_envs} != arch.num_envs={self.E}")
        if centralised_critic and not getattr(env, "add_global_state", False):
            raise ValueError("Global state must be provided to the centralised critic.")  # networks.py:315-316
        self.reps: List[_RecReplica] = []
        # rec_mappo.py:361-363: the action head of the configuration, sized by the env's action dimension
        action_head = make_action_head(config.network.get("action_head", None), env.action_dim)
        self.continuous = type(action_head).__name__ == "ContinuousActionHead"
        self.min_scale = float(getattr(action_head, "min_scale", 1e-3))
        # ContinuousActionHead(independent_std=False): the log_std layer is a second head of the actor's post-torso, which then
        # runs on the general layer kernels (mava_amd/generic_networks.py) whatever its torsos are
        self.dep_std = self.continuous and not getattr(action_head, "independent_std", True)
        for u in range(self.U):
            rep_env = env.clone(env_offset=getattr(env, "env_offset", 0) + (self.rank * self.U + u) * self.E)
            self.reps.append(_RecReplica(rep_env, self.T, self.n_upd, centralised_critic, self.device, self.continuous,
                                         hidden=int(config.network.get("hidden_state_dim", 128))))
        env0 = self.reps[0].env
        self.A, self.nA = env0.num_agents, env0.action_dim
        config.system.num_agents = self.A
        self.Oa = env0.obs_dim
        if centralised_critic:
            self.Oc = env0.state_dim
            self.critic_share = self.A if env0.global_state_shared else 1
        else:
            self.Oc, self.critic_share = self.Oa, 1
        self.Em = self.E // self.M
        self.Rm = self.Em * self.A
        if self.Rm % 32:
            raise ValueError(f"(num_envs / num_minibatches) * num_agents = {self.Rm} must be a multiple of 32")
        # Centralised critic on a global state shared by the A agents of an env: identical inputs, done flags and
        # initial hidden states give A identical hidden trajectories and values (the reference tiles the state and
        # runs all A, networks.py:306-331).  The critic then runs ONCE per env - E sequences instead of E*A - and the
        # sequence loss adds up the A agents' loss gradients per row (mava_seq_critic_loss_f32, agents_per_row = A).
        self.critic_agg = bool(centralised_critic and self.critic_share == self.A and self.A > 1
                               and self.E % 32 == 0 and self.Em % 32 == 0 and os.environ.get("MAVA_REC_CRITIC_AGG", "1") != "0")
        self.Ac = 1 if self.critic_agg else self.A            # critic sequences per env
        self.Rmc = self.Em * self.Ac                            # critic rows per time step of a minibatch
        if self.critic_agg:
            for rep in self.reps:
                rep.set_critic_rows(1)

        net = config.network
        from .generic_networks import CNNTorso, GenericMLPTorso

        def torsos(nc, general=False):
            """(pre, post) of one network: network/rnn.yaml's [128] relu torsos run on the dedicated kernels; any other
            configuration - MLPTorso layer sizes / tanh / layer norm, or network/rcnn.yaml's CNNTorso pre-torso - makes BOTH
            torsos of that network general ones."""
            is_cnn = lambda raw: str(dict(raw).get("_target_", "MLPTorso")).endswith("CNNTorso")
            cfgs = [{k: v for k, v in c.items() if k != "_target_"} for c in (nc.pre_torso, nc.post_torso)]
            if is_cnn(nc.post_torso):
                raise ValueError("post_torso cannot be a CNNTorso: it consumes the hidden features (network/rcnn.yaml uses an MLPTorso)")
            if is_cnn(nc.pre_torso):
                return CNNTorso(**cfgs[0]), GenericMLPTorso(**cfgs[1])
            default = not general and all(list(c.get("layer_sizes", [128])) == [128] and c.get("activation", "relu") == "relu"
                                          and not c.get("use_layer_norm", False) for c in cfgs)
            return tuple((MLPTorso if default else GenericMLPTorso)(**c) for c in cfgs)

        hsd = self.Hd = int(net.get("hidden_state_dim", 128))
        wide = hsd != H  # (a hidden width other than 128 runs the cell step by step on the general layer kernels: general torsos)
        env0 = self.reps[0].env
        obs_shape = getattr(env0, "obs_shape", None)  # (H, W, C) observations for CNN pre-torsos (env.synthetic.obs_shape)
        state_shape = getattr(env0, "state_shape", None) if centralised_critic else obs_shape
        self.actor_network = RecurrentActor(*torsos(net.actor_network, general=self.dep_std or wide), action_head, self.Oa, hsd, obs_shape)
        self.critic_network = RecurrentValueNet(*torsos(net.critic_network, general=wide), centralised_critic, self.Oc, hsd, state_shape)
        self.actor_network.ctx = self.critic_network.ctx = self.ctx
        self.generic_nets = self.actor_network.generic or self.critic_network.generic
        self.Pa, self.Pc = self.actor_network.num_params, self.critic_network.num_params
        self.P = self.Pa + self.Pc

        d = self.device
        self.p = torch.zeros(self.P, device=d)
        self.m = torch.zeros(self.P, device=d)
        self.v = torch.zeros(self.P, device=d)
        self.count = torch.zeros(2, dtype=torch.int32, device=d)
        self.g = torch.zeros(self.P + 4, device=d)
        self.seg_off = [0, self.Pa, self.P]
        self.seg_lr = [float(s.actor_lr), float(s.critic_lr)]
        self.ws_roll = RecWorkspace(self.E * self.A, max(self.nA, 1), d, training=False)
        self.ws = RecWorkspace(self.T * self.Rm, max(self.nA, 1), d, training=True, din_max=max(self.Oa, self.Oc))
        # the backward chain runs in units of a power of two near the row count (mava_seq_actor_loss_f32: f16 range)
        self.grad_scale = float(2 ** math.ceil(math.log2(self.T * self.Rm)))
        # f16x2: both networks' weights pre-split in MFMA-fragment order for the fused acting step (re-packed per rollout)
        self.fused_out = (self.matmul_mode == "f16x2" and not self.continuous and self.nA <= 16 and not self.generic_nets
                          and os.environ.get("MAVA_REC_FUSED_OUT", "1") != "0")
        self.pack_a = torch.empty(lib().mava_rec_step_pack_bytes(self.Oa), dtype=torch.uint8, device=d)
        self.pack_c = torch.empty(lib().mava_rec_step_pack_bytes(self.Oc), dtype=torch.uint8, device=d)
        # (with several ranks a few CUs stay free of the persistent X^T Y blocks for RCCL's kernels: learner.py)
        self.rccl_cus = int(s.get("rccl_cus", None) if s.get("rccl_cus", None) is not None
                            else os.environ.get("MAVA_RCCL_CUS", "8" if self.world > 1 else "0"))
        n_slab = max(1, min(NUM_CU - max(0, min(self.rccl_cus, NUM_CU - 1)), (self.T * self.Rm) // 32))
        self.slabs = torch.zeros((n_slab, H * 3 * H + 3 * H + 8), device=d)
        # The two networks are independent between the parameters of one Adam step and the next (rec_mappo.py:210-266 computes
        # both losses from the same params): the critic's forward / loss / backward runs on a second stream with its own
        # workspace and slabs, under the actor's.  Its once-per-env scans fill one CU in eight (32 workgroups at config 4)
        # and its products are short launches - alone on the device they left it mostly idle for ~14 ms per update.
        self.overlap_critic = (os.environ.get("MAVA_REC_OVERLAP", "1") != "0" and d.type == "cuda" and not self.generic_nets)
        if self.overlap_critic:
            from .streams import overlapping_stream

            self.side = overlapping_stream(d)  # (probed: a stream that shares the launch stream's hardware queue is no use)
            self.overlap_critic = self.side is not None
        if self.overlap_critic:
            c_rows = self.T * (self.Rmc if self.critic_agg else self.Rm)
            self.ws_c = RecWorkspace(c_rows, 1, d, training=True, din_max=self.Oc)
            self.slabs_c = torch.zeros_like(self.slabs)
        self.stats = torch.zeros((lib().mava_adv_stats_blocks(), 2), dtype=torch.float64, device=d)
        self.train_metrics = torch.zeros((self.n_upd, self.K, self.M, 4), device=d)
        self.perm_count = 0  # epoch permutations drawn so far (counter of mava_permutation_i32)
        self.t_global = 0
        self.ent_step = 0  # minibatches trained so far: counter of the continuous head's entropy sample
        self.dscale_partials = torch.zeros((self.ws.loss_partials.shape[0], max(self.nA, 1)), device=d)
        self.seed = int(s.seed)
        self._t_range = torch.arange(self.T, device=d, dtype=torch.int64)[:, None] * self.E
        self._t_range32 = self._t_range.to(torch.int32)
        self._learn_calls = 0  # guards.check_f16_range

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
            rep.dones.zero_()      # rec_mappo.py:555-558
            rep.h_actor.zero_()    # ScannedRNN.initialize_carry: zeros (:497-502)
            rep.h_critic.zero_()
        self.t_global = 0
        self.ent_step = 0
        self.perm_count = 0

    # ---------------------------------------------------------------------------- state views
    def learner_state(self) -> RNNLearnerState:
        lead = (1, self.U)
        params = Params(self.actor_network.tree(self.p[: self.Pa], lead), self.critic_network.tree(self.p[self.Pa :], lead))
        opts = []
        for i, (net, sl) in enumerate(((self.actor_network, slice(0, self.Pa)), (self.critic_network, slice(self.Pa, self.P)))):
            opts.append(AdamState(self.count[i].expand(1, self.U), net.tree(self.m[sl], lead), net.tree(self.v[sl], lead)))
        key = torch.tensor([[[self.seed, self.t_global]] * self.U], dtype=torch.int64)
        st = lambda f: torch.stack([f(r) for r in self.reps], 0).unsqueeze(0)
        E, A = self.E, self.A
        av, mask, sc = st(lambda r: r.agents_view[0]), st(lambda r: r.action_mask[0]).bool(), st(lambda r: r.step_count[0])
        if self.centralised:
            gs = st(lambda r: r.global_state[0].expand(-1, A, -1) if r.env.gs_tiles == 1 else r.global_state[0])
            obs: Any = ObservationGlobalState(av, mask, gs, sc)
        else:
            obs = Observation(av, mask, sc)
        dones = st(lambda r: r.dones).bool()
        ts = TimeStep(torch.where(dones[..., 0], 2, 1).to(torch.int8), st(lambda r: r.last_reward), 1.0 - dones.float(), obs, {})
        D = self.Hd
        hst = HiddenStates(st(lambda r: t32_to_rows(r.h_actor, D, E * A).view(E, A, D)),
                           st(lambda r: t32_to_rows(r.h_critic, D, E * self.Ac).view(E, self.Ac, D).expand(E, A, D)))
        env_state = {"step_count": st(lambda r: r.state.step_count), "episode_return": st(lambda r: r.state.ep_return),
                     "episode_length": st(lambda r: r.state.ep_length)}
        return RNNLearnerState(params, OptStates(*opts), key, env_state, ts, dones, hst)

    # ------------------------------------------------------------------------------------ update
    def _critic_x(self, rep, lo, hi):
        if self.centralised:
            return rep.global_state[lo:hi]
        return rep.agents_view[lo:hi]

    def _rollout(self, n: int) -> None:
        """rec_mappo.py:91-153: T acting steps; hidden states advance in place (T32), last_done is recorded."""
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        E, A = self.E, self.A
        EA = E * A
        ws = self.ws_roll
        EAc = E * self.Ac
        for rep in self.reps:  # hstates[0] of this rollout, the state the losses re-unroll from (:219-222)
            rep.h0_actor.view(EA, self.Hd).copy_(t32_to_rows(rep.h_actor, self.Hd, EA))
            rep.h0_critic.view(EAc, self.Hd).copy_(t32_to_rows(rep.h_critic, self.Hd, EAc))
        fused = os.environ.get("MAVA_REC_FUSED_STEP", "1") != "0" and not self.generic_nets  # (fused step: default torsos)
        packed = fused and self.matmul_mode == "f16x2" and self.nA <= 16
        if packed:  # the parameters are constant during a rollout: split them once (rec_step_h2.hip)
            check(lib().mava_rec_step_pack_f32(ptr(pa), self.Oa, ptr(self.pack_a), stream_ptr()), "mava_rec_step_pack_f32")
            check(lib().mava_rec_step_pack_f32(ptr(pc), self.Oc, ptr(self.pack_c), stream_ptr()), "mava_rec_step_pack_f32")
        for t in range(self.T):
            step = self.t_global + t
            for u, rep in enumerate(self.reps):
                if fused:
                    # ONE launch for both networks: pre_torso -> GRU cell -> post_torso -> head -> sample / value
                    # (mava_rec_step_f32); the critic runs once per env when the agents share its input.  The flags
                    # entering step t are the previous step's done flags, read in place (the once-per-env critic reads
                    # agent 0's flag of each env: stride A); done_in is materialised once after the rollout.
                    agg = self.critic_agg
                    d_prev = rep.dones if t == 0 else rep.done[t - 1]
                    critic_args = (ptr(pc), self.Oc, ptr(self._critic_x(rep, t, t + 1)), 1 if agg else self.critic_share,
                                   ptr(d_prev), A if agg else 1, ptr(rep.h_critic), ptr(rep.h_critic_next),
                                   E if agg else EA, A if agg else 1, ptr(rep.value[t]), stream_ptr())
                    rng_args = (self.seed & (2**64 - 1), step & 0xFFFFFFFF, ((self.rank * self.U + u) * EA) & 0xFFFFFFFF, 0)
                    if packed:
                        launch("rec_step", lib().mava_rec_step_packed_f32, ptr(self.pack_a), ptr(self.pack_c),
                               ptr(pa), self.Oa, self.nA, self.min_scale, ptr(rep.agents_view[t]), None if self.continuous else ptr(rep.action_mask[t]),
                               ptr(d_prev), ptr(rep.h_actor), ptr(rep.h_actor_next), EA, *rng_args,
                               None if self.continuous else ptr(rep.action[t]), ptr(rep.action[t]) if self.continuous else None,
                               ptr(rep.log_prob[t]), *critic_args)
                    elif self.continuous:
                        check(lib().mava_rec_step_continuous_f32(
                            ptr(pa), self.Oa, self.nA, self.min_scale, ptr(rep.agents_view[t]), ptr(d_prev), ptr(rep.h_actor),
                            ptr(rep.h_actor_next), EA, *rng_args, ptr(rep.action[t]), ptr(rep.log_prob[t]), *critic_args),
                            "mava_rec_step_continuous_f32")
                    else:
                        launch("rec_step", lib().mava_rec_step_f32,
                               ptr(pa), self.Oa, self.nA, ptr(rep.agents_view[t]), ptr(rep.action_mask[t]), ptr(d_prev),
                               ptr(rep.h_actor), ptr(rep.h_actor_next), EA, *rng_args, ptr(rep.action[t]), ptr(rep.log_prob[t]),
                               *critic_args)
                    rep.h_actor, rep.h_actor_next = rep.h_actor_next, rep.h_actor
                    rep.h_critic, rep.h_critic_next = rep.h_critic_next, rep.h_critic
                    rep.env.step_into(rep.state, step + 1, rep.obs_slot(t + 1), rep.reward[t], rep.done[t], rep.info_return[n, t],
                                      rep.info_length[n, t], rep.info_terminal[n, t], action=None if self.continuous else rep.action[t])
                    continue
                rep.done_in[t].copy_(rep.dones)
                d1 = rep.done_in[t : t + 1]
                if self.critic_agg:
                    rep.done_env.copy_(rep.dones[:, :1])
                    rep.done_env_in[t].copy_(rep.done_env)
                # layer-wise acting step (MAVA_REC_FUSED_STEP=0): the training kernels with T = 1
                self.actor_network.forward_sequence(pa, ws, rep.agents_view[t : t + 1], 1, d1, rep.h_actor, True, None, 1, EA, E,
                                                    A, training=False)
                if self.Hd == H:
                    rep.h_actor, ws.hs = ws.hs, rep.h_actor  # the scan's output becomes the carried hidden state
                else:
                    rep.h_actor[: EA * self.Hd].copy_(self.actor_network.last_hidden(ws, 1, EA))
                if self.continuous:
                    ind = not self.dep_std
                    check(lib().mava_seq_sample_continuous_f32(EA, self.nA, self.min_scale, ptr(ws.y), ptr(self.actor_network.log_std(pa)) if ind else None,
                                                               None if ind else ptr(ws.y2), self.seed & (2**64 - 1), step & 0xFFFFFFFF,
                                                               ((self.rank * self.U + u) * EA) & 0xFFFFFFFF, 0, ptr(rep.action[t]),
                                                               ptr(rep.log_prob[t]), stream_ptr()), "mava_seq_sample_continuous_f32")
                else:
                    check(lib().mava_seq_sample_f32(EA, self.nA, ptr(ws.y), ptr(rep.action_mask[t]), self.seed & (2**64 - 1),
                                                    step & 0xFFFFFFFF, ((self.rank * self.U + u) * EA) & 0xFFFFFFFF, 0,
                                                    ptr(rep.action[t]), ptr(rep.log_prob[t]), stream_ptr()), "mava_seq_sample_f32")
                # critic: a (rows x 1) T32 matrix IS row-major, so the head writes straight into the value slot
                if self.critic_agg:  # once per env, value broadcast to the A agent slots
                    self.critic_network.forward_sequence(pc, ws, self._critic_x(rep, t, t + 1), 1, rep.done_env_in[t : t + 1],
                                                         rep.h_critic, True, None, 1, E, E, 1, training=False,
                                                         y_out=rep.value_env)
                    rep.value[t].copy_(rep.value_env.view(E, 1).expand(E, A))
                else:
                    self.critic_network.forward_sequence(pc, ws, self._critic_x(rep, t, t + 1), self.critic_share, d1,
                                                         rep.h_critic, True, None, 1, EA, E, A, training=False,
                                                         y_out=rep.value[t])
                if self.Hd == H:
                    rep.h_critic, ws.hs = ws.hs, rep.h_critic
                else:
                    n_c = E if self.critic_agg else EA
                    rep.h_critic[: n_c * self.Hd].copy_(self.critic_network.last_hidden(ws, 1, n_c))
                rep.env.step_into(rep.state, step + 1, rep.obs_slot(t + 1), rep.reward[t], rep.done[t], rep.info_return[n, t],
                                  rep.info_length[n, t], rep.info_terminal[n, t], action=None if self.continuous else rep.action[t])
                rep.dones.copy_(rep.done[t])
                if t == self.T - 1:
                    rep.last_reward.copy_(rep.reward[t])
        if fused:
            for rep in self.reps:  # transition.done = the flag entering each step (rec_mappo.py:136-137), in two copies
                rep.done_in[0].copy_(rep.dones)
                rep.done_in[1:].copy_(rep.done[: self.T - 1])
                rep.dones.copy_(rep.done[self.T - 1])
                rep.last_reward.copy_(rep.reward[self.T - 1])
                if self.critic_agg:
                    rep.done_env_in.copy_(rep.done_in[:, :, :1])
        self.t_global += self.T

    def _bootstrap_and_gae(self) -> None:
        """rec_mappo.py:155-199: last_val from one more critic step (hidden state NOT advanced), then GAE with
        next_done masking seeded by the post-rollout dones."""
        s = self.config.system
        pc = self.p[self.Pa :]
        E, A, T = self.E, self.A, self.T
        EA = E * A
        for rep in self.reps:
            if self.critic_agg:
                rep.done_env.copy_(rep.dones[:, :1])
                self.critic_network.forward_sequence(pc, self.ws_roll, self._critic_x(rep, T, T + 1), 1, rep.done_env.view(1, E, 1),
                                                     rep.h_critic, True, None, 1, E, E, 1, training=False, y_out=rep.value_env)
                rep.last_val.view(E, A).copy_(rep.value_env.view(E, 1).expand(E, A))
            else:
                self.critic_network.forward_sequence(pc, self.ws_roll, self._critic_x(rep, T, T + 1), self.critic_share,
                                                     rep.dones.view(1, E, A), rep.h_critic, True, None, 1, EA, E, A, training=False,
                                                     y_out=rep.last_val)
            ops.gae(rep.reward.view(T, EA), rep.value.view(T, EA), rep.done_in.view(T, EA), rep.last_val.view(EA),
                    float(s.gamma), float(s.gae_lambda), last_done=rep.dones.view(EA), out=(rep.adv.view(T, EA), rep.tgt.view(T, EA)),
                    ctx=self.ctx)

    def _minibatch(self, n: int, k: int, mb: int, perm: torch.Tensor) -> None:
        s = self.config.system
        T, E, A, Em, Rm = self.T, self.E, self.A, self.Em, self.Rm
        pa, pc = self.p[: self.Pa], self.p[self.Pa :]
        idx = perm[mb * Em : (mb + 1) * Em].contiguous()
        flat_rows = (self._t_range32 + idx[None, :]).reshape(-1)  # (t*E + env) rows of the minibatch, int32 (T * E < 2^31): one launch
        L = lib()
        # f16x2: the output path (post_torso -> head -> loss -> backward) of both networks runs as ONE launch each
        # (mava_rec_out_f32); the continuous head and more than 16 actions stay on the layer-wise kernels
        fused_out = self.fused_out

        def actor_part(ws, slabs):
            """rec_mappo.py:210-242 for every replica, on the current stream."""
            st = stream_ptr()
            nblk = ws.loss_partials.shape[0]
            for u, rep in enumerate(self.reps):
                acc = u > 0
                self.actor_network.forward_sequence(pa, ws, rep.agents_view[:T], 1, rep.done_in, rep.h0_actor, False, idx, T, Rm, E, A,
                                                    training=True, stop_after_scan=fused_out)
                ops.adv_stats(rep.adv.view(-1), flat_rows, 0, T * Em, A, out=self.stats)
                if fused_out:
                    ok = self.actor_network.fused_output(pa, ws, idx, T, Rm, E, A, 1, True, rep.action_mask[:T], rep.action, rep.log_prob,
                                                         rep.adv, self.stats, float(s.clip_eps), float(s.ent_coef), slabs,
                                                         self.g[: self.Pa], self.g[self.P : self.P + 2], acc, self.grad_scale)
                    assert ok, "mava_rec_out_f32 refused a shape RecLearner.fused_out admitted"
                elif self.continuous:
                    ind = not self.dep_std
                    check(L.mava_seq_actor_loss_continuous_f32(
                        T, Rm, E, A, self.nA, self.min_scale, ptr(idx), ptr(ws.y), ptr(self.actor_network.log_std(pa)) if ind else None,
                        None if ind else ptr(ws.y2), ptr(rep.action),
                        ptr(rep.log_prob), ptr(rep.adv), ptr(self.stats), self.stats.shape[0], float(s.clip_eps), float(s.ent_coef),
                        self.seed & (2**64 - 1), self.ent_step & 0xFFFFFFFF, ((self.rank * self.U + u) * T * E * A) & 0xFFFFFFFF,
                        self.grad_scale, ptr(ws.dy), None if ind else ptr(ws.dy2), ptr(ws.loss_partials), ptr(self.dscale_partials), nblk, st),
                          "mava_seq_actor_loss_continuous_f32")
                    if ind:
                        ops.slab_reduce(self.dscale_partials, self.nA, self.actor_network.log_std(self.g[: self.Pa]), accumulate=acc)
                else:
                    check(L.mava_seq_actor_loss_f32(T, Rm, E, A, self.nA, ptr(idx), ptr(ws.y), ptr(rep.action_mask[:T]), ptr(rep.action),
                                                    ptr(rep.log_prob), ptr(rep.adv), ptr(self.stats), self.stats.shape[0],
                                                    float(s.clip_eps), float(s.ent_coef), self.grad_scale, ptr(ws.dy), ptr(ws.loss_partials),
                                                    nblk, st),
                          "mava_seq_actor_loss_f32")
                if not fused_out:
                    ops.slab_reduce(ws.loss_partials, 2, self.g[self.P : self.P + 2], accumulate=acc)
                self.actor_network.backward_sequence(pa, ws, rep.agents_view[:T], 1, rep.done_in, idx, T, Rm, E, A, slabs,
                                                     self.g[: self.Pa], accumulate=acc, grad_scale=self.grad_scale, from_scan=fused_out)

        def critic_part(ws, slabs):
            """rec_mappo.py:244-266 for every replica, on the current stream."""
            st = stream_ptr()
            nblk = ws.loss_partials.shape[0]
            for u, rep in enumerate(self.reps):
                acc = u > 0
                cx = self._critic_x(rep, 0, T)
                if self.critic_agg:  # E-row sequences: kernel view (E envs x 1 "agent"), A agent slots per row in the loss
                    c_share, c_done, c_Rm, c_A, c_apr = 1, rep.done_env_in, self.Rmc, 1, A
                else:
                    c_share, c_done, c_Rm, c_A, c_apr = self.critic_share, rep.done_in, Rm, A, 1
                self.critic_network.forward_sequence(pc, ws, cx, c_share, c_done, rep.h0_critic, False, idx, T, c_Rm, E, c_A,
                                                     training=True, stop_after_scan=fused_out)
                if fused_out:
                    ok = self.critic_network.fused_output(pc, ws, idx, T, c_Rm, E, c_A, c_apr, False, None, None, rep.value, rep.tgt, None,
                                                          float(s.clip_eps), float(s.vf_coef), slabs, self.g[self.Pa : self.P],
                                                          self.g[self.P + 2 : self.P + 3], acc, self.grad_scale)
                    assert ok, "mava_rec_out_f32 refused a shape RecLearner.fused_out admitted"
                else:
                    check(L.mava_seq_critic_loss_f32(T, c_Rm, E, c_A, c_apr, ptr(idx), ptr(ws.y), ptr(rep.value), ptr(rep.tgt),
                                                     float(s.clip_eps), float(s.vf_coef), self.grad_scale, ptr(ws.dy), ptr(ws.loss_partials),
                                                     nblk, st),
                          "mava_seq_critic_loss_f32")
                    ops.slab_reduce(ws.loss_partials, 1, self.g[self.P + 2 : self.P + 3], accumulate=acc)
                self.critic_network.backward_sequence(pc, ws, cx, c_share, c_done, idx, T, c_Rm, E, c_A, slabs,
                                                      self.g[self.Pa : self.P], accumulate=acc, grad_scale=self.grad_scale,
                                                      from_scan=fused_out)

        from . import _lib as _l
        if self.overlap_critic and _l.TIMERS is None:  # (bench.py's per-kernel timers measure the kernels one at a time)
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)  # parameters of the last Adam step, idx, advantages / targets
            with torch.cuda.stream(self.side):
                critic_part(self.ws_c, self.slabs_c)
            actor_part(self.ws, self.slabs)
            main.wait_stream(self.side)
        else:
            actor_part(self.ws, self.slabs)
            critic_part(self.ws, self.slabs)
        parallel.allreduce_sum_(self.g)
        ops.clip_adam(self.p, self.g, self.m, self.v, self.count, self.seg_off, self.seg_lr,
                      grad_scale=1.0 / (self.U * self.world), max_norm=float(s.max_grad_norm),
                      decay=bool(s.decay_learning_rates), steps_per_update=self.K * self.M,
                      num_updates=int(s.get("num_updates", 1) or 1), loss_sums=self.g[self.P :], vf_coef=float(s.vf_coef),
                      ent_coef=float(s.ent_coef), metrics_out=self.train_metrics[n, k, mb])
        self.ent_step += 1

    def update(self, n: int, permutations: Optional[List[torch.Tensor]] = None) -> None:
        self._rollout(n)
        self._bootstrap_and_gae()
        for k in range(self.K):
            if permutations is not None:
                perm = permutations[k]
            else:  # rec_mappo.py:277-279: a permutation of the env axis per epoch
                perm = ops.permutation(self.E, self.seed, self.perm_count)
                self.perm_count += 1
            for mb in range(self.M):
                self._minibatch(n, k, mb, perm)
        for rep in self.reps:
            rep.agents_view[0].copy_(rep.agents_view[self.T])
            rep.global_state[0].copy_(rep.global_state[self.T])
            rep.action_mask[0].copy_(rep.action_mask[self.T])
            rep.step_count[0].copy_(rep.step_count[self.T])

    def adopt(self, state: RNNLearnerState) -> None:
        """Make the parameter / optimiser buffers equal to `state` (no-op for trees that already alias them), so
        that learn() follows its argument like the reference's pure learner_fn - e.g. params restored from a
        checkpoint (rec_mappo.py:558-566).  Environment and hidden state stay with the learner."""
        pa = next(iter(state.params.actor_params["params"]["pre_torso"].values()))["kernel"]  # Dense_0 | Conv_0: the flat buffer's start
        if not isinstance(pa, torch.Tensor) or pa.data_ptr() != self.p.data_ptr():
            dev = self.p.device
            to = lambda tree: self._tree_to(tree, dev)
            self.actor_network.flat_from_tree(to(state.params.actor_params), self.p[: self.Pa])
            self.critic_network.flat_from_tree(to(state.params.critic_params), self.p[self.Pa :])
            for i, (net, sl, st) in enumerate(((self.actor_network, slice(0, self.Pa), state.opt_states.actor_opt_state),
                                               (self.critic_network, slice(self.Pa, self.P), state.opt_states.critic_opt_state))):
                net.flat_from_tree(to(st.mu), self.m[sl])
                net.flat_from_tree(to(st.nu), self.v[sl])
                self.count[i] = int(torch.as_tensor(st.count).reshape(-1)[0])

    @staticmethod
    def _tree_to(tree, device):
        if isinstance(tree, torch.Tensor):
            return tree.to(device)
        if isinstance(tree, dict):
            return {k: RecLearner._tree_to(v, device) for k, v in tree.items()}
        return tree

    def learn(self, learner_state: RNNLearnerState) -> ExperimentOutput:
        self.adopt(learner_state)
        if self.matmul_mode == "f16x2":
            obs = [t for r in self.reps for t in (r.agents_view[0], r.global_state[0] if self.centralised else None)]
            check_f16_range(self.p, obs, self.train_metrics if self._learn_calls else None, type(self).__name__)
        self._learn_calls += 1
        for n in range(self.n_upd):
            self.update(n)
        U = self.U
        episode_metrics = {
            "episode_return": torch.stack([r.info_return for r in self.reps], 1).unsqueeze(0),
            "episode_length": torch.stack([r.info_length for r in self.reps], 1).unsqueeze(0),
            "is_terminal_step": torch.stack([r.info_terminal for r in self.reps], 1).unsqueeze(0).bool(),
        }
        tm = self.train_metrics.unsqueeze(1).expand(self.n_upd, U, self.K, self.M, 4).unsqueeze(0)
        train_metrics = {"total_loss": tm[..., 0], "value_loss": tm[..., 1], "actor_loss": tm[..., 2], "entropy": tm[..., 3]}
        return ExperimentOutput(self.learner_state(), episode_metrics, train_metrics)


def learner_setup(env, keys, config, centralised_critic: bool, device=None):
    """Counterpart of learner_setup (rec_mappo.py:437-576): (learn, actor_network, init RNNLearnerState)."""
    key, actor_key, critic_key = (int(k) for k in keys)
    learner = RecLearner(env, config, centralised_critic, device)
    learner.seed = key
    learner.init_params(actor_key, critic_key)
    parallel.broadcast_(learner.p, src=0)
    learner.reset_envs()

    def learn(learner_state: RNNLearnerState) -> ExperimentOutput:
        return learner.learn(learner_state)

    learn.learner = learner  # type: ignore[attr-defined]
    return learn, learner.actor_network, learner.learner_state()
