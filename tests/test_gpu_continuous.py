"""GPU parity of the continuous action head (networks.py:127-169, distributions.py:24-91) against
oracle/tanh_normal.py, through the C ABI.  The oracle restates tensorflow_probability's published formulas
(checked against scipy in tests/test_oracle.py) - parity unpinned with respect to tfp itself."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import tanh_normal as tn
from tests.conftest import assert_close, check_and_sync_f16x2_state

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _params(rng, din, dim, head_scale=1.0):
    mlp = po.mlp_flatten(po.init_mlp(rng, din, dim, head_scale))
    return np.concatenate([mlp, rng.normal(size=dim) * 0.4]).astype(np.float32)


@pytest.fixture(params=[1e-3, 0.05], ids=["min_scale=1e-3", "min_scale=0.05"])
def min_scale(request, monkeypatch):
    """ContinuousActionHead.min_scale (mava/networks.py:134,162): the reference's default and another value; the oracle
    reads its module constant at call time."""
    monkeypatch.setattr(tn, "MIN_SCALE", request.param)
    return request.param


@pytest.mark.parametrize("rows,din,dim,share", [(64, 20, 2, 1), (1000, 31, 6, 1), (96, 50, 9, 4), (33, 7, 1, 1),
                                                (16384, 22, 2, 4)])
def test_continuous_policy_step_matches_oracle(dev, rows, din, dim, share, min_scale):
    from functools import partial

    from mava_amd import ops as _ops

    class ops:  # every call of this test with the fixture's min_scale
        policy_step_continuous = staticmethod(partial(_ops.policy_step_continuous, min_scale=min_scale))

    rng = np.random.default_rng(rows + dim)
    fa = _params(rng, din, dim)
    cdin = 3 * din
    fc = po.mlp_flatten(po.init_mlp(rng, cdin, 1, 1.0)).astype(np.float32)
    av = rng.standard_normal((rows, din)).astype(np.float32)
    cx = rng.standard_normal((rows // share, cdin)).astype(np.float32) if rows % share == 0 else None
    if cx is None:
        share, cx = 1, rng.standard_normal((rows, cdin)).astype(np.float32)
    seed, step, off = 0x1234567890, 17, 1000
    base = torch.tensor([5], dtype=torch.int32, device=dev)
    act, lp, val, mean = ops.policy_step_continuous(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(cx, dev), action_dim=dim,
                                                    critic_share=share, seed=seed, step=step - 5, step_base=base,
                                                    row_offset=off, want_mean=True)
    torch.cuda.synchronize()
    fm, ls = tn.split_params(fa.astype(np.float64), din, dim)
    mean64 = po.mlp_forward(po.mlp_unflatten(fm, din, dim), av.astype(np.float64))
    assert_close(mean.cpu().numpy(), mean64, 1e-5, "mean")
    eps = tn.normal_noise(seed, step, rows, dim, tn.STREAM_SAMPLE, row_offset=off)
    a64, lp64 = tn.sample(mean64, ls, eps.astype(np.float64))
    a = act.cpu().numpy()
    assert np.all(np.abs(a) <= 1.0)
    assert_close(a, a64, 1e-5, "actions", scale=1.0)
    # log-prob of the actions the kernel actually took (atanh near the clip magnifies a 1-ulp action difference)
    assert_close(lp.cpu().numpy(), tn.log_prob(a.astype(np.float64), mean64, ls), 2e-5, "log_prob", scale=1.0)
    v64 = po.mlp_forward(po.mlp_unflatten(fc.astype(np.float64), cdin, 1), cx.astype(np.float64))[:, 0]
    assert_close(val.cpu().numpy(), np.repeat(v64, share), 1e-5, "value")

    # greedy = mode; forced actions (including both clipped tails) are scored, not sampled
    g, _, _, _ = ops.policy_step_continuous(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(cx, dev), action_dim=dim,
                                            critic_share=share, seed=seed, step=step, greedy=True)
    assert_close(g.cpu().numpy(), np.tanh(mean64), 1e-5, "mode", scale=1.0)
    forced = rng.uniform(-1, 1, size=(rows, dim)).astype(np.float32)
    forced[0, 0], forced[1 % rows, dim - 1] = 0.99999, -1.0
    f, flp, _, _ = ops.policy_step_continuous(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(cx, dev), action_dim=dim,
                                              critic_share=share, seed=seed, step=step, forced_action=_t(forced, dev))
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy(), forced)
    assert_close(flp.cpu().numpy(), tn.log_prob(forced.astype(np.float64), mean64, ls), 2e-5, "forced log_prob", scale=1.0)


@pytest.mark.parametrize("TE,A,O,dim,Rb,use_idx,n_slab", [(64, 4, 20, 2, 64, False, 3), (200, 2, 60, 6, 77, True, 8),
                                                          (96, 3, 100, 9, 40, True, 2), (33, 1, 7, 1, 33, True, 1),
                                                          (4096, 4, 22, 2, 2048, True, 256), (80, 2, 180, 4, 50, True, 3)])
def test_continuous_actor_grad_matches_oracle(dev, TE, A, O, dim, Rb, use_idx, n_slab, min_scale):
    from mava_amd import ops

    rng = np.random.default_rng(TE + dim)
    rows, din = TE * A, O
    flat = _params(rng, din, dim)
    av = rng.standard_normal((rows, din)).astype(np.float32)
    fm, ls = tn.split_params(flat.astype(np.float64), din, dim)
    mean64 = po.mlp_forward(po.mlp_unflatten(fm, din, dim), av.astype(np.float64))
    action = np.tanh(mean64 + tn.scale_of(ls) * rng.standard_normal((rows, dim))).astype(np.float32)
    action[rng.random((rows, dim)) < 0.02] = 0.9995   # some actions in the clipped tails
    action[rng.random((rows, dim)) < 0.02] = -1.0
    # old log-probs close to the current ones: both sides of the clip range
    old_lp = (tn.log_prob(action.astype(np.float64), mean64, ls) + rng.standard_normal(rows) * 0.25).astype(np.float32)
    adv = rng.standard_normal(rows).astype(np.float32)
    idx = rng.permutation(TE)[:Rb].astype(np.int32) if use_idx else np.arange(Rb, dtype=np.int32)
    rows_sel = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
    seed, ent_step, off = 99, 12345, 777

    P = flat.size
    slab = torch.zeros((n_slab, P + 2), device=dev)
    stats = ops.adv_stats(_t(adv, dev), _t(idx, dev) if use_idx else None, 0, Rb, A)
    ops.ppo_actor_grad_continuous(_t(flat, dev), _t(av, dev), _t(action, dev), _t(old_lp, dev), _t(adv, dev), stats,
                                  _t(idx, dev) if use_idx else None, 0, Rb, A, dim, 0.2, 0.01, seed, ent_step, off, slab,
                                  min_scale=min_scale)
    out = torch.zeros(P + 2, device=dev)
    ops.slab_reduce(slab, P + 2, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()

    eps = tn.normal_noise(seed, ent_step, 0, dim, tn.STREAM_ENTROPY, row_offset=off, gid=rows_sel).astype(np.float64)
    tot, la, ent, g = tn.actor_loss_and_grad(flat.astype(np.float64), din, dim, av[rows_sel].astype(np.float64),
                                             action[rows_sel].astype(np.float64), old_lp[rows_sel].astype(np.float64),
                                             adv[rows_sel].astype(np.float64), 0.2, 0.01, eps)
    # north_star: PPO gradients within 1e-4 rtol (tolerance form of BASELINE.md §2)
    assert_close(got[: P - dim], g[: P - dim], 1e-4, "actor MLP grad")
    assert_close(got[P - dim : P], g[P - dim :], 1e-4, "log_std grad")
    assert_close(got[P:], np.array([la, ent]), 1e-5, "actor loss/entropy", scale=1.0)


def test_continuous_actor_grad_rejects_unsupported_shapes(dev):
    from mava_amd import ops
    from mava_amd._lib import MavaHipError

    z = torch.zeros
    nb = ops.lib().mava_adv_stats_blocks()
    for din, dim, msg in ((300, 2, "not instantiated"), (264, 2, "LDS"), (20, 17, "action_dim")):
        TE = 32
        flat = z(ops.continuous_param_count(din, dim), device=dev)
        with pytest.raises(MavaHipError, match=msg):
            ops.ppo_actor_grad_continuous(flat, z((TE, din), device=dev), z((TE, dim), device=dev), z(TE, device=dev),
                                          z(TE, device=dev), z((nb, 2), dtype=torch.float64, device=dev), None, 0, TE, 1,
                                          dim, 0.2, 0.01, 1, 1, 0, z((1, flat.numel() + 2), device=dev))


@pytest.mark.parametrize("system,U", [("ff_mappo", 2), ("ff_ippo", 1)])
@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
def test_continuous_learner_update_matches_oracle(dev, system, U, matmul):
    """The whole PPO update with network.action_head = ContinuousActionHead (the reference selects it with exactly this
    override for MaBrax) on a MaBrax-shaped synthetic env, against the NumPy whole-update oracle.  The continuous actor
    runs the exact-f32 kernels in either mode; matmul selects the critic's arithmetic (f16x2: end-to-end tolerances of
    tests/conftest.py:check_and_sync_f16x2_state)."""
    from tests.conftest import check_and_sync_f16x2_state
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_ippo, ff_mappo
    from oracle.ppo_loop import OracleLearner

    E, A, O, dim, T, K, M = 8, 2, 12, 3, 16, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}",
                                        "network.action_head._target_=mava.networks.ContinuousActionHead"])
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": dim}
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr = 1e-3
    cfg.system.critic_lr = 2e-3
    cfg.system.matmul_mode = matmul
    central = system == "ff_mappo"
    mod = ff_mappo if central else ff_ippo
    env, eval_env = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    assert L.continuous and L.reps[0].action.shape == (T, E, A, dim)
    head = state.params.actor_params["params"]["action_head"]
    assert head["mean"]["kernel"].shape == (1, U, 128, dim) and head["log_std"].shape == (1, U, dim)

    rng = np.random.default_rng(0)
    fa = np.concatenate([po.mlp_flatten(po.init_mlp(rng, A + O, dim, 1.0)), rng.normal(size=dim) * 0.3]).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, (A * O) if central else (A + O), 1, 1.0)).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleLearner(E=E, A=A, O=O, nA=dim, T=T, K=K, M=M, U=U, D=1, centralised=central, seed=42, actor_lr=1e-3,
                        critic_lr=2e-3, continuous=True)
    ora.set_params(fa, fc)
    for i in range(3):  # the third update replays the rollout graph captured for n = 0
        n = i % 2
        perms = [rng.permutation(T * E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[0][u]
            assert_close(rep.action.cpu().numpy(), tr["action"], 1e-5, "actions", scale=1.0)
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], 1e-4, "log_probs", scale=1.0)
            assert_close(rep.value.cpu().numpy(), tr["value"], 1e-5, "values")
            assert_close(rep.adv.cpu().numpy(), tr["adv"], 1e-5, "advantages")
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":
            assert_close(L.p[: L.Pa].cpu().numpy() - fa, ora.pa - fa, 2e-3, "actor update")
            assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, 1e-5, "actor params")
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, 1e-5, "critic params")
        else:
            check_and_sync_f16x2_state(L, ora)
    # evaluator seam: the host distribution view agrees with the kernels' acting step
    from mava_amd.evaluator import get_eval_fn, make_ff_eval_act_fn

    out = learn(L.learner_state())
    pi = actor_network.apply(out.learner_state.params.actor_params, out.learner_state.timestep.observation)
    a = pi.mode()
    assert a.shape == (1, U, E, A, dim) and float(a.abs().max()) <= 1.0
    fm, ls = tn.split_params(L.p[: L.Pa].double().cpu().numpy(), A + O, dim)
    mean = pi.loc.double().cpu().numpy()
    acts = np.tanh(mean + 0.3)
    assert_close(pi.log_prob(torch.from_numpy(acts).float().to(dev)).cpu().numpy(), tn.log_prob(acts, mean, ls), 1e-4,
                 "host log_prob", scale=1.0)
    ev = get_eval_fn(eval_env, make_ff_eval_act_fn(actor_network.apply, cfg), cfg, absolute_metric=False)
    m = ev(out.learner_state.params.actor_params, 0)
    assert m["episode_return"].shape[0] >= cfg.arch.num_eval_episodes


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,U,E,fused", [("rec_mappo", 1, 16, "1"), ("rec_ippo", 2, 16, "0"), ("rec_mappo", 1, 64, "1"),
                                              ("rec_mappo", 1, 64, "0")])
def test_continuous_rec_learner_update_matches_oracle(dev, monkeypatch, system, U, E, fused, matmul):
    """The recurrent systems with network.action_head = ContinuousActionHead against the whole-update oracle (BPTT
    gradients from torch autograd in float64, distribution formulas written independently of oracle/tanh_normal.py)."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo, rec_mappo
    from oracle import rec_oracle as ro
    from oracle.rec_loop import OracleRecLearner

    # fused acting step (mava_rec_step_continuous_f32) or the layer-wise one (mava_seq_sample_continuous_f32)
    monkeypatch.setenv("MAVA_REC_FUSED_STEP", fused)
    monkeypatch.setenv("MAVA_MATMUL", matmul)
    ftol = 1e-5 if matmul == "f32" else 5e-5  # values are network outputs: ~22-bit operands in the f16x2 mode
    A, O, dim, T, K, M = 4, 10, 3, 6, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}",
                                        "network.action_head._target_=mava.networks.ContinuousActionHead"])
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": dim}
    cfg.env.kwargs.time_limit = 4
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    central = system == "rec_mappo"
    mod = rec_mappo if central else rec_ippo
    env, eval_env = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    assert L.continuous and L.reps[0].action.shape == (T, E, A, dim)
    head = state.params.actor_params["params"]["action_head"]
    assert head["mean"]["kernel"].shape == (1, U, 128, dim) and head["log_std"].shape == (1, U, dim)

    rng = np.random.default_rng(1)
    Oc = A * O if central else A + O
    fa = np.concatenate([ro.init_rec(rng, A + O, dim, 1.0), rng.normal(size=dim) * 0.3]).astype(np.float32)
    fc = ro.init_rec(rng, Oc, 1, 1.0).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleRecLearner(E=E, A=A, O=O, nA=dim, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3,
                           critic_lr=2e-3, time_limit=4, continuous=True)
    ora.set_params(fa, fc)
    for n in range(2):
        perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[u]
            assert_close(rep.action.cpu().numpy(), tr["action"], 1e-5, "actions", scale=1.0)
            assert np.array_equal(rep.done_in.cpu().numpy().astype(bool), tr["done_in"]) and tr["done_in"].any()
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], 1e-4, "log_probs", scale=1.0)
            assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
            assert_close(rep.adv.cpu().numpy(), tr["adv"], ftol, "advantages")
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":
            assert_close(L.p[: L.Pa].cpu().numpy() - fa, ora.pa - fa, 2e-3, "actor update")
            assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, 1e-5, "actor params")
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, 1e-5, "critic params")
        else:
            check_and_sync_f16x2_state(L, ora)
    # learn() round trip and the recurrent evaluator seam with the continuous distribution view
    out = learn(L.learner_state())
    torch.cuda.synchronize()
    assert torch.isfinite(out.train_metrics["total_loss"]).all()
    from mava_amd.evaluator import get_eval_fn, make_rec_eval_act_fn

    ev = get_eval_fn(eval_env, make_rec_eval_act_fn(actor_network.apply, cfg), cfg, absolute_metric=False)
    init = {"hidden_state": torch.zeros((eval_env.num_envs, A, 128), device=dev)}
    m = ev(out.learner_state.params.actor_params, 0, init)
    assert m["episode_return"].shape[0] >= cfg.arch.num_eval_episodes


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,U", [("rec_mappo", 1), ("rec_ippo", 2)])
def test_rec_learner_state_dependent_std(dev, system, U, matmul):
    """ContinuousActionHead(independent_std=False) on the RECURRENT systems (mava/networks.py:137-141,161 inside
    RecurrentActor, :269-294): the log_std layer is a second head of the actor's post-torso (general layer kernels around the
    GRU scans), whole updates against the oracle with the same two heads (rec_oracle.rec_spec(two_heads=True)), the
    parameter tree with action_head/{mean, log_std}/{kernel, bias}, and the evaluator's distribution view."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo, rec_mappo
    from oracle import rec_oracle as ro
    from oracle.rec_loop import OracleRecLearner

    E, A, O, dim, T, K, M = 16, 4, 10, 3, 6, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}",
                                        "network.action_head._target_=mava.networks.ContinuousActionHead"])
    cfg.network.action_head.independent_std = False
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": dim}
    cfg.env.kwargs.time_limit = 4
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    cfg.system.matmul_mode = matmul
    central = system == "rec_mappo"
    mod = rec_mappo if central else rec_ippo
    env, eval_env = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    assert L.continuous and L.dep_std and L.actor_network.generic and not L.critic_network.generic
    head = state.params.actor_params["params"]["action_head"]
    assert head["mean"]["kernel"].shape == (1, U, 128, dim) and head["log_std"]["kernel"].shape == (1, U, 128, dim)
    assert head["log_std"]["bias"].shape == (1, U, dim)
    assert torch.equal(actor_network.flat_from_tree(state.params.actor_params), L.p[: L.Pa])
    Oc = A * O if central else A + O
    spec_a = ro.rec_spec(A + O, [128], [128], "relu", False, two_heads=True)
    assert L.Pa == ro.rec_param_count(spec_a, dim) and L.Pc == ro.rec_param_count(Oc, 1)

    rng = np.random.default_rng(3)
    fa = (rng.standard_normal(L.Pa) * 0.1).astype(np.float32)
    fc = ro.init_rec(rng, Oc, 1, 1.0).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleRecLearner(E=E, A=A, O=O, nA=dim, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3,
                           critic_lr=2e-3, time_limit=4, continuous=True, actor_net=spec_a)
    ora.set_params(fa, fc)
    ftol = 1e-5 if matmul == "f32" else 5e-5
    for n in range(2):
        perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[u]
            assert_close(rep.action.cpu().numpy(), tr["action"], 5e-5 if matmul == "f16x2" else 1e-5, "actions", scale=1.0)
            assert tr["done_in"].any()
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], 1e-4, "log_probs", scale=1.0)
            assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
            assert_close(rep.adv.cpu().numpy(), tr["adv"], ftol, "advantages")
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":
            assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, 1e-5, "actor params")
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, 1e-5, "critic params")
        else:
            check_and_sync_f16x2_state(L, ora)
    out = learn(L.learner_state())
    torch.cuda.synchronize()
    assert torch.isfinite(out.train_metrics["total_loss"]).all()
    # evaluator seam: the distribution's scale comes from the log_std layer's rows
    from mava_amd.evaluator import get_eval_fn, make_rec_eval_act_fn
    from mava_amd.types import Observation

    rep = L.reps[0]
    obs = Observation(rep.agents_view[:1], None, None)
    _, pi = actor_network.apply(out.learner_state.params.actor_params, torch.zeros((E, A, 128), device=dev),
                                (obs, torch.zeros((1, E, A), dtype=torch.bool, device=dev)))
    assert pi.scale.shape == pi.loc.shape == (1, E, A, dim) and (pi.scale > 0).all() and pi.scale.std() > 0
    h0 = torch.zeros((eval_env.num_envs, A, 128), device=dev)
    ev = get_eval_fn(eval_env, make_rec_eval_act_fn(actor_network.apply, cfg), cfg, absolute_metric=False)
    m = ev(out.learner_state.params.actor_params, 0, {"hidden_state": h0})
    assert m["episode_return"].shape[0] >= cfg.arch.num_eval_episodes
