"""GPU parity of the general network path (mava_amd/generic_networks.py, csrc/generic_layers.hip) against
oracle/generic_oracle.py (torch float64, autograd; torch's conv2d as the independent convolution)."""
import numpy as np
import pytest
import torch

from oracle import generic_oracle as go
from tests.conftest import assert_close

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


@pytest.fixture(params=[0, 1], ids=["f32", "f16x2"])
def mode(request):
    """The arithmetic of the products lives in a context handle (no process-wide setting): _net() hands it to the nets."""
    from mava_amd._lib import Ctx

    global _CTX
    _CTX = Ctx("f16x2" if request.param == 1 else "f32")
    yield request.param
    _CTX.close()
    _CTX = None


_CTX = None


def _net(case, din, heads, raw_tail=0):
    from mava_amd.generic_networks import CNNTorso, GenericMLPTorso, GenericNet

    if case["kind"] == "mlp":
        torso = GenericMLPTorso(case["sizes"], case["act"], case["ln"])
        spec = go.spec_mlp(din, case["sizes"], [h[1] for h in heads], case["act"], case["ln"], raw_tail)
        net = GenericNet(torso, din, heads, raw_tail=raw_tail)
    else:
        torso = CNNTorso(case["channels"], case["kernels"], case["strides"], case["act"], case["ln"])
        spec = go.spec_cnn(case["shape"], case["channels"], case["kernels"], case["strides"], [h[1] for h in heads], case["act"], case["ln"], raw_tail)
        net = GenericNet(torso, din, heads, obs_shape=case["shape"], raw_tail=raw_tail)
    net.ctx = _CTX
    return net, spec


CASES = [
    dict(kind="mlp", sizes=[64, 96], act="tanh", ln=True, din=37),
    dict(kind="mlp", sizes=[256, 128], act="relu", ln=False, din=70),
    dict(kind="mlp", sizes=[128], act="relu", ln=True, din=20),
    dict(kind="cnn", shape=(5, 5, 3), channels=[8, 16], kernels=[3, 3], strides=[1, 2], act="relu", ln=True, din=75),
    dict(kind="cnn", shape=(4, 6, 2), channels=[32, 32], kernels=[3, 3], strides=[1, 1], act="tanh", ln=False, din=48),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['kind']}-{c.get('sizes', c.get('channels'))}-{c['act']}-ln{int(c['ln'])}")
def test_generic_forward_and_gradients(dev, case, mode):
    """Discrete actor and critic of one configuration: head outputs, losses and flat gradients on a gathered minibatch."""
    from mava_amd import ops
    from mava_amd._lib import check, lib, ptr, stream_ptr
    from mava_amd.rec_networks import t32_to_rows

    rng = np.random.default_rng(7)
    E, A, Em, nA, din = 24, 4, 16, 6, case["din"]
    Rm = Em * A
    actor, spec_a = _net(case, din, [("logits", nA, 0.01)])
    critic, spec_c = _net(case, din, [("value", 1, 1.0)])
    assert actor.num_params == go.param_count(spec_a) and critic.num_params == go.param_count(spec_c)
    fa, fc = go.init(rng, spec_a).astype(np.float32), go.init(rng, spec_c).astype(np.float32)
    obs = rng.standard_normal((1, E, A, din)).astype(np.float32)
    idx = rng.permutation(E)[:Em].astype(np.int32)
    xg = obs[0, idx].reshape(Rm, din).astype(np.float64)
    mask = rng.random((E, A, nA)) > 0.25
    action = rng.integers(0, nA, (E, A)).astype(np.int32)
    np.put_along_axis(mask, action[..., None].astype(np.int64), True, -1)
    g = lambda x: x[idx].reshape(Rm, *x.shape[2:])

    d = lambda a, dt=None: _t(a, dev, dt)
    obs_d, idx_d, fa_d, fc_d = d(obs), d(idx), d(fa), d(fc)
    ws = actor.workspace(Rm, dev, True)
    logits = actor.forward(fa_d, ws, obs_d, 1, idx_d, Rm, E, A)[0]
    torch.cuda.synchronize()
    want = go.np_forward(fa, spec_a, xg)[0]
    assert_close(t32_to_rows(logits, nA, Rm).cpu().numpy(), want, 1e-5, "logits")

    # ---- actor loss + gradient
    z = np.where(g(mask), want, go.F32_MIN)
    lsm = z - (z.max(-1, keepdims=True) + np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)))
    lp_now = np.take_along_axis(lsm, g(action)[..., None].astype(np.int64), -1)[..., 0]
    old_lp = np.zeros((E, A), np.float32)
    old_lp[idx] = (lp_now + rng.standard_normal(lp_now.shape) * 0.25).reshape(Em, A)
    adv = (rng.standard_normal((E, A)) * 2 + 0.3).astype(np.float32)
    gs = float(2 ** int(np.ceil(np.log2(Rm))))
    adv_d, mask_d, act_d, olp_d = d(adv), d(mask).view(torch.uint8), d(action), d(old_lp)
    stats = ops.adv_stats(adv_d.view(-1), idx_d, 0, Em, A)
    check(lib().mava_seq_actor_loss_f32(1, Rm, E, A, nA, ptr(idx_d), ptr(logits), ptr(mask_d), ptr(act_d), ptr(olp_d), ptr(adv_d),
                                        ptr(stats), stats.shape[0], 0.2, 0.01, gs, ptr(ws.dout[0]), ptr(ws.loss_partials),
                                        ws.loss_partials.shape[0], stream_ptr()), "actor loss")
    ga = torch.zeros(actor.num_params, device=dev)
    actor.backward(fa_d, ws, [ws.dout[0]], ga, accumulate=False, grad_scale=gs)
    torch.cuda.synchronize()
    tot, la, ent, gw = go.actor_loss_grad(fa, spec_a, xg, g(mask), g(action), g(old_lp), g(adv), 0.2, 0.01)
    assert_close(ws.loss_partials.sum(0).cpu().numpy(), np.array([la, ent]), 1e-5, "actor loss / entropy", scale=1.0)
    assert_close(ga.cpu().numpy(), gw, 1e-4, "actor gradient")
    for ly in actor.layers + actor.heads:  # per segment: a small segment must not hide behind a large one
        assert_close(ga.cpu().numpy()[ly.w : ly.w + ly.K * ly.N], gw[ly.w : ly.w + ly.K * ly.N], 1e-4, f"kernel grad {ly.name}")
        assert_close(ga.cpu().numpy()[ly.b : ly.b + ly.N], gw[ly.b : ly.b + ly.N], 1e-4, f"bias grad {ly.name}")
        if ly.ln:
            assert_close(ga.cpu().numpy()[ly.lnb : ly.lnb + ly.N], gw[ly.lnb : ly.lnb + ly.N], 1e-4, f"layer-norm bias grad {ly.name}")

    # ---- critic
    wsc = critic.workspace(Rm, dev, True)
    v = critic.forward(fc_d, wsc, obs_d, 1, idx_d, Rm, E, A)[0]
    torch.cuda.synchronize()
    v_want = go.np_forward(fc, spec_c, xg)[0][:, 0]
    assert_close(v.cpu().numpy(), v_want, 1e-5, "values")  # a (rows x 1) T32 matrix is row-major
    old_v, tgt = np.zeros((E, A), np.float32), np.zeros((E, A), np.float32)
    old_v[idx] = (v_want + rng.standard_normal(v_want.shape) * 0.2).reshape(Em, A)
    tgt[idx] = (v_want + rng.standard_normal(v_want.shape)).reshape(Em, A)
    ov_d, tg_d = d(old_v), d(tgt)
    check(lib().mava_seq_critic_loss_f32(1, Rm, E, A, 1, ptr(idx_d), ptr(v), ptr(ov_d), ptr(tg_d), 0.2, 0.5, gs, ptr(wsc.dout[0]),
                                         ptr(wsc.loss_partials), wsc.loss_partials.shape[0], stream_ptr()), "critic loss")
    gc = torch.zeros(critic.num_params, device=dev)
    critic.backward(fc_d, wsc, [wsc.dout[0]], gc, accumulate=False, grad_scale=gs)
    torch.cuda.synchronize()
    tot, vl, gw = go.critic_loss_grad(fc, spec_c, xg, g(old_v), g(tgt), 0.2, 0.5)
    assert_close(wsc.loss_partials.sum(0).cpu().numpy()[:1], np.array([vl]), 1e-5, "value loss", scale=1.0)
    assert_close(gc.cpu().numpy(), gw, 1e-4, "critic gradient")
    # accumulate = True adds a second replica's gradient
    critic.backward(fc_d, wsc, [wsc.dout[0]], gc, accumulate=True, grad_scale=gs)
    torch.cuda.synchronize()
    assert_close(gc.cpu().numpy(), 2 * gw, 1e-4, "critic gradient, accumulated twice")


@pytest.mark.parametrize("independent", [True, False])
def test_generic_continuous_head(dev, independent, mode):
    """ContinuousActionHead on the general path: observation-independent log_std vector, and
    independent_std=False (networks.py:140,161: log_std = Dense(action_dim)(embedding))."""
    from mava_amd import ops
    from mava_amd._lib import check, lib, ptr, stream_ptr
    from mava_amd.rec_networks import t32_to_rows
    from oracle import tanh_normal as tn

    rng = np.random.default_rng(3)
    E, A, Em, dim, din = 16, 4, 8, 3, 22
    Rm = Em * A
    case = dict(kind="mlp", sizes=[64, 64], act="relu", ln=False)
    heads = [("mean", dim, 0.01)] + ([] if independent else [("log_std", dim, 0.01)])
    net, spec = _net(case, din, heads, raw_tail=dim if independent else 0)
    flat = go.init(rng, spec).astype(np.float32)
    obs = rng.standard_normal((1, E, A, din)).astype(np.float32)
    idx = rng.permutation(E)[:Em].astype(np.int32)
    xg = obs[0, idx].reshape(Rm, din).astype(np.float64)
    g = lambda x: x[idx].reshape(Rm, *x.shape[2:])
    action = np.tanh(rng.standard_normal((E, A, dim))).astype(np.float32)
    old_lp = (rng.standard_normal((E, A)) - 2.0).astype(np.float32)
    adv = (rng.standard_normal((E, A)) * 2 + 0.3).astype(np.float32)
    d = lambda a, dt=None: _t(a, dev, dt)
    obs_d, idx_d, flat_d = d(obs), d(idx), d(flat)
    ws = net.workspace(Rm, dev, True)
    outs = net.forward(flat_d, ws, obs_d, 1, idx_d, Rm, E, A)
    torch.cuda.synchronize()
    wants = go.np_forward(flat, spec, xg)
    for o, w_ in zip(outs, wants):
        assert_close(t32_to_rows(o, dim, Rm).cpu().numpy(), w_, 1e-5, "head output")
    gs = float(2 ** int(np.ceil(np.log2(Rm))))
    seed, ent_step = 1234, 5
    adv_d, act_d, olp_d = d(adv), d(action), d(old_lp)
    stats = ops.adv_stats(adv_d.view(-1), idx_d, 0, Em, A)
    dsp = torch.zeros((ws.loss_partials.shape[0], dim), device=dev)
    log_std_vec = flat_d[net.num_net_params :] if independent else None
    check(lib().mava_seq_actor_loss_continuous_f32(
        1, Rm, E, A, dim, 1e-3, ptr(idx_d), ptr(outs[0]), ptr(log_std_vec), None if independent else ptr(outs[1]), ptr(act_d), ptr(olp_d),
        ptr(adv_d), ptr(stats), stats.shape[0], 0.2, 0.01, seed, ent_step, 0, gs, ptr(ws.dout[0]),
        None if independent else ptr(ws.dout[1]), ptr(ws.loss_partials), ptr(dsp), ws.loss_partials.shape[0], stream_ptr()), "loss")
    gw_d = torch.zeros(net.num_params, device=dev)
    net.backward(flat_d, ws, ws.dout[: len(heads)], gw_d, accumulate=False, grad_scale=gs)
    if independent:
        ops.slab_reduce(dsp, dim, gw_d[net.num_net_params :])
    torch.cuda.synchronize()
    # entropy noise of the minibatch rows: Philox counter = trajectory row (env * A + agent), as the kernel draws it
    gid = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
    eps = tn.normal_noise(seed, ent_step, 0, dim, tn.STREAM_ENTROPY, row_offset=0, gid=gid).astype(np.float64)
    tot, la, ent, gw = go.actor_loss_grad_continuous(flat, spec, xg, g(action), g(old_lp), g(adv), 0.2, 0.01, eps, independent)
    assert_close(ws.loss_partials.sum(0).cpu().numpy(), np.array([la, ent]), 1e-5, "actor loss / entropy", scale=1.0)
    assert_close(gw_d.cpu().numpy(), gw, 1e-4, "continuous actor gradient")
    if not independent:
        hd = net.heads[1]
        assert_close(gw_d.cpu().numpy()[hd.w : hd.w + hd.K * hd.N], gw[hd.w : hd.w + hd.K * hd.N], 1e-4, "log_std head kernel grad")


E2E = [
    ("ff_mappo", dict(kind="mlp", sizes=[64, 96], act="tanh", ln=True), False, True),
    ("ff_ippo", dict(kind="cnn", shape=(2, 4, 2), channels=[8, 8], kernels=[3, 3], strides=[1, 1], act="relu", ln=False), False, True),
    ("ff_ippo", dict(kind="mlp", sizes=[128, 128], act="relu", ln=False), True, False),  # state-dependent log_std head
]


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,case,continuous,independent", E2E,
                         ids=["mappo-mlp-tanh-ln", "ippo-cnn", "ippo-continuous-dependent-std"])
def test_generic_learner_update_matches_oracle(dev, system, case, continuous, independent, matmul, monkeypatch):
    """ff_ippo / ff_mappo with a non-default network configuration: the whole update (rollout, GAE, epochs x minibatches,
    Adam) of the layer-wise path against the whole-update oracle on identical inputs, through the LearnerFn boundary."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_ippo, ff_mappo
    from oracle.ppo_loop import OracleLearner
    from tests.conftest import check_and_sync_f16x2_state

    monkeypatch.setenv("MAVA_MATMUL", matmul)
    E, A, O, nA, T, K, M, U = 16, 4, 12, 5, 8, 2, 2, 1
    dim = 3
    central = system == "ff_mappo"
    over = [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}", f"system.num_minibatches={M}",
            f"system.update_batch_size={U}", "env/scenario=tiny-4ag"]
    if continuous:
        over.append("network.action_head._target_=mava.networks.ContinuousActionHead")
    cfg = compose(f"default_{system}", over)
    cfg.env.synthetic = {"obs_dim": O, "num_actions": dim if continuous else nA}
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    if continuous and not independent:
        cfg.network.action_head.independent_std = False
    Oa, Oc = A + O, (A * O if central else A + O)
    n_act = dim if continuous else nA
    heads_a = [n_act] + ([n_act] if (continuous and not independent) else [])
    tail = n_act if (continuous and independent) else 0
    for netcfg in (cfg.network.actor_network.pre_torso, cfg.network.critic_network.pre_torso):
        if case["kind"] == "cnn":
            netcfg._target_ = "mava.networks.CNNTorso"
            for k_ in ("layer_sizes",):
                if k_ in netcfg:
                    del netcfg[k_]
            netcfg.channel_sizes, netcfg.kernel_sizes, netcfg.strides = case["channels"], case["kernels"], case["strides"]
        else:
            netcfg.layer_sizes = case["sizes"]
        netcfg.activation, netcfg.use_layer_norm = case["act"], case["ln"]
    if case["kind"] == "cnn":
        assert int(np.prod(case["shape"])) == Oa
        cfg.env.synthetic["obs_shape"] = list(case["shape"])
        spec_a = go.spec_cnn(case["shape"], case["channels"], case["kernels"], case["strides"], heads_a, case["act"], case["ln"], tail)
        spec_c = go.spec_cnn(case["shape"], case["channels"], case["kernels"], case["strides"], [1], case["act"], case["ln"])
    else:
        spec_a = go.spec_mlp(Oa, case["sizes"], heads_a, case["act"], case["ln"], tail)
        spec_c = go.spec_mlp(Oc, case["sizes"], [1], case["act"], case["ln"])
    mod = ff_mappo if central else ff_ippo
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    assert L.generic and L.Pa == go.param_count(spec_a) and L.Pc == go.param_count(spec_c)
    first = actor_network.net.layers[0]
    leaf = state.params.actor_params["params"]["torso"][first.name]["kernel"]
    assert leaf.shape[:2] == (1, U) and int(np.prod(leaf.shape[2:])) == first.K * first.N

    rng = np.random.default_rng(0)
    fa, fc = go.init(rng, spec_a).astype(np.float32), go.init(rng, spec_c).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleLearner(E=E, A=A, O=O, nA=n_act, T=T, K=K, M=M, U=U, D=1, centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3,
                        continuous=continuous, actor_spec=spec_a, critic_spec=spec_c, independent_std=independent)
    ora.set_params(fa, fc)
    ftol = 1e-5 if matmul == "f32" else 5e-5
    for n in range(2):
        perms = [rng.permutation(T * E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        rep, tr = L.reps[0], ora.last_traj[0][0]
        if continuous:
            assert_close(rep.action.cpu().numpy(), tr["action"], 1e-5 if matmul == "f32" else 1e-4, "actions", scale=1.0)
        else:
            assert np.array_equal(rep.action.cpu().numpy(), tr["action"]), "sampled actions differ"
        assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
        assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], 1e-4 if continuous else ftol, "log_probs", scale=1.0)
        assert_close(rep.adv.cpu().numpy(), tr["adv"], ftol, "advantages")
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":
            assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, 1e-5, "actor params")
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, 1e-5, "critic params")
        else:
            check_and_sync_f16x2_state(L, ora)
    out = learn(L.learner_state())  # the LearnerFn round trip (adopt -> updates -> state)
    torch.cuda.synchronize()
    assert torch.isfinite(out.train_metrics["total_loss"]).all()
    # the evaluator seam: actor_network.apply(params, observation) -> distribution
    obs = out.learner_state.timestep.observation
    dist = actor_network.apply(out.learner_state.params.actor_params, type(obs)(*[None if v is None else v[0, 0] for v in obs]))
    assert dist.mode().shape[:2] == (E, A)
