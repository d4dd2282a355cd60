"""GPU parity of the recurrent path (rec_ippo / rec_mappo kernels) against oracle/rec_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import rec_oracle as ro
from tests.conftest import assert_close, check_and_sync_f16x2_state

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def _to_t32(a):
    """(rows, N) row-major numpy -> T32 flat numpy."""
    rows, N = a.shape
    return a.reshape(rows // 32, 32, N).transpose(0, 2, 1).reshape(-1).copy()


def _from_t32(flat, rows, N):
    return np.asarray(flat).reshape(rows // 32, N, 32).transpose(0, 2, 1).reshape(rows, N)


@pytest.fixture(params=[0, 1], ids=["f32", "f16x2"])
def rec_mode(request):
    """Arithmetic of the dense / X^T Y products on T32 operands: 0 exact-f32 MFMAs, 1 split-f16 operands
    (rec_dense_h2.hip; three f16 MFMAs per product, f32 accumulation)."""
    from mava_amd._lib import Ctx

    global _CTX
    _CTX = Ctx("f16x2" if request.param == 1 else "f32")  # the mode lives in a context handle: no process-wide setting
    yield request.param
    _CTX.close()
    _CTX = None


_CTX = None


def _cp():
    """Context-handle argument of the direct C-ABI calls below (None = the library defaults: exact f32)."""
    return None if _CTX is None else _CTX.handle


@pytest.mark.parametrize("K,N,relu,gated", [(128, 384, False, False), (128, 128, True, False), (128, 13, False, False),
                                             (384, 128, False, True), (5, 128, False, True), (192, 128, True, False), (128, 256, False, False)])
def test_rec_dense_t32(dev, K, N, relu, gated, rec_mode):
    from mava_amd._lib import check, lib, ptr, stream_ptr

    rng = np.random.default_rng(K + N)
    rows = 96
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    g = rng.standard_normal((rows, N)).astype(np.float32)
    y = torch.zeros(rows * N, device=dev)
    xt, wt, bt, gt = _t(_to_t32(x), dev), _t(w, dev), _t(b, dev), _t(_to_t32(g), dev)
    check(lib().mava_rec_dense_f32(_cp(), ptr(xt), 0, None, 0, 0, 0, 1, K, 0, ptr(wt), N, ptr(bt), ptr(gt) if gated else None, ptr(y), 0, K, N,
                                   rows, int(relu), stream_ptr()), "dense")
    torch.cuda.synchronize()
    want = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        want = np.maximum(want, 0)
    if gated:
        want = want * (g > 0)
    assert_close(_from_t32(y.cpu().numpy(), rows, N), want, 1e-5, "dense")


def test_rec_dense_rowmajor_gather_and_xty(dev, rec_mode):
    from mava_amd._lib import check, lib, ptr, stream_ptr
    from mava_amd import ops

    rng = np.random.default_rng(3)
    T, E, A, K, N = 3, 20, 4, 70, 128
    Em = 8  # minibatch envs
    Rm, rows = Em * A, T * Em * A
    obs = rng.standard_normal((T, E, A, K)).astype(np.float32)
    idx = rng.permutation(E)[:Em].astype(np.int32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y = torch.zeros(rows * N, device=dev)
    # NB: every device tensor handed to the C ABI is kept alive in a named variable (a temporary would be
    # freed - and its block recycled by the caching allocator - before the asynchronous kernel reads it)
    obs_d, idx_d, w_d, b_d = _t(obs, dev), _t(idx, dev), _t(w, dev), _t(b, dev)
    check(lib().mava_rec_dense_f32(_cp(), ptr(obs_d), 1, ptr(idx_d), Rm, E, A, 1, K, 0, ptr(w_d), N, ptr(b_d), None,
                                   ptr(y), 0, K, N, rows, 1, stream_ptr()), "dense gather")
    xg = obs[:, idx].reshape(rows, K).astype(np.float64)  # time-major, env-major inside a step
    want = np.maximum(xg @ w.astype(np.float64) + b, 0)
    got = _from_t32(y.cpu().numpy(), rows, N)
    assert_close(got, want, 1e-5, "dense gather")
    # shared input rows (global state stored once per env)
    gs = rng.standard_normal((T, E, K)).astype(np.float32)
    gs_d = _t(gs, dev)
    check(lib().mava_rec_dense_f32(_cp(), ptr(gs_d), 1, ptr(idx_d), Rm, E, A, A, K, 0, ptr(w_d), N, ptr(b_d), None,
                                   ptr(y), 0, K, N, rows, 0, stream_ptr()), "dense gather shared")
    want2 = np.repeat(gs[:, idx], A, 1).reshape(rows, K).astype(np.float64) @ w.astype(np.float64) + b
    assert_close(_from_t32(y.cpu().numpy(), rows, N), want2, 1e-5, "dense gather shared")
    # X^T Y with both input kinds
    dy = rng.standard_normal((rows, N)).astype(np.float32)
    dy_d = _t(_to_t32(dy), dev)
    slab = torch.zeros((5, K * N + N), device=dev)
    check(lib().mava_rec_xty_f32(_cp(), ptr(obs_d), 1, ptr(idx_d), Rm, E, A, 1, K, ptr(dy_d), 0, None, 0, 0, K, N, rows, 1, 1.0,
                                 ptr(slab), slab.shape[1], 5, stream_ptr()), "xty")
    out = torch.zeros(K * N + N, device=dev)
    ops.slab_reduce(slab, K * N + N, out)
    got = out.cpu().numpy()
    assert_close(got[: K * N].reshape(K, N), xg.T @ dy.astype(np.float64), 1e-5, "xty dW")
    assert_close(got[K * N :], dy.astype(np.float64).sum(0), 1e-5, "xty db")
    x2 = rng.standard_normal((rows, 128)).astype(np.float32)
    dy2 = rng.standard_normal((rows, 384)).astype(np.float32)
    x2_d, dy2_d = _t(_to_t32(x2), dev), _t(_to_t32(dy2), dev)
    slab = torch.zeros((3, 128 * 384 + 384), device=dev)
    check(lib().mava_rec_xty_f32(_cp(), ptr(x2_d), 0, None, 0, 0, 0, 1, 128, ptr(dy2_d), 0, None, 0, 0, 128, 384, rows, 1, 0.25,
                                 ptr(slab), slab.shape[1], 3, stream_ptr()), "xty t32")
    out = torch.zeros(128 * 384 + 384, device=dev)
    ops.slab_reduce(slab, out.numel(), out)
    assert_close(out.cpu().numpy()[: 128 * 384].reshape(128, 384), 0.25 * x2.astype(np.float64).T @ dy2.astype(np.float64), 1e-5,
                 "xty t32 (out_scale 0.25)")
    assert_close(out.cpu().numpy()[128 * 384 :], 0.25 * dy2.astype(np.float64).sum(0), 1e-5, "xty t32 db")
    # Y in two pieces (y_tail): features [0, 256) from a 384-wide matrix whose last third holds something else, [256, 384) from
    # a 128-wide one - how the BPTT scan leaves dgh (its r and z thirds are dgi's)
    head = dy2.copy()
    head[:, 256:] = 99.0
    head_d, tail_d = _t(_to_t32(head), dev), _t(_to_t32(np.ascontiguousarray(dy2[:, 256:])), dev)
    slab = torch.zeros((3, 128 * 384 + 384), device=dev)
    check(lib().mava_rec_xty_f32(_cp(), ptr(x2_d), 0, None, 0, 0, 0, 1, 128, ptr(head_d), 0, ptr(tail_d), 256, 128, 128, 384, rows, 1, 0.25,
                                 ptr(slab), slab.shape[1], 3, stream_ptr()), "xty t32, y_tail")
    out2 = torch.zeros(128 * 384 + 384, device=dev)
    ops.slab_reduce(slab, out2.numel(), out2)
    assert torch.equal(out2, out), "X^T Y with Y in two pieces == the same product on the whole matrix, bit for bit"
    # the gathered observations as a padded T32 matrix (mava_rec_gather_t32_f32), then both products on T32 operands:
    # the path the f16x2 arithmetic takes (exact copies in either mode)
    for src_d, share, xs in ((obs_d, 1, xg), (gs_d, A, np.repeat(gs[:, idx], A, 1).reshape(rows, K).astype(np.float64))):
        kp = -(-K // 32) * 32
        xin = torch.full((rows * kp,), 7.0, device=dev)
        check(lib().mava_rec_gather_t32_f32(ptr(src_d), ptr(idx_d), Rm, E, A, share, K, K, rows, kp, ptr(xin), stream_ptr()), "gather")
        got_x = _from_t32(xin.cpu().numpy(), rows, kp)
        assert np.array_equal(got_x[:, :K], xs.astype(np.float32)) and not got_x[:, K:].any(), "gathered T32 input"
        if rec_mode == 1:  # (the exact-f32 kernel reads padded T32 inputs only when K is a multiple of 16)
            check(lib().mava_rec_dense_f32(_cp(), ptr(xin), 0, None, 0, 0, 0, 1, kp, 0, ptr(w_d), N, ptr(b_d), None, ptr(y), 0, K, N, rows, 1,
                                           stream_ptr()), "dense on the gathered input")
            assert_close(_from_t32(y.cpu().numpy(), rows, N), np.maximum(xs @ w.astype(np.float64) + b, 0), 1e-5,
                         "dense, gathered T32")
        slab = torch.zeros((5, K * N + N), device=dev)
        check(lib().mava_rec_xty_f32(_cp(), ptr(xin), 0, None, 0, 0, 0, 1, kp, ptr(dy_d), 0, None, 0, 0, K, N, rows, 1, 1.0, ptr(slab), slab.shape[1], 5,
                                     stream_ptr()), "xty on the gathered input")
        out = torch.zeros(K * N + N, device=dev)
        ops.slab_reduce(slab, K * N + N, out)
        assert_close(out.cpu().numpy()[: K * N].reshape(K, N), xs.T @ dy.astype(np.float64), 1e-5, "xty dW, gathered T32")


def _seq_case(rng, T, E, A, Em, din, nA, shared):
    obs = rng.standard_normal((T, E, 1 if shared else A, din)).astype(np.float32)
    done = rng.random((T, E)) < 0.15
    done = np.repeat(done[:, :, None], A, 2)
    h0 = (rng.standard_normal((E, A, 128)) * 0.5).astype(np.float32)
    idx = rng.permutation(E)[:Em].astype(np.int32)
    return obs, done, h0, idx


def _gather(x, idx, A, shared):
    """external (T,E,A|1,...) -> (T, Em*A, ...) time-major minibatch rows"""
    g = x[:, idx]
    if shared:
        g = np.repeat(g, A, 2)
    return g.reshape(g.shape[0], g.shape[1] * g.shape[2], *g.shape[3:])


# the last case is BASELINE config 4's shape at full sequence length: rec_mappo SMAX 3s5z, 8 agents, actor input
# 155, 13 actions, seq_len = 128 (f32 error growth over 128 GRU steps against the float64 oracle)
@pytest.mark.parametrize("shared,T,E,A,Em,din,nA", [(False, 9, 12, 4, 8, 37, 6), (True, 9, 12, 4, 8, 37, 6),
                                                    (False, 128, 8, 8, 4, 155, 13), (True, 128, 8, 8, 4, 188, 1)])
def test_recurrent_forward_matches_oracle(dev, shared, T, E, A, Em, din, nA, rec_mode):
    from mava_amd.networks import DiscreteActionHead, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, RecWorkspace, t32_to_rows

    rng = np.random.default_rng(11)
    obs, done, h0, idx = _seq_case(rng, T, E, A, Em, din, nA, shared)
    net = RecurrentActor(MLPTorso([128]), MLPTorso([128]), DiscreteActionHead(nA), din)
    net.ctx = _CTX
    flat = ro.init_rec(rng, din, nA, 1.0).astype(np.float32)
    flat[net.off["bi"][0] : net.off["bi"][0] + 384] = rng.standard_normal(384) * 0.1
    flat[net.off["bhn"][0] : net.off["bhn"][0] + 128] = rng.standard_normal(128) * 0.1
    assert flat.size == net.num_params == ro.rec_param_count(din, nA)
    Rm = Em * A
    ws = RecWorkspace(T * Rm, nA, dev, din_max=din)
    flat_d, obs_d, done_d, h0_d, idx_d = _t(flat, dev), _t(obs, dev), _t(done, dev).view(torch.uint8), _t(h0, dev), _t(idx, dev)
    y = net.forward_sequence(flat_d, ws, obs_d, A if shared else 1, done_d, h0_d, False, idx_d, T, Rm, E, A, training=True)
    torch.cuda.synchronize()
    want, hs_in, h_last = ro.rec_forward(flat, din, nA, _gather(obs, idx, A, shared), _gather(done, idx, A, False),
                                         _gather(h0[None], idx, A, False)[0])
    got = t32_to_rows(y, nA, T * Rm).cpu().numpy().reshape(T, Rm, nA)
    assert_close(got, want, 1e-5, "recurrent logits")
    hs = t32_to_rows(ws.hs, 128, T * Rm).cpu().numpy().reshape(T, Rm, 128)
    assert_close(hs[-1], h_last, 1e-5, "final hidden state")
    assert_close(hs[:-1], hs_in[1:], 1e-5, "hidden states of every step")


# (128, 8, 8, 4, 155, 13): BPTT over the FULL sequence length of BASELINE config 4 (seq_len = 128, 8 agents, input 155,
# 13 actions) - actor and critic gradients at the north-star 1e-4 against float64 autograd
@pytest.mark.parametrize("T,E,A,Em,din,nA", [(6, 8, 4, 8, 20, 5), (12, 16, 8, 4, 40, 13), (128, 8, 8, 4, 155, 13)])
@pytest.mark.parametrize("fused_out", [False, True], ids=["layerwise-out", "fused-out"])
def test_recurrent_gradients_match_autograd(dev, T, E, A, Em, din, nA, rec_mode, fused_out):
    """fused_out: post_torso -> head -> loss -> backward in one launch (mava_rec_out_f32) instead of eight."""
    if fused_out and rec_mode == 0:
        pytest.skip("the fused output path belongs to the f16x2 arithmetic")
    from mava_amd import ops
    from mava_amd._lib import check, lib, ptr, stream_ptr
    from mava_amd.networks import DiscreteActionHead, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, RecurrentValueNet, RecWorkspace, t32_to_rows

    rng = np.random.default_rng(T + nA)
    obs, done, h0, idx = _seq_case(rng, T, E, A, Em, din, nA, False)
    Rm, rows = Em * A, T * Em * A
    actor = RecurrentActor(MLPTorso([128]), MLPTorso([128]), DiscreteActionHead(nA), din)
    critic = RecurrentValueNet(MLPTorso([128]), MLPTorso([128]), False, din)
    actor.ctx = critic.ctx = _CTX
    fa = ro.init_rec(rng, din, nA, 1.0).astype(np.float32)
    fc = ro.init_rec(rng, din, 1, 1.0).astype(np.float32)
    mask = rng.random((T, E, A, nA)) > 0.25
    action = rng.integers(0, nA, (T, E, A)).astype(np.int32)
    np.put_along_axis(mask, action[..., None].astype(np.int64), True, -1)
    go = lambda x: _gather(x, idx, A, False)
    y, _, _ = ro.rec_forward(fa, din, nA, go(obs), go(done), go(h0[None])[0])
    z = np.where(go(mask), y, ro.F32_MIN)
    lsm = z - (z.max(-1, keepdims=True) + np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)))
    lp_now = np.take_along_axis(lsm, go(action)[..., None].astype(np.int64), -1)[..., 0]
    old_lp = np.zeros((T, E, A), np.float32)
    old_lp[:, idx] = (lp_now + rng.standard_normal(lp_now.shape) * 0.25).reshape(T, Em, A)
    adv = (rng.standard_normal((T, E, A)) * 2 + 0.3).astype(np.float32)
    v_now, _, _ = ro.rec_forward(fc, din, 1, go(obs), go(done), go(h0[None])[0])
    old_v = np.zeros((T, E, A), np.float32)
    tgt = np.zeros((T, E, A), np.float32)
    old_v[:, idx] = (v_now[..., 0] + rng.standard_normal(lp_now.shape) * 0.2).reshape(T, Em, A)
    tgt[:, idx] = (v_now[..., 0] + rng.standard_normal(lp_now.shape)).reshape(T, Em, A)

    ws = RecWorkspace(rows, max(nA, 1), dev, din_max=din)
    gscale = float(2 ** int(np.ceil(np.log2(rows))))  # the backward chain runs in these units (RecLearner.grad_scale)
    d = lambda a, dt=None: _t(a, dev, dt)
    idx_d, done_d, h0_d, obs_d = d(idx), d(done).view(torch.uint8), d(h0), d(obs)
    slabs = torch.zeros((4, 128 * 384 + 384 + 8), device=dev)
    # ---- actor
    fa_d = d(fa)
    actor.forward_sequence(fa_d, ws, obs_d, 1, done_d, h0_d, False, idx_d, T, Rm, E, A, training=True, stop_after_scan=fused_out)
    flat_rows = (torch.arange(T, device=dev)[:, None] * E + idx_d[None, :].long()).reshape(-1).to(torch.int32)
    adv_d, mask_d, act_d, olp_d = d(adv), d(mask).view(torch.uint8), d(action), d(old_lp)  # kept alive (see above)
    stats = ops.adv_stats(adv_d.view(-1), flat_rows, 0, T * Em, A)
    ga = torch.zeros(actor.num_params, device=dev)
    lsum_d = torch.zeros(2, device=dev)
    if fused_out:
        assert actor.fused_output(fa_d, ws, idx_d, T, Rm, E, A, 1, True, mask_d, act_d, olp_d, adv_d, stats, 0.2, 0.01, slabs, ga,
                                  lsum_d, False, gscale)
    else:
        check(lib().mava_seq_actor_loss_f32(T, Rm, E, A, nA, ptr(idx_d), ptr(ws.y), ptr(mask_d), ptr(act_d),
                                            ptr(olp_d), ptr(adv_d), ptr(stats), stats.shape[0], 0.2, 0.01, gscale, ptr(ws.dy),
                                            ptr(ws.loss_partials), ws.loss_partials.shape[0], stream_ptr()), "actor loss")
    actor.backward_sequence(fa_d, ws, obs_d, 1, done_d, idx_d, T, Rm, E, A, slabs, ga, accumulate=False, grad_scale=gscale,
                            from_scan=fused_out)
    torch.cuda.synchronize()
    tot, la, ent, g = ro.rec_actor_loss_grad(fa, din, nA, go(obs), go(done), go(h0[None])[0], go(mask), go(action), go(old_lp),
                                             go(adv), 0.2, 0.01)
    lsum = lsum_d.cpu().numpy() if fused_out else ws.loss_partials.sum(0).cpu().numpy()
    assert_close(lsum, np.array([la, ent]), 1e-5, "actor loss / entropy", scale=1.0)
    assert_close(ga.cpu().numpy(), g, 1e-4, "recurrent actor gradient")  # north_star: PPO gradients 1e-4
    for name in ("Wpre", "Wi", "Wh", "bhn", "Wpost", "bpost", "Whead", "bhead"):
        o, s = actor.off[name]
        n = int(np.prod(s))
        assert_close(ga.cpu().numpy()[o : o + n], g[o : o + n], 1e-4, f"actor grad {name}")
    # ---- critic
    fc_d = d(fc)
    critic.forward_sequence(fc_d, ws, obs_d, 1, done_d, h0_d, False, idx_d, T, Rm, E, A, training=True, stop_after_scan=fused_out)
    ov_d, tg_d = d(old_v), d(tgt)
    gc = torch.zeros(critic.num_params, device=dev)
    vsum_d = torch.zeros(1, device=dev)
    if fused_out:
        assert critic.fused_output(fc_d, ws, idx_d, T, Rm, E, A, 1, False, None, None, ov_d, tg_d, None, 0.2, 0.5, slabs, gc, vsum_d,
                                   False, gscale)
    else:
        check(lib().mava_seq_critic_loss_f32(T, Rm, E, A, 1, ptr(idx_d), ptr(ws.y), ptr(ov_d), ptr(tg_d), 0.2, 0.5, gscale, ptr(ws.dy),
                                             ptr(ws.loss_partials), ws.loss_partials.shape[0], stream_ptr()), "critic loss")
    critic.backward_sequence(fc_d, ws, obs_d, 1, done_d, idx_d, T, Rm, E, A, slabs, gc, accumulate=False, grad_scale=gscale,
                             from_scan=fused_out)
    torch.cuda.synchronize()
    tot, vl, g = ro.rec_critic_loss_grad(fc, din, go(obs), go(done), go(h0[None])[0], go(old_v), go(tgt), 0.2, 0.5)
    vsum = vsum_d.cpu().numpy() if fused_out else ws.loss_partials.sum(0).cpu().numpy()[:1]
    assert_close(vsum, np.array([vl]), 1e-5, "value loss", scale=1.0)
    assert_close(gc.cpu().numpy(), g, 1e-4, "recurrent critic gradient")


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,U,E", [("rec_mappo", 1, 16), ("rec_ippo", 2, 16), ("rec_mappo", 1, 64)])
def test_rec_learner_update_matches_oracle(dev, system, U, E, matmul):
    """End to end: the HIP recurrent learner against the whole-update oracle on identical inputs.  E = 64 switches the
    centralised critic to one sequence per ENV (the agents share the global state; the oracle, like the reference,
    evaluates all E*A tiled rows)."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo, rec_mappo
    from oracle import ppo_oracle as po
    from oracle.rec_loop import OracleRecLearner

    A, O, nA, T, K, M = 4, 10, 5, 6, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}"])
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": nA}
    cfg.env.kwargs.time_limit = 4  # forces resets inside the rollout (hidden-state resets, GAE masking)
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    cfg.system.matmul_mode = matmul
    central = system == "rec_mappo"
    mod = rec_mappo if central else rec_ippo
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    assert L.critic_agg == (central and E == 64)
    assert state.hstates.policy_hidden_state.shape == (1, U, E, A, 128) and state.dones.shape == (1, U, E, A)
    assert state.hstates.critic_hidden_state.shape == (1, U, E, A, 128)
    k = state.params.actor_params["params"]["ScannedRNN_0"]["GRUCell_0"]["hz"]["kernel"]
    assert k.shape == (1, U, 128, 128)

    rng = np.random.default_rng(1)
    Oc = A * O if central else A + O
    fa = ro.init_rec(rng, A + O, nA, 1.0).astype(np.float32)
    fc = ro.init_rec(rng, Oc, 1, 1.0).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleRecLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3,
                           time_limit=4)
    ora.set_params(fa, fc)
    for n in range(2):
        perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[u]
            assert np.array_equal(rep.action.cpu().numpy(), tr["action"]), "sampled actions differ"
            assert np.array_equal(rep.done_in.cpu().numpy().astype(bool), tr["done_in"])
            assert tr["done_in"].any(), "the test must exercise hidden-state resets"
            assert_close(rep.value.cpu().numpy(), tr["value"], 1e-5, "values")
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], 1e-5, "log_probs")
            assert_close(rep.adv.cpu().numpy(), tr["adv"], 1e-5, "advantages")
            assert_close(rep.tgt.cpu().numpy(), tr["tgt"], 1e-5, "targets")
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":
            assert_close(L.p[: L.Pa].cpu().numpy() - fa, ora.pa - fa, 2e-3, "actor update")
            assert_close(L.p[L.Pa :].cpu().numpy() - fc, ora.pc - fc, 2e-3, "critic update")
            assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, 1e-5, "actor params")
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, 1e-5, "critic params")
        else:
            check_and_sync_f16x2_state(L, ora)  # f16x2: see tests/conftest.py
    out = learn(L.learner_state())
    torch.cuda.synchronize()
    assert out.train_metrics["total_loss"].shape == (1, 2, U, K, M) and torch.isfinite(out.train_metrics["total_loss"]).all()


def test_rec_dense_any_t32_width(dev, rec_mode):
    """T32 inputs of any width (155 = the config-4 observation; the general network path produces arbitrary layer sizes):
    clamped operand addresses in the exact-f32 kernel, staged zeros in the f16x2 kernel."""
    from mava_amd._lib import check, lib, ptr, stream_ptr

    rng = np.random.default_rng(155)
    for K, N in ((155, 128), (100, 96), (37, 13)):
        rows = 64
        x = rng.standard_normal((rows, K)).astype(np.float32)
        w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        xt, wt = _t(_to_t32(x), dev), _t(w, dev)
        y = torch.zeros(rows * N, device=dev)
        check(lib().mava_rec_dense_f32(_cp(), ptr(xt), 0, None, 0, 0, 0, 1, K, 0, ptr(wt), N, None, None, ptr(y), 0, K, N, rows, 0,
                                       stream_ptr()), "dense")
        torch.cuda.synchronize()
        assert_close(_from_t32(y.cpu().numpy(), rows, N), x.astype(np.float64) @ w.astype(np.float64), 1e-5, f"dense K={K} N={N}")


def test_recurrent_apply_and_eval_act_fn(dev):
    """actor_network.apply(params, hstate, (obs[None], done[None])) -> (hstate, dist) and the evaluator's recurrent
    act function (mava/evaluator.py:189-207) against the oracle, stepping a hidden state over several calls; the row
    count (E*A = 20) is not a multiple of the kernels' 32-row tile."""
    from types import SimpleNamespace

    from mava_amd.evaluator import make_rec_eval_act_fn
    from mava_amd.networks import DiscreteActionHead, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, RecurrentValueNet
    from mava_amd.types import Observation, ObservationGlobalState, TimeStep

    rng = np.random.default_rng(5)
    E, A, din, nA, steps = 5, 4, 23, 7, 4
    actor = RecurrentActor(MLPTorso([128]), MLPTorso([128]), DiscreteActionHead(nA), din)
    actor.ctx = _CTX
    flat = ro.init_rec(rng, din, nA, 1.0).astype(np.float32)
    tree = actor.tree(_t(flat, dev), (1, 1))  # Flax-shaped, with (device, update_batch) leading dims
    assert torch.equal(actor.flat_from_tree(tree).cpu(), torch.from_numpy(flat))
    config = SimpleNamespace(arch=SimpleNamespace(evaluation_greedy=True))
    act = make_rec_eval_act_fn(actor.apply, config)
    h_or = np.zeros((E * A, 128))
    state = {"hidden_state": torch.zeros((E, A, 128), device=dev)}
    for k in range(steps):
        av = rng.standard_normal((E, A, din)).astype(np.float32)
        mask = rng.random((E, A, nA)) > 0.3
        mask[..., 0] = True
        last = rng.random(E) < 0.4
        ts = TimeStep(_t(np.where(last, 2, 1).astype(np.int8), dev), None, None,
                      Observation(_t(av, dev), _t(mask, dev), torch.zeros((E, A), dtype=torch.int32, device=dev)), {})
        action, state = act(tree, ts, None, state)
        done = np.repeat(last[:, None], A, 1).reshape(1, E * A)
        want, _, h_or = ro.rec_forward(flat, din, nA, av.reshape(1, E * A, din), done, h_or)
        logits = np.where(mask.reshape(E * A, nA), want[0], np.finfo(np.float32).min)
        assert action.shape == (E, A)
        # greedy action = argmax of the masked logits (ties are measure-zero for random inputs)
        assert np.array_equal(action.cpu().numpy().reshape(-1), logits.argmax(-1))
        assert_close(state["hidden_state"].cpu().numpy().reshape(E * A, 128), h_or, 1e-5, "eval hidden state")

    # critic: centralised input, value shape (T, E, A), error without a global state
    S = 31
    critic = RecurrentValueNet(MLPTorso([128]), MLPTorso([128]), True, S)
    critic.ctx = _CTX
    fc = ro.init_rec(rng, S, 1, 1.0).astype(np.float32)
    gs = rng.standard_normal((2, E, A, S)).astype(np.float32)
    dn = rng.random((2, E, A)) < 0.3
    h0 = rng.standard_normal((E, A, 128)).astype(np.float32)
    obs = ObservationGlobalState(torch.zeros((2, E, A, din), device=dev), None, _t(gs, dev), None)
    h1, v = critic.apply(_t(fc, dev), _t(h0, dev), (obs, _t(dn, dev)))
    want, _, h_last = ro.rec_forward(fc, S, 1, gs.reshape(2, E * A, S), dn.reshape(2, E * A), h0.reshape(E * A, 128))
    assert v.shape == (2, E, A)
    assert_close(v.cpu().numpy().reshape(2, E * A), want[..., 0], 1e-5, "recurrent value")
    assert_close(h1.cpu().numpy().reshape(E * A, 128), h_last, 1e-5, "critic hidden state")
    with pytest.raises(ValueError):
        critic.apply(_t(fc, dev), _t(h0, dev), (Observation(torch.zeros((2, E, A, din), device=dev), None, None), _t(dn, dev)))


def test_rec_learner_adopts_foreign_params(dev):
    """learn(state) must follow its argument: parameter trees that do not alias the learner's buffers (e.g. restored
    from a checkpoint, rec_mappo.py:558-566) are copied in; the learner's own trees are a no-op."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_mappo
    from mava_amd.types import Params

    cfg = compose("default_rec_mappo", ["env/scenario=tiny-2ag", "arch.num_envs=32", "system.rollout_length=4",
                                       "system.update_batch_size=1", "system.num_minibatches=2", "system.ppo_epochs=1"])
    cfg.system.num_updates_per_eval = 1
    env, _ = envs.make(cfg, add_global_state=True)
    learn, actor_network, state = rec_mappo.learner_setup(env, (1, 2, 3), cfg)
    L = learn.learner
    before = L.p.clone()
    L.adopt(state)
    assert torch.equal(L.p, before)
    g = torch.Generator().manual_seed(0)
    fa = torch.randn(L.Pa, generator=g)
    fc = torch.randn(L.P - L.Pa, generator=g)
    foreign = Params(L.actor_network.tree(fa), L.critic_network.tree(fc))  # host tensors, no replica dims
    L.adopt(state._replace(params=foreign))
    assert torch.equal(L.p[: L.Pa].cpu(), fa) and torch.equal(L.p[L.Pa :].cpu(), fc)
    assert torch.equal(state.params.actor_params["params"]["pre_torso"]["Dense_0"]["kernel"][0, 0].cpu(),
                       fa[: L.Oa * 128].view(L.Oa, 128))  # the state's trees are views of the adopted buffers


@pytest.mark.parametrize("E,A,din,S,nA,share", [(16, 4, 37, 23, 6, False), (32, 8, 155, 188, 13, True), (8, 4, 20, 20, 20, False)])
def test_fused_rec_step_matches_oracle(dev, E, A, din, S, nA, share):
    """mava_rec_step_f32 (one launch: both networks, GRU cell with reset, head, sampling) against the oracle's recurrent
    forward for one step, and its sampled actions against mava_seq_sample_f32 on the oracle-checked logits path."""
    from mava_amd._lib import check, lib, ptr, stream_ptr
    from mava_amd.networks import DiscreteActionHead, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, rows_to_t32, t32_to_rows

    rng = np.random.default_rng(E + din)
    R = E * A
    Rc = E if share else R                      # critic rows: one per env when the agents share its input
    fa = ro.init_rec(rng, din, nA, 1.0).astype(np.float32)
    fc = ro.init_rec(rng, S, 1, 1.0).astype(np.float32)
    for f, d in ((fa, din), (fc, S)):           # non-zero biases everywhere
        f[d * 128 : d * 128 + 128] = rng.standard_normal(128) * 0.1
    x = rng.standard_normal((R, din)).astype(np.float32)
    xc = rng.standard_normal((Rc, S)).astype(np.float32)
    mask = rng.random((R, nA)) > 0.3
    mask[:, 0] = True
    done_a = rng.random(R) < 0.3
    done_c = rng.random(Rc) < 0.3
    ha = rng.standard_normal((R, 128)).astype(np.float32)
    hc = rng.standard_normal((Rc, 128)).astype(np.float32)
    pad = lambda a, n: np.concatenate([a, np.zeros((n - a.shape[0],) + a.shape[1:], a.dtype)])  # rows up to a multiple of 32
    Rp, Rcp = -(-R // 32) * 32, -(-Rc // 32) * 32
    xa_d, xc_d = _t(pad(x, Rp), dev), _t(pad(xc, Rcp), dev)
    mask_d = _t(pad(mask, Rp), dev).view(torch.uint8)
    da_d, dc_d = _t(pad(done_a, Rp), dev).view(torch.uint8), _t(pad(done_c, Rcp), dev).view(torch.uint8)
    ha_in, hc_in = rows_to_t32(_t(pad(ha, Rp), dev)), rows_to_t32(_t(pad(hc, Rcp), dev))
    ha_out, hc_out = torch.zeros_like(ha_in), torch.zeros_like(hc_in)
    action = torch.zeros(Rp, dtype=torch.int32, device=dev)
    logp = torch.zeros(Rp, device=dev)
    vb = A if share else 1
    value = torch.zeros(Rcp * vb, device=dev)
    fa_d, fc_d = _t(fa, dev), _t(fc, dev)
    check(lib().mava_rec_step_f32(ptr(fa_d), din, nA, ptr(xa_d), ptr(mask_d), ptr(da_d), ptr(ha_in), ptr(ha_out), Rp, 77, 5, 11, 0,
                                  ptr(action), ptr(logp), ptr(fc_d), S, ptr(xc_d), 1, ptr(dc_d), 1, ptr(hc_in), ptr(hc_out), Rcp, vb,
                                  ptr(value), stream_ptr()), "mava_rec_step_f32")
    torch.cuda.synchronize()
    ya, _, ha_new = ro.rec_forward(fa, din, nA, x[None], done_a[None], ha)
    yc, _, hc_new = ro.rec_forward(fc, S, 1, xc[None], done_c[None], hc)
    assert_close(t32_to_rows(ha_out, 128, Rp).cpu().numpy()[:R], ha_new, 1e-5, "actor hidden state")
    assert_close(t32_to_rows(hc_out, 128, Rcp).cpu().numpy()[:Rc], hc_new, 1e-5, "critic hidden state")
    assert_close(value.cpu().numpy().reshape(Rcp, vb)[:Rc], np.repeat(yc[0], vb, 1), 1e-5, "value")
    # sampling: the same Philox stream as mava_seq_sample_f32, fed with the oracle's logits
    logits_t32 = rows_to_t32(_t(pad(ya[0].astype(np.float32), Rp), dev))
    a2 = torch.zeros(Rp, dtype=torch.int32, device=dev)
    lp2 = torch.zeros(Rp, device=dev)
    check(lib().mava_seq_sample_f32(Rp, nA, ptr(logits_t32), ptr(mask_d), 77, 5, 11, 0, ptr(a2), ptr(lp2), stream_ptr()), "sample")
    torch.cuda.synchronize()
    a, a2 = action.cpu().numpy()[:R], a2.cpu().numpy()[:R]
    assert (a != a2).mean() < 0.02  # f32 logits vs f64-derived logits: near-ties of the Gumbel scores only
    assert mask[np.arange(R), a].all(), "sampled an illegal action"
    z = np.where(mask, ya[0], np.finfo(np.float32).min)
    lsm = z - np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)) - z.max(-1, keepdims=True)
    assert_close(logp.cpu().numpy()[:R], lsm[np.arange(R), a], 1e-5, "log_prob")


RCNN = dict(shape=(2, 7, 1), channels=[8, 8], kernels=[3, 3], strides=[1, 1])  # network/rcnn.yaml-style CNN pre-torso: 14 = A + O inputs


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,pre,post,act,ln", [("rec_mappo", [64, 96], [64], "tanh", True), ("rec_ippo", [128], [96, 32], "relu", False),
                                                    ("rec_ippo", RCNN, [64], "tanh", False)], ids=["mappo-mlp-tanh-ln", "ippo-mlp-relu", "ippo-rcnn"])
def test_rec_learner_general_torsos(dev, system, pre, post, act, ln, matmul):
    """Recurrent systems with pre / post torsos other than network/rnn.yaml's [128] relu (mava/networks.py:39-58 inside
    RecurrentActor / RecurrentValueNet): the general layer kernels around the same GRU scans, whole updates against the
    oracle with the same torsos (oracle/rec_oracle.py rec_spec)."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo, rec_mappo
    from oracle.rec_loop import OracleRecLearner

    E, U, A, O, nA, T, K, M = 16, 1, 4, 10, 5, 6, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}"])
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": nA}
    cfg.env.kwargs.time_limit = 4
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    cfg.system.matmul_mode = matmul
    cnn = isinstance(pre, dict)
    for nc in (cfg.network.actor_network, cfg.network.critic_network):
        nc.post_torso.layer_sizes = post
        if cnn:  # configs/network/rcnn.yaml: CNNTorso pre-torso (mava/networks.py:61-85), MLPTorso post-torso
            nc.pre_torso._target_ = "mava.networks.CNNTorso"
            del nc.pre_torso["layer_sizes"]
            nc.pre_torso.channel_sizes, nc.pre_torso.kernel_sizes, nc.pre_torso.strides = pre["channels"], pre["kernels"], pre["strides"]
        else:
            nc.pre_torso.layer_sizes = pre
        for t in (nc.pre_torso, nc.post_torso):
            t.activation, t.use_layer_norm = act, ln
    if cnn:
        cfg.env.synthetic["obs_shape"] = list(pre["shape"])
    central = system == "rec_mappo"
    mod = rec_mappo if central else rec_ippo
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    Oc = A * O if central else A + O
    if cnn:
        spec_a, spec_c = ro.rec_spec(A + O, None, post, act, ln, pre_cnn=pre), ro.rec_spec(Oc, None, post, act, ln, pre_cnn=pre)
    else:
        spec_a, spec_c = ro.rec_spec(A + O, pre, post, act, ln), ro.rec_spec(Oc, pre, post, act, ln)
    assert L.generic_nets and not L.fused_out
    assert L.Pa == ro.rec_param_count(spec_a, nA) and L.Pc == ro.rec_param_count(spec_c, 1)
    tree = state.params.actor_params["params"]
    if cnn:
        assert int(np.prod(tree["pre_torso"]["Conv_0"]["kernel"].shape[2:])) == 3 * 3 * 1 * 8
        assert tree["ScannedRNN_0"]["GRUCell_0"]["ir"]["kernel"].shape == (1, U, 2 * 7 * 8, 128)
    else:
        assert tree["pre_torso"]["Dense_0"]["kernel"].shape == (1, U, A + O, pre[0])
        assert tree["ScannedRNN_0"]["GRUCell_0"]["ir"]["kernel"].shape == (1, U, pre[-1], 128)
    assert ("LayerNorm_0" in tree["post_torso"]) == ln
    assert torch.equal(actor_network.flat_from_tree(state.params.actor_params), L.p[: L.Pa])

    # (relu has a kink: with parameter seed 2 one post-torso pre-activation of the second update is 1.2e-7 in f32 / f64 and
    # -3e-8 in f16x2, which moves that unit's whole gradient column; the seed below keeps every pre-activation clear of 0)
    rng = np.random.default_rng(3 if act == "relu" else 2)
    fa = (rng.standard_normal(L.Pa) * 0.1).astype(np.float32)
    fc = (rng.standard_normal(L.Pc) * 0.1).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleRecLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3,
                           time_limit=4, actor_net=spec_a, critic_net=spec_c)
    ora.set_params(fa, fc)
    ftol = 1e-5 if matmul == "f32" else 5e-5
    for n in range(2):
        perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        rep, tr = L.reps[0], ora.last_traj[0]
        assert np.array_equal(rep.action.cpu().numpy(), tr["action"]), "sampled actions differ"
        assert tr["done_in"].any()
        assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
        assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], ftol, "log_probs")
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


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
@pytest.mark.parametrize("system,hidden,U", [("rec_mappo", 64, 1), ("rec_ippo", 96, 2), ("rec_ippo", 160, 1)])
def test_rec_learner_hidden_state_dim(dev, system, hidden, U, matmul):
    """network.hidden_state_dim != 128 (mava/networks.py:222-266: ScannedRNN(hidden_state_dim) inside RecurrentActor /
    RecurrentValueNet; rec_mappo.py:623-629 initialises the carry with it): the GRU cell runs one time step at a time on the
    general layer kernels (mava_t32_gru_* around T32 dense launches).  Whole updates against the oracle with the same width
    (rec_oracle.rec_spec(hidden=...)), the parameter tree's shapes, learn() and the evaluator's carried state."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo, rec_mappo
    from oracle.rec_loop import OracleRecLearner

    E, A, O, nA, T, K, M = 16, 4, 10, 5, 6, 2, 2
    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}"])
    cfg.env.scenario.task_config.num_agents = A
    cfg.env.synthetic = {"obs_dim": O, "num_actions": nA}
    cfg.env.kwargs.time_limit = 4
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    cfg.system.matmul_mode = matmul
    cfg.network.hidden_state_dim = hidden
    central = system == "rec_mappo"
    mod = rec_mappo if central else rec_ippo
    env, eval_env = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    Oc = A * O if central else A + O
    spec_a = ro.rec_spec(A + O, [128], [128], "relu", False, hidden=hidden)
    spec_c = ro.rec_spec(Oc, [128], [128], "relu", False, hidden=hidden)
    assert L.generic_nets and L.Pa == ro.rec_param_count(spec_a, nA) and L.Pc == ro.rec_param_count(spec_c, 1)
    cell = state.params.actor_params["params"]["ScannedRNN_0"]["GRUCell_0"]
    assert cell["ir"]["kernel"].shape == (1, U, 128, hidden) and cell["hn"]["kernel"].shape == (1, U, hidden, hidden)
    assert cell["hn"]["bias"].shape == (1, U, hidden)
    assert state.hstates.policy_hidden_state.shape[-3:] == (E, A, hidden)
    assert torch.equal(actor_network.flat_from_tree(state.params.actor_params), L.p[: L.Pa])

    rng = np.random.default_rng(3)
    fa = (rng.standard_normal(L.Pa) * 0.1).astype(np.float32)
    fc = (rng.standard_normal(L.Pc) * 0.1).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleRecLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3,
                           time_limit=4, actor_net=spec_a, critic_net=spec_c)
    ora.set_params(fa, fc)
    ftol = 1e-5 if matmul == "f32" else 5e-5
    for n in range(2):
        perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
        L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[u]
            assert np.array_equal(rep.action.cpu().numpy(), tr["action"]), "sampled actions differ"
            assert tr["done_in"].any()
            assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], ftol, "log_probs")
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
    assert out.learner_state.hstates.policy_hidden_state.shape[-3:] == (E, A, hidden)
    from mava_amd.evaluator import get_eval_fn, make_rec_eval_act_fn

    ev = get_eval_fn(eval_env, make_rec_eval_act_fn(actor_network.apply, cfg), cfg, absolute_metric=False)
    m = ev(out.learner_state.params.actor_params, 0, {"hidden_state": torch.zeros((eval_env.num_envs, A, hidden), device=dev)})
    assert m["episode_return"].shape[0] >= cfg.arch.num_eval_episodes
