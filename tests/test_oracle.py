"""CPU: the oracle against (a) published known answers where they exist (Philox), (b) analytic
properties of the reference algorithm, (c) an independent torch-autograd restatement, and (d) the
committed golden fixtures (regression pin; parity with Mava itself is unpinned, see oracle headers)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import philox, ppo_oracle as po, torch_ref
from oracle.synth_env import SynthRware
from tests.conftest import assert_close

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for c, k, want in kat:
        got = tuple(int(x) for x in philox.philox4x32_10(*c, *k))
        assert got == want


@settings(max_examples=25, deadline=None)
@given(T=st.integers(1, 40), N=st.integers(1, 9), seed=st.integers(0, 10_000))
def test_gae_properties(T, N, seed):
    rng = np.random.default_rng(seed)
    r, v, lv = rng.standard_normal((T, N)), rng.standard_normal((T, N)), rng.standard_normal(N)
    z = np.zeros((T, N), bool)
    adv, tgt = po.gae(r, v, z, lv, 1.0, 1.0)  # lambda=gamma=1, no dones: telescoping sum
    assert np.allclose(adv, np.cumsum(r[::-1], 0)[::-1] + lv - v)
    adv, tgt = po.gae(r, v, ~z, lv, 0.99, 0.95)  # all done: adv = r - V, target = r
    assert np.allclose(adv, r - v) and np.allclose(tgt, r)
    # recurrent masking == feed-forward masking with the done array shifted by one step
    d = rng.random((T, N)) < 0.3
    ld = rng.random(N) < 0.3
    a_rec, _ = po.gae(r, v, d, lv, 0.99, 0.95, last_done=ld)
    a_ff, _ = po.gae(r, v, np.concatenate([d[1:], ld[None]], 0), lv, 0.99, 0.95)
    assert np.allclose(a_rec, a_ff)
    # chunked affine composition == sequential scan (the GPU kernel's algebra)
    c = 0.99 * 0.95 * (1 - d)
    delta = r + 0.99 * np.concatenate([v[1:], lv[None]], 0) * (1 - d) - v
    L = max(1, T // 3)
    a_in = np.zeros(N)
    out = np.zeros((T, N))
    for hi in range(T, 0, -L):
        lo = max(0, hi - L)
        S, P = np.zeros((hi - lo, N)), np.zeros((hi - lo, N))
        a, p = np.zeros(N), np.ones(N)
        for t in range(hi - 1, lo - 1, -1):
            a = delta[t] + c[t] * a
            p = c[t] * p
            S[t - lo], P[t - lo] = a, p
        out[lo:hi] = S + P * a_in
        a_in = out[lo]
    assert np.allclose(out, po.gae(r, v, d, lv, 0.99, 0.95)[0])


def test_gae_f32_vs_f64_tolerance_form():
    """a pure elementwise rtol of 1e-5 is unattainable even for sequential f32; the stated
    form |a-b| <= rtol*|b| + rtol*rms(b) holds."""
    rng = np.random.default_rng(0)
    r, v = rng.standard_normal((128, 4096)).astype(np.float32), rng.standard_normal((128, 4096)).astype(np.float32)
    d = rng.random((128, 4096)) < 1 / 500
    lv = rng.standard_normal(4096).astype(np.float32)
    a64, _ = po.gae(r, v, d, lv, 0.99, 0.95)
    a32, _ = po.gae(r, v, d, lv, 0.99, 0.95, dtype=np.float32)
    assert_close(a32, a64, 1e-5, "f32 vs f64")


@pytest.mark.parametrize("seed", [0, 1])
def test_manual_gradients_match_autograd(seed):
    rng = np.random.default_rng(seed)
    R, din, nA = 200, 17, 6
    p = po.mlp_flatten(po.init_mlp(rng, din, nA, 1.0)) + rng.standard_normal(po.mlp_param_count(din, nA)) * 0.05
    obs = rng.standard_normal((R, din))
    mask = rng.random((R, nA)) > 0.3
    act = rng.integers(0, nA, R)
    mask[np.arange(R), act] = True
    lsm = po.log_softmax(po.masked_logits(po.mlp_forward(po.mlp_unflatten(p, din, nA), obs), mask))
    olp = lsm[np.arange(R), act] + rng.standard_normal(R) * 0.25
    adv = rng.standard_normal(R)
    a = po.actor_loss_and_grad(p, din, nA, obs, mask, act, olp, adv, 0.2, 0.01)
    b = torch_ref.actor_grad(p, din, nA, obs, mask, act, olp, adv, 0.2, 0.01)
    assert abs(a[0] - b[0]) < 1e-12 and abs(a[1] - b[1]) < 1e-12 and abs(a[2] - b[2]) < 1e-12
    assert np.abs(a[3] - b[3]).max() < 1e-12
    pc = po.mlp_flatten(po.init_mlp(rng, din, 1, 1.0))
    v = po.mlp_forward(po.mlp_unflatten(pc, din, 1), obs)[:, 0]
    ov, tg = v + rng.standard_normal(R) * 0.2, v + rng.standard_normal(R)
    a = po.critic_loss_and_grad(pc, din, obs, ov, tg, 0.2, 0.5)
    b = torch_ref.critic_grad(pc, din, obs, ov, tg, 0.2, 0.5)
    assert abs(a[0] - b[0]) < 1e-12 and np.abs(a[2] - b[2]).max() < 1e-12


def test_actor_gradient_finite_difference():
    rng = np.random.default_rng(5)
    R, din, nA = 40, 6, 4
    p = po.mlp_flatten(po.init_mlp(rng, din, nA, 1.0))
    obs, mask = rng.standard_normal((R, din)), np.ones((R, nA), bool)
    act = rng.integers(0, nA, R)
    olp, adv = -1.3 + rng.standard_normal(R) * 0.1, rng.standard_normal(R)
    f = lambda q: po.actor_loss_and_grad(q, din, nA, obs, mask, act, olp, adv, 0.2, 0.01)[0]
    g = po.actor_loss_and_grad(p, din, nA, obs, mask, act, olp, adv, 0.2, 0.01)[3]
    for i in rng.integers(0, p.size, 12):
        e = np.zeros_like(p)
        e[i] = 1e-6
        fd = (f(p + e) - f(p - e)) / 2e-6
        assert abs(fd - g[i]) < 1e-6 + 1e-4 * abs(g[i]), (i, fd, g[i])


def test_clip_adam_semantics():
    rng = np.random.default_rng(1)
    p, g = rng.standard_normal(50), rng.standard_normal(50) * 10
    pn, m, v, c = po.clip_adam(p, g, np.zeros(50), np.zeros(50), 0, 1e-3, 0.5)
    gc = g / np.sqrt((g * g).sum()) * 0.5  # clipped to norm 0.5
    assert np.allclose(m, 0.1 * gc) and np.allclose(v, 0.001 * gc * gc) and c == 1
    # first step of Adam moves by lr * g/(|g| + eps*...) ~ lr * sign(g)
    assert np.allclose(pn, p - 1e-3 * gc / (np.abs(gc) + 1e-5))
    assert po.learning_rate(1.0, 17, True, 4, 2, 10) == 1.0 - (17 // 8) / 10
    assert po.learning_rate(1.0, 17, False, 4, 2, 10) == 1.0


def test_minibatch_rows_follow_reference_reshape():
    perm = np.random.default_rng(0).permutation(24)
    got = np.stack([po.minibatch_rows(perm, 3, i) for i in range(3)])
    assert np.array_equal(got, perm.reshape(3, -1))  # jnp.reshape(x, (num_minibatches, -1, ...)), ff_mappo.py:277-280


def test_synth_env_semantics():
    env = SynthRware(9, 3, 21, 5, time_limit=4, seed=7)
    o = env.reset(0)
    assert o["agents_view"].shape == (9, 3, 24) and o["global_state"].shape == (9, 1, 63)
    assert np.array_equal(o["agents_view"][:, :, :3], np.broadcast_to(np.eye(3, dtype=np.float32), (9, 3, 3)))
    assert np.array_equal(o["global_state"][:, 0], o["agents_view"][:, :, 3:].reshape(9, 63))
    lens = []
    for t in range(1, 30):
        o, r, d, info = env.step(t)
        assert (r == r[:, :1]).all() and (d == d[:, :1]).all()  # team reward / shared done
        assert ((o["step_count"] == 0) | ~d).all()  # auto-reset: terminal step returns a reset observation
        lens.append(info["episode_length"][info["is_terminal_step"]])
    lens = np.concatenate(lens)
    assert lens.max() <= 4 and lens.min() >= 1


# ------------------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", ["gae_small.npz", "gae_cfg1.npz"])
def test_golden_gae(name):
    z = np.load(os.path.join(G, name))
    adv, tgt = po.gae(z["reward"], z["value"], z["done"], z["last_val"], float(z["gamma"]), float(z["gae_lambda"]))
    assert np.array_equal(adv, z["adv"]) and np.array_equal(tgt, z["tgt"])
    adv, tgt = po.gae(z["reward"], z["value"], z["done"], z["last_val"], float(z["gamma"]), float(z["gae_lambda"]),
                      last_done=z["last_done"])
    assert np.array_equal(adv, z["adv_rec"]) and np.array_equal(tgt, z["tgt_rec"])


@pytest.mark.parametrize("name", ["loss_small.npz", "loss_rware.npz"])
def test_golden_losses(name):
    z = np.load(os.path.join(G, name))
    O, A, nA = int(z["O"]), int(z["A"]), int(z["nA"])
    ta, la, ent, ga = po.actor_loss_and_grad(z["actor_params"].astype(np.float64), O + A, nA, z["obs"].astype(np.float64),
                                             z["mask"], z["action"], z["old_log_prob"].astype(np.float64),
                                             z["adv"].astype(np.float64), 0.2, 0.01)
    assert np.allclose(ga, z["actor_grad"], rtol=0, atol=1e-14) and abs(la - float(z["actor_loss"])) < 1e-14
    tc, vl, gc = po.critic_loss_and_grad(z["critic_params"].astype(np.float64), A * O, z["global_state"].astype(np.float64),
                                         z["old_value"].astype(np.float64), z["targets"].astype(np.float64), 0.2, 0.5)
    assert np.allclose(gc, z["critic_grad"], rtol=0, atol=1e-14) and abs(vl - float(z["value_loss"])) < 1e-14
    b = torch_ref.critic_grad(z["critic_params"].astype(np.float64), A * O, z["global_state"].astype(np.float64),
                              z["old_value"].astype(np.float64), z["targets"].astype(np.float64), 0.2, 0.5)
    assert np.abs(b[2] - z["critic_grad"]).max() < 1e-12


def test_golden_adam_and_rng():
    z = np.load(os.path.join(G, "adam.npz"))
    for tag in ("small", "big"):
        pn, mn, vn, c = po.clip_adam(z["p"], z[f"g_{tag}"], z["m"], z["v"], int(z["count"]), float(z["lr"]), float(z["max_norm"]))
        assert np.array_equal(pn, z[f"p_{tag}"]) and np.array_equal(mn, z[f"m_{tag}"]) and np.array_equal(vn, z[f"v_{tag}"])
    z = np.load(os.path.join(G, "rng.npz"))
    assert np.array_equal(philox.policy_uniforms(0x1234ABCD5678EF01, 77, 16, 5, row_offset=1000), z["policy_uniforms"])
    env = SynthRware(6, 3, 21, 5, time_limit=4, seed=1234, env_offset=10)
    assert np.array_equal(env.reset(0)["agents_view"], z["av0"])
    for t in range(1, 7):
        o, r, d, info = env.step(t)
        assert np.array_equal(o["agents_view"], z["av"][t - 1]) and np.array_equal(r, z["reward"][t - 1])
        assert np.array_equal(info["episode_length"], z["ep_length"][t - 1])


# ------------------------------------------------------------------------------ recurrent oracle
def test_recurrent_oracle_numpy_vs_torch_and_golden():
    import torch

    from oracle import rec_oracle as ro

    z = np.load(os.path.join(G, "rec_small.npz"))
    din, nA = int(z["din"]), int(z["nA"])
    y, hs, hl = ro.rec_forward(z["actor_params"], din, nA, z["obs"], z["done"], z["h0"])
    assert np.allclose(y, z["logits"], rtol=0, atol=1e-13) and np.allclose(hl, z["h_last"], rtol=0, atol=1e-13)
    yt, ht = ro.t_rec_forward(torch.tensor(z["actor_params"].astype(np.float64)), din, nA, torch.tensor(z["obs"].astype(np.float64)),
                              torch.tensor(z["done"]), torch.tensor(z["h0"].astype(np.float64)))
    assert np.abs(yt.numpy() - y).max() < 1e-12  # two independent forward implementations
    # hidden state is zeroed where done enters the step (networks.py:253-257): with all-done the output depends on x only
    y1, _, _ = ro.rec_forward(z["actor_params"], din, nA, z["obs"], np.ones_like(z["done"]), z["h0"])
    y2, _, _ = ro.rec_forward(z["actor_params"], din, nA, z["obs"], np.ones_like(z["done"]), z["h0"] * 0 + 7.0)
    assert np.array_equal(y1, y2)
    _, la, ent, ga = ro.rec_actor_loss_grad(z["actor_params"], din, nA, z["obs"], z["done"], z["h0"], z["mask"], z["action"],
                                            z["old_log_prob"], z["adv"], 0.2, 0.01)
    assert abs(la - float(z["actor_loss"])) < 1e-13 and np.abs(ga - z["actor_grad"]).max() < 1e-13
    # finite-difference check of the BPTT gradient on a few recurrent-weight coordinates
    f = lambda q: ro.rec_actor_loss_grad(q, din, nA, z["obs"], z["done"], z["h0"], z["mask"], z["action"], z["old_log_prob"],
                                         z["adv"], 0.2, 0.01)[0]
    p = z["actor_params"].astype(np.float64)
    off_wh = din * 128 + 128 + 128 * 384 + 384
    for i in (off_wh + 5, off_wh + 20000, 100):
        e = np.zeros_like(p)
        e[i] = 1e-6
        fd = (f(p + e) - f(p - e)) / 2e-6
        assert abs(fd - ga[i]) < 1e-7 + 1e-4 * abs(ga[i]), (i, fd, ga[i])


def test_recurrent_param_layout():
    from mava_amd.networks import DiscreteActionHead, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, RecurrentValueNet
    from oracle import rec_oracle as ro

    a = RecurrentActor(MLPTorso([128]), MLPTorso([128]), DiscreteActionHead(13), 155)
    c = RecurrentValueNet(MLPTorso([128]), MLPTorso([128]), True, 188)
    assert a.num_params == ro.rec_param_count(155, 13) and c.num_params == ro.rec_param_count(188, 1)
    flat = a.init_flat(0)
    t = a.tree(flat, (1, 2))["params"]
    cell = t["ScannedRNN_0"]["GRUCell_0"]
    assert set(cell) == {"ir", "iz", "in", "hr", "hz", "hn"} and "bias" not in cell["hr"] and "bias" in cell["hn"]
    assert cell["iz"]["kernel"].shape == (1, 2, 128, 128) and t["action_head"]["Dense_0"]["kernel"].shape == (1, 2, 128, 13)
    # the tree leaves are views of the flat vector in the oracle's segment order
    seg = ro.rec_unflatten(flat.numpy(), 155, 13)
    assert np.array_equal(cell["in"]["kernel"][0, 0].numpy(), seg["Wi"][:, 256:])
    assert np.array_equal(cell["hz"]["kernel"][0, 0].numpy(), seg["Wh"][:, 128:256])


# ------------------------------------------------------------------ continuous action head
def test_tanh_normal_formulas_against_scipy():
    """oracle/tanh_normal.py against scipy's independent implementations of the same published formulas (the
    reference delegates them to tensorflow_probability, which is not installable here: parity unpinned)."""
    from scipy import special, stats

    from oracle import tanh_normal as tn

    z = np.array([-30.0, -12.0, -10.0, -9.99, -5.4, -1.0, 0.0, 2.0, 4.99, 5.01, 9.0])
    np.testing.assert_allclose(tn.log_ndtr(z), special.log_ndtr(z), rtol=2e-4, atol=1e-12)
    rng = np.random.default_rng(0)
    mean, ls = rng.normal(size=(64, 3)), rng.normal(size=3) * 0.5
    scale = tn.scale_of(ls)
    np.testing.assert_allclose(scale, np.log1p(np.exp(ls)) + 1e-3, rtol=1e-12)
    eps = rng.normal(size=(64, 3))
    a, lp = tn.sample(mean, ls, eps)
    a[0, 0], a[1, 1], a[2, 2] = 0.9995, -1.0, 0.999  # beyond / at the clipping threshold
    lp = tn.log_prob(a, mean, ls)
    x = np.arctanh(np.clip(a, -0.999, 0.999))
    ref = stats.norm.logpdf(x, mean, scale) - np.log1p(-np.tanh(x) ** 2)
    ath = np.arctanh(0.999)
    ref = np.where(a >= 0.999, stats.norm.logsf(ath, mean, scale) - np.log(1 - 0.999), ref)
    ref = np.where(a <= -0.999, stats.norm.logcdf(-ath, mean, scale) - np.log(1 - 0.999), ref)
    np.testing.assert_allclose(lp, ref.sum(-1), rtol=1e-9, atol=1e-9)
    # fldj of tanh
    xs = np.linspace(-6, 6, 25)
    np.testing.assert_allclose(tn.tanh_fldj(xs), np.log1p(-np.tanh(xs) ** 2), rtol=1e-7, atol=1e-10)
    # Box-Muller noise: standard normal moments, distinct per dimension and row
    n = tn.normal_noise(7, 3, 20000, 4, tn.STREAM_SAMPLE)
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1.0) < 0.02 and abs(np.corrcoef(n[:, 0], n[:, 1])[0, 1]) < 0.03


def test_continuous_actor_gradient_finite_difference():
    from oracle import tanh_normal as tn

    rng = np.random.default_rng(1)
    din, dim, R = 7, 3, 24
    flat = np.concatenate([po.mlp_flatten(po.init_mlp(rng, din, dim, 1.0)), rng.normal(size=dim) * 0.3])
    obs = rng.normal(size=(R, din))
    mean = po.mlp_forward(po.mlp_unflatten(flat[:-dim], din, dim), obs)
    a, old_lp = tn.sample(mean, flat[-dim:], rng.normal(size=(R, dim)))
    a[0, 0], a[1, 1] = 0.9999, -0.9999  # both tail branches
    old_lp = tn.log_prob(a, mean, flat[-dim:]) + rng.normal(size=R) * 0.1
    adv, eps = rng.normal(size=R), rng.normal(size=(R, dim))
    f = lambda p: tn.actor_loss_and_grad(p, din, dim, obs, a, old_lp, adv, 0.2, 0.01, eps)
    total, la, ent, g = f(flat)
    assert np.isfinite(total) and g.shape == flat.shape
    idx = np.concatenate([rng.choice(flat.size - dim, 40, replace=False), np.arange(flat.size - dim, flat.size)])
    for i in idx:
        e = np.zeros_like(flat)
        e[i] = 1e-6
        fd = (f(flat + e)[0] - f(flat - e)[0]) / 2e-6
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_continuous_head_numpy_vs_torch_autograd():
    """Two independent restatements of the continuous head agree: oracle/tanh_normal.py (NumPy, hand-derived gradients)
    and the torch-autograd formulas in oracle/rec_oracle.py (torch.special.log_ndtr, softplus, atanh)."""
    import torch

    from oracle import rec_oracle as ro
    from oracle import tanh_normal as tn

    rng = np.random.default_rng(3)
    R, dim = 40, 4
    mean, ls = rng.normal(size=(R, dim)), rng.normal(size=dim) * 0.4
    a = np.tanh(mean + tn.scale_of(ls) * rng.normal(size=(R, dim)))
    a[0, 0], a[1, 1], a[2, 2] = 0.9999, -1.0, 0.999
    m_t = torch.tensor(mean, requires_grad=True)
    ls_t = torch.tensor(ls, requires_grad=True)
    lp_t = ro.t_tanh_normal_log_prob(torch.tensor(a), m_t, torch.nn.functional.softplus(ls_t) + 1e-3)
    np.testing.assert_allclose(lp_t.detach().numpy(), tn.log_prob(a, mean, ls), rtol=1e-9, atol=1e-9)
    w = rng.normal(size=R)
    (lp_t * torch.tensor(w)).sum().backward()
    _, dmean, dscale = tn.log_prob_terms(a, mean, tn.scale_of(ls))
    np.testing.assert_allclose(m_t.grad.numpy(), w[:, None] * dmean, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(ls_t.grad.numpy(), (w[:, None] * dscale).sum(0) * tn.sigmoid(ls), rtol=1e-7, atol=1e-10)


# ------------------------------------------------------------------ third-party restatements pinned to torch
# The reference's arithmetic that lives in flax / optax / tfp cannot be executed here (oracle headers).  What IS
# importable is torch, whose GRUCell, Adam and Categorical are independent implementations of the same published
# algebra - they pin the oracle's restatements of those three (not Mava itself: "parity unpinned" stands).
def test_gru_step_matches_torch_grucell():
    """flax.linen.GRUCell (networks.py:258) restated in oracle/rec_oracle.py:gru_step == torch.nn.GRUCell with
    b_hr = b_hz = 0 (flax's hr / hz sub-modules carry no bias), weights transposed to torch's (3H, in) layout."""
    import torch

    from oracle import rec_oracle as ro

    rng = np.random.default_rng(3)
    H = ro.H
    p = {"Wi": rng.standard_normal((H, 3 * H)) / np.sqrt(H), "bi": rng.standard_normal(3 * H) * 0.1,
         "Wh": rng.standard_normal((H, 3 * H)) / np.sqrt(H), "bhn": rng.standard_normal(H) * 0.1}
    x, h = rng.standard_normal((37, H)), rng.standard_normal((37, H))
    cell = torch.nn.GRUCell(H, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(p["Wi"].T.copy()))
        cell.weight_hh.copy_(torch.from_numpy(p["Wh"].T.copy()))
        cell.bias_ih.copy_(torch.from_numpy(p["bi"]))
        cell.bias_hh.copy_(torch.from_numpy(np.concatenate([np.zeros(2 * H), p["bhn"]])))
        want = cell(torch.from_numpy(x), torch.from_numpy(h)).numpy()
    assert np.allclose(ro.gru_step(p, x, h), want, rtol=1e-12, atol=1e-13)
    # several steps with resets, through the torch (autograd) restatement the BPTT gradients come from
    flat = ro.init_rec(rng, 11, 5, 1.0)
    flat[11 * H + H + H * 3 * H:][: 3 * H] = rng.standard_normal(3 * H) * 0.1  # bi
    xs, dn, h0 = rng.standard_normal((6, 9, 11)), rng.random((6, 9)) < 0.3, rng.standard_normal((9, H))
    q = ro.rec_unflatten(flat, 11, 5)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(q["Wi"].T.copy()))
        cell.weight_hh.copy_(torch.from_numpy(q["Wh"].T.copy()))
        cell.bias_ih.copy_(torch.from_numpy(q["bi"]))
        cell.bias_hh.copy_(torch.from_numpy(np.concatenate([np.zeros(2 * H), q["bhn"]])))
        hh = torch.from_numpy(h0)
        for t in range(6):
            hh = torch.where(torch.from_numpy(dn[t])[:, None], torch.zeros_like(hh), hh)
            hh = cell(torch.relu(torch.from_numpy(xs[t]) @ torch.from_numpy(q["Wpre"]) + torch.from_numpy(q["bpre"])), hh)
    _, _, h_last = ro.rec_forward(flat, 11, 5, xs, dn, h0)
    assert np.allclose(h_last, hh.numpy(), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("big", [False, True])
def test_clip_adam_matches_torch_optim(big):
    """optax.chain(clip_by_global_norm(0.5), adam(lr, eps=1e-5)) restated in po.clip_adam == torch.optim.Adam
    (betas (.9, .999), eps 1e-5: eps added outside the square root, like optax with eps_root = 0) fed with the
    gradient clipped by torch.nn.utils.clip_grad_norm_ (which divides by norm + 1e-6: agreement to 4e-6)."""
    import torch

    rng = np.random.default_rng(5)
    n = 300
    p0 = rng.standard_normal(n)
    w = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([w], lr=2.5e-4, betas=(0.9, 0.999), eps=1e-5)
    p, m, v, c = p0.copy(), np.zeros(n), np.zeros(n), 0
    for step in range(25):
        g = rng.standard_normal(n) * (1.0 if big else 0.01)  # big: norm >> 0.5, the clip is active
        w.grad = torch.from_numpy(g.copy())
        torch.nn.utils.clip_grad_norm_([w], 0.5)
        opt.step()
        p, m, v, c = po.clip_adam(p, g, m, v, c, 2.5e-4, 0.5)
        assert np.allclose(p - p0, w.detach().numpy() - p0, rtol=1e-5, atol=1e-12), step
    st = opt.state[w]
    assert np.allclose(m, st["exp_avg"].numpy(), rtol=1e-5, atol=1e-15)
    assert np.allclose(v, st["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-18) and c == 25


def test_masked_categorical_matches_torch_distributions():
    """tfd.Categorical(logits=where(mask, logits, finfo(f32).min)) (networks.py:116-124) restated in po ==
    torch.distributions.Categorical on the same masked logits: log_prob, entropy, mode."""
    import torch

    rng = np.random.default_rng(7)
    y = rng.standard_normal((64, 7)) * 3
    mask = rng.random((64, 7)) > 0.4
    mask[:, 2] = True
    z = po.masked_logits(y, mask)
    d = torch.distributions.Categorical(logits=torch.from_numpy(z))
    a = rng.integers(0, 7, 64)
    a = np.where(mask[np.arange(64), a], a, 2)
    lsm = po.log_softmax(z)
    assert np.allclose(lsm[np.arange(64), a], d.log_prob(torch.from_numpy(a)).numpy(), rtol=1e-12, atol=1e-12)
    assert np.allclose(po.categorical_entropy(lsm), d.entropy().numpy(), rtol=1e-12, atol=1e-12)
    assert np.array_equal(z.argmax(-1), d.probs.argmax(-1).numpy())
    assert (np.exp(lsm)[~mask] == 0).all()


def test_chunked_loss_evaluation_equals_whole_minibatch():
    """part_of / R_total: summing chunk shares reproduces the whole-minibatch loss and gradient (used by the
    full-launch-shape GPU tests)."""
    rng = np.random.default_rng(11)
    R, din, nA = 300, 13, 5
    flat = po.mlp_flatten(po.init_mlp(rng, din, nA, 1.0))
    obs, mask = rng.standard_normal((R, din)), rng.random((R, nA)) > 0.2
    act = rng.integers(0, nA, R)
    mask[np.arange(R), act] = True
    olp, adv = -np.abs(rng.standard_normal(R)), rng.standard_normal(R) * 2 + 0.3
    tot, la, ent, g = po.actor_loss_and_grad(flat, din, nA, obs, mask, act, olp, adv, 0.2, 0.01)
    acc = np.zeros(4, object)
    acc = [0.0, 0.0, 0.0, np.zeros_like(g)]
    for lo in range(0, R, 77):
        sl = slice(lo, lo + 77)
        out = po.actor_loss_and_grad(flat, din, nA, obs[sl], mask[sl], act[sl], olp[sl], adv[sl], 0.2, 0.01,
                                     part_of=(R, adv.mean(), adv.std()))
        acc = [a + b for a, b in zip(acc, out)]
    assert np.allclose([tot, la, ent], acc[:3], rtol=1e-12) and np.allclose(g, acc[3], rtol=1e-10, atol=1e-15)
    fc = po.mlp_flatten(po.init_mlp(rng, din, 1, 1.0))
    ov, tg = rng.standard_normal(R), rng.standard_normal(R)
    tot, vl, g = po.critic_loss_and_grad(fc, din, obs, ov, tg, 0.2, 0.5)
    acc = [0.0, 0.0, np.zeros_like(g)]
    for lo in range(0, R, 64):
        sl = slice(lo, lo + 64)
        acc = [a + b for a, b in zip(acc, po.critic_loss_and_grad(fc, din, obs[sl], ov[sl], tg[sl], 0.2, 0.5, R_total=R))]
    assert np.allclose([tot, vl], acc[:2], rtol=1e-12) and np.allclose(g, acc[2], rtol=1e-10, atol=1e-15)


def test_oracle_learns_match_task():
    """The oracle's whole-update loop LEARNS the synthetic env's action-dependent "match" task (team reward = fraction
    of agents whose action equals the first grid coordinate they observed, mod n_actions): the signs of the restated
    loss / gradient / Adam chain are right end to end.  The GPU counterpart is tests/test_gpu_learning.py."""
    from oracle.ppo_loop import OracleLearner

    E, A, O, nA, T, K, M = 32, 2, 10, 5, 16, 4, 2
    rng = np.random.default_rng(0)
    ora = OracleLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=1, centralised=True, seed=42, actor_lr=3e-3, critic_lr=3e-3,
                        gamma=0.5, reward_mode="match", time_limit=50)
    ora.set_params(po.mlp_flatten(po.init_mlp(rng, A + O, nA, 0.01)), po.mlp_flatten(po.init_mlp(rng, A * O, 1, 1.0)))
    rs = []
    for _ in range(60):
        ora.update([rng.permutation(T * E).astype(np.int32) for _ in range(K)])
        rs.append(ora.last_traj[0][0]["reward"].mean())
    assert np.mean(rs[:3]) < 0.35 and np.mean(rs[-5:]) > 0.55, (rs[:3], rs[-5:])


def test_permutation_oracle_is_a_uniform_bijection():
    """oracle/permutation.py (the restatement the kernel is held to bit for bit): a bijection of [0, n) for every n and
    key - the contract of jax.random.permutation (ff_mappo.py:272-273) - and position statistics over many keys that a
    uniform shuffle has: chi-square of where element 0 lands / what slot 0 holds (63 dof: 99.9 % quantile 103.4), the
    share of ascents, and no rank correlation with the identity."""
    from oracle.permutation import permutation

    for n in (1, 2, 3, 5, 7, 64, 1000, 4096, 4097, 65537, 524288):
        for counter in (0, 1):
            p = permutation(n, 42, counter)
            assert p.dtype == np.int32 and np.array_equal(np.sort(p), np.arange(n)), n
    assert not np.array_equal(permutation(4096, 42, 0), permutation(4096, 42, 1))
    assert not np.array_equal(permutation(4096, 42, 0), permutation(4096, 43, 0))
    n, keys = 64, 6400
    slot0, elem0 = np.zeros(n), np.zeros(n)
    for k in range(keys):
        p = permutation(n, 7, k)
        slot0[p[0]] += 1
        elem0[np.nonzero(p == 0)[0][0]] += 1
    for c in (slot0, elem0):
        assert ((c - keys / n) ** 2 / (keys / n)).sum() < 103.4
    p = permutation(524288, 42, 5).astype(np.float64)
    assert abs((np.diff(p) > 0).mean() - 0.5) < 5e-3
    assert abs(np.corrcoef(p, np.arange(p.size))[0, 1]) < 5e-3
    assert abs(np.corrcoef(p[1:], p[:-1])[0, 1]) < 5e-3  # neighbours are unrelated
