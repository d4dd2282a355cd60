"""Generates tests/golden/*.npz from the CPU oracle (oracle/ppo_oracle.py, float64).

The reference (Mava) cannot be executed in the build container (jax/flax/optax/chex absent) and its
own tests hold no numeric vectors, so these fixtures are produced by this repository's restatement
of the reference's algorithm - PARITY UNPINNED with respect to the reference itself.  They pin the
oracle against regressions and give the GPU tests inputs/expected outputs that do not depend on
re-running the oracle.  Shapes: the small case (T=8,E=4,A=2) and BASELINE config-1 shape
(T=128,E=16,A=2).   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import philox, ppo_oracle as po  # noqa: E402
from oracle.synth_env import SynthRware  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gae_case(name, T, E, A, seed):
    rng = np.random.default_rng(seed)
    r = rng.standard_normal((T, E, A)).astype(np.float32)
    v = rng.standard_normal((T, E, A)).astype(np.float32)
    d = np.repeat(rng.random((T, E, 1)) < 0.1, A, 2)
    lv = rng.standard_normal((E, A)).astype(np.float32)
    ld = np.repeat(rng.random((E, 1)) < 0.2, A, 1)
    adv, tgt = po.gae(r, v, d, lv, 0.99, 0.95)
    adv_r, tgt_r = po.gae(r, v, d, lv, 0.99, 0.95, last_done=ld)
    np.savez(os.path.join(OUT, name), reward=r, value=v, done=d, last_val=lv, last_done=ld, adv=adv, tgt=tgt, adv_rec=adv_r,
             tgt_rec=tgt_r, gamma=0.99, gae_lambda=0.95)


def loss_case(name, R, O, A, nA, seed):
    rng = np.random.default_rng(seed)
    din_a, din_c = O + A, A * O
    fa = po.mlp_flatten(po.init_mlp(rng, din_a, nA, 1.0)).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, din_c, 1, 1.0)).astype(np.float32)
    obs = rng.standard_normal((R, din_a)).astype(np.float32)
    gs = rng.standard_normal((R, din_c)).astype(np.float32)
    mask = rng.random((R, nA)) > 0.25
    action = rng.integers(0, nA, R).astype(np.int32)
    mask[np.arange(R), action] = True
    y = po.mlp_forward(po.mlp_unflatten(fa.astype(np.float64), din_a, nA), obs.astype(np.float64))
    lsm = po.log_softmax(po.masked_logits(y, mask))
    old_lp = (lsm[np.arange(R), action] + rng.standard_normal(R) * 0.25).astype(np.float32)
    adv = (rng.standard_normal(R) * 2 + 0.3).astype(np.float32)
    v = po.mlp_forward(po.mlp_unflatten(fc.astype(np.float64), din_c, 1), gs.astype(np.float64))[:, 0]
    old_v = (v + rng.standard_normal(R) * 0.2).astype(np.float32)
    tgt = (v + rng.standard_normal(R)).astype(np.float32)
    ta, la, ent, ga = po.actor_loss_and_grad(fa.astype(np.float64), din_a, nA, obs.astype(np.float64), mask, action,
                                             old_lp.astype(np.float64), adv.astype(np.float64), 0.2, 0.01)
    tc, vl, gc = po.critic_loss_and_grad(fc.astype(np.float64), din_c, gs.astype(np.float64), old_v.astype(np.float64),
                                         tgt.astype(np.float64), 0.2, 0.5)
    np.savez(os.path.join(OUT, name), actor_params=fa, critic_params=fc, obs=obs, global_state=gs, mask=mask, action=action,
             old_log_prob=old_lp, adv=adv, old_value=old_v, targets=tgt, logits=y, actor_total=ta, actor_loss=la, entropy=ent,
             actor_grad=ga, critic_total=tc, value_loss=vl, critic_grad=gc, O=O, A=A, nA=nA, clip_eps=0.2, ent_coef=0.01,
             vf_coef=0.5)


def adam_case(name, n, seed):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal(n).astype(np.float32) * 0.1
    m = rng.standard_normal(n).astype(np.float32) * 1e-3
    v = rng.random(n).astype(np.float32) * 1e-5 + 1e-8
    g_small = rng.standard_normal(n).astype(np.float32) * 1e-3
    g_big = rng.standard_normal(n).astype(np.float32)
    out = {}
    for tag, g in (("small", g_small), ("big", g_big)):
        pn, mn, vn, c = po.clip_adam(p, g, m, v, 7, 2.5e-4, 0.5)
        out.update({f"p_{tag}": pn, f"m_{tag}": mn, f"v_{tag}": vn})
    np.savez(os.path.join(OUT, name), p=p, m=m, v=v, g_small=g_small, g_big=g_big, count=7, lr=2.5e-4, max_norm=0.5, **out)


def rng_case(name):
    u = philox.policy_uniforms(0x1234ABCD5678EF01, 77, 16, 5, row_offset=1000)
    env = SynthRware(6, 3, 21, 5, time_limit=4, seed=1234, env_offset=10)
    o0 = env.reset(0)
    steps = [env.step(t) for t in range(1, 7)]
    np.savez(os.path.join(OUT, name), policy_uniforms=u, av0=o0["agents_view"], mask0=o0["action_mask"],
             av=np.stack([s[0]["agents_view"] for s in steps]), reward=np.stack([s[1] for s in steps]),
             done=np.stack([s[2] for s in steps]), ep_return=np.stack([s[3]["episode_return"] for s in steps]),
             ep_length=np.stack([s[3]["episode_length"] for s in steps]))


def rec_case(name, T, R, din, nA, seed):
    from oracle import rec_oracle as ro

    rng = np.random.default_rng(seed)
    fa = ro.init_rec(rng, din, nA, 1.0).astype(np.float32)
    fc = ro.init_rec(rng, din, 1, 1.0).astype(np.float32)
    obs = rng.standard_normal((T, R, din)).astype(np.float32)
    done = rng.random((T, R)) < 0.2
    h0 = (rng.standard_normal((R, 128)) * 0.5).astype(np.float32)
    mask = rng.random((T, R, nA)) > 0.25
    action = rng.integers(0, nA, (T, R)).astype(np.int32)
    np.put_along_axis(mask, action[..., None].astype(np.int64), True, -1)
    y, hs, hl = ro.rec_forward(fa, din, nA, obs, done, h0)
    old_lp = (-1.2 + rng.standard_normal((T, R)) * 0.3).astype(np.float32)
    adv = rng.standard_normal((T, R)).astype(np.float32)
    v, _, _ = ro.rec_forward(fc, din, 1, obs, done, h0)
    old_v = (v[..., 0] + rng.standard_normal((T, R)) * 0.2).astype(np.float32)
    tgt = (v[..., 0] + rng.standard_normal((T, R))).astype(np.float32)
    ta, la, ent, ga = ro.rec_actor_loss_grad(fa, din, nA, obs, done, h0, mask, action, old_lp, adv, 0.2, 0.01)
    tc, vl, gc = ro.rec_critic_loss_grad(fc, din, obs, done, h0, old_v, tgt, 0.2, 0.5)
    np.savez(os.path.join(OUT, name), actor_params=fa, critic_params=fc, obs=obs, done=done, h0=h0, mask=mask, action=action,
             old_log_prob=old_lp, adv=adv, old_value=old_v, targets=tgt, logits=y, h_last=hl, actor_loss=la, entropy=ent,
             actor_grad=ga, value_loss=vl, critic_grad=gc, din=din, nA=nA)


if __name__ == "__main__":
    rec_case("rec_small.npz", 5, 32, 12, 5, 6)
    gae_case("gae_small.npz", 8, 4, 2, 1)
    gae_case("gae_cfg1.npz", 128, 16, 2, 2)
    loss_case("loss_small.npz", 48, 10, 2, 5, 3)
    loss_case("loss_rware.npz", 96, 66, 4, 5, 4)
    adam_case("adam.npz", 1000, 5)
    rng_case("rng.npz")
    print("golden fixtures written to", OUT)
