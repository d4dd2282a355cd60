"""GPU parity of the individual HIP kernels against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import philox, ppo_oracle as po
from tests.conftest import assert_close

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


# ------------------------------------------------------------------------------------------ GAE
@pytest.mark.parametrize(
    "T,N,rec",
    [(8, 8, False), (128, 32, False), (128, 16384, False), (128, 16384, True), (5, 7, False), (37, 130, True),
     (300, 66, False), (1, 64, False), (128, 4100, True)],
)
def test_gae_matches_oracle(dev, T, N, rec):
    from mava_amd import ops

    rng = np.random.default_rng(T * 1000 + N)
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    d = rng.random((T, N)) < (0.05 if N < 1000 else 1 / 500)
    lv = rng.standard_normal(N).astype(np.float32)
    ld = (rng.random(N) < 0.1) if rec else None
    adv64, tgt64 = po.gae(r, v, d, lv, 0.99, 0.95, last_done=ld)
    adv32, tgt32 = po.gae(r, v, d, lv, 0.99, 0.95, last_done=ld, dtype=np.float32)
    adv, tgt = ops.gae(_t(r, dev), _t(v, dev), _t(d, dev), _t(lv, dev), 0.99, 0.95,
                       last_done=None if ld is None else _t(ld, dev))
    torch.cuda.synchronize()
    # north_star: advantages/returns within 1e-5 rtol (tolerance form of BASELINE.md §2)
    assert_close(adv.cpu().numpy(), adv64, 1e-5, "adv vs f64")
    assert_close(tgt.cpu().numpy(), tgt64, 1e-5, "tgt vs f64")
    assert_close(adv.cpu().numpy(), adv32, 1e-5, "adv vs f32")


def test_gae_properties(dev):
    """lambda=gamma=1, no dones: adv_t = sum_{s>=t} r_s + last_val - V_t; all-done: adv = r - V."""
    from mava_amd import ops

    T, N = 128, 256
    rng = np.random.default_rng(3)
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    lv = rng.standard_normal(N).astype(np.float32)
    z = np.zeros((T, N), bool)
    adv, _ = ops.gae(_t(r, dev), _t(v, dev), _t(z, dev), _t(lv, dev), 1.0, 1.0)
    want = np.cumsum(r[::-1].astype(np.float64), 0)[::-1] + lv - v
    assert_close(adv.cpu().numpy(), want, 1e-5, "telescoping")
    adv, tgt = ops.gae(_t(r, dev), _t(v, dev), _t(~z, dev), _t(lv, dev), 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy(), r - v)
    assert_close(tgt.cpu().numpy(), r.astype(np.float64), 1e-6, "targets all-done")


def test_gae_empty(dev):
    from mava_amd import ops

    e = torch.empty((0, 16), device=dev)
    adv, tgt = ops.gae(e, e.clone(), torch.empty((0, 16), dtype=torch.uint8, device=dev), torch.zeros(16, device=dev), 0.99, 0.95)
    assert adv.shape == (0, 16)


# ---------------------------------------------------------------------------------- permutation
@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 64, 255, 256, 257, 1000, 4096, 4097, 65537, 524288, 1048576 + 3])
def test_permutation_matches_oracle_bit_for_bit(dev, n):
    """mava_permutation_i32 (the epoch shuffle, ff_mappo.py:272-273) against oracle/permutation.py: integer work, exact;
    and a bijection of [0, n) at every size, including the non-powers of two that need the cycle walk."""
    from mava_amd import ops
    from oracle.permutation import permutation

    for seed, counter in ((42, 0), (42, 7), (2**63 + 11, 2**40 + 5)):
        got = ops.permutation(n, seed, counter).cpu().numpy()
        assert got.dtype == np.int32 and np.array_equal(got, permutation(n, seed, counter))
        assert np.array_equal(np.sort(got), np.arange(n))
    out = torch.empty(n, dtype=torch.int32, device=dev)
    assert ops.permutation(n, 42, 7, out=out) is out
    with pytest.raises(ValueError):
        ops.permutation(n, 42, 7, out=torch.empty(n + 1, dtype=torch.int32, device=dev))


def test_adv_stats_batched_equals_per_minibatch(dev):
    """mava_adv_stats_batched_f64 (all epochs x minibatches of an update in one launch) = the per-minibatch launches bit
    for bit, and their sums give the float64 mean / variance of the gathered advantages (ff_mappo.py:164)."""
    from mava_amd import ops

    rng = np.random.default_rng(3)
    TE, A, K, M = 4096, 3, 2, 4
    Rb = TE // M
    adv = rng.standard_normal((TE, A)).astype(np.float32)
    perms = np.stack([rng.permutation(TE) for _ in range(K)]).astype(np.int32)
    adv_d, perm_d = _t(adv, dev), _t(perms, dev)
    got = ops.adv_stats_batched(adv_d.view(-1), perm_d.view(-1), Rb, A, K * M)
    for k in range(K):
        for mb in range(M):
            one = ops.adv_stats(adv_d.view(-1), perm_d[k, mb * Rb : (mb + 1) * Rb].contiguous(), 0, Rb, A)
            assert torch.equal(got[k * M + mb], one)
            sel = adv[perms[k, mb * Rb : (mb + 1) * Rb]].astype(np.float64)
            s1, s2 = got[k * M + mb].sum(0).cpu().numpy()
            np.testing.assert_allclose([s1, s2], [sel.sum(), (sel * sel).sum()], rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        ops.adv_stats_batched(adv_d.view(-1), perm_d.view(-1), Rb, A, K * M + 1)


# ----------------------------------------------------------------------------------------- Adam
@pytest.mark.parametrize("decay", [False, True])
@pytest.mark.parametrize("big_grad", [False, True])
def test_clip_adam_matches_oracle(dev, decay, big_grad):
    from mava_amd import ops

    rng = np.random.default_rng(11 + decay + 2 * big_grad)
    sizes = [26245, 50561]
    off = [0, sizes[0], sizes[0] + sizes[1]]
    P = off[-1]
    p0 = rng.standard_normal(P).astype(np.float32) * 0.1
    m0 = rng.standard_normal(P).astype(np.float32) * 1e-3
    v0 = (rng.random(P).astype(np.float32)) * 1e-5
    U_D = 4
    lrs = [2.5e-4, 1e-3]
    count0 = [9, 9]
    p, m, v = _t(p0, dev), _t(m0, dev), _t(v0, dev)
    count = torch.tensor(count0, dtype=torch.int32, device=dev)
    pr, mr, vr = p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64)
    cr = list(count0)
    loss = rng.random(3).astype(np.float32)
    metrics = torch.zeros(4, device=dev)
    for it in range(3):
        gsum = rng.standard_normal(P).astype(np.float32) * (10.0 if big_grad else 1e-3)
        ops.clip_adam(p, _t(gsum, dev), m, v, count, off, lrs, grad_scale=1.0 / U_D, max_norm=0.5, decay=decay,
                      steps_per_update=8, num_updates=10, loss_sums=_t(loss, dev), vf_coef=0.5, ent_coef=0.01,
                      metrics_out=metrics)
        for s in range(2):
            sl = slice(off[s], off[s + 1])
            lr = po.learning_rate(lrs[s], cr[s], decay, 4, 2, 10)
            pr[sl], mr[sl], vr[sl], cr[s] = po.clip_adam(pr[sl], gsum[sl].astype(np.float64) / U_D, mr[sl], vr[sl],
                                                        cr[s], lr, 0.5)
    torch.cuda.synchronize()
    assert count.cpu().tolist() == cr
    # f32 arithmetic vs the f64 oracle: elements with v ~ 1e-10 take steps of ~1e-2, whose f32 rounding
    # (~1e-7 relative per op, also present in a float32 NumPy run of the oracle) bounds the agreement.
    assert_close(p.cpu().numpy(), pr, 1e-5, "params")
    assert_close(p.cpu().numpy() - p0, pr - p0, 1e-3, "param update")  # limited by ulp(p) ~ 1e-8 on 3e-4 steps
    assert_close(m.cpu().numpy(), mr, 1e-5, "mu")
    assert_close(v.cpu().numpy(), vr, 1e-5, "nu")
    a, e, vl = (loss / U_D).tolist()
    assert_close(metrics.cpu().numpy(), np.array([a - 0.01 * e + 0.5 * vl, vl, a, e]), 1e-6, "metrics")


def test_slab_reduce(dev):
    from mava_amd import ops

    rng = np.random.default_rng(5)
    slab = rng.standard_normal((37, 1000)).astype(np.float32)
    out = torch.zeros(900, device=dev)
    ops.slab_reduce(_t(slab, dev), 900, out)
    assert_close(out.cpu().numpy(), slab[:, :900].astype(np.float64).sum(0), 1e-6, "slab")
    ops.slab_reduce(_t(slab, dev), 900, out, accumulate=True)
    assert_close(out.cpu().numpy(), 2 * slab[:, :900].astype(np.float64).sum(0), 1e-6, "slab acc")


# ------------------------------------------------------------------------------------------ MLP
def _net(rng, din, no, head_scale, bias_noise=0.1):
    p = po.init_mlp(rng, din, no, head_scale)
    p = p._replace(b1=rng.standard_normal(128) * bias_noise, b2=rng.standard_normal(128) * bias_noise,
                   b3=rng.standard_normal(no) * bias_noise)
    return po.mlp_flatten(p)


@pytest.mark.parametrize("din,no,rows,share", [(70, 5, 64, 1), (264, 1, 100, 4), (66, 14, 33, 1), (7, 3, 1, 1),
                                               (129, 32, 257, 1), (264, 1, 16384, 4)])
def test_mlp_forward_matches_oracle(dev, din, no, rows, share):
    from mava_amd import ops

    rng = np.random.default_rng(din * 7 + no)
    flat = _net(rng, din, no, 1.0)
    rows_x = (rows + share - 1) // share
    x = rng.standard_normal((rows_x, din))
    y = ops.mlp_forward(_t(flat, dev, torch.float32), din, no, _t(x, dev, torch.float32), rows=rows, x_share=share)
    torch.cuda.synchronize()
    x32 = x.astype(np.float32).astype(np.float64)
    f32 = flat.astype(np.float32).astype(np.float64)
    want = po.mlp_forward(po.mlp_unflatten(f32, din, no), x32[np.arange(rows) // share])
    assert_close(y.cpu().numpy(), want, 1e-5, "mlp forward")


def test_policy_step_matches_oracle(dev):
    from mava_amd import ops

    rng = np.random.default_rng(2024)
    E, A, O, nA = 96, 4, 66, 5
    rows = E * A
    fa = _net(rng, O + A, nA, 1.0).astype(np.float32)  # head scale 1.0 => non-trivial distribution
    fc = _net(rng, A * O, 1, 1.0).astype(np.float32)
    av = rng.standard_normal((rows, O + A)).astype(np.float32)
    gs = rng.standard_normal((E, A * O)).astype(np.float32)
    mask = rng.random((rows, nA)) > 0.2
    mask[:, 0] = True
    seed, step = 0x1234ABCD5678EF01, 77
    action, logp, value, logits = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(gs, dev),
                                                  n_actions=nA, critic_share=A, seed=seed, step=step,
                                                  row_offset=1000, want_logits=True)
    torch.cuda.synchronize()
    pa = po.mlp_unflatten(fa.astype(np.float64), O + A, nA)
    pc = po.mlp_unflatten(fc.astype(np.float64), A * O, 1)
    y = po.mlp_forward(pa, av.astype(np.float64))
    assert_close(logits.cpu().numpy(), y, 1e-5, "logits")
    z = po.masked_logits(y, mask)
    lsm = po.log_softmax(z)
    u = philox.policy_uniforms(seed, step, rows, nA, row_offset=1000)
    a_or = po.gumbel_argmax(z, u)
    a = action.cpu().numpy()
    # f32 vs f64 Gumbel scores may flip near-ties only
    g = -np.log(-np.log(u.astype(np.float64)))
    sc = z + g
    diff = a != a_or
    if diff.any():
        gap = np.abs(sc[np.arange(rows), a] - sc[np.arange(rows), a_or])[diff]
        assert gap.max() < 1e-4, gap.max()
    assert diff.mean() < 0.01
    assert mask[np.arange(rows), a].all(), "sampled an illegal action"
    assert_close(logp.cpu().numpy(), lsm[np.arange(rows), a], 1e-5, "log_prob")
    v = po.mlp_forward(pc, gs.astype(np.float64))[:, 0]
    assert_close(value.cpu().numpy(), np.repeat(v, A), 1e-5, "value")
    # forced actions + greedy
    forced = torch.from_numpy(a_or).to(dev)
    _, logp2, _, _ = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(gs, dev),
                                     n_actions=nA, critic_share=A, seed=seed, step=step, forced_action=forced)
    assert_close(logp2.cpu().numpy(), lsm[np.arange(rows), a_or], 1e-5, "forced log_prob")
    ag, _, _, _ = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(gs, dev), n_actions=nA,
                                  critic_share=A, seed=seed, step=step, greedy=True)
    assert (ag.cpu().numpy() == np.argmax(z, -1)).mean() > 0.995


def test_policy_sampling_distribution(dev):
    """Sampled actions follow softmax(masked logits): chi-square on one repeated state."""
    from mava_amd import ops

    rng = np.random.default_rng(9)
    O, nA, rows = 20, 5, 1 << 16
    fa = _net(rng, O, nA, 2.0).astype(np.float32)
    fc = _net(rng, O, 1, 1.0).astype(np.float32)
    x1 = rng.standard_normal((1, O)).astype(np.float32)
    av = np.repeat(x1, rows, 0)
    mask = np.ones((rows, nA), bool)
    mask[:, 3] = False
    action, _, _, _ = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(av, dev), n_actions=nA,
                                      seed=5, step=1)
    a = action.cpu().numpy()
    y = po.mlp_forward(po.mlp_unflatten(fa.astype(np.float64), O, nA), x1.astype(np.float64))
    p = np.exp(po.log_softmax(po.masked_logits(y, mask[:1])))[0]
    cnt = np.bincount(a, minlength=nA)
    assert cnt[3] == 0
    exp = p * rows
    keep = exp > 0
    chi2 = (((cnt - exp) ** 2)[keep] / exp[keep]).sum()
    assert chi2 < 30.0, (chi2, cnt, exp)  # 3 dof; 30 is far in the tail


# ---------------------------------------------------------------------------- PPO gradient kernels
@pytest.fixture(params=[0, 1, 2], ids=["f32", "f16x2", "f16x2-w4"])
def matmul_mode(request):
    """0: exact-f32 MFMA kernels (ppo_train.hip); 1: split-f16 operands (ppo_train_h2.hip) for the shapes it
    instantiates (others fall back to the exact kernel).  Yields (mode, launches-on-the-f16x2-kernel counter, the context
    handle that carries the mode: the library has no process-wide setting)."""
    from mava_amd._lib import Ctx

    ctx = Ctx("f16x2" if request.param >= 1 else "f32")
    if request.param == 2:  # the four-wave kernels only (ppo_train_h2.hip); default: the eight-wave actor kernel where instantiated
        ctx.set(ctx.TRAIN_VARIANT, 1)
    yield min(request.param, 1), (lambda: ctx.h2_launches), ctx
    ctx.close()


def _kink_free(rng, flat, din, no, x_rows, TE, per_index, Rb, margin=5e-5):
    """Rb of the TE (t,e) indices, none of which has a hidden pre-activation within `margin` of zero in the float64 network
    `flat` on its `per_index` input rows: at >= 10^4 rows a few of the ~10^7 pre-activations sit closer to the ReLU kink
    than the arithmetic's own rounding, f32 and f64 then legitimately take different branches, and one such unit moves
    ~330 gradient entries by a whole row's contribution (see test_train_kernels_full_launch_shape)."""
    _, (_, z1, _, z2, _) = po.mlp_forward(po.mlp_unflatten(flat.astype(np.float64), din, no), x_rows.astype(np.float64), keep=True)
    near = (np.minimum(np.abs(z1).min(1), np.abs(z2).min(1)) < margin).reshape(TE, per_index).any(1)
    cand = rng.permutation(TE)
    cand = cand[~near[cand]]
    assert cand.size >= Rb, "not enough kink-free rows"
    return cand[:Rb].astype(np.int32)


def _traj(rng, TE, A, O, nA, shared_gs=True):
    rows = TE * A
    av = rng.standard_normal((rows, O + A)).astype(np.float32)
    gs = rng.standard_normal((TE if shared_gs else rows, A * O)).astype(np.float32)
    mask = rng.random((rows, nA)) > 0.25
    action = rng.integers(0, nA, rows).astype(np.int32)
    mask[np.arange(rows), action] = True
    old_lp = (-np.abs(rng.standard_normal(rows)) - 0.5).astype(np.float32)
    adv = (rng.standard_normal(rows) * 2.0 + 0.3).astype(np.float32)
    old_v = rng.standard_normal(rows).astype(np.float32)
    tgt = (old_v + rng.standard_normal(rows) * 0.5).astype(np.float32)
    return av, gs, mask, action, old_lp, adv, old_v, tgt


@pytest.mark.parametrize("TE,A,O,nA,Rb,use_idx,n_slab", [(64, 4, 66, 5, 64, False, 3), (200, 4, 66, 5, 77, True, 8),
                                                          (96, 2, 30, 14, 40, True, 2), (33, 1, 7, 3, 33, True, 1),
                                                          (4096, 4, 66, 5, 2048, True, 256),
                                                          (4096, 4, 66, 5, 4096, True, 16),  # 32 tiles per block
                                                          # the slab count of multi-rank jobs (system.rccl_cus = 8 CUs left
                                                          # to RCCL: 248 persistent blocks), 1024 tiles dealt unevenly
                                                          (9000, 4, 66, 5, 8192, True, 248)])
def test_actor_grad_matches_oracle(dev, TE, A, O, nA, Rb, use_idx, n_slab, matmul_mode):
    from mava_amd import ops
    from oracle import torch_ref

    rng = np.random.default_rng(TE + nA)
    av, gs, mask, action, old_lp, adv, old_v, tgt = _traj(rng, TE, A, O, nA)
    din = O + A
    flat = _net(rng, din, nA, 1.0).astype(np.float32)
    if use_idx:
        idx = rng.permutation(TE)[:Rb].astype(np.int32) if Rb < 8192 else _kink_free(rng, flat, din, nA, av, TE, A, Rb)
        base = 0
    else:
        idx = np.arange(Rb, dtype=np.int32)
        base = 0
    rows_sel = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
    # make old log-probs close to the current ones so that the clip range is exercised on both sides
    y = po.mlp_forward(po.mlp_unflatten(flat.astype(np.float64), din, nA), av.astype(np.float64))
    lsm = po.log_softmax(po.masked_logits(y, mask))
    old_lp = (lsm[np.arange(TE * A), action] + rng.standard_normal(TE * A) * 0.25).astype(np.float32)

    P = flat.size
    slab = torch.zeros((n_slab, P + 2), device=dev)
    stats = ops.adv_stats(_t(adv, dev), _t(idx, dev) if use_idx else None, base, Rb, A)
    ops.ppo_actor_grad(_t(flat, dev), _t(av, dev), _t(mask, dev), _t(action, dev), _t(old_lp, dev), _t(adv, dev), stats,
                       _t(idx, dev) if use_idx else None, base, Rb, A, nA, 0.2, 0.01, slab, ctx=matmul_mode[2])
    out = torch.zeros(P + 2, device=dev)
    ops.slab_reduce(slab, P + 2, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()

    args = (flat.astype(np.float64), din, nA, av[rows_sel].astype(np.float64), mask[rows_sel], action[rows_sel],
            old_lp[rows_sel].astype(np.float64), adv[rows_sel].astype(np.float64), 0.2, 0.01)
    tot, la, ent, g = po.actor_loss_and_grad(*args)
    tot2, la2, ent2, g2 = torch_ref.actor_grad(*args)
    assert_close(g, g2, 1e-9, "oracle vs torch autograd")  # the two CPU restatements agree
    # north_star: PPO gradients within 1e-4 rtol (tolerance form of BASELINE.md §2)
    assert_close(got[:P], g, 1e-4, "actor grad")
    assert_close(got[P:], np.array([la, ent]), 1e-5, "actor loss/entropy", scale=1.0)
    if matmul_mode[0] == 1:
        assert matmul_mode[1]() == 1, "the f16x2 kernel did not run for this shape"
        ctx = matmul_mode[2]
        want_w8 = ctx.get(ctx.TRAIN_VARIANT) == 0 and din + 1 <= 128 and nA <= 16
        assert ctx.get(ctx.W8_LAUNCHES) == (1 if want_w8 else 0), "eight-wave / four-wave kernel selection"


@pytest.mark.parametrize("TE,A,O,Rb,use_idx,shared,n_slab", [(64, 4, 66, 64, False, True, 3), (200, 4, 66, 77, True, True, 8),
                                                             (96, 2, 30, 40, True, False, 2), (33, 1, 9, 33, True, True, 1),
                                                             (4096, 4, 66, 2048, True, True, 256),
                                                             (150, 8, 20, 101, True, True, 5), (70, 2, 50, 70, False, True, 4),
                                                             (4096, 4, 66, 4096, True, True, 16),  # 8 / 32 tiles per block
                                                             (2048, 4, 66, 2048, True, True, 4),
                                                             (9000, 4, 66, 8192, True, True, 248)])  # rccl_cus = 8: 248 slabs
@pytest.mark.parametrize("agg", [1, 0])
def test_critic_grad_matches_oracle(dev, TE, A, O, Rb, use_idx, shared, n_slab, agg, matmul_mode):
    """agg=1: agents that share one critic input row are aggregated (one network pass per (t,e) row, the sum of
    their loss gradients back-propagated); agg=0: one pass per agent row.  Both must match the oracle, which
    follows the reference and evaluates every agent row."""
    from mava_amd import ops
    from mava_amd._lib import lib
    from oracle import torch_ref

    if agg == 0 and not shared:
        pytest.skip("aggregation only applies to shared critic inputs")
    matmul_mode[2].set(matmul_mode[2].CRITIC_AGGREGATION, agg)

    rng = np.random.default_rng(TE + O)
    av, gs, mask, action, old_lp, adv, old_v, tgt = _traj(rng, TE, A, O, 5, shared_gs=shared)
    din = A * O
    flat = _net(rng, din, 1, 1.0).astype(np.float32)
    idx = rng.permutation(TE)[:Rb].astype(np.int32) if use_idx else np.arange(Rb, dtype=np.int32)
    if use_idx and Rb >= 8192:
        idx = _kink_free(rng, flat, din, 1, gs, TE, 1 if shared else A, Rb)
    rows_sel = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
    share = A if shared else 1
    xsel = gs[rows_sel // share]
    v_now = po.mlp_forward(po.mlp_unflatten(flat.astype(np.float64), din, 1), gs.astype(np.float64))[:, 0]
    v_rows = v_now[np.arange(TE * A) // share]
    old_v = (v_rows + rng.standard_normal(TE * A) * 0.2).astype(np.float32)  # both sides of the clip range
    tgt = (v_rows + rng.standard_normal(TE * A)).astype(np.float32)

    P = flat.size
    slab = torch.zeros((n_slab, P + 2), device=dev)
    ops.ppo_critic_grad(_t(flat, dev), _t(gs, dev), share, _t(old_v, dev), _t(tgt, dev), _t(idx, dev) if use_idx else None,
                        0, Rb, A, 0.2, 0.5, slab, ctx=matmul_mode[2])
    out = torch.zeros(P + 2, device=dev)
    ops.slab_reduce(slab, P + 2, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    args = (flat.astype(np.float64), din, xsel.astype(np.float64), old_v[rows_sel].astype(np.float64),
            tgt[rows_sel].astype(np.float64), 0.2, 0.5)
    tot, vl, g = po.critic_loss_and_grad(*args)
    tot2, vl2, g2 = torch_ref.critic_grad(*args)
    assert_close(g, g2, 1e-9, "oracle vs torch autograd")
    assert_close(got[:P], g, 1e-4, "critic grad")
    assert_close(got[P:P + 1], np.array([vl]), 1e-5, "value loss", scale=1.0)
    if matmul_mode[0] == 1:  # input widths up to 287 run on the f16x2 kernel (> 95: streamed W1), wider ones fall back
        assert matmul_mode[1]() == (1 if din <= 287 else 0)
        # the eight-wave kernel serves the value network on inputs up to 127 wide when no agents are aggregated into a row
        # (ff_ippo's critic); the four-wave kernels take the rest (and everything under MAVA_CTX_TRAIN_VARIANT bit 0)
        ctx = matmul_mode[2]
        aggregated = agg == 1 and shared and A > 1
        want_w8 = ctx.get(ctx.TRAIN_VARIANT) == 0 and din + 1 <= 128 and not aggregated
        assert ctx.get(ctx.W8_LAUNCHES) == (1 if want_w8 else 0), "eight-wave / four-wave kernel selection (critic)"


def test_train_kernels_full_launch_shape(dev, matmul_mode):
    """BASELINE config 2's REAL launch shape: one minibatch of Rb = 262 144 (t,e) indices out of T*E = 524 288,
    A = 4 -> 1 048 576 agent rows, 256 slabs (one persistent block per CU), 32-bit row cursors over the 2 M-row
    trajectory - against the float64 oracle evaluated in chunks (oracle/ppo_oracle.py part_of / R_total).

    Rows that sit on a KINK of the loss are kept out of the comparison: the derivative is discontinuous there, so
    float32 (these kernels, or the reference's own f32 XLA program) and float64 legitimately take different branches,
    and at this size ONE such row moves whole gradient blocks by ~1e-3 of the gradient's rms (1 / sqrt(R); measured
    with tools/debug_fullshape.py: identical errors for the exact-f32 and the f16x2 kernel, any slab count, any row
    order, while the losses agree to 1e-8 - a property of the comparison, not of the arithmetic).  Kinks: a hidden
    pre-activation within 5e-5 of zero (ReLU; ~4 % of the (t,e) indices are dropped), the PPO ratio within 1e-3 of
    1 +- clip_eps, |value - old_value| within 1e-3 of clip_eps, and targets midway between the value and its clipped
    version (the old log-probs / old values / targets are drawn outside those bands)."""
    from mava_amd import ops
    from tests.fullshape_case import build_case, oracle_gradients

    case = build_case()
    A, nA, Rb, n_slab = case["A"], case["nA"], case["Rb"], case["n_slab"]
    av, gs, mask, action, adv = case["av"], case["gs"], case["mask"], case["action"], case["adv"]
    fa, fc, old_lp, old_v, tgt, idx = case["fa"], case["fc"], case["old_lp"], case["old_v"], case["tgt"], case["idx"]

    idx_d, adv_d = _t(idx, dev), _t(adv, dev)
    Pa, Pc = fa.size, fc.size
    slab_a = torch.zeros((n_slab, Pa + 2), device=dev)
    slab_c = torch.zeros((n_slab, Pc + 2), device=dev)
    stats = ops.adv_stats(adv_d, idx_d, 0, Rb, A)
    ops.ppo_actor_grad(_t(fa, dev), _t(av, dev), _t(mask, dev), _t(action, dev), _t(old_lp, dev), adv_d, stats, idx_d, 0,
                       Rb, A, nA, 0.2, 0.01, slab_a, ctx=matmul_mode[2])
    ops.ppo_critic_grad(_t(fc, dev), _t(gs, dev), A, _t(old_v, dev), _t(tgt, dev), idx_d, 0, Rb, A, 0.2, 0.5, slab_c, ctx=matmul_mode[2])
    out_a = torch.zeros(Pa + 2, device=dev)
    out_c = torch.zeros(Pc + 2, device=dev)
    ops.slab_reduce(slab_a, Pa + 2, out_a)
    ops.slab_reduce(slab_c, Pc + 2, out_c)
    torch.cuda.synchronize()

    acc_a, acc_c = oracle_gradients(case)
    ga, gc = out_a.cpu().numpy(), out_c.cpu().numpy()
    assert_close(ga[:Pa], acc_a[3], 1e-4, "actor grad, full launch shape")  # north_star: PPO gradients 1e-4
    assert_close(ga[Pa:], np.array([acc_a[1], acc_a[2]]), 1e-5, "actor loss/entropy", scale=1.0)
    # The value-loss gradient is the hardest case of this shape: its entries cancel to ~1/sqrt(R) of their terms, so
    # 1e-4 of the gradient's rms is 4e-9 absolute - 1.5e-8 of the sum of the term magnitudes, the float32 rounding
    # floor.  Both arithmetic modes are held to the north-star 1e-4.  Measured over five seeds (tools/debug_fullshape.py,
    # profiles/r03_fullshape_debug.txt; worst entry over its tolerance, always in dW3): exact f32 0.18 - 0.94 (0.24 at this
    # seed); f16x2 1.1 - 1.6 before the critic's weights were split as 16 w (their low terms sat in f16's subnormal range:
    # h2_core.h W_SCALE_CRITIC), 0.5 - 1.4 since (0.52 at this seed), with an rms error equal to exact f32's.
    assert_close(gc[:Pc], acc_c[2], 1e-4, "critic grad, full launch shape")
    assert_close(gc[Pc : Pc + 1], np.array([acc_c[1]]), 1e-5, "value loss", scale=1.0)


@pytest.mark.parametrize("variant", [1, 11, 21, 22, 23, 24, 41, 42, 43, 44, 45, 46, 47])
@pytest.mark.parametrize("T,N,rec", [(128, 16384, False), (37, 136, True), (300, 264, False)])
def test_gae_variants(dev, variant, T, N, rec):
    """Every chunk/lane mapping of the GAE kernel gives the same scan (incl. ragged T and multi-slab T)."""
    from mava_amd import ops
    from mava_amd._lib import Ctx

    rng = np.random.default_rng(variant * 7 + T)
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    d = rng.random((T, N)) < 0.03
    lv = rng.standard_normal(N).astype(np.float32)
    ld = (rng.random(N) < 0.1) if rec else None
    want, want_t = po.gae(r, v, d, lv, 0.99, 0.95, last_done=ld)
    ctx = Ctx()
    ctx.set(ctx.GAE_VARIANT, variant)
    adv, tgt = ops.gae(_t(r, dev), _t(v, dev), _t(d, dev), _t(lv, dev), 0.99, 0.95,
                       last_done=None if ld is None else _t(ld, dev), ctx=ctx)
    torch.cuda.synchronize()
    assert_close(adv.cpu().numpy(), want, 1e-5, f"adv variant {variant}")
    assert_close(tgt.cpu().numpy(), want_t, 1e-5, f"tgt variant {variant}")


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_policy_kernels_both_variants(dev, variant):
    """The default acting step (per-wave actor; hybrid launch with block-cooperative critic blocks when the critic has
    few tiles), the per-wave kernel alone (1) and the block-cooperative kernels (2) agree with the oracle on logits,
    values, log-probs and sampled actions - with one critic pass per agent row and with one pass per env broadcast
    to the agents."""
    from mava_amd import ops
    from mava_amd._lib import Ctx

    rng = np.random.default_rng(77)
    E, A, O, nA = 257, 4, 66, 5  # ragged last tile
    rows = E * A
    fa = _net(rng, O + A, nA, 1.0).astype(np.float32)
    fc = _net(rng, A * O, 1, 1.0).astype(np.float32)
    av = rng.standard_normal((rows, O + A)).astype(np.float32)
    gs = rng.standard_normal((E, A * O)).astype(np.float32)
    mask = rng.random((rows, nA)) > 0.2
    mask[:, 0] = True
    ctx = Ctx()
    ctx.set(ctx.POLICY_VARIANT, variant)
    action, logp, value, logits = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(gs, dev),
                                                  n_actions=nA, critic_share=A, seed=99, step=5, row_offset=7,
                                                  want_logits=True, ctx=ctx)
    raw = ops.mlp_forward(_t(fc, dev), A * O, 1, _t(gs, dev), rows=rows, x_share=A, ctx=ctx)
    # one critic pass per env, value written to all A agent slots
    action_b, logp_b, value_b, _ = ops.policy_step(_t(fa, dev), _t(fc, dev), _t(av, dev), _t(mask, dev), _t(gs, dev),
                                                   n_actions=nA, critic_share=1, critic_rows=E, value_broadcast=A,
                                                   seed=99, step=5, row_offset=7, ctx=ctx)
    torch.cuda.synchronize()
    y = po.mlp_forward(po.mlp_unflatten(fa.astype(np.float64), O + A, nA), av.astype(np.float64))
    v = po.mlp_forward(po.mlp_unflatten(fc.astype(np.float64), A * O, 1), gs.astype(np.float64))[:, 0]
    assert_close(logits.cpu().numpy(), y, 1e-5, "logits")
    assert_close(value.cpu().numpy(), np.repeat(v, A), 1e-5, "value")
    assert_close(raw.cpu().numpy()[:, 0], np.repeat(v, A), 1e-5, "raw forward")
    assert_close(value_b.cpu().numpy(), np.repeat(v, A), 1e-5, "broadcast value")
    assert torch.equal(action_b, action) and torch.equal(logp_b, logp)
    z = po.masked_logits(y, mask)
    u = philox.policy_uniforms(99, 5, rows, nA, row_offset=7)
    a_or = po.gumbel_argmax(z, u)
    a = action.cpu().numpy()
    assert (a != a_or).mean() < 0.005
    assert_close(logp.cpu().numpy(), po.log_softmax(z)[np.arange(rows), a], 1e-5, "log_prob")


@pytest.mark.parametrize("din_c,mode,big", [(264, "f16x2", True), (70, "f16x2", False), (264, "f32", False), (150, "f16x2", True)])
def test_fused_tail_equals_separate_launches(dev, din_c, mode, big):
    """mava_ppo_finish_f32 (both slab sums + clip + Adam + count increment in two launches, the wide critic's W1 re-split by
    the last-arriving Adam block) against the six launches it replaces: the reduced gradient bit for bit, parameters and
    moments to the last few ulps (the squared norm is summed in another fixed order), and - through the next critic launch -
    the re-split W1 bit for bit equal to the stand-alone pack kernel's."""
    from mava_amd import ops
    from mava_amd._lib import Ctx

    rng = np.random.default_rng(din_c)
    din_a, nA, n_slab = 70, 5, 37
    Pa, Pc = ops.mlp_param_count(din_a, nA), ops.mlp_param_count(din_c, 1)
    P = Pa + Pc
    scale = 3.0 if big else 1e-3  # clip active / inactive
    slab_a = _t((rng.standard_normal((n_slab, Pa + 2)) * scale / n_slab).astype(np.float32), dev)
    slab_c = _t((rng.standard_normal((n_slab, Pc + 2)) * scale / n_slab).astype(np.float32), dev)
    p0 = (rng.standard_normal(P) * 0.1).astype(np.float32)
    state = lambda: (_t(p0, dev), torch.zeros(P, device=dev), torch.zeros(P, device=dev), torch.zeros(2, dtype=torch.int32, device=dev))
    kw = dict(max_norm=0.5, decay=True, steps_per_update=4, num_updates=10, vf_coef=0.5, ent_coef=0.01)
    # reference: the separate launches
    p1, m1, v1, c1 = state()
    g1 = torch.zeros(P + 4, device=dev)
    met1 = torch.zeros((3, 4), device=dev)
    ctx = Ctx(mode)
    p2, m2, v2, c2 = state()
    g2 = torch.zeros(P + 4, device=dev)
    met2 = torch.zeros((3, 4), device=dev)
    ws = ops.ppo_finish_workspace(Pa, Pc, dev)
    for step in range(3):  # (the arrival ticket must be ready again for the second and third launch)
        ops.slab_reduce2(slab_a, Pa, g1[:Pa], 2, g1[P : P + 2])
        ops.slab_reduce2(slab_c, Pc, g1[Pa:P], 1, g1[P + 2 : P + 3])
        ops.clip_adam(p1, g1, m1, v1, c1, [0, Pa, P], [1e-3, 2e-3], grad_scale=1.0, loss_sums=g1[P:], metrics_out=met1[step], **kw)
        ops.ppo_finish(ctx, slab_a, slab_c, Pa, Pc, g2, p2, m2, v2, c2, 1e-3, 2e-3, grad_scale=1.0, metrics_out=met2[step],
                       critic_din=din_c, workspace=ws, **kw)
        torch.cuda.synchronize()
        assert torch.equal(g1[: P + 3], g2[: P + 3]), "reduced gradient / loss sums"
        assert c1.tolist() == c2.tolist() == [step + 1, step + 1]
        assert torch.equal(met1[step], met2[step])
        for name, a, b in (("p", p1, p2), ("m", m1, m2), ("v", v1, v2)):
            assert_close(b.cpu().numpy(), a.cpu().numpy(), 2e-6, f"{name} after step {step}")
    wide = mode == "f16x2" and 96 <= din_c <= 287
    assert ctx.get(ctx.W1_SPLIT_FRESH) == (1 if wide else 0)
    if wide:
        # the next critic launch of this handle reads the W1 copy the Adam launch wrote; a fresh handle packs it itself
        TE, A, Rb = 64, 4, 64
        gs = _t(rng.standard_normal((TE, din_c)).astype(np.float32), dev)
        ov, tg = _t(rng.standard_normal(TE * A).astype(np.float32), dev), _t(rng.standard_normal(TE * A).astype(np.float32), dev)
        outs = []
        for c in (ctx, Ctx(mode)):
            slab = torch.zeros((3, Pc + 2), device=dev)
            ops.ppo_critic_grad(p2[Pa:], gs, A, ov, tg, None, 0, Rb, A, 0.2, 0.5, slab, ctx=c)
            torch.cuda.synchronize()
            outs.append(slab.clone())
        assert ctx.get(ctx.W1_SPLIT_FRESH) == 0, "the fresh flag is one-shot"
        assert torch.equal(outs[0], outs[1]), "critic gradient on the W1 copy re-split inside the Adam launch"


@pytest.mark.parametrize("exact_rows", ["all", "some"])
def test_f16_exact_inputs_skip_the_low_term_with_the_same_bits(dev, exact_rows):
    """Observations that are exact in f16 (flags, one-hot ids, small integers: RobotWarehouse's agents_view) have a zero low
    term: the eight-wave actor kernel and the wide critic's dW1 product then run two MFMAs per product instead of three and
    never read the low plane.  The skipped product is exactly 0, so slabs must be BIT-identical to the run that is forced
    through the three-product path (MAVA_CTX_TRAIN_VARIANT bit 1) - with every tile exact, and with exact and inexact tiles
    mixed in one launch (the flag is per tile)."""
    from mava_amd import ops
    from mava_amd._lib import Ctx

    TE, A, O, nA, Rb = 2048, 4, 66, 5, 1536
    rng = np.random.default_rng(11)
    av, gs, mask, action, old_lp, adv, old_v, tgt = _traj(rng, TE, A, O, nA)
    # f16-exact observations: bits, a one-hot id, two small integer coordinates
    av = (rng.random(av.shape) < 0.2).astype(np.float32)
    av[:, O : O + A] = np.eye(A, dtype=np.float32)[np.arange(TE * A) % A]
    av[:, :2] = rng.integers(0, 10, (TE * A, 2)).astype(np.float32)
    gs = av[:, :O].reshape(TE, A * O).copy()
    if exact_rows == "some":  # every third (t,e) index carries values with a low term
        noisy = np.arange(TE) % 3 == 0
        gs[noisy] += rng.standard_normal((int(noisy.sum()), A * O)).astype(np.float32) * 1e-3
        av.reshape(TE, A, O + A)[noisy] += rng.standard_normal((int(noisy.sum()), A, O + A)).astype(np.float32) * 1e-3
    din, dc = O + A, A * O
    fa, fc = _net(rng, din, nA, 1.0).astype(np.float32), _net(rng, dc, 1, 1.0).astype(np.float32)
    idx = rng.permutation(TE)[:Rb].astype(np.int32)
    n_slab = 48
    out = {}
    for variant in (0, 2):
        ctx = Ctx("f16x2")
        ctx.set(ctx.TRAIN_VARIANT, variant)
        sa, sc = torch.zeros((n_slab, fa.size + 2), device=dev), torch.zeros((n_slab, fc.size + 2), device=dev)
        stats = ops.adv_stats(_t(adv, dev), _t(idx, dev), 0, Rb, A)
        ops.ppo_actor_grad(_t(fa, dev), _t(av, dev), _t(mask, dev), _t(action, dev), _t(old_lp, dev), _t(adv, dev), stats,
                           _t(idx, dev), 0, Rb, A, nA, 0.2, 0.01, sa, ctx=ctx)
        ops.ppo_critic_grad(_t(fc, dev), _t(gs, dev), A, _t(old_v, dev), _t(tgt, dev), _t(idx, dev), 0, Rb, A, 0.2, 0.5, sc, ctx=ctx)
        torch.cuda.synchronize()
        assert ctx.get(ctx.W8_LAUNCHES) == 1 and ctx.h2_launches == 2
        out[variant] = (sa.clone(), sc.clone())
        ctx.close()
    assert torch.equal(out[0][0], out[2][0]), "actor slabs: two-product path != three-product path"
    assert torch.equal(out[0][1], out[2][1]), "critic slabs: two-product path != three-product path"
    assert float(out[0][0].abs().sum()) > 0 and float(out[0][1].abs().sum()) > 0
