"""The full-launch-shape gradient case of tests/test_gpu_kernels.py::test_train_kernels_full_launch_shape, shared with
tools/debug_fullshape.py: inputs at BASELINE config 2's launch shape with the rows on a kink of the loss kept out of the
minibatch, and the float64 oracle gradients evaluated in chunks (TEST INFRASTRUCTURE: imports oracle/)."""
import numpy as np

from oracle import ppo_oracle as po


def _net(rng, din, no, head_scale, bias_noise=0.1):
    p = po.init_mlp(rng, din, no, head_scale)
    p = p._replace(b1=rng.standard_normal(128) * bias_noise, b2=rng.standard_normal(128) * bias_noise,
                   b3=rng.standard_normal(no) * bias_noise)
    return po.mlp_flatten(p)


def build_case(seed=2024):
    TE, A, O, nA, Rb, n_slab = 524288, 4, 66, 5, 262144, 256
    KINK = 5e-5
    rng = np.random.default_rng(seed)
    rows, din, dc = TE * A, O + A, A * O
    av = rng.standard_normal((rows, din), dtype=np.float32)
    gs = rng.standard_normal((TE, dc), dtype=np.float32)
    mask = rng.random((rows, nA), dtype=np.float32) > 0.25
    action = rng.integers(0, nA, rows).astype(np.int32)
    mask[np.arange(rows), action] = True
    adv = (rng.standard_normal(rows, dtype=np.float32) * 2.0 + 0.3).astype(np.float32)
    fa = _net(rng, din, nA, 1.0).astype(np.float32)
    fc = _net(rng, dc, 1, 1.0).astype(np.float32)
    pa = po.mlp_unflatten(fa.astype(np.float64), din, nA)
    pc = po.mlp_unflatten(fc.astype(np.float64), dc, 1)
    old_lp = np.zeros(rows, np.float32)
    old_v = np.zeros(rows, np.float32)
    tgt = np.zeros(rows, np.float32)
    cand = rng.permutation(TE)
    keep = []
    CH = 1 << 14  # (t,e) indices per chunk = 65 536 agent rows
    n_keep = 0
    for lo in range(0, TE, CH):
        ii = cand[lo : lo + CH]
        r = (ii[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
        y, (_, z1, _, z2, _) = po.mlp_forward(pa, av[r].astype(np.float64), keep=True)
        vv, (_, c1, _, c2, _) = po.mlp_forward(pc, gs[ii].astype(np.float64), keep=True)
        near = (np.minimum(np.abs(z1).min(1), np.abs(z2).min(1)) < KINK).reshape(-1, A).any(1)
        near |= np.minimum(np.abs(c1).min(1), np.abs(c2).min(1)) < KINK
        # old log-probs / values near the current ones: both sides of the clip ranges, never within 1e-3 of a boundary
        lsm = po.log_softmax(po.masked_logits(y, mask[r]))
        dl = rng.standard_normal(r.size) * 0.25  # log ratio = lp - old_lp
        for edge in (np.log(1.2), np.log(0.8)):
            dl = np.where(np.abs(dl - edge) < 1e-3, edge + 2e-3, dl)
        old_lp[r] = (lsm[np.arange(r.size), action[r]] - dl).astype(np.float32)
        v = np.repeat(vv[:, 0], A)
        dv = rng.standard_normal(r.size) * 0.2   # v - old_v
        dv = np.where(np.abs(np.abs(dv) - 0.2) < 1e-3, np.sign(dv) * 0.203, dv)
        old_v[r] = (v - dv).astype(np.float32)
        # ... and the max(l1, l2) kink of the clipped value loss: outside the clip range the gradient jumps where
        # |v - tgt| == |v_clip - tgt|, i.e. where the target sits midway between v and v_clip
        e1 = -rng.standard_normal(r.size)          # v - tgt
        e2 = e1 - dv + np.clip(dv, -0.2, 0.2)      # v_clip - tgt
        e1 = np.where((np.abs(dv) > 0.2) & (np.abs(np.abs(e1) - np.abs(e2)) < 2e-3), e1 + 5e-3, e1)
        tgt[r] = (v - e1).astype(np.float32)
        keep.append(ii[~near])
        n_keep += int((~near).sum())
        if n_keep >= Rb:
            break
    idx = np.concatenate(keep)[:Rb].astype(np.int32)
    assert idx.size == Rb
    sel = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)  # agent rows of the minibatch, kernel order
    R = sel.size

    return dict(TE=TE, A=A, O=O, nA=nA, Rb=Rb, n_slab=n_slab, din=din, dc=dc, av=av, gs=gs, mask=mask, action=action, adv=adv,
                fa=fa, fc=fc, old_lp=old_lp, old_v=old_v, tgt=tgt, idx=idx, sel=sel, R=R)


def oracle_gradients(case):
    """(actor: [total, loss, entropy, grad], critic: [total, value_loss, grad]) in float64, chunked."""
    A, din, dc, nA = case["A"], case["din"], case["dc"], case["nA"]
    av, gs, mask, action, adv = case["av"], case["gs"], case["mask"], case["action"], case["adv"]
    fa, fc, old_lp, old_v, tgt, sel, R = case["fa"], case["fc"], case["old_lp"], case["old_v"], case["tgt"], case["sel"], case["R"]
    Pa, Pc = fa.size, fc.size
    a64 = adv[sel].astype(np.float64)
    part = (R, a64.mean(), a64.std())
    acc_a = [0.0, 0.0, 0.0, np.zeros(Pa)]
    acc_c = [0.0, 0.0, np.zeros(Pc)]
    for lo in range(0, R, 1 << 16):
        r = sel[lo : lo + (1 << 16)]
        o = po.actor_loss_and_grad(fa.astype(np.float64), din, nA, av[r].astype(np.float64), mask[r], action[r],
                                   old_lp[r].astype(np.float64), adv[r].astype(np.float64), 0.2, 0.01, part_of=part)
        acc_a = [x + y for x, y in zip(acc_a, o)]
        o = po.critic_loss_and_grad(fc.astype(np.float64), dc, gs[r // A].astype(np.float64), old_v[r].astype(np.float64),
                                    tgt[r].astype(np.float64), 0.2, 0.5, R_total=R)
        acc_c = [x + y for x, y in zip(acc_c, o)]
    return acc_a, acc_c
