"""Diagnostic: where does the full-launch-shape gradient error come from?  Runs the config-2 launch shape through the
critic / actor gradient kernels in several variants against ONE float64 oracle evaluation and prints per-segment errors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd import ops
from mava_amd._lib import Ctx
from oracle import ppo_oracle as po

dev = torch.device("cuda", 0)
TE, A, O, nA, Rb = 524288, 4, 66, 5, 262144
rng = np.random.default_rng(7)
rows, din, dc = TE * A, O + A, A * O
av = rng.standard_normal((rows, din), dtype=np.float32)
gs = rng.standard_normal((TE, dc), dtype=np.float32)
mask = rng.random((rows, nA), dtype=np.float32) > 0.25
action = rng.integers(0, nA, rows).astype(np.int32)
mask[np.arange(rows), action] = True
adv = (rng.standard_normal(rows, dtype=np.float32) * 2.0 + 0.3).astype(np.float32)
def net(din, no):
    p = po.init_mlp(rng, din, no, 1.0)
    p = p._replace(b1=rng.standard_normal(128) * 0.1, b2=rng.standard_normal(128) * 0.1, b3=rng.standard_normal(no) * 0.1)
    return po.mlp_flatten(p).astype(np.float32)
fa, fc = net(din, nA), net(dc, 1)
pa, pc = po.mlp_unflatten(fa.astype(np.float64), din, nA), po.mlp_unflatten(fc.astype(np.float64), dc, 1)
idx = rng.permutation(TE)[:Rb].astype(np.int32)
sel = (idx[:, None].astype(np.int64) * A + np.arange(A)).reshape(-1)
R = sel.size
old_lp, old_v, tgt = np.zeros(rows, np.float32), np.zeros(rows, np.float32), np.zeros(rows, np.float32)
CH = 1 << 16
t0 = time.time()
for lo in range(0, R, CH):
    r = sel[lo:lo + CH]
    lsm = po.log_softmax(po.masked_logits(po.mlp_forward(pa, av[r].astype(np.float64)), mask[r]))
    old_lp[r] = (lsm[np.arange(r.size), action[r]] + rng.standard_normal(r.size) * 0.25).astype(np.float32)
    v = po.mlp_forward(pc, gs[r // A].astype(np.float64))[:, 0]
    old_v[r] = (v + rng.standard_normal(r.size) * 0.2).astype(np.float32)
    tgt[r] = (v + rng.standard_normal(r.size)).astype(np.float32)
a64 = adv[sel].astype(np.float64)
part = (R, a64.mean(), a64.std())
acc_a, acc_c = [0.0, 0.0, 0.0, np.zeros(fa.size)], [0.0, 0.0, np.zeros(fc.size)]
for lo in range(0, R, CH):
    r = sel[lo:lo + CH]
    o = po.actor_loss_and_grad(fa.astype(np.float64), din, nA, av[r].astype(np.float64), mask[r], action[r],
                               old_lp[r].astype(np.float64), adv[r].astype(np.float64), 0.2, 0.01, part_of=part)
    acc_a = [x + y for x, y in zip(acc_a, o)]
    o = po.critic_loss_and_grad(fc.astype(np.float64), dc, gs[r // A].astype(np.float64), old_v[r].astype(np.float64),
                                tgt[r].astype(np.float64), 0.2, 0.5, R_total=R)
    acc_c = [x + y for x, y in zip(acc_c, o)]
print(f"oracle done in {time.time() - t0:.0f} s", flush=True)

d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
av_d, gs_d, mask_d, act_d, olp_d, adv_d, ov_d, tg_d = d(av), d(gs), d(mask), d(action), d(old_lp), d(adv), d(old_v), d(tgt)
fa_d, fc_d = d(fa), d(fc)

def seg_report(name, got, want, din_, no_):
    offs = [0, din_ * 128, din_ * 128 + 128, din_ * 128 + 128 + 16384, din_ * 128 + 256 + 16384, din_ * 128 + 256 + 16384 + 128 * no_, want.size]
    rms = np.sqrt((want ** 2).mean())
    out = []
    for nm, a, b in zip(("W1", "b1", "W2", "b2", "W3", "b3"), offs[:-1], offs[1:]):
        e = np.abs(got[a:b] - want[a:b])
        i = int(np.argmax(e))
        out.append(f"{nm}: maxerr/rms {e.max() / rms:.2e} (rel {e[i] / max(abs(want[a + i]), 1e-30):.1e})")
    print(f"  {name}: " + "; ".join(out), flush=True)

def run_critic(tag, idx_np, n_slab, agg, mode=0):
    ctx = Ctx("f16x2" if mode == 1 else "f32", critic_aggregation=bool(agg))
    idx_d = d(idx_np)
    slab = torch.zeros((n_slab, fc.size + 2), device=dev)
    out = torch.zeros(fc.size + 2, device=dev)
    ops.ppo_critic_grad(fc_d, gs_d, A, ov_d, tg_d, idx_d, 0, Rb, A, 0.2, 0.5, slab, ctx=ctx)
    ops.slab_reduce(slab, fc.size + 2, out)
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    print(f"[critic {tag}] value loss got {g[fc.size]:.9f} want {acc_c[1]:.9f} rel {abs(g[fc.size] - acc_c[1]) / acc_c[1]:.2e}")
    seg_report(tag, g[:fc.size], acc_c[2], dc, 1)
    o3 = dc * 128 + 128 + 16384 + 128
    e3 = g[o3:o3 + 128] - acc_c[2][o3:o3 + 128]
    print(f"    dW3 signed error: mean {e3.mean():+.2e} rms {np.sqrt((e3 ** 2).mean()):.2e}  corr with mean-h2 proxy |want| {np.corrcoef(e3, acc_c[2][o3:o3 + 128])[0, 1]:+.2f};"
          f" db3 err {g[o3 + 128] - acc_c[2][o3 + 128]:+.2e} (want {acc_c[2][o3 + 128]:+.3e})")

def run_actor(tag, idx_np, n_slab, mode):
    ctx = Ctx("f16x2" if mode == 1 else "f32")
    idx_d = d(idx_np)
    slab = torch.zeros((n_slab, fa.size + 2), device=dev)
    out = torch.zeros(fa.size + 2, device=dev)
    stats = ops.adv_stats(adv_d, idx_d, 0, Rb, A)
    ops.ppo_actor_grad(fa_d, av_d, mask_d, act_d, olp_d, adv_d, stats, idx_d, 0, Rb, A, nA, 0.2, 0.01, slab, ctx=ctx)
    ops.slab_reduce(slab, fa.size + 2, out)
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    print(f"[actor {tag}] loss got {g[fa.size]:.9f} want {acc_a[1]:.9f}; entropy got {g[fa.size + 1]:.9f} want {acc_a[2]:.9f}")
    seg_report(tag, g[:fa.size], acc_a[3], din, nA)

run_critic("f32 agg 256 slabs", idx, 256, 1)
run_critic("f16x2 agg 256 slabs", idx, 256, 1, mode=1)
run_critic("f16x2 no-agg 256 slabs", idx, 256, 0, mode=1)
run_actor("f32 256 slabs", idx, 256, 0)
run_actor("f16x2 256 slabs", idx, 256, 1)
