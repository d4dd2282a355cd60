"""Diagnostic: where does the full-launch-shape gradient error come from?  Runs BASELINE config 2's launch shape (the case
of tests/test_gpu_kernels.py::test_train_kernels_full_launch_shape, kink rows excluded) through the critic / actor
gradient kernels in both arithmetic modes against ONE float64 oracle evaluation and prints per-segment errors, the worst
entries of each mode side by side, and how far the two modes' error vectors agree (a common, arithmetic-independent part
points at the comparison itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd import ops
from mava_amd._lib import Ctx
from tests.fullshape_case import build_case, oracle_gradients

dev = torch.device("cuda", 0)
t0 = time.time()
case = build_case(int(os.environ.get("SEED", "2024")))
acc_a, acc_c = oracle_gradients(case)
print(f"case + oracle in {time.time() - t0:.0f} s", flush=True)
A, nA, Rb, din, dc = case["A"], case["nA"], case["Rb"], case["din"], case["dc"]
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
av_d, gs_d, mask_d, act_d, olp_d, adv_d, ov_d, tg_d = (d(case[k]) for k in ("av", "gs", "mask", "action", "old_lp", "adv", "old_v", "tgt"))
fa, fc = case["fa"], case["fc"]
fa_d, fc_d, idx_d = d(fa), d(fc), d(case["idx"])


def segs(din_, no_):
    o = [0, din_ * 128, din_ * 128 + 128, din_ * 128 + 128 + 16384, din_ * 128 + 256 + 16384, din_ * 128 + 256 + 16384 + 128 * no_]
    return list(zip(("W1", "b1", "W2", "b2", "W3", "b3"), o, o[1:] + [o[-1] + no_]))


def where(i, din_, no_):
    for nm, a, b in segs(din_, no_):
        if a <= i < b:
            j = i - a
            return f"{nm}[{j // (128 if nm in ('W1', 'W2') else no_ if nm == 'W3' else 1)}][{j % (128 if nm in ('W1', 'W2') else no_ if nm == 'W3' else 1)}]"
    return "?"


def report(tag, got, want, din_, no_):
    rms = np.sqrt((want ** 2).mean())
    e = got - want
    out = []
    for nm, a, b in segs(din_, no_):
        x = np.abs(e[a:b]) / (1e-4 * (np.abs(want[a:b]) + rms))
        out.append(f"{nm} {x.max():.2f}")
    tol = 1e-4 * (np.abs(want) + rms)
    print(f"  {tag}: worst entry / tolerance per segment: " + "; ".join(out) + f"; entries over 1e-4: {int((np.abs(e) > tol).sum())}", flush=True)
    return e, tol


def run_critic(mode, n_slab=256, agg=True):
    ctx = Ctx(mode, critic_aggregation=agg)
    slab = torch.zeros((n_slab, fc.size + 2), device=dev)
    out = torch.zeros(fc.size + 2, device=dev)
    ops.ppo_critic_grad(fc_d, gs_d, A, ov_d, tg_d, idx_d, 0, Rb, A, 0.2, 0.5, slab, ctx=ctx)
    ops.slab_reduce(slab, fc.size + 2, out)
    torch.cuda.synchronize()
    g = out.cpu().numpy().astype(np.float64)
    # the same slabs summed in float64 on the host: separates the slab reduction's rounding from the kernel's
    g64 = slab.cpu().numpy().astype(np.float64).sum(0)
    return g[: fc.size], g64[: fc.size]


def run_actor(mode, variant=0):
    ctx = Ctx(mode)
    ctx.set(ctx.TRAIN_VARIANT, variant)
    slab = torch.zeros((256, fa.size + 2), device=dev)
    out = torch.zeros(fa.size + 2, device=dev)
    stats = ops.adv_stats(adv_d, idx_d, 0, Rb, A)
    ops.ppo_actor_grad(fa_d, av_d, mask_d, act_d, olp_d, adv_d, stats, idx_d, 0, Rb, A, nA, 0.2, 0.01, slab, ctx=ctx)
    ops.slab_reduce(slab, fa.size + 2, out)
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64)[: fa.size]


print("== critic (value-loss gradient)")
res = {}
for mode in ("f32", "f16x2"):
    g, g64 = run_critic(mode)
    res[mode] = report(f"{mode} agg, 256 slabs", g, acc_c[2], dc, 1)
    report(f"{mode} agg, slabs summed in f64 ", g64, acc_c[2], dc, 1)
e32, tol = res["f32"]
e16, _ = res["f16x2"]
print(f"  correlation of the two modes' error vectors: {np.corrcoef(e32, e16)[0, 1]:+.3f}; rms error / rms gradient: f32 {np.sqrt((e32**2).mean()) / np.sqrt((acc_c[2]**2).mean()):.2e}, f16x2 {np.sqrt((e16**2).mean()) / np.sqrt((acc_c[2]**2).mean()):.2e}")
order = np.argsort(-np.abs(e16) / tol)[:12]
for i in order:
    print(f"    {where(int(i), dc, 1):16s} want {acc_c[2][i]:+.4e}  f16x2 err {e16[i]:+.2e} ({abs(e16[i]) / tol[i]:.2f} tol)  f32 err {e32[i]:+.2e} ({abs(e32[i]) / tol[i]:.2f} tol)")
g, _ = run_critic("f16x2", agg=False)
report("f16x2 no-agg        ", g, acc_c[2], dc, 1)
print("== actor")
for mode, variant, tag in (("f32", 0, "f32"), ("f16x2", 0, "f16x2 eight-wave"), ("f16x2", 1, "f16x2 four-wave")):
    report(tag, run_actor(mode, variant), acc_a[3], din, nA)
