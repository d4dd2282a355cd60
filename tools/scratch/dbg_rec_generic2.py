import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mava_amd import envs
from mava_amd.config import compose
from mava_amd.systems.ppo import rec_ippo, rec_mappo
from oracle.rec_loop import OracleRecLearner
from oracle import rec_oracle as ro
dev = torch.device("cuda", 0)
system, pre, post, act, ln, matmul = "rec_ippo", [128], [96, 32], "relu", False, sys.argv[1] if len(sys.argv) > 1 else "f16x2"
E, U, A, O, nA, T, K, M = 16, 1, 4, 10, 5, 6, 2, 2
cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                    f"system.num_minibatches={M}", f"system.update_batch_size={U}"])
cfg.env.scenario.task_config.num_agents = A
cfg.env.synthetic = {"obs_dim": O, "num_actions": nA}
cfg.env.kwargs.time_limit = 4
cfg.system.num_updates_per_eval = 2
cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
cfg.system.matmul_mode = matmul
for nc in (cfg.network.actor_network, cfg.network.critic_network):
    nc.pre_torso.layer_sizes, nc.post_torso.layer_sizes = pre, post
    for t in (nc.pre_torso, nc.post_torso):
        t.activation, t.use_layer_norm = act, ln
central = False
from mava_amd._lib import lib
def mk(mode):
    cfg.system.matmul_mode = mode
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = rec_ippo.learner_setup(env, (42, 7, 8), cfg, device=dev)
    return learn.learner
L = mk("f16x2"); L32 = mk("f32")
spec_a, spec_c = ro.rec_spec(A + O, pre, post, act, ln), ro.rec_spec(A + O, pre, post, act, ln)
rng = np.random.default_rng(2)
fa = (rng.standard_normal(L.Pa) * 0.1).astype(np.float32)
fc = (rng.standard_normal(L.Pc) * 0.1).astype(np.float32)
for l in (L, L32):
    l.p[: L.Pa].copy_(torch.from_numpy(fa)); l.p[L.Pa :].copy_(torch.from_numpy(fc))
ora = OracleRecLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3,
                       time_limit=4, actor_net=spec_a, critic_net=spec_c)
ora.set_params(fa, fc)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import check_and_sync_f16x2_state
import mava_amd.rec_learner as RL
orig = RL.ops.clip_adam
grads = []
def spy(p, g, *a, **k):
    grads.append(g.clone()); return orig(p, g, *a, **k)
RL.ops.clip_adam = spy
snaps = {}
def hook(l, tag):
    an = l.actor_network
    ob = an.backward_sequence
    def bs(*a, **k):
        wpre, wpost, _ = an._gen_ws(l.ws, True)
        snaps.setdefault(tag, []).append((wpost.y[0].clone(), wpost.z[0].clone()))
        return ob(*a, **k)
    an.backward_sequence = bs
hook(L, "16"); hook(L32, "32")
for n in range(2):
    perms = [rng.permutation(E).astype(np.int32) for _ in range(K)]
    grads.clear()
    lib().mava_ppo_set_matmul_mode(1)
    L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
    torch.cuda.synchronize()
    g16 = list(grads); grads.clear()
    lib().mava_ppo_set_matmul_mode(0)
    L32.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
    torch.cuda.synchronize()
    g32 = list(grads); grads.clear()
    res = ora.update(perms)
    for i, (x, y) in enumerate(zip(g16, g32)):
        for nm, lo, hi in (("actor", 0, L.Pa), ("critic", L.Pa, L.P), ("loss", L.P, L.P + 3)):
            xa, ya = x[lo:hi].double(), y[lo:hi].double()
            print(n, i, nm, "grad relmax", float((xa - ya).abs().max() / ya.pow(2).mean().sqrt()), "rms", float(ya.pow(2).mean().sqrt()))
    if n == 1:
        y16, z16 = snaps["16"][4]; y32, z32 = snaps["32"][4]
        rows = 192
        a16 = y16.view(-1, 96, 32)[:, 23, :].reshape(-1).cpu().numpy(); a32 = y32.view(-1, 96, 32)[:, 23, :].reshape(-1).cpu().numpy()
        zz16 = z16.view(-1, 96, 32)[:, 23, :].reshape(-1).cpu().numpy(); zz32 = z32.view(-1, 96, 32)[:, 23, :].reshape(-1).cpu().numpy()
        flip = np.nonzero((a16 > 0) != (a32 > 0))[0]
        print("col 23 flips at rows", flip, "z16", zz16[flip], "z32", zz32[flip], "a16", a16[flip], "a32", a32[flip])
        allflip = ((y16 > 0) != (y32 > 0)).sum().item()
        print("total sign flips in post layer 0:", allflip, " max |z16-z32|", float((z16 - z32).abs().max()))
        an = L.actor_network
        segs = []
        for ly in an.pre.layers: segs += [(f"pre.{ly.name}.w", ly.w, ly.K * ly.N), (f"pre.{ly.name}.b", ly.b, ly.N)]
        for n_ in ("Wi", "bi", "Wh", "bhn"):
            o, shp = an.off[n_]; segs.append((n_, o, int(np.prod(shp))))
        for ly in an.post.layers + an.post.heads: segs += [(f"post.{ly.name}.w", an.post_off + ly.w, ly.K * ly.N), (f"post.{ly.name}.b", an.post_off + ly.b, ly.N)]
        x, y = g16[0].double().cpu().numpy(), g32[0].double().cpu().numpy()
        for n_, o, c in segs:
            d = np.abs(x[o:o+c] - y[o:o+c]); r = np.sqrt((y[o:o+c]**2).mean())
            bad = np.nonzero(d > 1e-3 * r)[0]
            print(f"{n_:20s} n={c:6d} relmax {d.max()/r:.2e} nbad {bad.size} first {bad[:12]}")
    print("p diff 16 vs 32", float((L.p - L32.p).abs().max()), " 32 vs ora", np.abs(L32.p[:L.Pa].cpu().numpy() - ora.pa).max())
    if n == 0:
        check_and_sync_f16x2_state(L, ora)
        check_and_sync_f16x2_state(L32, ora)
    print("metrics diff", np.abs(L.train_metrics[n].cpu().numpy() - res["train_metrics"]).max())
sys.exit(0)
an = L.actor_network
segs = []
for ly in an.pre.layers: segs += [(f"pre.{ly.name}.w", ly.w, ly.K * ly.N), (f"pre.{ly.name}.b", ly.b, ly.N)]
for n_ in ("Wi", "bi", "Wh", "bhn"):
    o, shp = an.off[n_]; segs.append((n_, o, int(np.prod(shp))))
for ly in an.post.layers + an.post.heads: segs += [(f"post.{ly.name}.w", an.post_off + ly.w, ly.K * ly.N), (f"post.{ly.name}.b", an.post_off + ly.b, ly.N)]
got = L.p[: L.Pa].cpu().numpy().astype(np.float64); want = ora.pa
rms = np.sqrt(np.mean(want * want))
g0 = grads[0][: L.Pa].cpu().numpy()
print("grad0 actor rms", np.sqrt((g0 * g0).mean()), "absmax", np.abs(g0).max(), "grad_scale", L.grad_scale)
for n_, o, c in segs:
    e = np.abs(got[o:o+c] - want[o:o+c]) / (np.abs(want[o:o+c]) + rms)
    print(f"{n_:22s} n={c:6d} max {e.max():.2e} med {np.median(e):.2e} frac>1e-4 {(e > 1e-4).mean():.3f}  g0 rms {np.sqrt((g0[o:o+c]**2).mean()):.2e}")
