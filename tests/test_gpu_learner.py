"""End-to-end GPU parity: the HIP learner (through the Mava-shaped boundary) against the NumPy
whole-update oracle on identical inputs (same parameters, same Philox action noise, same epoch
permutations), plus bit-exact parity of the synthetic environment generator."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle.ppo_loop import OracleLearner
from oracle.synth_env import SynthRware
from tests.conftest import assert_close

pytestmark = pytest.mark.gpu


def _cfg(system, scenario_agents, E, T, K, M, U, extra=()):
    from mava_amd.config import compose

    cfg = compose(f"default_{system}", [f"arch.num_envs={E}", f"system.rollout_length={T}", f"system.ppo_epochs={K}",
                                        f"system.num_minibatches={M}", f"system.update_batch_size={U}", *extra])
    cfg.env.scenario.task_config.num_agents = scenario_agents
    return cfg


@pytest.mark.parametrize("E,A,O,nA,S,tiled", [(37, 3, 21, 5, 0, False), (16, 4, 66, 5, 0, False), (9, 5, 30, 6, 27, False),
                                              (12, 2, 2, 3, 0, False), (10, 4, 66, 5, 0, True), (7, 3, 19, 4, 40, True)])
def test_synthetic_env_bit_exact(dev, E, A, O, nA, S, tiled):
    """Row widths that take the 8-byte store path (even) and the scalar one (odd), a raw view with no bit chunk
    (O = 2), an independent state vector (S > 0, SMAX-shaped) and the tiled (per-agent) global state."""
    from mava_amd.envs import SyntheticRware

    env = SyntheticRware(E, A, O, nA, time_limit=6, add_global_state=True, seed=1234, env_offset=1000, device=dev,
                         state_dim=S, tile_global_state=tiled)
    ora = SynthRware(E, A, O, nA, time_limit=6, seed=1234, env_offset=1000, state_dim=S, gs_tiles=A if tiled else 1)
    state, ts = env.reset()
    o = ora.reset(0)
    assert np.array_equal(ts.observation.agents_view.cpu().numpy(), o["agents_view"])
    want_gs = o["global_state"] if tiled else np.repeat(o["global_state"], A, 1)
    assert np.array_equal(ts.observation.global_state.cpu().numpy(), want_gs)
    assert np.array_equal(ts.observation.action_mask.cpu().numpy(), o["action_mask"])
    n_term = 0
    for t in range(1, 40):
        state, ts = env.step(state, torch.zeros((E, A), dtype=torch.int32, device=dev))
        o, r, d, info = ora.step(t)
        assert np.array_equal(ts.observation.agents_view.cpu().numpy(), o["agents_view"]), t
        assert np.array_equal(ts.observation.global_state.cpu().numpy(),
                              o["global_state"] if tiled else np.repeat(o["global_state"], A, 1)), t
        assert np.array_equal(ts.observation.action_mask.cpu().numpy(), o["action_mask"]), t
        assert np.array_equal(ts.observation.step_count.cpu().numpy(), o["step_count"]), t
        assert np.array_equal(ts.reward.cpu().numpy(), r), t
        assert np.array_equal(ts.last().cpu().numpy(), d[:, 0]), t
        m = ts.extras["episode_metrics"]
        assert np.array_equal(m["episode_return"].cpu().numpy(), info["episode_return"]), t
        assert np.array_equal(m["episode_length"].cpu().numpy(), info["episode_length"]), t
        assert np.array_equal(m["is_terminal_step"].cpu().numpy(), info["is_terminal_step"]), t
        n_term += int(d[:, 0].sum())
    assert n_term >= E * 5  # the time limit of 6 forces resets
    # action-dependent "match" reward (reward_mode 1): bit-exact against the restatement on random actions
    env_m = SyntheticRware(E, A, O, nA, time_limit=6, add_global_state=True, seed=77, env_offset=3, device=dev,
                           state_dim=S, tile_global_state=tiled, reward_mode="match")
    ora_m = SynthRware(E, A, O, nA, time_limit=6, seed=77, env_offset=3, state_dim=S, gs_tiles=A if tiled else 1,
                       reward_mode="match")
    state, ts = env_m.reset()
    ora_m.reset(0)
    rng = np.random.default_rng(E)
    seen = set()
    for t in range(1, 12):
        act = rng.integers(0, nA, (E, A)).astype(np.int32)
        # half of the agents play the target: first coordinate of the observation they see, mod n_actions
        tgt = (ts.observation.agents_view[:, :, A].cpu().numpy().astype(np.int64) % nA).astype(np.int32)
        act = np.where(rng.random((E, A)) < 0.5, tgt, act)
        state, ts = env_m.step(state, torch.from_numpy(act).to(dev))
        o, r, d, info = ora_m.step(t, action=act)
        assert np.array_equal(ts.reward.cpu().numpy(), r), t
        assert np.array_equal(ts.extras["episode_metrics"]["episode_return"].cpu().numpy(), info["episode_return"]), t
        seen.update(np.unique(r).tolist())
    assert len(seen) > 2 and max(seen) == 1.0  # fractions of A, including full hits
    # agent ids are one-hot, global state is the concatenation of the raw views
    av = ts.observation.agents_view.cpu().numpy()
    assert np.array_equal(av[:, :, :A], np.broadcast_to(np.eye(A, dtype=np.float32), (E, A, A)))
    if S == 0:
        assert np.array_equal(ts.observation.global_state.cpu().numpy()[:, 0], av[:, :, A:].reshape(E, A * O))


@pytest.mark.parametrize("rollout,matmul", [("fused", "f16x2"), ("per-step", "f16x2"), ("per-step", "f32")])
@pytest.mark.parametrize("system,U", [("ff_mappo", 2), ("ff_ippo", 1)])
def test_learner_update_matches_oracle(dev, system, U, rollout, matmul, monkeypatch):
    """fused: the whole rollout in one launch (rollout_h2.hip) + the f16x2 gradient kernels (the default configuration);
    per-step: mava_policy_step_f32 + mava_synth_rware_step per time step (HIP graph from the second update on), with
    the f16x2 or the exact-f32 gradient kernels."""
    from mava_amd import envs
    from mava_amd.systems.ppo import ff_ippo, ff_mappo

    monkeypatch.setenv("MAVA_FUSED_ROLLOUT", "1" if rollout == "fused" else "0")
    monkeypatch.setenv("MAVA_MATMUL", matmul)
    # End-to-end tolerances.  The north-star figures (advantages 1e-5 on identical inputs, gradients 1e-4) are pinned at
    # kernel level (test_gpu_kernels.py: GAE on identical reward / value / done inputs, both gradient kernels in both
    # arithmetic modes).  Here values and log-probs are OUTPUTS of the acting networks, small numbers (rms ~0.1) formed
    # from O(1..10) pre-activations: exact f32 lands at ~4e-6 of their rms, the f16x2 mode (22 mantissa bits per
    # operand instead of 24) at ~1.6e-5, and the advantages inherit that through the values.  Adam's normalised step
    # g / (sqrt(v) + eps) then amplifies the relative error of the few gradient entries near eps (1e-5).
    ftol = 1e-5 if matmul == "f32" else 5e-5
    utol = 1e-3 if matmul == "f32" else 1e-2
    ptol = 1e-5 if matmul == "f32" else 1e-4

    E, A, O, nA, T, K, M = 8, 2, 10, 5, 16, 2, 2
    cfg = _cfg(system, A, E, T, K, M, U)
    cfg.env.synthetic = {"obs_dim": O, "num_actions": nA}
    cfg.system.num_updates_per_eval = 2
    cfg.system.actor_lr = 1e-3
    cfg.system.critic_lr = 2e-3
    central = system == "ff_mappo"
    mod = ff_mappo if central else ff_ippo
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner

    # make the networks non-trivial (head scale 0.01 would give a near-uniform policy)
    rng = np.random.default_rng(0)
    fa = po.mlp_flatten(po.init_mlp(rng, A + O, nA, 1.0)).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, (A * O) if central else (A + O), 1, 1.0)).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))

    ora = OracleLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, D=1, centralised=central, seed=42, actor_lr=1e-3,
                        critic_lr=2e-3)
    ora.set_params(fa, fc)

    for i in range(4):  # update 2 captures the rollout graph, updates 3 and 4 replay it (for n = 0 and n = 1)
        n = i % 2
        if i < 3:
            perms = [rng.permutation(T * E).astype(np.int32) for _ in range(K)]
            L.update(n, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        else:
            # the production path: the learner draws its own epoch permutations (mava_permutation_i32, keyed by the seed
            # and a running counter) and takes the advantage statistics of all K x M minibatches from one batched launch
            from oracle.permutation import permutation

            c0 = L.perm_count
            L.update(n)
            assert L._stats_batched and L.perm_count == c0 + K
            perms = [b.cpu().numpy() for b in L._perm_bufs]
            for k_, p in enumerate(perms):
                assert np.array_equal(p, permutation(T * E, L.seed, c0 + k_)), "epoch permutation differs from oracle/permutation.py"
        torch.cuda.synchronize()
        res = ora.update(perms)
        for u in range(U):
            rep, tr = L.reps[u], ora.last_traj[0][u]
            assert np.array_equal(rep.action.cpu().numpy(), tr["action"]), "sampled actions differ"
            assert np.array_equal(rep.done.cpu().numpy().astype(bool), tr["done"])
            assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
            assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], ftol, "log_probs")
            assert_close(rep.adv.cpu().numpy(), tr["adv"], ftol, "advantages")
            assert_close(rep.tgt.cpu().numpy(), tr["tgt"], ftol, "targets")
            # the GAE kernel itself on the values it was given: north_star 1e-5 on identical inputs
            a64, t64 = po.gae(rep.reward.cpu().numpy(), rep.value.cpu().numpy(), rep.done.cpu().numpy().astype(bool),
                              rep.last_val.cpu().numpy(), 0.99, 0.95)
            assert_close(rep.adv.cpu().numpy(), a64, 1e-5, "advantages on identical inputs")
            assert_close(rep.tgt.cpu().numpy(), t64, 1e-5, "targets on identical inputs")
            assert np.array_equal(rep.info_terminal[n].cpu().numpy().astype(bool), tr["term"])
            assert np.array_equal(rep.info_length[n].cpu().numpy(), tr["len"])
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        if matmul == "f32":  # (f16x2: see the parameter checks below)
            assert_close(L.p[: L.Pa].cpu().numpy() - fa, ora.pa - fa, utol, "actor update")
            assert_close(L.p[L.Pa :].cpu().numpy() - fc, ora.pc - fc, utol, "critic update")
        assert_close(L.p[: L.Pa].cpu().numpy(), ora.pa, ptol, "actor params")
        if matmul == "f32":
            assert_close(L.p[L.Pa :].cpu().numpy(), ora.pc, ptol, "critic params")
        else:
            # f16x2: a gradient entry far below Adam's eps (1e-5) turns a 1e-4-of-rms gradient difference into a
            # visible step (lr * dg / eps): all but a handful of the 18 K critic parameters at 1e-4, every one at 1e-3
            got, want = L.p[L.Pa :].cpu().numpy().astype(np.float64), ora.pc
            tol = 1e-4 * (np.abs(want) + np.sqrt(np.mean(want * want)))
            bad = np.abs(got - want) > tol
            assert bad.sum() <= 3, f"critic params: {int(bad.sum())} entries outside 1e-4"
            assert_close(got, want, 1e-3, "critic params (hard bound)")
            # every update is compared from an IDENTICAL state: the learner takes over the oracle's parameters and Adam
            # moments (rounded to f32, which the oracle then adopts too), so that the handful of eps-amplified entries
            # above does not compound over the four updates
            Pa = L.Pa
            for dst, a_, c_ in ((L.p, ora.pa, ora.pc), (L.m, ora.ma, ora.mc), (L.v, ora.va, ora.vc)):
                dst[:Pa].copy_(torch.from_numpy(a_.astype(np.float32)))
                dst[Pa:].copy_(torch.from_numpy(c_.astype(np.float32)))
            ora.pa, ora.pc = L.p[:Pa].cpu().numpy().astype(np.float64), L.p[Pa:].cpu().numpy().astype(np.float64)
            ora.ma, ora.mc = L.m[:Pa].cpu().numpy().astype(np.float64), L.m[Pa:].cpu().numpy().astype(np.float64)
            ora.va, ora.vc = L.v[:Pa].cpu().numpy().astype(np.float64), L.v[Pa:].cpu().numpy().astype(np.float64)
    assert L.count.cpu().tolist() == [4 * K * M, 4 * K * M]
    if rollout == "per-step":
        assert len(L._graphs) == 1, "ONE rollout graph must serve every update index n"
    else:
        assert L.fused_rollout and not L._graphs, "the fused rollout kernel did not run"


def _check_traj(L, ora, n, ftol, max_flips=4):
    """Trajectory of update n on every replica against the oracle's (which took the learner's sampled actions as
    inputs): integer / byte leaves bit-exact, floats at ftol; every action the oracle itself would have sampled
    differently must be a near-tie of the Gumbel-max (score margin < 1e-3)."""
    flips = 0
    for u in range(L.U):
        rep, tr = L.reps[u], ora.last_traj[0][u]
        own, margin = tr["own_action"], tr["action_margin"]
        diff = own != tr["action"]
        flips += int(diff.sum())
        assert (margin[diff] < 1e-3).all(), f"sampled actions differ beyond a near-tie: margins {margin[diff]}"
        assert np.array_equal(rep.done.cpu().numpy().astype(bool), tr["done"])
        assert np.array_equal(rep.reward.cpu().numpy(), tr["reward"].astype(np.float32))
        # recorded observations: slots 1 .. T-1 as the rollout wrote them; slot T is the bootstrap observation, which
        # update() has already copied to slot 0 for the next rollout (slot 0's old content is gone) - both must equal the
        # observation the oracle's env holds now
        T_ = L.T
        nxt = ora.obs[0][u]
        assert np.array_equal(rep.agents_view[1:T_].cpu().numpy(), tr["av"][1:].astype(np.float32)), "recorded observations"
        assert np.array_equal(rep.action_mask[1:T_].cpu().numpy().astype(bool), tr["mask"][1:].astype(bool))
        for slot in (0, T_):
            assert np.array_equal(rep.agents_view[slot].cpu().numpy(), nxt["agents_view"]), f"observation in slot {slot}"
            assert np.array_equal(rep.action_mask[slot].cpu().numpy().astype(bool), nxt["action_mask"].astype(bool))
        if L.centralised:
            assert np.array_equal(rep.global_state[1:T_, :, 0].cpu().numpy(), tr["cx"][1:, :, 0].astype(np.float32)), "recorded state"
            assert np.array_equal(rep.global_state[T_, :, 0].cpu().numpy(), nxt["global_state"][:, 0]), "bootstrap state"
        assert np.array_equal(rep.info_terminal[n].cpu().numpy().astype(bool), tr["term"])
        assert np.array_equal(rep.info_length[n].cpu().numpy(), tr["len"])
        assert np.array_equal(rep.info_return[n].cpu().numpy(), tr["ret"].astype(np.float32))
        assert_close(rep.value.cpu().numpy(), tr["value"], ftol, "values")
        assert_close(rep.log_prob.cpu().numpy(), tr["log_prob"], ftol, "log_probs")
        assert_close(rep.last_val.cpu().numpy(), tr["last_val"], ftol, "bootstrap value")
        assert_close(rep.adv.cpu().numpy(), tr["adv"], ftol, "advantages")
        assert_close(rep.tgt.cpu().numpy(), tr["tgt"], ftol, "targets")
        a64, t64 = po.gae(rep.reward.cpu().numpy(), rep.value.cpu().numpy(), rep.done.cpu().numpy().astype(bool),
                          rep.last_val.cpu().numpy(), 0.99, 0.95)
        assert_close(rep.adv.cpu().numpy(), a64, 1e-5, "advantages on identical inputs")  # north_star
        assert_close(rep.tgt.cpu().numpy(), t64, 1e-5, "targets on identical inputs")
    assert flips <= max_flips, f"{flips} sampled actions differ from the oracle's own draw"
    return flips


# The rollout_h2_kernel instantiations the benchmarked configurations dispatch to (csrc/rollout_h2.hip, end of file):
# instance id = NO * 100000 + S1A * 1000 + S1C * 10 + SHARED.
@pytest.mark.parametrize("system,A,E,U,K,inst", [
    ("ff_mappo", 4, 40, 1, 1, 805171),   # <8,5,17,true>: BASELINE config 2 / 3 (tiny- / small-4ag); 16 envs per block: 3 blocks, last ragged
    ("ff_mappo", 2, 40, 1, 1, 805091),   # <8,5,9,true>: tiny-2ag with the centralised critic; 32 envs per block: 2 blocks, last ragged
    ("ff_ippo", 2, 16, 2, 4, 805050),    # <8,5,5,false>: BASELINE config 1 at its OWN shape (16 envs, T = 128, U = 2, K = 4, M = 2)
    ("ff_ippo", 4, 40, 1, 1, 805050),    # <8,5,5,false> with 4 agents, 3 blocks
])
def test_fused_rollout_instantiations_match_oracle(dev, system, A, E, U, K, inst, monkeypatch):
    """ff_mappo.py:76-139 on the one-launch rollout at the BENCHMARKED template instances: O = 66, 5 actions, T = 128,
    several workgroups with a ragged last one, a time limit of 20 (every env resets ~6 times per rollout), two whole
    updates (the second starts from slot T of the first and from updated parameters) through learn()'s update(), own epoch
    permutations, against OracleLearner on identical inputs.  The launch is asserted to be the intended instantiation."""
    from mava_amd import envs
    from mava_amd._lib import lib
    from mava_amd.systems.ppo import ff_ippo, ff_mappo
    from tests.conftest import check_and_sync_f16x2_state

    monkeypatch.setenv("MAVA_FUSED_ROLLOUT", "1")
    monkeypatch.setenv("MAVA_MATMUL", "f16x2")
    O, nA, T, M, TL = 66, 5, 128, 2, 20
    cfg = _cfg(system, A, E, T, K, M, U)
    cfg.env.kwargs.time_limit = TL
    cfg.system.num_updates_per_eval = 2
    central = system == "ff_mappo"
    mod = ff_mappo if central else ff_ippo
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, _, state = mod.learner_setup(env, (42, 7, 8), cfg, device=dev)
    L = learn.learner
    rng = np.random.default_rng(A * 100 + E)
    fa = po.mlp_flatten(po.init_mlp(rng, A + O, nA, 1.0)).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, (A * O) if central else (A + O), 1, 1.0)).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    ora = OracleLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=U, D=1, centralised=central, seed=42, time_limit=TL)
    ora.set_params(fa, fc)
    fn = lib().mava_debug_rollout_last_instance
    n_term = 0
    for n in range(2):
        L.update(n)
        torch.cuda.synchronize()
        assert L.fused_rollout and fn() == inst, f"rollout instance {fn()} ran, wanted {inst}"
        perms = [b.cpu().numpy() for b in L._perm_bufs]
        res = ora.update(perms, forced_actions=[[r.action.cpu().numpy() for r in L.reps]])
        _check_traj(L, ora, n, 5e-5)
        n_term += int(sum(r.info_terminal[n].sum().item() for r in L.reps))
        assert_close(L.train_metrics[n].cpu().numpy(), res["train_metrics"], 1e-4, "train metrics", scale=1.0)
        check_and_sync_f16x2_state(L, ora)
    assert n_term >= 2 * U * E * (T // TL - 1), "the time limit must force resets inside the rollout"


def test_fused_rollout_full_size_against_per_step_kernels(dev, monkeypatch):
    """BASELINE config 2 at full size (4096 envs x 4 agents x 128 steps, 256 workgroups): the one-launch rollout against the
    per-step kernels (mava_policy_step_f32 in exact f32 + mava_synth_rware_step + mava_gae_f32) from the same state.  The
    environment outputs are bit-identical; sampled actions may differ only on near-ties of the Gumbel-max (two arithmetics);
    values / log-probs / advantages / targets within 5e-5 of their rms."""
    from mava_amd import envs
    from mava_amd.systems.ppo import ff_mappo

    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("MAVA_FUSED_ROLLOUT", fused)
        monkeypatch.setenv("MAVA_MATMUL", "f16x2")
        cfg = _cfg("ff_mappo", 4, 4096, 128, 1, 2, 1)
        cfg.env.kwargs.time_limit = 50
        env, _ = envs.make(cfg, add_global_state=True, device=dev)
        learn, _, state = ff_mappo.learner_setup(env, (42, 7, 8), cfg, device=dev)
        L = learn.learner
        rng = np.random.default_rng(5)
        L.p[: L.Pa].copy_(torch.from_numpy(po.mlp_flatten(po.init_mlp(rng, 70, 5, 1.0)).astype(np.float32)))
        L.p[L.Pa :].copy_(torch.from_numpy(po.mlp_flatten(po.init_mlp(rng, 264, 1, 1.0)).astype(np.float32)))
        L._rollout(0)
        torch.cuda.synchronize()
        assert L.fused_rollout == (fused == "1")
        r = L.reps[0]
        outs.append({k: getattr(r, k).cpu().numpy() for k in ("agents_view", "global_state", "action_mask", "step_count", "action",
                                                              "value", "reward", "log_prob", "done", "last_val", "adv", "tgt")}
                    | {"ret": r.info_return[0].cpu().numpy(), "len": r.info_length[0].cpu().numpy(),
                       "term": r.info_terminal[0].cpu().numpy()})
        del learn, L, env
        torch.cuda.empty_cache()
    f, s = outs
    for k in ("agents_view", "global_state", "action_mask", "step_count", "reward", "done", "ret", "len", "term"):
        assert np.array_equal(f[k], s[k]), k
    same = f["action"] == s["action"]
    assert (~same).mean() < 2e-5, f"{int((~same).sum())} of {same.size} sampled actions differ"
    assert_close(f["value"], s["value"], 5e-5, "values")
    assert_close(f["last_val"], s["last_val"], 5e-5, "bootstrap values")
    assert_close(np.where(same, f["log_prob"], 0.0), np.where(same, s["log_prob"], 0.0), 5e-5, "log-probs")
    assert_close(f["adv"], s["adv"], 5e-5, "advantages")
    assert_close(f["tgt"], s["tgt"], 5e-5, "targets")


def test_graph_rollout_is_bit_identical(dev, monkeypatch):
    """The captured rollout (device-side step counter) against the eager launches, 5 updates."""
    from mava_amd import envs
    from mava_amd.systems.ppo import ff_mappo

    finals = []
    monkeypatch.setenv("MAVA_FUSED_ROLLOUT", "0")  # this test is about the per-step launches and their HIP graph
    for flag in ("1", "0"):
        monkeypatch.setenv("MAVA_GRAPH_ROLLOUT", flag)
        cfg = _cfg("ff_mappo", 4, 16, 8, 2, 2, 2)
        cfg.system.num_updates_per_eval = 1
        cfg.env.kwargs.time_limit = 5
        env, _ = envs.make(cfg, add_global_state=True, device=dev)
        learn, _, state = ff_mappo.learner_setup(env, (3, 4, 5), cfg, device=dev)
        L = learn.learner
        assert L.graph_rollout == (flag == "1")
        for _ in range(5):
            out = learn(state)
            state = out.learner_state
        torch.cuda.synchronize()
        assert len(L._graphs) == (1 if flag == "1" else 0)
        finals.append((L.p.clone(), L.reps[1].action.clone(), L.reps[0].adv.clone(), L.reps[0].info_return.clone(),
                       int(L.step_dev.item()), L.t_global))
    for a, b in zip(finals[0][:4], finals[1][:4]):
        assert torch.equal(a, b)
    assert finals[0][4:] == finals[1][4:] == (40, 40)


def test_learn_contract_shapes(dev):
    """learn(LearnerState) -> ExperimentOutput with the reference's leaf layout."""
    from mava_amd import envs
    from mava_amd.learner import get_final_step_metrics
    from mava_amd.systems.ppo import ff_mappo

    E, A, T, K, M, U, N = 16, 4, 8, 2, 2, 2, 3
    cfg = _cfg("ff_mappo", A, E, T, K, M, U)
    cfg.system.num_updates_per_eval = N
    cfg.env.kwargs.time_limit = 5
    env, eval_env = envs.make(cfg, add_global_state=True, device=dev)
    learn, actor_network, state = ff_mappo.learner_setup(env, (1, 2, 3), cfg, device=dev)
    assert cfg.system.num_agents == A
    k0 = state.params.actor_params["params"]["torso"]["Dense_0"]["kernel"]
    assert k0.shape == (1, U, A + 66, 128)
    assert state.params.critic_params["params"]["Dense_0"]["kernel"].shape == (1, U, 128, 1)
    assert state.timestep.observation.global_state.shape == (1, U, E, A, A * 66)
    before = k0[:, 0].clone()  # unreplicate_batch_dim
    out = learn(state)
    torch.cuda.synchronize()
    assert out.episode_metrics["episode_return"].shape == (1, N, U, T, E)
    assert out.train_metrics["total_loss"].shape == (1, N, U, K, M)
    for k in ("total_loss", "value_loss", "actor_loss", "entropy"):
        assert torch.isfinite(out.train_metrics[k]).all()
    after = out.learner_state.params.actor_params["params"]["torso"]["Dense_0"]["kernel"][0, 0]  # unreplicate_n_dims
    assert not torch.equal(before[0], after)
    assert int(out.learner_state.opt_states.actor_opt_state.count[0, 0]) == N * K * M
    fm, done = get_final_step_metrics(out.episode_metrics)
    assert done and (fm["episode_length"] <= 5).all() and (fm["episode_length"] >= 1).all()
    # entropy of a freshly initialised policy (head scale 0.01) is close to log(n_legal)
    ent = float(out.train_metrics["entropy"][0, 0, 0, 0, 0])
    assert 1.3 < ent < np.log(5) + 1e-3
    # evaluator seam
    from mava_amd.evaluator import get_eval_fn, make_ff_eval_act_fn

    ev = get_eval_fn(eval_env, make_ff_eval_act_fn(actor_network.apply, cfg), cfg, absolute_metric=False)
    m = ev(out.learner_state.params.actor_params, 0)
    assert m["episode_return"].shape[0] >= cfg.arch.num_eval_episodes


@pytest.mark.parametrize("system", ["ff_mappo", "rec_mappo"])
def test_run_experiment_with_eval_and_checkpoint(dev, system, tmp_path, monkeypatch):
    """The host loop of ff_mappo.py:435-553 / rec_mappo.py:579-713 on the synthetic env: train, evaluate through the
    (feed-forward / recurrent) act function, save a checkpoint, and start a second run from it."""
    import importlib

    from mava_amd.config import compose
    from mava_amd.utils.checkpointing import Checkpointer

    monkeypatch.chdir(tmp_path)
    mod = importlib.import_module(f"mava_amd.systems.ppo.{system}")
    cfg = compose(f"default_{system}", ["env/scenario=tiny-2ag", "arch.num_envs=32", "system.rollout_length=8",
                                        "system.update_batch_size=1", "system.num_updates=4", "arch.num_evaluation=2",
                                        "arch.num_eval_episodes=16", "system.num_minibatches=2", "system.ppo_epochs=1"])
    cfg.env.kwargs.time_limit = 12  # short evaluation episodes
    cfg.logger.checkpointing.save_model = True
    cfg.logger.checkpointing.save_args.checkpoint_uid = "t"
    cfg.arch.num_absolute_metric_eval_episodes = 64
    recs = []
    mod.run_experiment(cfg, log=recs.append)
    absolute = [r for r in recs if "absolute_episode_return" in r]
    recs = [r for r in recs if "total_loss" in r]
    assert len(recs) == 2 and recs[1]["timestep"] == 2 * recs[0]["timestep"]
    assert len(absolute) == 1 and np.isfinite(absolute[0]["absolute_episode_return"])  # ff_mappo.py:546-553
    for r in recs:
        assert r["steps_per_second"] > 0 and np.isfinite(r["total_loss"]) and np.isfinite(r["eval_episode_return"])
    ck = Checkpointer(model_name=system, checkpoint_uid="t")
    assert ck.latest_step() is not None
    raw = ck.restore_learner_state_raw()
    assert "params" in raw and raw["params"]["actor_params"]["params"]
    first_kernel = next(iter(next(iter(raw["params"]["actor_params"]["params"].values())).values()))["kernel"]
    assert first_kernel.dim() == 2  # unreplicated: no (device, update_batch) leading dims

    cfg2 = compose(f"default_{system}", ["env/scenario=tiny-2ag", "arch.num_envs=32", "system.rollout_length=8",
                                         "system.update_batch_size=1", "system.num_updates=2", "arch.num_evaluation=1",
                                         "arch.num_eval_episodes=16", "system.num_minibatches=2", "system.ppo_epochs=1"])
    cfg2.env.kwargs.time_limit = 12
    cfg2.logger.checkpointing.load_model = True
    cfg2.logger.checkpointing.load_args.checkpoint_uid = "t"
    cfg2.arch.num_absolute_metric_eval_episodes = 32
    # this run logs through the configuration's MavaLogger (marl-eval JSON writer) instead of a callback
    cfg2.logger.use_json, cfg2.logger.use_console = True, False
    cfg2.logger.base_exp_path = str(tmp_path / "results")
    cfg2.logger.kwargs.json_path = "run"
    mod.run_experiment(cfg2)
    import json as _json

    with open(tmp_path / "results" / "json" / "run" / "metrics.json") as f:
        data = _json.load(f)
    run = data[str(cfg2.env.env_name)][str(cfg2.env.scenario.task_name)][system]["seed_42"]
    assert run["step_0"]["step_count"] > 0 and len(run["step_0"]["mean_episode_return"]) == 1
    assert "steps_per_second" not in run["step_0"] or run["step_0"]["steps_per_second"][0] > 0
    assert len(run["absolute_metrics"]["mean_episode_return"]) == 1


def _small_learner(dev, matmul, seed_keys=(42, 7, 8), E=16, T=8):
    from mava_amd import envs
    from mava_amd.systems.ppo import ff_mappo

    cfg = _cfg("ff_mappo", 4, E, T, 2, 2, 1)
    cfg.system.matmul_mode = matmul
    cfg.system.num_updates_per_eval = 1
    cfg.env.kwargs.time_limit = 5
    env, _ = envs.make(cfg, add_global_state=True, device=dev)
    learn, _, state = ff_mappo.learner_setup(env, seed_keys, cfg, device=dev)
    return learn, state


def test_f16_range_guard(dev):
    """An operand beyond f16's range must not come back as finite-looking garbage (mava_amd/guards.py): the f16x2 learner
    refuses observations / parameters >= 6e4 and reports a previous call's non-finite losses; the exact-f32 learner takes
    the same inputs."""
    from mava_amd._lib import MavaHipError

    learn, state = _small_learner(dev, "f16x2")
    L = learn.learner
    out = learn(state)  # in range: fine
    torch.cuda.synchronize()
    assert torch.isfinite(out.train_metrics["total_loss"]).all()
    L.reps[0].agents_view[0, 3, 1, 7] = 1.0e5
    with pytest.raises(MavaHipError, match="observation"):
        learn(out.learner_state)
    L.reps[0].agents_view[0, 3, 1, 7] = 1.0
    L.p[11] = -7.0e4
    with pytest.raises(MavaHipError, match="parameters"):
        learn(L.learner_state())
    L.p[11] = 0.01
    L.train_metrics[0, 0, 0, 0] = float("nan")  # what an overflowed activation leaves behind
    with pytest.raises(MavaHipError, match="non-finite"):
        learn(L.learner_state())
    L.train_metrics.zero_()
    learn(L.learner_state())
    # exact f32: the same observation is legal and the update stays finite
    learn32, state32 = _small_learner(dev, "f32")
    learn32.learner.reps[0].agents_view[0, 3, 1, 7] = 1.0e5
    out32 = learn32(state32)
    torch.cuda.synchronize()
    assert torch.isfinite(out32.train_metrics["total_loss"]).all() and torch.isfinite(learn32.learner.p).all()


def test_two_learners_with_different_arithmetic_share_nothing(dev):
    """The library keeps no process-wide mode (include/mava_hip.h, mava_ctx_*): an f16x2 learner and an exact-f32 learner
    interleaved in one process give bit for bit what each gives alone, and each handle counts its own f16x2 launches."""
    solo = {}
    for mm in ("f16x2", "f32"):
        learn, state = _small_learner(dev, mm)
        for _ in range(3):
            state = learn(state).learner_state
        torch.cuda.synchronize()
        solo[mm] = learn.learner.p.clone()
    la, sa = _small_learner(dev, "f16x2")
    lb, sb = _small_learner(dev, "f32")
    for _ in range(3):
        sa = la(sa).learner_state
        sb = lb(sb).learner_state
    torch.cuda.synchronize()
    assert torch.equal(la.learner.p, solo["f16x2"]) and torch.equal(lb.learner.p, solo["f32"])
    assert not torch.equal(solo["f16x2"], solo["f32"])
    assert la.learner.ctx.h2_launches == 3 * 2 * 2 * 2 and lb.learner.ctx.h2_launches == 0


def test_f16x2_free_running_drift(dev):
    """The default arithmetic against exact f32 over TEN updates with NO re-synchronisation: learners from the same seeds
    (same initial parameters, same env streams, same action noise, same epoch permutations) - one on the f16x2 kernels with
    the one-launch rollout, one on the exact-f32 kernels with the per-step rollout, and, as the yardstick, a second
    exact-f32 learner whose initial parameters are one float32 rounding (2^-24 relative, random sign) away: PPO + Adam is
    itself sensitive to perturbations of that size (a sampled action flips on a near-tie, Adam's g / (sqrt(v) + eps)
    turns a tiny gradient difference on a near-zero entry into a full-size step).  The f16x2 learner must stay within a small
    multiple of that yardstick - drift is a number, not a hope.  Printed for the record (profiles/r03_f16x2_drift.txt)."""
    from mava_amd import envs
    from mava_amd.systems.ppo import ff_mappo

    E, A, T, K, M, N = 64, 4, 32, 2, 2, 10
    runs = {}
    for tag, mm in (("f16x2", "f16x2"), ("f32", "f32"), ("f32 + 1 ulp", "f32")):
        cfg = _cfg("ff_mappo", A, E, T, K, M, 1)
        cfg.system.matmul_mode = mm
        cfg.system.num_updates_per_eval = 1
        cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 1e-3  # 4x Mava's default step: drift shows earlier
        cfg.env.kwargs.time_limit = 20
        env, _ = envs.make(cfg, add_global_state=True, device=dev)
        learn, _, state = ff_mappo.learner_setup(env, (42, 7, 8), cfg, device=dev)
        runs[tag] = [learn, state]
    ref = runs["f32"][0].learner
    assert runs["f16x2"][0].learner.fused_rollout and not ref.fused_rollout and torch.equal(runs["f16x2"][0].learner.p, ref.p)
    g = torch.Generator(device="cpu").manual_seed(0)
    sign = (torch.randint(0, 2, (ref.P,), generator=g).float() * 2 - 1).to(dev)
    Lp = runs["f32 + 1 ulp"][0].learner
    Lp.p.mul_(1.0 + sign * 2.0 ** -24)
    runs["f32 + 1 ulp"][1] = Lp.learner_state()
    dist = {k: [] for k in runs if k != "f32"}
    flips = {k: [] for k in dist}
    for n in range(N):
        for r in runs.values():
            r[1] = r[0](r[1]).learner_state
        torch.cuda.synchronize()
        pb = ref.p.double()
        for k in dist:
            L = runs[k][0].learner
            dist[k].append(float((L.p.double() - pb).norm() / pb.norm()))
            flips[k].append(float((L.reps[0].action != ref.reps[0].action).float().mean()))
    for k in dist:
        print(f"\n{k:12s} vs f32, free running: relative parameter distance per update " + " ".join(f"{d:.1e}" for d in dist[k])
              + "; share of differing sampled actions " + " ".join(f"{f:.0e}" for f in flips[k]))
    d16, d1 = dist["f16x2"], dist["f32 + 1 ulp"]
    assert all(np.isfinite(d16)) and d16[0] < 2e-5, d16
    assert d16[-1] < 1e-2 and max(flips["f16x2"]) < 3e-2, (d16, flips)
    assert d16[-1] < 20 * max(d1[-1], 1e-4), f"f16x2 drifts {d16[-1] / d1[-1]:.1f}x faster than a one-ulp perturbation of exact f32"
    ma, mb = runs["f16x2"][0].learner.train_metrics[0].cpu().numpy(), ref.train_metrics[0].cpu().numpy()
    assert_close(ma, mb, 5e-3, "train metrics after ten free-running updates", scale=1.0)


def test_side_stream_is_probed_for_concurrency(dev):
    """mava_amd/streams.py: the stream handed to the learners ran a spin kernel beside the launch stream's in the time of
    one (a stream that shares the launch stream's hardware queue would take the time of two and is passed over)."""
    from mava_amd import streams

    for _ in range(3):  # also after other streams exist: the case that lost the overlap in bench.py's secondary runs
        s = streams.overlapping_stream(dev)
        assert s is not None
        main = torch.cuda.current_stream(dev)
        single = streams._pair_ms(main, None, dev)
        assert streams._pair_ms(main, s, dev) < 1.5 * single
