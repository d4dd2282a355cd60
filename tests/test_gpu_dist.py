"""GPU, world_size 2: the HIP learner itself with a process group.  Two fresh child processes (gloo rendezvous on
127.0.0.1, both on cuda:0 - a one-GPU box cannot host two RCCL ranks) each run FFLearner.update on their env shard:
rank-offset env ids and noise rows, local advantage normalisation, the split asynchronous exchange of
`FFLearner._minibatch` (actor slice under the critic's backward, then the rest) and the 1/(U*D) scale in the Adam
kernel.  Both ranks must end with bit-identical parameters, equal to OracleLearner(D=2) - the reference's pmap over
devices + pmean("device") (mava/systems/ppo/ff_mappo.py:224-238, :388-403)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CASE = dict(E=8, A=2, O=10, nA=5, T=16, K=2, M=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, system, U, n_updates, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from mava_amd import envs, parallel
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_ippo, ff_mappo
    from oracle import ppo_oracle as po

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    parallel.init_from_env(backend="gloo")
    assert parallel.rank_world() == (rank, world)
    c = CASE
    cfg = compose(f"default_{system}", [f"arch.num_envs={c['E']}", f"system.rollout_length={c['T']}",
                                        f"system.ppo_epochs={c['K']}", f"system.num_minibatches={c['M']}",
                                        f"system.update_batch_size={U}"])
    cfg.env.scenario.task_config.num_agents = c["A"]
    cfg.env.synthetic = {"obs_dim": c["O"], "num_actions": c["nA"]}
    cfg.system.num_updates_per_eval = 1
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    central = system == "ff_mappo"
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, _, _ = (ff_mappo if central else ff_ippo).learner_setup(env, (42, 7 + rank, 8 + rank), cfg, device=dev)
    L = learn.learner
    assert (L.rank, L.world) == (rank, world)
    # learner_setup broadcast rank 0's parameters (flax.jax_utils.replicate): different init seeds, equal params
    p0 = L.p.clone()
    dist.broadcast(p0, src=0)
    assert torch.equal(p0, L.p)
    rng = np.random.default_rng(0)  # the same stream on every rank: identical parameters and permutations (Q2)
    fa = po.mlp_flatten(po.init_mlp(rng, c["A"] + c["O"], c["nA"], 1.0)).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, c["A"] * c["O"] if central else c["A"] + c["O"], 1, 1.0)).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    out = {}
    for n in range(n_updates):
        perms = [rng.permutation(c["T"] * c["E"]).astype(np.int32) for _ in range(c["K"])]
        L.update(0, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        out[f"p{n}"] = L.p.cpu().numpy()
        out[f"metrics{n}"] = L.train_metrics[0].cpu().numpy()
        for u in range(U):
            out[f"action{n}_{u}"] = L.reps[u].action.cpu().numpy()
            out[f"adv{n}_{u}"] = L.reps[u].adv.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("system,U", [("ff_mappo", 1), ("ff_ippo", 2)])
def test_two_rank_hip_learner_matches_oracle(dev, system, U, tmp_path):
    import torch.multiprocessing as mp

    from oracle import ppo_oracle as po
    from oracle.ppo_loop import OracleLearner
    from tests.conftest import assert_close

    world, port, n_updates = 2, _free_port(), 2
    ctx = mp.get_context("spawn")  # fresh interpreters: nothing of this process's HIP state is inherited
    procs = [ctx.Process(target=_worker, args=(r, world, port, system, U, n_updates, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    c = CASE
    central = system == "ff_mappo"
    rng = np.random.default_rng(0)
    fa = po.mlp_flatten(po.init_mlp(rng, c["A"] + c["O"], c["nA"], 1.0)).astype(np.float32)
    fc = po.mlp_flatten(po.init_mlp(rng, c["A"] * c["O"] if central else c["A"] + c["O"], 1, 1.0)).astype(np.float32)
    ora = OracleLearner(E=c["E"], A=c["A"], O=c["O"], nA=c["nA"], T=c["T"], K=c["K"], M=c["M"], U=U, D=world,
                        centralised=central, seed=42, actor_lr=1e-3, critic_lr=2e-3)
    ora.set_params(fa, fc)
    Pa = fa.size
    for n in range(n_updates):
        perms = [rng.permutation(c["T"] * c["E"]).astype(np.int32) for _ in range(c["K"])]
        res = ora.update(perms)
        # every rank applied the same reduced gradient: parameters stay bit-identical (ff_mappo.py:241-250)
        assert np.array_equal(got[0][f"p{n}"], got[1][f"p{n}"])
        assert np.array_equal(got[0][f"metrics{n}"], got[1][f"metrics{n}"])
        for d in range(world):
            for u in range(U):
                tr = ora.last_traj[d][u]
                assert np.array_equal(got[d][f"action{n}_{u}"], tr["action"]), (n, d, u)
                assert_close(got[d][f"adv{n}_{u}"], tr["adv"], 1e-5, f"advantages rank {d} replica {u}")
        assert not np.array_equal(got[0][f"action{n}_0"], got[1][f"action{n}_0"])  # the shards are different data
        p = got[0][f"p{n}"]
        assert_close(p[:Pa] - fa, ora.pa - fa, 1e-3, "actor update (2 ranks)")
        assert_close(p[Pa:] - fc, ora.pc - fc, 1e-3, "critic update (2 ranks)")
        assert_close(p[:Pa], ora.pa, 1e-5, "actor params (2 ranks)")
        assert_close(p[Pa:], ora.pc, 1e-5, "critic params (2 ranks)")
        assert_close(got[0][f"metrics{n}"], res["train_metrics"], 1e-4, "train metrics (2 ranks)", scale=1.0)


# BASELINE config 5's widths (rec_ippo on SMAX 3s5z_vs_3s6z: 8 agents, 5 + 9 = 14 actions, per-agent critic input), small
# in every other dimension so that the float64 BPTT oracle finishes in seconds
REC_CASE = dict(E=8, A=8, O=40, nA=14, T=8, K=1, M=2)


def _rec_worker(rank, world, port, matmul, n_updates, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from mava_amd import envs, parallel
    from mava_amd.config import compose
    from mava_amd.systems.ppo import rec_ippo
    from oracle import rec_oracle as ro

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    parallel.init_from_env(backend="gloo")
    c = REC_CASE
    cfg = compose("default_rec_ippo", [f"arch.num_envs={c['E']}", f"system.rollout_length={c['T']}", f"system.ppo_epochs={c['K']}",
                                       f"system.num_minibatches={c['M']}", "system.update_batch_size=1"])
    cfg.env.scenario.task_config.num_agents = c["A"]
    cfg.env.synthetic = {"obs_dim": c["O"], "num_actions": c["nA"]}
    cfg.env.kwargs.time_limit = 5
    cfg.system.num_updates_per_eval = 1
    cfg.system.actor_lr, cfg.system.critic_lr = 1e-3, 2e-3
    cfg.system.matmul_mode = matmul
    env, _ = envs.make(cfg, add_global_state=False, device=dev)
    learn, _, _ = rec_ippo.learner_setup(env, (42, 7 + rank, 8 + rank), cfg, device=dev)
    L = learn.learner
    assert (L.rank, L.world) == (rank, world) and L.rccl_cus == 8  # (the CU reserve of multi-rank jobs: 248 slabs at most)
    p0 = L.p.clone()
    dist.broadcast(p0, src=0)
    assert torch.equal(p0, L.p)  # learner_setup broadcast rank 0's parameters
    rng = np.random.default_rng(0)
    din = c["A"] + c["O"]
    fa = ro.init_rec(rng, din, c["nA"], 1.0).astype(np.float32)
    fc = ro.init_rec(rng, din, 1, 1.0).astype(np.float32)
    L.p[: L.Pa].copy_(torch.from_numpy(fa))
    L.p[L.Pa :].copy_(torch.from_numpy(fc))
    out = {}
    for n in range(n_updates):
        perms = [rng.permutation(c["E"]).astype(np.int32) for _ in range(c["K"])]
        L.update(0, permutations=[torch.from_numpy(p).to(dev) for p in perms])
        torch.cuda.synchronize()
        out[f"p{n}"] = L.p.cpu().numpy()
        out[f"metrics{n}"] = L.train_metrics[0].cpu().numpy()
        out[f"action{n}"] = L.reps[0].action.cpu().numpy()
        out[f"adv{n}"] = L.reps[0].adv.cpu().numpy()
        out[f"done_in{n}"] = L.reps[0].done_in.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("matmul", ["f32", "f16x2"])
def test_two_rank_rec_learner_matches_oracle(dev, matmul, tmp_path):
    """RecLearner with world size 2 (rec_mappo.py:301-306 pmean "device"): env shards with rank-offset env ids and noise
    rows, the all-reduce of the flat gradient, the 1/(U*D) scale - bit-identical parameters on both ranks, equal to the
    oracle.  OracleRecLearner has no device axis; its U replicas ARE the two ranks here (replica u = env ids [u*E, (u+1)*E),
    noise rows offset by u*E*A, gradients averaged over U - exactly what rank u of a U = 1, D = 2 job holds)."""
    import torch.multiprocessing as mp

    from oracle import rec_oracle as ro
    from oracle.rec_loop import OracleRecLearner
    from tests.conftest import assert_close

    world, port, n_updates = 2, _free_port(), 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rec_worker, args=(r, world, port, matmul, n_updates, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    c = REC_CASE
    rng = np.random.default_rng(0)
    din = c["A"] + c["O"]
    fa = ro.init_rec(rng, din, c["nA"], 1.0).astype(np.float32)
    fc = ro.init_rec(rng, din, 1, 1.0).astype(np.float32)
    ora = OracleRecLearner(E=c["E"], A=c["A"], O=c["O"], nA=c["nA"], T=c["T"], K=c["K"], M=c["M"], U=world, centralised=False,
                           seed=42, actor_lr=1e-3, critic_lr=2e-3, time_limit=5)
    ora.set_params(fa, fc)
    Pa = fa.size
    for n in range(n_updates):
        perms = [rng.permutation(c["E"]).astype(np.int32) for _ in range(c["K"])]
        res = ora.update(perms)
        assert np.array_equal(got[0][f"p{n}"], got[1][f"p{n}"]), "ranks must hold bit-identical parameters"
        assert np.array_equal(got[0][f"metrics{n}"], got[1][f"metrics{n}"])
        for d in range(world):
            tr = ora.last_traj[d]
            assert np.array_equal(got[d][f"action{n}"], tr["action"]), (n, d)
            assert np.array_equal(got[d][f"done_in{n}"].astype(bool), tr["done_in"])
            assert_close(got[d][f"adv{n}"], tr["adv"], 1e-5 if matmul == "f32" else 5e-5, f"advantages rank {d}")
        assert not np.array_equal(got[0][f"action{n}"], got[1][f"action{n}"])  # the shards are different data
        p = got[0][f"p{n}"]
        ptol = 1e-5 if matmul == "f32" else 1e-4
        for name, g_, w_ in (("actor", p[:Pa], ora.pa), ("critic", p[Pa:], ora.pc)):
            bad = np.abs(g_ - w_) > ptol * (np.abs(w_) + np.sqrt(np.mean(w_ * w_)))
            assert bad.sum() <= (0 if matmul == "f32" else 12), f"{name} params (2 ranks): {int(bad.sum())} outside {ptol}"
            assert_close(g_, w_, 1e-3, f"{name} params, hard bound (2 ranks)")
        assert_close(got[0][f"metrics{n}"], res["train_metrics"], 1e-4, "train metrics (2 ranks)", scale=1.0)
        if matmul != "f32":  # f16x2: later updates are compared from the learner's own (rounded) state, like conftest's helper
            break


def test_comm_abi_single_rank(dev, monkeypatch):
    """The C-ABI exchange step on a one-rank RCCL communicator (a box has one GPU): the sum over one rank is the
    identity, broadcast from root 0 likewise; then a learner update with MAVA_COMM=abi routing its three all-reduces
    per minibatch through mava_allreduce_sum_f32 gives bit-identical parameters to the plain single-rank update."""
    import numpy as np
    import torch

    from mava_amd import envs, parallel
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_mappo

    c = parallel.AbiComm(0, 1)
    x = torch.arange(1000, dtype=torch.float32, device=dev) * 0.5
    want = x.clone()
    c.allreduce_sum_(x)
    c.broadcast_(x, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    c.close()

    finals = []
    for mode in ("abi", ""):
        monkeypatch.setenv("MAVA_COMM", mode)
        parallel._abi_comm = None
        cfg = compose("default_ff_mappo", ["env/scenario=tiny-4ag", "arch.num_envs=16", "system.rollout_length=8",
                                           "system.ppo_epochs=2", "system.num_minibatches=2", "system.update_batch_size=1"])
        cfg.env.synthetic = {"obs_dim": 12, "num_actions": 5}
        env, _ = envs.make(cfg, add_global_state=True, device=dev)
        learn, _, state = ff_mappo.learner_setup(env, (3, 4, 5), cfg, device=dev)
        out = learn(state)
        torch.cuda.synchronize()
        finals.append(learn.learner.p.clone())
        if mode == "abi":
            assert parallel._abi_comm is not None, "the ABI communicator was not used"
            parallel._abi_comm.close()
            parallel._abi_comm = None
    assert torch.equal(finals[0], finals[1])
