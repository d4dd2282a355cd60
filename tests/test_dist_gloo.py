"""CPU, world_size 2 (gloo): the N>1 exchange step.  Two processes each own a disjoint env shard,
compute their local minibatch gradients with the oracle, and run the product's
`parallel.allreduce_sum_` + 1/(U*D) scaling; the result must equal the oracle's D=2 update
(mean of per-rank gradients with per-rank advantage normalisation, ff_mappo.py:224-238, Q5)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mava_amd import parallel
    from oracle import ppo_oracle as po
    from oracle.ppo_loop import OracleLearner

    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.rank_world() == (rank, world)
    E, A, O, nA, T, K, M = 4, 2, 6, 5, 8, 1, 2
    rng = np.random.default_rng(0)
    fa = po.mlp_flatten(po.init_mlp(rng, A + O, nA, 1.0))
    fc = po.mlp_flatten(po.init_mlp(rng, A * O, 1, 1.0))
    perm = rng.permutation(T * E)
    # this rank's shard: the oracle with D=1 but this rank's env ids / noise offsets
    loc = OracleLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=1, D=world, seed=42)
    loc.set_params(fa, fc)
    tr = loc._rollout(rank, 0)
    rows = po.minibatch_rows(perm, M, 0)
    flat = lambda x: x.reshape((T * E,) + x.shape[2:])
    sel = lambda x: flat(x)[rows]
    R = rows.size * A
    _, la, ent, ga = po.actor_loss_and_grad(loc.pa, A + O, nA, sel(tr["av"]).reshape(R, -1), sel(tr["mask"]).reshape(R, nA),
                                            sel(tr["action"]).reshape(R), sel(tr["log_prob"]).reshape(R),
                                            sel(tr["adv"]).reshape(R), 0.2, 0.01)
    _, vl, gc = po.critic_loss_and_grad(loc.pc, A * O, sel(tr["cx"]).reshape(R, -1), sel(tr["value"]).reshape(R),
                                        sel(tr["tgt"]).reshape(R), 0.2, 0.5)
    g = torch.from_numpy(np.concatenate([ga, gc, [la, ent, vl, 0.0]]))
    g_sync = g.clone()
    # the learner's exchange: actor slice first (in flight during the critic's backward), then the rest
    w1 = parallel.allreduce_sum_async(g[: ga.size])
    w2 = parallel.allreduce_sum_async(g[ga.size :])
    assert w1 is not None and w2 is not None
    w1.wait()
    w2.wait()
    parallel.allreduce_sum_(g_sync)
    assert torch.equal(g, g_sync)
    g = g * parallel.grad_scale(update_batch_size=1)
    pb = torch.from_numpy(fa.copy() if rank == 0 else np.zeros_like(fa))
    parallel.broadcast_(pb, src=0)
    assert np.array_equal(pb.numpy(), fa)
    q.put((rank, g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_matches_oracle():
    sys.path.insert(0, ROOT)
    from oracle import ppo_oracle as po
    from oracle.ppo_loop import OracleLearner

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])  # every rank holds the same reduced buffer

    # single-process oracle with D=2 virtual ranks: mean of the per-rank gradients
    E, A, O, nA, T, K, M = 4, 2, 6, 5, 8, 1, 2
    rng = np.random.default_rng(0)
    fa = po.mlp_flatten(po.init_mlp(rng, A + O, nA, 1.0))
    fc = po.mlp_flatten(po.init_mlp(rng, A * O, 1, 1.0))
    perm = rng.permutation(T * E)
    ref = OracleLearner(E=E, A=A, O=O, nA=nA, T=T, K=K, M=M, U=1, D=2, seed=42)
    ref.set_params(fa, fc)
    trs = [ref._rollout(d, 0) for d in range(2)]
    rows = po.minibatch_rows(perm, M, 0)
    acc = np.zeros(fa.size + fc.size + 4)
    for tr in trs:
        sel = lambda x: x.reshape((T * E,) + x.shape[2:])[rows]
        R = rows.size * A
        _, la, ent, ga = po.actor_loss_and_grad(fa, A + O, nA, sel(tr["av"]).reshape(R, -1), sel(tr["mask"]).reshape(R, nA),
                                                sel(tr["action"]).reshape(R), sel(tr["log_prob"]).reshape(R),
                                                sel(tr["adv"]).reshape(R), 0.2, 0.01)
        _, vl, gc = po.critic_loss_and_grad(fc, A * O, sel(tr["cx"]).reshape(R, -1), sel(tr["value"]).reshape(R),
                                            sel(tr["tgt"]).reshape(R), 0.2, 0.5)
        acc += np.concatenate([ga, gc, [la, ent, vl, 0.0]])
    assert np.allclose(got[0], acc / 2, rtol=1e-12, atol=1e-15)
    # the two shards really are different data (different env ids / noise streams)
    assert not np.array_equal(trs[0]["av"], trs[1]["av"])
