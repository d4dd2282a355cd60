"""Single-process rehearsal of the FF update loop WITH a process group and a forced gradient exchange per minibatch
(world_size 1, backend nccl = RCCL or gloo), with the rollout graph on and off.  A one-GPU box cannot host two RCCL
ranks, so this is how the interplay of HIP-graph replay with the collective's streams / watchdog is checked.

    python tools/graph_pg_rehearsal.py nccl|gloo|none [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
elif backend == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)

from mava_amd import envs, parallel
from mava_amd.config import compose
from mava_amd.systems.ppo import ff_mappo

if backend != "none":
    def forced(flat):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    parallel.allreduce_sum_async = forced  # the learner looks it up through the module at call time

for graph in ("1", "0"):
    os.environ["MAVA_GRAPH_ROLLOUT"] = graph
    cfg = compose("default_ff_mappo", ["env=rware", "env/scenario=tiny-4ag", "arch.num_envs=4096", "system.update_batch_size=1"])
    cfg.system.num_updates_per_eval = 1
    cfg.system.num_updates = steps + 3
    env, _ = envs.make(cfg, add_global_state=True, device=dev)
    learn, _, _ = ff_mappo.learner_setup(env, (42, 43, 44), cfg, device=dev)
    L = learn.learner
    for i in range(3):
        t0 = time.perf_counter()
        L.update(0)
        torch.cuda.synchronize()
        print(f"[{backend} graph={graph}] warmup {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        L.update(0)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print(f"[{backend} graph={graph}] {ms:.2f} ms per update, {L.T * L.E / ms * 1e3:,.0f} env-steps/s", flush=True)
if backend != "none":
    dist.destroy_process_group()
