"""Long-run sanity of a learner: N updates, then every parameter / optimiser moment / logged loss must be finite.

    python tools/stability_run.py ff_mappo|ff_ippo|rec_mappo|rec_ippo discrete|continuous [updates] [envs]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mava_amd import envs
from mava_amd.config import compose
from mava_amd.systems.ppo import ff_ippo, ff_mappo, rec_ippo, rec_mappo

system = sys.argv[1] if len(sys.argv) > 1 else "ff_mappo"
head = sys.argv[2] if len(sys.argv) > 2 else "discrete"
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 100
E = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dev = torch.device("cuda", 0)
over = [f"arch.num_envs={E}", "system.update_batch_size=1", "env/scenario=tiny-4ag"]
if head == "continuous":
    over.append("network.action_head._target_=mava.networks.ContinuousActionHead")
cfg = compose(f"default_{system}", over)
if head == "continuous":
    cfg.env.synthetic = {"obs_dim": 27, "num_actions": 3}
cfg.system.num_updates_per_eval = 1
cfg.system.num_updates = updates
central = system.endswith("mappo")
mod = {"ff_mappo": ff_mappo, "ff_ippo": ff_ippo, "rec_mappo": rec_mappo, "rec_ippo": rec_ippo}[system]
env, _ = envs.make(cfg, add_global_state=central, device=dev)
learn, _, state = mod.learner_setup(env, (42, 43, 44), cfg, device=dev)
L = learn.learner
t0 = time.perf_counter()
first = last = None
for i in range(updates):
    L.update(0)
    if i in (0, updates - 1):
        torch.cuda.synchronize()
        m = L.train_metrics[0].mean((0, 1)).cpu().tolist()
        first, last = (m if first is None else first), m
torch.cuda.synchronize()
ok = all(bool(torch.isfinite(t).all()) for t in (L.p, L.m, L.v, L.train_metrics))
print(f"{system} {head}: {updates} updates in {time.perf_counter() - t0:.1f} s, finite={ok}, "
      f"[total, value, actor, entropy] first {['%.4f' % x for x in first]} last {['%.4f' % x for x in last]}", flush=True)
sys.exit(0 if ok else 1)
