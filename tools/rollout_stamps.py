"""Diagnostic: per-phase cycles of the fused rollout kernel (library built by tools/build_stamps.sh, MAVA_LIB_PATH).
Prints cycles per env step for wave 0 of the actor role and of the critic role of block 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mava_amd import envs
from mava_amd._lib import lib
from mava_amd.config import compose
from mava_amd.systems.ppo import ff_mappo

dev = torch.device("cuda", 0)
cfg = compose("default_ff_mappo", ["arch.num_envs=4096", "system.update_batch_size=1", "env/scenario=tiny-4ag"])
cfg.system.num_updates_per_eval = 3
env, _ = envs.make(cfg, add_global_state=True, device=dev)
learn, _, state = ff_mappo.learner_setup(env, (42, 7, 8), cfg, device=dev)
L = learn.learner
L.graph_rollout = False
for n in range(2):
    L._rollout(n)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
f = lib().mava_debug_get_rollout_stamps
f.argtypes = [C.c_void_p]
assert f(out) == 0
T = L.T
names = ["P1 layer 1 + image", "barrier 1", "P2 layer 2 + head", "barrier 2", "S sample / values", "barrier 3", "E env step", "barrier 4"]
tot = [sum(out[0:8]), sum(out[8:16])]
print(f"cycles per env step: actor role {tot[0] / T:.0f}, critic role {tot[1] / T:.0f} (T = {T})")
for i, n in enumerate(names):
    print(f"   {n:22s} actor role {out[i] / T:8.0f}   critic role {out[8 + i] / T:8.0f}")
