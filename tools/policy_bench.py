"""Device time of one acting step (mava_policy_step_f32) at the BASELINE config-2 shape, per kernel variant
(0 = default: hybrid launch when the critic has few tiles; 1 = per-wave kernel always; 2 = cooperative kernels),
timed through a captured HIP graph (no host launch path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mava_amd import ops
from mava_amd._lib import Ctx, lib

dev = torch.device("cuda", 0)
E, A, O, nA = 4096, 4, 66, 5
EA = E * A
pa = torch.randn(ops.mlp_param_count(A + O, nA), device=dev) * 0.05
pc = torch.randn(ops.mlp_param_count(A * O, 1), device=dev) * 0.05
av = torch.randn(EA, A + O, device=dev)
gs = torch.randn(E, A * O, device=dev)
mask = torch.ones(EA, nA, dtype=torch.uint8, device=dev)
out = (torch.empty(EA, dtype=torch.int32, device=dev), torch.empty(EA, device=dev), torch.empty(EA, device=dev))
REPS = 32


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(REPS):
                fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side); g.replay(); b.record(side); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / REPS)
    return best


for variant in (0, 1, 2):
    ctx = Ctx()
    ctx.set(ctx.POLICY_VARIANT, variant)
    for name, kw in (("critic per agent row", dict(critic_share=A, critic_rows=EA, value_broadcast=1)),
                     ("critic per env row", dict(critic_share=1, critic_rows=E, value_broadcast=A)),
                     ("actor only", dict(critic_share=1, critic_rows=0, value_broadcast=A))):
        o = (out[0], out[1], out[2][:0]) if kw["critic_rows"] == 0 else out
        t = graph_time(lambda: ops.policy_step(pa, pc, av, mask, gs, n_actions=nA, seed=1, step=3, out=o, ctx=ctx, **kw))
        print(f"variant {variant} {name:22s}: {t:6.1f} us")
    t = graph_time(lambda: ops.mlp_forward(pc, A * O, 1, gs, rows=E, x_share=1, ctx=ctx))
    print(f"variant {variant} value only (E rows)   : {t:6.1f} us")
