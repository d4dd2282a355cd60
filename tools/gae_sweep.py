"""Times the GAE kernel variants (MAVA_CTX_GAE_VARIANT of a context handle) with HIP events around a captured HIP graph of
back-to-back launches, so the host launch path (Python + ctypes, ~10 us) is not what is measured.

 warm: every launch re-reads the same 35.7 MB (served by L2 / the 256 MB infinity cache)
 cold: launches rotate over enough buffer sets (> 512 MB) that every launch streams from HBM, as in the
       training loop, where reward/value/done were written over 128 rollout steps with ~2.8 GB of
       observation traffic in between.
A device-to-device copy moving the same number of bytes is timed the same way as the ceiling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mava_amd import ops
from mava_amd._lib import Ctx, lib

T = 128
REPS = 32
dev = torch.device("cuda", 0)
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 11, 21, 24, 41, 42, 43, 45, 46, 47]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16384, 65536]


def graph_time(fn, n_sets):
    """fn(i) launches on the current stream using buffer set i; returns us per launch."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(n_sets):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(REPS):
                fn(k % n_sets)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side); g.replay(); b.record(side); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / REPS)
    return best


for N in sizes:
    bytes_ = 17 * T * N + 4 * N
    n_cold = max(2, int(600e6 // bytes_) + 1)
    sets = []
    for i in range(n_cold):
        r = torch.randn(T, N, device=dev); v = torch.randn(T, N, device=dev)
        d = (torch.rand(T, N, device=dev) < 1 / 500).to(torch.uint8); lv = torch.randn(N, device=dev)
        sets.append((r, v, d, lv, torch.empty_like(r), torch.empty_like(r)))
    srcs = [torch.randn(bytes_ // 8, device=dev) for _ in range(n_cold)]
    dsts = [torch.empty_like(s) for s in srcs]
    print(f"--- N={N}: {bytes_/1e6:.1f} MB algorithmic per launch; cold = {n_cold} buffer sets")
    for mode, ns in (("warm", 1), ("cold", n_cold)):
        t = graph_time(lambda i: dsts[i].copy_(srcs[i]), ns)
        print(f"copy of equal traffic [{mode}]: {t:6.2f} us -> {bytes_/t/1e3:7.1f} GB/s")
    for variant in variants:
        ctx = Ctx()
        ctx.set(ctx.GAE_VARIANT, variant)
        def run(i):
            r, v, d, lv, adv, tgt = sets[i]
            ops.gae(r, v, d, lv, 0.99, 0.95, out=(adv, tgt), ctx=ctx)
        tw = graph_time(run, 1)
        tc = graph_time(run, n_cold)
        print(f"variant {variant:2d}: warm {tw:6.2f} us -> {bytes_/tw/1e3:7.1f} GB/s   cold {tc:6.2f} us -> {bytes_/tc/1e3:7.1f} GB/s ({bytes_/tc/1e3/80:.0f} % of 8 TB/s)")
    del sets, srcs, dsts
