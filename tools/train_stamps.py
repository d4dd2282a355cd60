"""Diagnostic: per-phase cycle shares of the fused PPO train kernels (needs a library built with
MAVA_HIPCC_EXTRA="-DMAVA_STAMPS").  Prints, per wave of block 0, the cycles spent in each phase."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd import ops
from mava_amd._lib import Ctx, lib

dev = torch.device("cuda", 0)
MODE = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f16x2") else 0
TE, A, O, nA = 128 * 4096, 4, 66, 5
Rb = TE // 2
rng = np.random.default_rng(0)
names = ["P1 layer1", "P2 layer2+head", "P3 loss+dz2", "P4 dh1+sweeps", "P4b dz1 write", "P5a gW2", "P5b gW1", "barrier E",
         "commit+F", "loop top", "P2a prefetch issue", "P2b mfma loop", "P1a mfma loop", "barrier A", "x issue", "row issue"]
# note: with sub-stamps, "P1 layer1" = epilogue after P1a, "P2 layer2+head" = head part after P2b
if MODE == 1:  # phases of ppo_train_h2.hip
    names = ["P1 mfma", "P1 relu+image", "gather issue", "barrier A", "P2 mfma", "P2 image+head", "barrier B", "P3 loss",
             "B2+dz2+image", "barrier C", "P4 mfma+dz1+commit", "gW3+gW2", "barrier D", "gW1 (wide chain: gW2)", "loop top", "barrier B2"]
# phases of ppo_train_w8.hip (the eight-wave actor kernel, default for the discrete actor; `w4` as a second argument keeps
# the four-wave kernel)
names_w8 = ["P1 mfma", "P1 relu+image", "gather issue", "barrier A", "P2 mfma", "P2 image+head", "barrier B", "P3 loss | x commit",
            "barrier B2", "gW3+dz2+image", "gW2", "barrier C", "dh1+dz1 image", "gW1", "loop top", "-"]
l = lib()
ctx = Ctx("f16x2" if MODE == 1 else "f32")
if len(sys.argv) > 2 and sys.argv[2] == "w4":
    ctx.set(ctx.TRAIN_VARIANT, 1)
l.mava_debug_set_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(128, dtype=torch.int64, device=dev)  # waves 0-3 (chain / only group), 4-7 (loader group of wide launches)
l.mava_debug_set_stamps(stamps.data_ptr())
perm = torch.randperm(TE, device=dev).to(torch.int32)
for which in ("critic", "actor"):
    if which == "critic":
        din = A * O
        params = torch.randn(ops.mlp_param_count(din, 1), device=dev) * 0.05
        x = torch.randn(TE, din, device=dev)
        ov, tg = torch.randn(TE * A, device=dev), torch.randn(TE * A, device=dev)
        slab = torch.zeros(256, params.numel() + 2, device=dev)
        run = lambda: ops.ppo_critic_grad(params, x, A, ov, tg, perm, 0, Rb, A, 0.2, 0.5, slab, ctx=ctx)
    else:
        din = A + O
        params = torch.randn(ops.mlp_param_count(din, nA), device=dev) * 0.05
        x = torch.randn(TE * A, din, device=dev)
        mask = (torch.rand(TE * A, nA, device=dev) > 0.2).to(torch.uint8); mask[:, 0] = 1
        act = torch.zeros(TE * A, dtype=torch.int32, device=dev)
        olp, adv = -torch.rand(TE * A, device=dev) - 1.0, torch.randn(TE * A, device=dev)
        st = ops.adv_stats(adv, perm, 0, Rb, A)
        slab = torch.zeros(256, params.numel() + 2, device=dev)
        run = lambda: ops.ppo_actor_grad(params, x, mask, act, olp, adv, st, perm, 0, Rb, A, nA, 0.2, 0.01, slab, ctx=ctx)
    w8_before = ctx.get(ctx.W8_LAUNCHES)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    s8 = stamps.cpu().numpy().reshape(8, 16)
    if ctx.get(ctx.W8_LAUNCHES) > w8_before:  # eight equal waves: print waves 0-3 and wave 4 (the SIMD partner of wave 0)
        tot = s8.sum(1)
        ntile = -(-(Rb * A) // 32 // 256)
        print(f"== {which} (eight-wave kernel): launch {a.elapsed_time(b):.3f} ms, block 0 cycles per wave {tot.tolist()}, {ntile} tiles per block")
        for i, n in enumerate(names_w8):
            print(f"   {n:24s} " + "  ".join(f"{s8[w, i] / ntile:8.0f}" for w in (0, 1, 4, 5)) + f"   ({100 * s8[0, i] / tot[0]:5.1f} %)")
        stamps.zero_()
        continue
    s = s8[:4]
    tot = s.sum(1)
    # row tiles per block: agent rows for the actor; (t,e) rows for the critic when the agents of a row are aggregated
    rows = Rb * A if which == "actor" else Rb
    ntile = -(-rows // 32 // 256)
    print(f"== {which}: launch {a.elapsed_time(b):.3f} ms, block 0 cycles per wave {tot.tolist()}, {ntile} tiles per block")
    for i, n in enumerate(names):
        print(f"   {n:24s} " + "  ".join(f"{s[w, i] / ntile:8.0f}" for w in range(4)) + f"   ({100 * s[0, i] / tot[0]:5.1f} %)"
              + (f"   loader wave 4: {s8[4, i] / ntile:8.0f}" if s8[4].sum() > 0 else ""))
    stamps.zero_()
