"""Per-launch time of the recurrent path's dense / X^T Y / scan kernels at the config-4 actor shapes (1 M row-steps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mava_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device("cuda", 0)
L = lib()
L.mava_ppo_set_matmul_mode(int(os.environ.get("MODE", "1")))
rows = 128 * 8192
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K, N, relu, gated, name) in ((128, 384, 0, 0, "gi"), (128, 128, 1, 0, "post"), (384, 128, 0, 1, "dxpre"), (13, 128, 0, 1, "dpost"),
                                  (160, 128, 1, 0, "pre"), (128, 13, 0, 0, "head")):
    x = torch.randn(rows * K, device=dev); w = torch.randn(K * N, device=dev) * 0.1; b = torch.randn(N, device=dev)
    g = torch.randn(rows * N, device=dev) if gated else None
    y = torch.empty(rows * N, device=dev)
    us = t(lambda: check(L.mava_rec_dense_f32(ptr(x), 0, None, 0, 0, 0, 1, K, 0, ptr(w), N, ptr(b), ptr(g), ptr(y), 0, K, N, rows, relu, stream_ptr()), "d"))
    by = rows * 4 * (K + N + (N if gated else 0))
    print(f"dense {name:6s} K={K:3d} N={N:3d}: {us:7.1f} us  {by / us / 1e6:6.2f} TB/s")
    del x, y, g
slab = torch.zeros((256, 128 * 384 + 384 + 8), device=dev)
for (K, N, name) in ((128, 384, "Wi/Wh"), (128, 128, "Wpost"), (128, 13, "Whead"), (160, 128, "Wpre")):
    x = torch.randn(rows * K, device=dev); y = torch.randn(rows * N, device=dev)
    us = t(lambda: check(L.mava_rec_xty_f32(ptr(x), 0, None, 0, 0, 0, 1, K, ptr(y), 0, K, N, rows, 1, 1.0, ptr(slab), slab.shape[1], 256, stream_ptr()), "x"))
    by = rows * 4 * (K + N)
    print(f"xty   {name:6s} K={K:3d} N={N:3d}: {us:7.1f} us  {by / us / 1e6:6.2f} TB/s")
    del x, y
