import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device("cuda", 0)
L = lib()
L.mava_debug_set_out_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(8, dtype=torch.int64, device=dev)
L.mava_debug_set_out_stamps(stamps.data_ptr())
T, Rm, E, A, no = 128, 8192, 2048, 8, 13
rows = T * Rm
hs = torch.randn(rows * 128, device=dev) * 0.5
params = torch.randn(128 * 128 + 128 + 128 * no + no, device=dev) * 0.05
idx = torch.randperm(E, device=dev)[: Rm // A].to(torch.int32).contiguous()
mask = torch.ones((T, E, A, no), dtype=torch.uint8, device=dev)
action = torch.randint(0, no, (T, E, A), dtype=torch.int32, device=dev)
f0 = -torch.rand((T, E, A), device=dev) - 1
f1 = torch.randn((T, E, A), device=dev)
stats = torch.zeros((128, 2), dtype=torch.float64, device=dev); stats[0, 1] = rows
dh = torch.empty(rows * 128, device=dev)
slab = torch.zeros((256, 128 * 128 + 128 + 128 * no + no + 2), device=dev)
def run():
    check(L.mava_rec_out_f32(T, Rm, E, A, no, 1, ptr(idx), ptr(hs), ptr(params), ptr(mask), ptr(action), ptr(f0), ptr(f1), ptr(stats), 128,
                             0.2, 0.01, float(2 ** 20), 1, ptr(dh), ptr(slab), slab.shape[1], 256, stream_ptr()), "out")
for _ in range(2): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy(); nt = rows // 32 // 256
print(f"launch {a.elapsed_time(b):.3f} ms, {nt} tiles per block, cycles per tile by phase:")
for n, v in zip(["P1 post+image+head", "barrier B", "P3 loss", "B2+P4 gW3/dpost/image", "barrier C", "P5 dh+store+gWp+commit", "barrier D", "loop top (rotate+row loads)"], s):
    print(f"  {n:30s} {v / nt:8.0f}  ({100 * v / s.sum():5.1f} %)")
