import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd._lib import Ctx, check, lib, ptr, stream_ptr
dev = torch.device("cuda", 0)
L = lib()
CTX = Ctx("f16x2")
L.mava_debug_set_scan_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(8, dtype=torch.int64, device=dev)
L.mava_debug_set_scan_stamps(stamps.data_ptr())
for Rm, A in ((8192, 8), (1024, 1)):
    T, E = 128, 2048
    rows = T * Rm
    gi = torch.randn(rows * 384, device=dev) * 0.5
    hs, hprev, saved = torch.empty(rows * 128, device=dev), torch.empty(rows * 128, device=dev), torch.empty(rows * 512, device=dev)
    wh = torch.randn(128 * 384, device=dev) * 0.1; bhn = torch.zeros(128, device=dev)
    idx = torch.randperm(E, device=dev)[: Rm // A].to(torch.int32).contiguous()
    done = (torch.rand((T, E, A), device=dev) < 0.02).to(torch.uint8)
    h0 = torch.zeros((E, A, 128), device=dev)
    run = lambda: check(L.mava_gru_scan_fwd_f32(CTX.handle, T, Rm, E, A, ptr(idx), ptr(done), ptr(h0), 0, ptr(wh), ptr(bhn), ptr(gi), ptr(hs), ptr(hprev), ptr(saved), stream_ptr()), "scan")
    for _ in range(2): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    print(f"Rm={Rm}: launch {a.elapsed_time(b):.3f} ms; cycles per step by phase:")
    for n, v in zip(["loop top", "acc init + prefetch issue", "barrier", "MFMA loop", "gates + stores", "put16"], s):
        print(f"  {n:28s} {v / T:8.0f}  ({100 * v / s.sum():5.1f} %)")
print("---- backward scan")
for Rm, A in ((8192, 8), (1024, 1)):
    T, E = 128, 2048
    rows = T * Rm
    saved, hprev, dh = torch.rand(rows * 512, device=dev), torch.randn(rows * 128, device=dev), torch.randn(rows * 128, device=dev)
    dgi, dgh = torch.empty(rows * 384, device=dev), torch.empty(rows * 384, device=dev)
    wh = torch.randn(128 * 384, device=dev) * 0.1
    idx = torch.randperm(E, device=dev)[: Rm // A].to(torch.int32).contiguous()
    done = (torch.rand((T, E, A), device=dev) < 0.02).to(torch.uint8)
    run = lambda: check(L.mava_gru_scan_bwd_f32(CTX.handle, T, Rm, E, A, ptr(idx), ptr(done), ptr(wh), ptr(saved), ptr(hprev), ptr(dh), ptr(dgi), ptr(dgh), 0, stream_ptr()), "scanb")
    for _ in range(2): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    print(f"Rm={Rm}: launch {a.elapsed_time(b):.3f} ms; cycles per step by phase:")
    for n, v in zip(["loop top", "gate grads + stores", "put16 x3", "load_step issue", "barrier", "MFMA loop + dhc"], s):
        print(f"  {n:28s} {v / T:8.0f}  ({100 * v / s.sum():5.1f} %)")
