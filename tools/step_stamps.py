import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mava_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device("cuda", 0)
L = lib()
L.mava_debug_set_step_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(12, dtype=torch.int64, device=dev)
L.mava_debug_set_step_stamps(stamps.data_ptr())
E, A, Oa, Oc, nA = 2048, 8, 155, 188, 13
EA = E * A
def nparams(din, no): return din * 128 + 128 + 128 * 384 + 384 + 128 * 384 + 128 + 128 * 128 + 128 + 128 * no + no
pa, pc = torch.randn(nparams(Oa, nA), device=dev) * 0.05, torch.randn(nparams(Oc, 1), device=dev) * 0.05
pka = torch.empty(L.mava_rec_step_pack_bytes(Oa), dtype=torch.uint8, device=dev)
pkc = torch.empty(L.mava_rec_step_pack_bytes(Oc), dtype=torch.uint8, device=dev)
check(L.mava_rec_step_pack_f32(ptr(pa), Oa, ptr(pka), stream_ptr()), "pack"); check(L.mava_rec_step_pack_f32(ptr(pc), Oc, ptr(pkc), stream_ptr()), "pack")
av, gs = torch.randn((EA, Oa), device=dev), torch.randn((E, Oc), device=dev)
mask = torch.ones((EA, nA), dtype=torch.uint8, device=dev); done = torch.zeros(EA, dtype=torch.uint8, device=dev)
ha, ha2 = torch.zeros(EA * 128, device=dev), torch.zeros(EA * 128, device=dev)
hc, hc2 = torch.zeros(EA * 128, device=dev), torch.zeros(EA * 128, device=dev)
act, lp, val = torch.zeros(EA, dtype=torch.int32, device=dev), torch.zeros(EA, device=dev), torch.zeros(EA, device=dev)
def run():
    check(L.mava_rec_step_packed_f32(ptr(pka), ptr(pkc), ptr(pa), Oa, nA, 1e-3, ptr(av), ptr(mask), ptr(done), ptr(ha), ptr(ha2), EA, 42, 0, 0, 0,
                                     ptr(act), None, ptr(lp), ptr(pc), Oc, ptr(gs), 1, ptr(done), A, ptr(hc), ptr(hc2), E, A, ptr(val), stream_ptr()), "step")
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
b.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy()
print(f"launch {a.elapsed_time(b) / 10 * 1e3:.1f} us; cycles of actor group 0 by phase (total {s.sum()}):")
for n, v in zip(["frag loads + staging x/h", "barrier 1", "pre MFMA", "barrier + E image", "barrier", "GRU MFMA", "barrier + gates + H2 image", "barrier", "post + head partial", "barrier", "epilogue"], s):
    print(f"  {n:30s} {v:8d}  ({100 * v / s.sum():5.1f} %)")
