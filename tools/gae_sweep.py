"""Times the GAE kernel variants (mava_gae_set_variant) at the BASELINE config-2 shape with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mava_amd import ops
from mava_amd._lib import lib

T = 128
dev = torch.device("cuda", 0)
for N in (16384, 65536, 262144):
  r = torch.randn(T, N, device=dev); v = torch.randn(T, N, device=dev)
  d = (torch.rand(T, N, device=dev) < 1 / 500).to(torch.uint8); lv = torch.randn(N, device=dev)
  adv = torch.empty_like(r); tgt = torch.empty_like(r)
  bytes_ = 17 * T * N + 4 * N
  print(f"--- N={N}: {bytes_/1e6:.1f} MB")
  for variant in (1, 11, 21, 22, 24, 41, 42, 43, 45, 46, 47):
    lib().mava_gae_set_variant(variant)
    for _ in range(5):
        ops.gae(r, v, d, lv, 0.99, 0.95, out=(adv, tgt))
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gae(r, v, d, lv, 0.99, 0.95, out=(adv, tgt)); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    # back-to-back launches: amortises the event/launch overhead of a single short kernel
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.gae(r, v, d, lv, 0.99, 0.95, out=(adv, tgt))
    b.record(); torch.cuda.synchronize()
    b2b = a.elapsed_time(b) * 1e3 / 20
    print(f"variant {variant:2d}: min {ts[0]:6.1f} us  median {ts[len(ts)//2]:6.1f} us  -> {bytes_/ts[len(ts)//2]/1e3:7.1f} GB/s (median); back-to-back {b2b:6.1f} us -> {bytes_/b2b/1e3:7.1f} GB/s")
lib().mava_gae_set_variant(0)
