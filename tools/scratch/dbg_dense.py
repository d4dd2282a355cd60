import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mava_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device("cuda", 0)
def to_t32(a):
    rows, N = a.shape
    return a.reshape(rows // 32, 32, N).transpose(0, 2, 1).reshape(-1).copy()
def from_t32(flat, rows, N):
    return np.asarray(flat).reshape(rows // 32, N, 32).transpose(0, 2, 1).reshape(rows, N)
lib().mava_ppo_set_matmul_mode(1)
for K in (5, 13, 16, 20, 32):
    N = 128
    rng = np.random.default_rng(K + N)
    rows = 96
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y = torch.zeros(rows * N, device=dev)
    xt, wt, bt = torch.from_numpy(to_t32(x)).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    check(lib().mava_rec_dense_f32(ptr(xt), 0, None, 0, 0, 0, 1, K, 0, ptr(wt), N, ptr(bt), None, ptr(y), 0, K, N, rows, 0, stream_ptr()), "dense")
    torch.cuda.synchronize()
    want = x.astype(np.float64) @ w.astype(np.float64) + b
    got = from_t32(y.cpu().numpy(), rows, N)
    err = np.abs(got - want)
    bad = err > 1e-5
    print("K", K, "bad", bad.sum(), "max err", err.max(), "rows with bad", np.unique(np.where(bad)[0])[:40], "cols", np.unique(np.where(bad)[1])[:40])
    # which input feature explains the error?  regress err on x columns
    if bad.any():
        r, c = np.unravel_index(err.argmax(), err.shape)
        d = got[r, c] - want[r, c]
        print("   worst", r, c, d, " x row", x[r], " w col", w[:, c])
