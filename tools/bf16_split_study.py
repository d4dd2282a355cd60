"""CPU study for DESIGN §9 item 2: accuracy of f32 matrix products rebuilt from bf16 MFMAs (f32 accumulation).

Every f32 operand is split into bf16 terms (hi + mid + lo, round-to-nearest); bf16 x bf16 products are exact in f32, so
emulating the matrix pipe needs only f32 matmuls of bf16-valued matrices.  Compared on the ff_mappo actor network's
forward and backward products at the BASELINE config-2 shapes against float64:

  f32        : plain f32 matmul (what v_mfma_f32_32x32x2_f32 computes today)
  bf16x3     : hi*hi + hi*mid + mid*hi                     (3 bf16 products per f32 product)
  bf16x6     : + mid*mid + hi*lo + lo*hi                   (6 products)

Run: python tools/bf16_split_study.py   (no GPU, no library)
"""
import torch

torch.manual_seed(0)


def split(a, n):
    terms, r = [], a.clone()
    for _ in range(n):
        t = r.to(torch.bfloat16).to(torch.float32)
        terms.append(t)
        r = r - t
    return terms


def mm(a, b, mode):
    if mode == "f32":
        return a @ b
    n = 2 if mode == "bf16x3" else 3
    A, B = split(a, n), split(b, n)
    pairs = [(0, 0), (0, 1), (1, 0)] if mode == "bf16x3" else [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    out = torch.zeros(a.shape[0], b.shape[1])
    for i, j in reversed(pairs):  # small terms first
        out += A[i] @ B[j]
    return out


def rel(x, ref):
    ref = ref.double()
    return float((x.double() - ref).abs().max() / ref.pow(2).mean().sqrt())


R, din, H, no = 8192, 70, 128, 5
x = (torch.rand(R, din) < 0.2).float() + 0.0
x[:, :6] = torch.rand(R, 6) * 9
W1, W2, W3 = torch.randn(din, H) * 0.17, torch.randn(H, H) * 0.12, torch.randn(H, no) * 0.1
dy = torch.randn(R, no) / R
ref = {}
for mode in ("f64", "f32", "bf16x3", "bf16x6"):
    if mode == "f64":
        f = lambda a, b, m=None: a.double() @ b.double()
    else:
        f = lambda a, b, m=mode: mm(a.float(), b.float(), m)
    h1 = torch.relu(f(x, W1))
    h2 = torch.relu(f(h1, W2))
    y = f(h2, W3)
    dz2 = f(dy, W3.t()) * (h2 > 0)
    gW2 = f(h1.t(), dz2)
    dz1 = f(dz2, W2.t()) * (h1 > 0)
    gW1 = f(x.t(), dz1)
    out = {"logits": y, "gW2": gW2, "gW1": gW1}
    if mode == "f64":
        ref = out
        continue
    print(f"{mode:7s} " + "  ".join(f"{k}: {rel(v, ref[k]):.2e}" for k, v in out.items()))
