"""In-tree build of libmavahip.so (hipcc, gfx950 only).

`python -m mava_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
Objects are rebuilt only when a source or header is newer than the object.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
OUT_LIB = os.path.join(ROOT, "libmavahip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_SOURCES = [
    "gae.hip",
    "permutation.hip",
    "adam.hip",
    "mlp_policy.hip",
    "mlp_coop.hip",
    "ppo_train.hip",
    "ppo_train_h2.hip",
    "ppo_train_w8.hip",
    "rollout_h2.hip",
    "synth_rware.hip",
    "rec_dense.hip",
    "rec_dense_h2.hip",
    "rec_gru.hip",
    "rec_gru_h2.hip",
    "rec_step.hip",
    "rec_step_h2.hip",
    "rec_out_h2.hip",
    "generic_layers.hip",
]
CPP_SOURCES = ["api.cpp", "comm.cpp"]

COMMON_FLAGS = ["-O3", "-fPIC", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "..", "include")]
HIP_FLAGS = [f"--offload-arch={ARCH}", "-ffp-contract=fast", "-Wno-unused-result"]
# developer knobs: MAVA_HIPCC_EXTRA="-DMAVA_FAST_BUILD -DMAVA_STAMPS" builds a diagnostic library
EXTRA_FLAGS = os.environ.get("MAVA_HIPCC_EXTRA", "").split()
FLAG_STAMP = os.path.join(CSRC, ".build_flags")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "..", "include", "mava_hip.h"))
    return [h for h in hs if os.path.exists(h)]


def _flags_changed() -> bool:
    cur = " ".join(EXTRA_FLAGS)
    old = open(FLAG_STAMP).read() if os.path.exists(FLAG_STAMP) else ""
    return cur != old


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj) or _flags_changed():
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.abspath(__file__)] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, verbose: bool) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    if not _stale(path, obj):
        return obj
    cmd = [HIPCC, "-c", path, "-o", obj] + COMMON_FLAGS
    if src.endswith(".hip"):
        cmd += HIP_FLAGS + EXTRA_FLAGS
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(verbose: bool = False, jobs: int = 4) -> str:
    srcs = HIP_SOURCES + CPP_SOURCES
    missing = [s for s in srcs if not os.path.exists(os.path.join(CSRC, s))]
    if missing:
        raise RuntimeError(f"missing sources under {CSRC}: {missing}")
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    flags_changed = _flags_changed()
    with open(FLAG_STAMP, "w") as f:
        f.write(" ".join(EXTRA_FLAGS))
    need_link = flags_changed or (not os.path.exists(OUT_LIB)) or any(
        os.path.getmtime(o) > os.path.getmtime(OUT_LIB) for o in objs
    )
    if need_link:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT_LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT_LIB


if __name__ == "__main__":
    print(build(verbose=True))
