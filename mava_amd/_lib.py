"""ctypes binding of libmavahip.so (include/mava_hip.h).

The product path has NO fallback: if the HIP library is missing or an entry point fails, an
exception is raised.  torch is imported first so the HIP runtime already resident in the process
(torch/lib/libamdhip64.so, soname libamdhip64.so.7) is the one the library binds to.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must precede loading libmavahip.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAVA_LIB_PATH: a diagnostic build of the same library (e.g. -DMAVA_STAMPS, tools/train_stamps.py); never a fallback
LIB_PATH = os.environ.get("MAVA_LIB_PATH") or os.path.join(_HERE, "libmavahip.so")

_lib: Optional[C.CDLL] = None


class MavaHipError(RuntimeError):
    pass


vp = C.c_void_p
i32 = C.c_int
u32 = C.c_uint32
u64 = C.c_uint64
f32 = C.c_float
lng = C.c_long

# name -> argtypes (all return int unless listed in _RESTYPES)
_SIGNATURES = {
    "mava_abi_version": [],
    "mava_ctx_create": [vp],
    "mava_ctx_destroy": [vp],
    "mava_ctx_set": [vp, i32, lng],
    "mava_ctx_get": [vp, i32, vp],
    "mava_gae_f32": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp],
    "mava_permutation_i32": [lng, u64, u64, vp, vp],
    "mava_clip_adam": [vp, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(f32), i32, f32, f32, i32, i32, i32,
                       f32, f32, f32, vp, f32, f32, vp, vp],
    "mava_slab_reduce_f32": [vp, i32, lng, i32, i32, vp, vp],
    "mava_ppo_finish_workspace_bytes": [i32, i32],
    "mava_ppo_finish_f32": [vp, vp, lng, vp, lng, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, i32, i32, i32, f32, f32, f32,
                            f32, f32, vp, i32, vp, C.c_size_t, vp],
    "mava_slab_reduce2_f32": [vp, i32, lng, i32, vp, i32, vp, i32, vp],
    "mava_mlp_param_count": [i32, i32],
    "mava_mlp_forward_f32": [vp, vp, i32, i32, vp, i32, i32, vp, vp],
    "mava_policy_step_f32": [vp, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, u64, u32, vp, u32, i32,
                             vp, vp, vp, vp, vp, vp],
    "mava_policy_step_continuous_f32": [vp, vp, i32, i32, f32, vp, vp, i32, vp, i32, i32, i32, i32, u64, u32, vp, u32, i32, vp, vp, vp, vp, vp, vp],
    "mava_ppo_actor_grad_continuous_f32": [vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, lng, i32, i32, f32, f32, u64, u32, u32, vp, lng, i32, vp],
    "mava_seq_actor_loss_continuous_f32": [i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, u64, u32, u32, f32, vp, vp, vp, vp, i32, vp],
    "mava_seq_sample_continuous_f32": [i32, i32, f32, vp, vp, vp, u64, u32, u32, i32, vp, vp, vp],
    "mava_adv_stats_blocks": [],
    "mava_adv_stats_f64": [vp, vp, lng, i32, i32, vp, vp],
    "mava_adv_stats_batched_f64": [vp, vp, lng, i32, i32, i32, vp, vp],
    "mava_ppo_actor_grad_f32": [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, lng, i32, i32, f32, f32, vp, lng, i32,
                                vp],
    "mava_ppo_critic_grad_f32": [vp, vp, i32, vp, i32, vp, vp, vp, lng, i32, i32, f32, f32, vp, lng, i32, vp],
    "mava_synth_rware_step": [i32, i32, i32, i32, i32, i32, i32, u64, u32, vp, u32, i32] + [vp] * 14 + [vp, i32, vp],
    "mava_rollout_ff_f32": [vp, i32, vp, i32, i32, i32, i32, i32, i32, u64, u64, u32, u32, u32, i32] + [vp] * 18 + [vp, vp, f32, f32, vp],
    "mava_rec_dense_f32": [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mava_rec_xty_f32": [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp, lng, i32, vp],
    "mava_rec_gather_t32_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "mava_gru_scan_fwd_f32": [vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "mava_t32_gru_mask_f32": [vp, vp, vp, i32, i32, i32, lng, vp, vp],
    "mava_t32_gru_gates_f32": [vp, vp, vp, vp, i32, lng, vp, vp, vp, vp, vp, i32, i32, vp],
    "mava_t32_gru_gates_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, lng, vp, vp, vp, vp],
    "mava_gru_scan_bwd_f32": [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    "mava_seq_actor_loss_f32": [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, i32, vp],
    "mava_rec_step_continuous_f32": [vp, i32, i32, f32, vp, vp, vp, vp, i32, u64, u32, u32, i32, vp, vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, vp, vp],
    "mava_rec_step_f32": [vp, i32, i32, vp, vp, vp, vp, vp, i32, u64, u32, u32, i32, vp, vp, vp, i32, vp, i32, vp, i32, vp, vp,
                          i32, i32, vp, vp],
    "mava_seq_critic_loss_f32": [i32, i32, i32, i32, i32, vp, vp, vp, vp, f32, f32, f32, vp, vp, i32, vp],
    "mava_rec_out_f32": [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, vp, vp, lng, i32, vp],
    "mava_rec_step_pack_bytes": [i32],
    "mava_t32_norm_act_f32": [vp, i32, lng, i32, vp, i32, vp, vp, vp, vp],
    "mava_t32_norm_act_bwd_f32": [vp, vp, i32, lng, i32, vp, vp, i32, vp, vp, vp],
    "mava_t32_colsum_f32": [vp, i32, lng, f32, vp, lng, i32, vp],
    "mava_t32_im2col_f32": [vp, i32, lng, i32, i32, i32, i32, i32, vp, vp],
    "mava_t32_col2im_f32": [vp, i32, lng, i32, i32, i32, i32, i32, vp, vp],
    "mava_t32_flatten_f32": [vp, lng, i32, i32, i32, vp, vp],
    "mava_comm_unique_id": [vp],
    "mava_comm_create": [vp, i32, i32, vp],
    "mava_allreduce_sum_f32": [vp, vp, C.c_size_t, vp],
    "mava_broadcast_f32": [vp, vp, C.c_size_t, i32, vp],
    "mava_comm_destroy": [vp],
    "mava_rec_step_pack_f32": [vp, i32, vp, vp],
    "mava_rec_step_packed_f32": [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, i32, u64, u32, u32, i32, vp, vp, vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, vp, vp],
    "mava_seq_sample_f32": [i32, i32, vp, vp, u64, u32, u32, i32, vp, vp, vp],
    "mava_t32_convert_f32": [vp, i32, i32, i32, vp, vp],
}
_RESTYPES = {"mava_last_error": C.c_char_p, "mava_rec_step_pack_bytes": C.c_long, "mava_ppo_finish_workspace_bytes": C.c_size_t}


def declared_symbols():
    """Every symbol include/mava_hip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(set(_SIGNATURES) | set(_RESTYPES))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MavaHipError(
                f"{LIB_PATH} is missing: build it with `python -m mava_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.mava_last_error.restype = C.c_char_p
        _lib.mava_last_error.argtypes = []
        for name, args in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
    return _lib


class Ctx:
    """Context handle of the C ABI (include/mava_hip.h, mava_ctx_*): the arithmetic mode, the critic aggregation, the
    bench-only kernel variants and the library-owned workspaces of ONE learner.  The library has no process-wide
    settings: two Ctx objects (two learners in one process) share nothing.  `handle` is what the entry points take as
    their first argument (None = the library defaults: exact f32)."""

    MATMUL_MODE, CRITIC_AGGREGATION, GAE_VARIANT, POLICY_VARIANT, H2_LAUNCHES, TRAIN_VARIANT, W8_LAUNCHES, W1_SPLIT_FRESH = 0, 1, 2, 3, 4, 5, 6, 7

    def __init__(self, matmul_mode: str = "f32", critic_aggregation: bool = True):
        if matmul_mode not in ("f32", "f16x2"):
            raise ValueError(f"matmul_mode must be 'f16x2' or 'f32', got {matmul_mode!r}")
        h = C.c_void_p()
        check(lib().mava_ctx_create(C.byref(h)), "mava_ctx_create")
        self.handle: Optional[int] = h.value
        self.set(self.MATMUL_MODE, 1 if matmul_mode == "f16x2" else 0)
        self.set(self.CRITIC_AGGREGATION, int(bool(critic_aggregation)))

    def set(self, key: int, value: int) -> None:
        check(lib().mava_ctx_set(self.handle, key, int(value)), "mava_ctx_set")

    def get(self, key: int) -> int:
        out = C.c_long()
        check(lib().mava_ctx_get(self.handle, key, C.byref(out)), "mava_ctx_get")
        return int(out.value)

    @property
    def matmul_mode(self) -> str:
        return "f16x2" if self.get(self.MATMUL_MODE) == 1 else "f32"

    @property
    def h2_launches(self) -> int:
        return self.get(self.H2_LAUNCHES)

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and _lib is not None:
            _lib.mava_ctx_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ctx_ptr(ctx: Optional["Ctx"]) -> Optional[int]:
    return None if ctx is None else ctx.handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mava_last_error().decode("utf-8", "replace")
        raise MavaHipError(f"{what} failed with code {rc}: {msg}")


# bench.py: when set to a dict, every launch() call is bracketed by a HIP event pair on the current stream
# (name -> [(start, end)]); None in production: launch() is then a plain call + check
TIMERS: Optional[dict] = None


def launch(what: str, fn, *args, ok=(0,)) -> int:
    """check(fn(*args), what), optionally timed with HIP events on the launch stream.  Return codes listed in `ok` are
    passed back to the caller instead of raising (1 = "shape not instantiated" of the optional fused kernels)."""
    if TIMERS is None:
        rc = fn(*args)
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        TIMERS.setdefault(what, []).append((a, b))
    if rc not in ok:
        check(rc, what)
    return rc


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
