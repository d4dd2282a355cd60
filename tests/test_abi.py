"""CPU: the C-ABI library loads and exports every symbol include/mava_hip.h declares (no compute)."""
import os
import re

from mava_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mava_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mava_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mava_hip.h but not exported by libmavahip.so"
    assert lib.mava_abi_version() == 3


def test_python_binding_matches_header():
    assert _header_symbols() == _lib.declared_symbols()


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.lib()
    # rejected on the host before any launch
    rc = lib.mava_gae_f32(None, None, None, None, None, None, -1, 4, 0.99, 0.95, None, None, None)
    assert rc <= -1000
    assert b"negative shape" in lib.mava_last_error()
    assert lib.mava_mlp_param_count(70, 5) == 26245 and lib.mava_mlp_param_count(264, 1) == 50561
    rc = lib.mava_mlp_forward_f32(None, None, 70, 99, None, 1, 8, None, None)
    assert rc <= -1000


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch

    from mava_amd import ops

    r = torch.zeros(4, 8)
    with pytest.raises(_lib.MavaHipError):
        ops.gae(r, r.clone(), torch.zeros(4, 8, dtype=torch.uint8), torch.zeros(8), 0.99, 0.95)


def test_comm_abi_argument_errors():
    """mava_comm_* (csrc/comm.cpp): argument checks run before RCCL is touched."""
    import ctypes as C

    lib = _lib.lib()
    assert lib.mava_comm_create(None, 0, 1, None) <= -1000
    h = C.c_void_p()
    idb = (C.c_uint8 * 128)()
    assert lib.mava_comm_create(C.byref(h), 3, 2, idb) <= -1000 and b"rank 3 of 2" in lib.mava_last_error()
    assert lib.mava_allreduce_sum_f32(None, None, 4, None) <= -1000
    assert lib.mava_comm_destroy(None) == 0


def test_permutation_abi_argument_errors():
    """mava_permutation_i32: range and pointer checks come before the launch."""
    lib = _lib.lib()
    assert lib.mava_permutation_i32(0, 1, 0, None, None) <= -1000 and b"n=0" in lib.mava_last_error()
    assert lib.mava_permutation_i32(1 << 31, 1, 0, None, None) <= -1000
    assert lib.mava_permutation_i32(8, 1, 0, None, None) <= -1000 and b"null pointer" in lib.mava_last_error()


def test_context_handles_are_independent():
    """mava_ctx_*: settings live in handles, not in the process - two handles (two learners) keep their own values, and a
    NULL handle means the defaults."""
    a, b = _lib.Ctx("f16x2"), _lib.Ctx("f32", critic_aggregation=False)
    assert a.matmul_mode == "f16x2" and b.matmul_mode == "f32"
    assert a.get(a.CRITIC_AGGREGATION) == 1 and b.get(b.CRITIC_AGGREGATION) == 0
    b.set(b.GAE_VARIANT, 13)
    assert a.get(a.GAE_VARIANT) == 0 and b.get(b.GAE_VARIANT) == 13 and a.h2_launches == 0
    lib = _lib.lib()
    assert lib.mava_ctx_set(a.handle, 99, 0) <= -1000 and b"unknown key" in lib.mava_last_error()
    assert lib.mava_ctx_set(a.handle, a.MATMUL_MODE, 7) <= -1000
    assert lib.mava_ctx_set(None, 0, 0) <= -1000 and lib.mava_ctx_destroy(None) == 0
    a.close()
    b.close()
    assert a.handle is None
    for name in dir(lib):  # no process-wide setters are left in the ABI
        assert "set_variant" not in name and "set_matmul" not in name
    for gone in ("mava_ppo_set_matmul_mode", "mava_gae_set_variant", "mava_policy_set_variant", "mava_ppo_set_critic_aggregation"):
        assert not hasattr(lib, gone), gone
