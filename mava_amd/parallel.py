"""Data-parallel exchange step: the only cross-GPU traffic of the hot path.

Reference: mava/systems/ppo/ff_mappo.py:224-238 - four `jax.lax.pmean` calls per minibatch
((grads, loss_info) of actor and critic over the "batch" and "device" axes).  Here every rank owns
`update_batch_size * num_envs` environments end to end (no data-path collective during rollout,
GAE, shuffling, forward or backward; advantage normalisation stays local) and
the only exchange is a sum all-reduce per minibatch of the flat buffer
[actor grads | critic grads | actor_loss, entropy, value_loss, pad], followed by the 1/(U*D) scale
inside the fused Adam kernel.  With backend "nccl" this is RCCL over xGMI; at ~307 KB the collective
is latency-bound, so flat messages (not one per leaf) are the design point: the actor's slice goes out
as soon as its gradient is reduced and travels while the critic's backward kernel runs, the critic's
slice and the loss scalars follow, and the Adam kernel waits for both.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(device: Optional[torch.device] = None, backend: Optional[str] = None) -> Tuple[int, int]:
    """Join the job torch.distributed.run / torchrun described in RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


class AbiComm:
    """The exchange step through libmavahip.so alone (mava_comm_* / mava_allreduce_sum_f32, csrc/comm.cpp): what a host
    without torch.distributed binds.  MAVA_COMM=abi routes the learners' all-reduces through it; the 128-byte RCCL id
    travels from rank 0 over whatever host channel exists (here: the torch.distributed store the launcher set up)."""

    def __init__(self, rank: int, world: int, unique_id: Optional[bytes] = None):
        import ctypes as C

        from ._lib import check, lib

        self._lib, self._check = lib(), check
        self.rank, self.world = rank, world
        if unique_id is None:
            buf = (C.c_uint8 * 128)()
            if rank == 0:
                check(self._lib.mava_comm_unique_id(buf), "mava_comm_unique_id")
            if world > 1:
                obj = [bytes(buf)]
                dist.broadcast_object_list(obj, src=0)
                unique_id = obj[0]
            else:
                unique_id = bytes(buf)
        self.unique_id = unique_id
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._h = C.c_void_p()
        check(self._lib.mava_comm_create(C.byref(self._h), rank, world, idb), "mava_comm_create")
        # the communicator is torn down with the process at the latest (mava_comm_destroy: ncclCommDestroy)
        import atexit

        atexit.register(self._close_quietly)

    def _close_quietly(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def allreduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        from ._lib import stream_ptr

        self._check(self._lib.mava_allreduce_sum_f32(self._h, flat.data_ptr(), flat.numel(), stream_ptr()), "mava_allreduce_sum_f32")
        return flat

    def broadcast_(self, flat: torch.Tensor, src: int = 0) -> torch.Tensor:
        from ._lib import stream_ptr

        self._check(self._lib.mava_broadcast_f32(self._h, flat.data_ptr(), flat.numel(), src, stream_ptr()), "mava_broadcast_f32")
        return flat

    def close(self) -> None:
        if self._h:
            self._check(self._lib.mava_comm_destroy(self._h), "mava_comm_destroy")
            self._h = None


_abi_comm: Optional[AbiComm] = None


def abi_comm() -> Optional[AbiComm]:
    """The process-wide AbiComm when MAVA_COMM=abi (created on first use, on the current device), else None."""
    global _abi_comm
    if os.environ.get("MAVA_COMM", "") != "abi":
        return None
    if _abi_comm is None:
        rank, world = rank_world()
        _abi_comm = AbiComm(rank, world)
    return _abi_comm


class _Done:
    def wait(self) -> None:  # enqueued on the launch stream itself: already ordered
        return None


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the flat gradient/loss buffer (no-op on a single rank)."""
    c = abi_comm()
    if c is not None:
        return c.allreduce_sum_(flat)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_async(flat: torch.Tensor):
    """Start the in-place sum over ranks and return a handle whose .wait() orders the CURRENT stream after the
    collective (RCCL runs it on its own stream, behind everything already enqueued on the current one), or None
    on a single rank.  Used to hide the actor's exchange under the critic's backward pass."""
    c = abi_comm()
    if c is not None:
        c.allreduce_sum_(flat)
        return _Done()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Replicate rank `src`'s parameters (flax.jax_utils.replicate, ff_mappo.py:426)."""
    c = abi_comm()
    if c is not None:
        return c.broadcast_(flat, src)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def grad_scale(update_batch_size: int) -> float:
    """1 / (U * D): turns the summed gradient into pmean over "batch" and "device"."""
    return 1.0 / (update_batch_size * rank_world()[1])
