"""Tensor-level wrappers over the C ABI: shape/dtype/device validation on the host, then one
asynchronous launch on torch's current HIP stream.  Tensors only carry device memory; all
arithmetic happens in the hand-written kernels of libmavahip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import Ctx, check, ctx_ptr, lib, ptr, stream_ptr


def _req(t: torch.Tensor, dtype: torch.dtype, name: str, shape: Optional[Tuple[int, ...]] = None) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.MavaHipError(f"{name}: tensor must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def _as_u8(t: torch.Tensor) -> torch.Tensor:
    """bool tensors are reinterpreted (not copied) as uint8."""
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def permutation(n: int, seed: int, counter: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 permutation of range(n) determined by (seed, counter): the epoch shuffle that stands where the reference
    calls jax.random.permutation (ff_mappo.py:272-273); see mava_permutation_i32."""
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    _req(out, torch.int32, "out", (n,))
    check(lib().mava_permutation_i32(n, seed & (2**64 - 1), counter & (2**64 - 1), ptr(out), stream_ptr()), "mava_permutation_i32")
    return out


def gae(reward, value, done, last_val, gamma: float, gae_lambda: float, last_done=None, out=None, ctx: Optional[Ctx] = None):
    """(advantages, targets) for time-major (T, ...) inputs; see mava_gae_f32."""
    T = reward.shape[0]
    N = reward[0].numel() if T > 0 else last_val.numel()
    done = _as_u8(done)
    _req(reward, torch.float32, "reward")
    _req(value, torch.float32, "value", reward.shape)
    _req(done, torch.uint8, "done", reward.shape)
    _req(last_val, torch.float32, "last_val")
    if last_val.numel() != N:
        raise ValueError(f"last_val: expected {N} elements, got {last_val.numel()}")
    if last_done is not None:
        last_done = _as_u8(last_done)
        _req(last_done, torch.uint8, "last_done")
        if last_done.numel() != N:
            raise ValueError(f"last_done: expected {N} elements, got {last_done.numel()}")
    if out is None:
        adv = torch.empty_like(reward)
        tgt = torch.empty_like(reward)
    else:
        adv, tgt = out
        _req(adv, torch.float32, "adv", reward.shape)
        _req(tgt, torch.float32, "tgt", reward.shape)
    check(
        lib().mava_gae_f32(ctx_ptr(ctx), ptr(reward), ptr(value), ptr(done), ptr(last_val), ptr(last_done), T, N,
                           gamma, gae_lambda, ptr(adv), ptr(tgt), stream_ptr()),
        "mava_gae_f32",
    )
    return adv, tgt


def clip_adam(p, g, m, v, count, seg_off: Sequence[int], seg_lr: Sequence[float], *, grad_scale: float,
              max_norm: float, decay: bool = False, steps_per_update: int = 1, num_updates: int = 1,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-5, loss_sums=None, vf_coef: float = 0.0,
              ent_coef: float = 0.0, metrics_out=None) -> None:
    n_seg = len(seg_lr)
    if len(seg_off) != n_seg + 1:
        raise ValueError("seg_off must have len(seg_lr)+1 entries")
    total = int(seg_off[-1])
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _req(t, torch.float32, name)
        if t.numel() < total:
            raise ValueError(f"{name}: needs at least {total} elements, has {t.numel()}")
    _req(count, torch.int32, "count")
    if count.numel() < n_seg:
        raise ValueError("count: one int32 per segment required")
    if loss_sums is not None:
        _req(loss_sums, torch.float32, "loss_sums")
        if loss_sums.numel() < 3:
            raise ValueError("loss_sums: 3 floats required")
    if metrics_out is not None:
        _req(metrics_out, torch.float32, "metrics_out")
        if metrics_out.numel() < 4:
            raise ValueError("metrics_out: 4 floats required")
    off = (C.c_int * (n_seg + 1))(*[int(x) for x in seg_off])
    lr = (C.c_float * n_seg)(*[float(x) for x in seg_lr])
    check(
        lib().mava_clip_adam(ptr(p), ptr(g), ptr(m), ptr(v), ptr(count), off, lr, n_seg, grad_scale, max_norm,
                             int(decay), steps_per_update, num_updates, b1, b2, eps, ptr(loss_sums), vf_coef,
                             ent_coef, ptr(metrics_out), stream_ptr()),
        "mava_clip_adam",
    )


def slab_reduce(slab: torch.Tensor, n: int, out: torch.Tensor, accumulate: bool = False) -> None:
    _req(slab, torch.float32, "slab")
    _req(out, torch.float32, "out")
    if slab.dim() != 2 or slab.shape[1] < n or out.numel() < n:
        raise ValueError("slab must be (n_slab, stride>=n) and out must hold n floats")
    check(
        lib().mava_slab_reduce_f32(ptr(slab), slab.shape[0], slab.shape[1], n, int(accumulate), ptr(out),
                                   stream_ptr()),
        "mava_slab_reduce_f32",
    )


def slab_reduce_cols(slab: torch.Tensor, col0: int, n: int, out: torch.Tensor, accumulate: bool = False) -> None:
    """out[:n] (+)= sum over slabs of columns [col0, col0 + n) - the same kernel on a column window, no copy of the window."""
    _req(slab, torch.float32, "slab")
    _req(out, torch.float32, "out")
    if slab.dim() != 2 or col0 < 0 or slab.shape[1] < col0 + n or out.numel() < n:
        raise ValueError("slab must be (n_slab, stride >= col0 + n) and out must hold n floats")
    check(lib().mava_slab_reduce_f32(slab.data_ptr() + 4 * col0, slab.shape[0], slab.shape[1], n, int(accumulate), ptr(out), stream_ptr()),
          "mava_slab_reduce_f32")


def ppo_finish_workspace(Pa: int, Pc: int, device) -> torch.Tensor:
    """Zeroed workspace of ppo_finish (norm partials + arrival ticket); allocate once per learner."""
    n = int(lib().mava_ppo_finish_workspace_bytes(Pa, Pc))
    return torch.zeros((n + 7) // 8, dtype=torch.float64, device=device)


def ppo_finish(ctx: Optional[Ctx], slab_a, slab_c, Pa: int, Pc: int, g, p, m, v, count, lr_a: float, lr_c: float, *,
               grad_scale: float, max_norm: float, decay: bool, steps_per_update: int, num_updates: int, vf_coef: float,
               ent_coef: float, metrics_out, critic_din: int, workspace: torch.Tensor, b1: float = 0.9, b2: float = 0.999,
               eps: float = 1e-5) -> None:
    """Both slab reductions + clip + Adam + count increment (+ the wide critic's W1 re-split) in two launches
    (mava_ppo_finish_f32): the tail of a minibatch on one rank with one update-batch replica."""
    reduced = slab_a is None  # g already holds the summed (all-reduced) gradient: the Adam launch alone (+ counts, W1 re-split)
    if not reduced:
        _req(slab_a, torch.float32, "slab_a")
        _req(slab_c, torch.float32, "slab_c")
        if slab_a.dim() != 2 or slab_c.dim() != 2 or slab_a.shape[0] != slab_c.shape[0] or slab_a.shape[1] < Pa + 2 or slab_c.shape[1] < Pc + 1:
            raise ValueError("ppo_finish: slabs must be (n_slab, >= P + 2) with equal slab counts")
    for name, t, n in (("g", g, Pa + Pc + 3), ("p", p, Pa + Pc), ("m", m, Pa + Pc), ("v", v, Pa + Pc)):
        _req(t, torch.float32, name)
        if t.numel() < n:
            raise ValueError(f"{name}: needs at least {n} elements")
    _req(count, torch.int32, "count")
    _req(workspace, torch.float64, "workspace")
    if metrics_out is not None:
        _req(metrics_out, torch.float32, "metrics_out")
    check(lib().mava_ppo_finish_f32(ctx_ptr(ctx), ptr(slab_a), 0 if reduced else slab_a.shape[1], ptr(slab_c),
                                    0 if reduced else slab_c.shape[1], 0 if reduced else slab_a.shape[0], Pa, Pc,
                                    ptr(g), ptr(p), ptr(m), ptr(v), ptr(count), lr_a, lr_c, grad_scale, max_norm, int(decay),
                                    steps_per_update, num_updates, b1, b2, eps, vf_coef, ent_coef, ptr(metrics_out), critic_din,
                                    ptr(workspace), workspace.numel() * 8, stream_ptr()), "mava_ppo_finish_f32")


def mlp_param_count(din: int, n_out: int) -> int:
    return lib().mava_mlp_param_count(din, n_out)


def mlp_forward(params: torch.Tensor, din: int, n_out: int, x: torch.Tensor, rows: Optional[int] = None,
                x_share: int = 1, out: Optional[torch.Tensor] = None, ctx: Optional[Ctx] = None) -> torch.Tensor:
    """Raw network outputs (rows, n_out); x is (rows_x, din) and output row r reads x[r // x_share]."""
    _req(params, torch.float32, "params")
    if params.numel() != mlp_param_count(din, n_out):
        raise ValueError(f"params: expected {mlp_param_count(din, n_out)} floats, got {params.numel()}")
    _req(x, torch.float32, "x")
    if x.dim() != 2 or x.shape[1] != din:
        raise ValueError(f"x: expected (rows, {din}), got {tuple(x.shape)}")
    if rows is None:
        rows = x.shape[0] * x_share
    if (rows + x_share - 1) // x_share > x.shape[0]:
        raise ValueError("x has too few rows for rows/x_share")
    if out is None:
        out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    else:
        _req(out, torch.float32, "out")
        if out.numel() != rows * n_out:
            raise ValueError(f"out: expected {rows * n_out} elements, got {out.numel()}")
    check(lib().mava_mlp_forward_f32(ctx_ptr(ctx), ptr(params), din, n_out, ptr(x), x_share, rows, ptr(out), stream_ptr()),
          "mava_mlp_forward_f32")
    return out


def policy_step(actor_params, critic_params, agents_view, action_mask, critic_input, *, n_actions: int,
                critic_share: int = 1, critic_rows: Optional[int] = None, value_broadcast: int = 1, seed: int,
                step: int, row_offset: int = 0, greedy: bool = False, forced_action=None, out=None,
                want_logits: bool = False, step_base: Optional[torch.Tensor] = None, ctx: Optional[Ctx] = None):
    """One acting step: returns (action i32 (rows), log_prob (rows), value, logits|None).  `step_base` (a device
    int32 word) is added to `step` on the device."""
    if step_base is not None:
        _req(step_base, torch.int32, "step_base")
    _req(agents_view, torch.float32, "agents_view")
    rows, actor_din = agents_view.shape
    _req(critic_input, torch.float32, "critic_input")
    critic_din = critic_input.shape[1]
    if critic_rows is None:
        critic_rows = critic_input.shape[0] * critic_share
    if (critic_rows + critic_share - 1) // critic_share > critic_input.shape[0]:
        raise ValueError("critic_input has too few rows")
    _req(actor_params, torch.float32, "actor_params")
    _req(critic_params, torch.float32, "critic_params")
    if actor_params.numel() != mlp_param_count(actor_din, n_actions):
        raise ValueError("actor_params: wrong size")
    if critic_params.numel() != mlp_param_count(critic_din, 1):
        raise ValueError("critic_params: wrong size")
    if action_mask is not None:
        action_mask = _as_u8(action_mask)
        _req(action_mask, torch.uint8, "action_mask", (rows, n_actions))
    if forced_action is not None:
        _req(forced_action, torch.int32, "forced_action", (rows,))
    dev = agents_view.device
    if out is None:
        action = torch.empty((rows,), dtype=torch.int32, device=dev)
        log_prob = torch.empty((rows,), dtype=torch.float32, device=dev)
        value = torch.empty((critic_rows * value_broadcast,), dtype=torch.float32, device=dev)
    else:
        action, log_prob, value = out
        _req(action, torch.int32, "action")
        _req(log_prob, torch.float32, "log_prob")
        _req(value, torch.float32, "value")
        if action.numel() != rows or log_prob.numel() != rows or value.numel() != critic_rows * value_broadcast:
            raise ValueError("policy_step: output buffers have the wrong size")
    logits = torch.empty((rows, n_actions), dtype=torch.float32, device=dev) if want_logits else None
    check(
        lib().mava_policy_step_f32(ctx_ptr(ctx), ptr(actor_params), actor_din, n_actions, ptr(agents_view), ptr(action_mask),
                                   ptr(critic_params), critic_din, ptr(critic_input), critic_share, critic_rows,
                                   value_broadcast, rows, seed & 0xFFFFFFFFFFFFFFFF, step & 0xFFFFFFFF,
                                   ptr(step_base), row_offset & 0xFFFFFFFF, int(greedy), ptr(forced_action), ptr(action),
                                   ptr(log_prob), ptr(value), ptr(logits), stream_ptr()),
        "mava_policy_step_f32",
    )
    return action, log_prob, value, logits


def continuous_param_count(din: int, action_dim: int) -> int:
    """[MLP(din -> 128 -> 128 -> action_dim) | log_std(action_dim)] (networks.py:127-169, independent_std)."""
    return mlp_param_count(din, action_dim) + action_dim


def policy_step_continuous(actor_params, critic_params, agents_view, critic_input, *, action_dim: int,
                           critic_share: int = 1, critic_rows: Optional[int] = None, value_broadcast: int = 1,
                           seed: int, step: int, row_offset: int = 0, greedy: bool = False, forced_action=None,
                           out=None, want_mean: bool = False, step_base: Optional[torch.Tensor] = None,
                           ctx: Optional[Ctx] = None, min_scale: float = 1e-3):
    """One acting step with the continuous head: returns (action f32 (rows, action_dim), log_prob (rows), value,
    mean|None)."""
    if step_base is not None:
        _req(step_base, torch.int32, "step_base")
    _req(agents_view, torch.float32, "agents_view")
    rows, actor_din = agents_view.shape
    _req(critic_input, torch.float32, "critic_input")
    critic_din = critic_input.shape[1]
    if critic_rows is None:
        critic_rows = critic_input.shape[0] * critic_share
    if (critic_rows + critic_share - 1) // critic_share > critic_input.shape[0]:
        raise ValueError("critic_input has too few rows")
    _req(actor_params, torch.float32, "actor_params")
    _req(critic_params, torch.float32, "critic_params")
    if actor_params.numel() != continuous_param_count(actor_din, action_dim):
        raise ValueError("actor_params: wrong size")
    if critic_params.numel() != mlp_param_count(critic_din, 1):
        raise ValueError("critic_params: wrong size")
    if forced_action is not None:
        _req(forced_action, torch.float32, "forced_action", (rows, action_dim))
    dev = agents_view.device
    if out is None:
        action = torch.empty((rows, action_dim), dtype=torch.float32, device=dev)
        log_prob = torch.empty(rows, dtype=torch.float32, device=dev)
        value = torch.empty(critic_rows * value_broadcast, dtype=torch.float32, device=dev)
    else:
        action, log_prob, value = out
        _req(action, torch.float32, "action")
        _req(log_prob, torch.float32, "log_prob")
        _req(value, torch.float32, "value")
        if action.numel() != rows * action_dim or log_prob.numel() != rows or value.numel() != critic_rows * value_broadcast:
            raise ValueError("out: wrong sizes")
    mean = torch.empty((rows, action_dim), dtype=torch.float32, device=dev) if want_mean else None
    check(
        lib().mava_policy_step_continuous_f32(ctx_ptr(ctx), ptr(actor_params), actor_din, action_dim, float(min_scale), ptr(agents_view),
                                              ptr(critic_params), critic_din, ptr(critic_input), critic_share,
                                              critic_rows, value_broadcast, rows, seed & 0xFFFFFFFFFFFFFFFF,
                                              step & 0xFFFFFFFF, ptr(step_base), row_offset & 0xFFFFFFFF, int(greedy),
                                              ptr(forced_action), ptr(action), ptr(log_prob), ptr(value), ptr(mean),
                                              stream_ptr()),
        "mava_policy_step_continuous_f32",
    )
    return action, log_prob, value, mean


def ppo_actor_grad_continuous(params, agents_view, action, old_log_prob, advantages, stats, idx, idx_base: int,
                              Rb: int, A: int, action_dim: int, clip_eps: float, ent_coef: float, seed: int,
                              ent_step: int, row_offset: int, slab: torch.Tensor, min_scale: float = 1e-3) -> None:
    """Fills slab (n_slab, stride) with partial [MLP gradient | d log_std | actor_loss, entropy] sums."""
    _req(agents_view, torch.float32, "agents_view")
    rows, din = agents_view.shape
    if rows % A:
        raise ValueError("agents_view rows must be a multiple of A")
    TE = rows // A
    _req(params, torch.float32, "params")
    P = continuous_param_count(din, action_dim)
    if params.numel() != P:
        raise ValueError(f"params: expected {P} floats")
    _req(action, torch.float32, "action")
    _req(old_log_prob, torch.float32, "old_log_prob")
    _req(advantages, torch.float32, "advantages")
    for n, t, k in (("action", action, action_dim), ("old_log_prob", old_log_prob, 1), ("advantages", advantages, 1)):
        if t.numel() != rows * k:
            raise ValueError(f"{n}: expected {rows * k} elements, got {t.numel()}")
    _req(stats, torch.float64, "stats", (lib().mava_adv_stats_blocks(), 2))
    _check_idx(idx, idx_base, Rb, TE)
    if rows >= 2 ** 31:
        raise ValueError("trajectory too large: TE*A must be < 2^31 (32-bit row arithmetic in the kernel)")
    _req(slab, torch.float32, "slab")
    if slab.dim() != 2 or slab.shape[1] < P + 2:
        raise ValueError("slab must be (n_slab, >= P+2)")
    check(
        lib().mava_ppo_actor_grad_continuous_f32(ptr(params), din, action_dim, float(min_scale), ptr(agents_view), ptr(action),
                                                 ptr(old_log_prob), ptr(advantages), ptr(stats), ptr(idx), idx_base,
                                                 Rb, A, clip_eps, ent_coef, seed & 0xFFFFFFFFFFFFFFFF,
                                                 ent_step & 0xFFFFFFFF, row_offset & 0xFFFFFFFF, ptr(slab),
                                                 slab.shape[1], slab.shape[0], stream_ptr()),
        "mava_ppo_actor_grad_continuous_f32",
    )


def adv_stats_batched(advantages: torch.Tensor, idx: torch.Tensor, Rb: int, A: int, n_batch: int,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """adv_stats of n_batch consecutive Rb-slices of idx in one launch: out[j] = the partials of idx[j*Rb : (j+1)*Rb]."""
    _req(advantages, torch.float32, "advantages")
    _req(idx, torch.int32, "idx")
    if idx.numel() < n_batch * Rb:
        raise ValueError("idx shorter than n_batch * Rb")
    nb = lib().mava_adv_stats_blocks()
    if out is None:
        out = torch.empty((n_batch, nb, 2), dtype=torch.float64, device=advantages.device)
    _req(out, torch.float64, "out", (n_batch, nb, 2))
    check(lib().mava_adv_stats_batched_f64(ptr(advantages), ptr(idx), Rb, Rb, A, n_batch, ptr(out), stream_ptr()),
          "mava_adv_stats_batched_f64")
    return out


def adv_stats(advantages: torch.Tensor, idx: Optional[torch.Tensor], idx_base: int, Rb: int, A: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f64 (sum, sumsq) partials of the minibatch advantages, consumed by ppo_actor_grad."""
    _req(advantages, torch.float32, "advantages")
    if idx is not None:
        _req(idx, torch.int32, "idx")
        if idx.numel() < Rb:
            raise ValueError("idx shorter than Rb")
    elif (idx_base + Rb) * A > advantages.numel():
        raise ValueError("minibatch exceeds the trajectory")
    nb = lib().mava_adv_stats_blocks()
    if out is None:
        out = torch.empty((nb, 2), dtype=torch.float64, device=advantages.device)
    else:
        _req(out, torch.float64, "out", (nb, 2))
    check(lib().mava_adv_stats_f64(ptr(advantages), ptr(idx), idx_base, Rb, A, ptr(out), stream_ptr()),
          "mava_adv_stats_f64")
    return out


def _check_idx(idx, idx_base, Rb, TE):
    if idx is not None:
        _req(idx, torch.int32, "idx")
        if idx.numel() < Rb:
            raise ValueError("idx shorter than Rb")
    elif idx_base < 0 or idx_base + Rb > TE:
        raise ValueError("minibatch exceeds the trajectory")


def ppo_actor_grad(params, agents_view, action_mask, action, old_log_prob, advantages, stats, idx, idx_base: int,
                   Rb: int, A: int, n_actions: int, clip_eps: float, ent_coef: float, slab: torch.Tensor,
                   ctx: Optional[Ctx] = None) -> None:
    """Fills slab (n_slab, stride) with partial [actor gradient | actor_loss, entropy] sums.  `ctx` selects the arithmetic
    (None: exact f32)."""
    _req(agents_view, torch.float32, "agents_view")
    rows, din = agents_view.shape
    if rows % A:
        raise ValueError("agents_view rows must be a multiple of A")
    TE = rows // A
    _req(params, torch.float32, "params")
    P = mlp_param_count(din, n_actions)
    if params.numel() != P:
        raise ValueError(f"params: expected {P} floats")
    if action_mask is not None:
        action_mask = _as_u8(action_mask)
        _req(action_mask, torch.uint8, "action_mask", (rows, n_actions))
    _req(action, torch.int32, "action")
    _req(old_log_prob, torch.float32, "old_log_prob")
    _req(advantages, torch.float32, "advantages")
    for n, t in (("action", action), ("old_log_prob", old_log_prob), ("advantages", advantages)):
        if t.numel() != rows:
            raise ValueError(f"{n}: expected {rows} elements, got {t.numel()}")
    _req(stats, torch.float64, "stats", (lib().mava_adv_stats_blocks(), 2))
    _check_idx(idx, idx_base, Rb, TE)
    if rows >= 2 ** 31:
        raise ValueError("trajectory too large: TE*A must be < 2^31 (32-bit row arithmetic in the kernel)")
    _req(slab, torch.float32, "slab")
    if slab.dim() != 2 or slab.shape[1] < P + 2:
        raise ValueError("slab must be (n_slab, >= P+2)")
    check(
        lib().mava_ppo_actor_grad_f32(ctx_ptr(ctx), ptr(params), din, n_actions, ptr(agents_view), ptr(action_mask), ptr(action),
                                      ptr(old_log_prob), ptr(advantages), ptr(stats), ptr(idx), idx_base, Rb, A,
                                      clip_eps, ent_coef, ptr(slab), slab.shape[1], slab.shape[0], stream_ptr()),
        "mava_ppo_actor_grad_f32",
    )


def ppo_critic_grad(params, critic_input, x_share: int, old_value, targets, idx, idx_base: int, Rb: int, A: int,
                    clip_eps: float, vf_coef: float, slab: torch.Tensor, ctx: Optional[Ctx] = None) -> None:
    """Fills slab (n_slab, stride) with partial [critic gradient | value_loss, 0] sums.  `ctx` selects the arithmetic and
    the aggregation of shared input rows (None: exact f32, aggregation on)."""
    _req(critic_input, torch.float32, "critic_input")
    din = critic_input.shape[1]
    _req(params, torch.float32, "params")
    P = mlp_param_count(din, 1)
    if params.numel() != P:
        raise ValueError(f"params: expected {P} floats")
    _req(old_value, torch.float32, "old_value")
    _req(targets, torch.float32, "targets")
    rows = old_value.numel()
    if targets.numel() != rows or rows % A:
        raise ValueError("old_value/targets must hold TE*A elements")
    if (rows + x_share - 1) // x_share > critic_input.shape[0]:
        raise ValueError("critic_input has too few rows")
    _check_idx(idx, idx_base, Rb, rows // A)
    if rows >= 2 ** 31:
        raise ValueError("trajectory too large: TE*A must be < 2^31 (32-bit row arithmetic in the kernel)")
    _req(slab, torch.float32, "slab")
    if slab.dim() != 2 or slab.shape[1] < P + 2:
        raise ValueError("slab must be (n_slab, >= P+2)")
    check(
        lib().mava_ppo_critic_grad_f32(ctx_ptr(ctx), ptr(params), din, ptr(critic_input), x_share, ptr(old_value), ptr(targets),
                                       ptr(idx), idx_base, Rb, A, clip_eps, vf_coef, ptr(slab), slab.shape[1],
                                       slab.shape[0], stream_ptr()),
        "mava_ppo_critic_grad_f32",
    )


def slab_reduce2(slab: torch.Tensor, n_main: int, out_main: torch.Tensor, n_tail: int, out_tail: torch.Tensor,
                 accumulate: bool = False) -> None:
    """Sum slab rows: columns [0,n_main) -> out_main, [n_main, n_main+n_tail) -> out_tail."""
    _req(slab, torch.float32, "slab")
    _req(out_main, torch.float32, "out_main")
    _req(out_tail, torch.float32, "out_tail")
    if slab.dim() != 2 or slab.shape[1] < n_main + n_tail or out_main.numel() < n_main or out_tail.numel() < n_tail:
        raise ValueError("slab_reduce2: inconsistent sizes")
    check(
        lib().mava_slab_reduce2_f32(ptr(slab), slab.shape[0], slab.shape[1], n_main, ptr(out_main), n_tail, ptr(out_tail),
                                    int(accumulate), stream_ptr()),
        "mava_slab_reduce2_f32",
    )


def rollout_ff(actor_params, critic_params, *, n_actions: int, critic_shared: bool, E: int, A: int, O: int, T: int,
               time_limit: int, policy_seed: int, env_seed: int, t0: int, row_offset: int, env_offset: int,
               reward_mode: int, env_state, agents_view, global_state, action_mask, obs_step_count, action, value,
               reward, log_prob, done, last_val, info_return, info_length, info_terminal, adv=None, tgt=None,
               gamma: float = 0.99, gae_lambda: float = 0.95) -> bool:
    """The whole rollout of one replica in one launch (mava_rollout_ff_f32).  Returns False when the library does not
    instantiate the shape - the caller then steps policy_step / env.step_into per time step."""
    W = A + O
    _req(actor_params, torch.float32, "actor_params")
    _req(critic_params, torch.float32, "critic_params")
    if actor_params.numel() != mlp_param_count(W, n_actions):
        raise ValueError("actor_params: wrong size")
    if critic_params.numel() != mlp_param_count(A * O if critic_shared else W, 1):
        raise ValueError("critic_params: wrong size")
    _req(agents_view, torch.float32, "agents_view", (T + 1, E, A, W))
    if critic_shared:
        _req(global_state, torch.float32, "global_state", (T + 1, E, 1, A * O))
    _req(action_mask, torch.uint8, "action_mask", (T + 1, E, A, n_actions))
    _req(obs_step_count, torch.int32, "obs_step_count", (T + 1, E, A))
    _req(action, torch.int32, "action", (T, E, A))
    for name, t, dt in (("value", value, torch.float32), ("reward", reward, torch.float32),
                        ("log_prob", log_prob, torch.float32), ("done", done, torch.uint8)):
        _req(t, dt, name, (T, E, A))
    _req(last_val, torch.float32, "last_val", (E, A))
    if adv is not None or tgt is not None:
        _req(adv, torch.float32, "adv", (T, E, A))
        _req(tgt, torch.float32, "tgt", (T, E, A))
    _req(info_return, torch.float32, "info_return", (T, E))
    _req(info_length, torch.int32, "info_length", (T, E))
    _req(info_terminal, torch.uint8, "info_terminal", (T, E))
    _req(env_state.step_count, torch.int32, "step_count", (E, A))
    for name, t, dt in (("run_return", env_state.run_return, torch.float32), ("run_length", env_state.run_length, torch.int32),
                        ("ep_return", env_state.ep_return, torch.float32), ("ep_length", env_state.ep_length, torch.int32)):
        _req(t, dt, name, (E,))
    rc = lib().mava_rollout_ff_f32(
        ptr(actor_params), n_actions, ptr(critic_params), int(critic_shared), E, A, O, T, time_limit,
        policy_seed & 0xFFFFFFFFFFFFFFFF, env_seed & 0xFFFFFFFFFFFFFFFF, t0 & 0xFFFFFFFF, row_offset & 0xFFFFFFFF,
        env_offset & 0xFFFFFFFF, reward_mode, ptr(env_state.step_count), ptr(env_state.run_return),
        ptr(env_state.run_length), ptr(env_state.ep_return), ptr(env_state.ep_length), ptr(agents_view),
        ptr(global_state) if critic_shared else None, ptr(action_mask), ptr(obs_step_count), ptr(action), ptr(value),
        ptr(reward), ptr(log_prob), ptr(done), ptr(last_val), ptr(info_return), ptr(info_length), ptr(info_terminal),
        ptr(adv), ptr(tgt), gamma, gae_lambda, stream_ptr())
    if rc == 1:
        return False
    check(rc, "mava_rollout_ff_f32")
    return True
