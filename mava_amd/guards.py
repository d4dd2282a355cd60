"""Range guard of the split-f16 ("f16x2") arithmetic.

The f16x2 kernels (csrc/h2_core.h) represent every matrix operand as two f16 terms, so an operand must stay below
f16's largest finite value (65504): a parameter or an observation beyond it becomes inf in the high term and the
products that follow are inf / NaN.  The reference computes in f32 and has no such limit, so a learner running f16x2
checks what it is handed instead of returning finite-looking garbage:

  * at the start of every learn() call (one device reduction + one host read, per call - not per update): the largest
    |parameter|, the largest |observation| / |global state| of the slot the rollout starts from, and - from the second
    call on - the train metrics the PREVIOUS call produced (an activation that overflowed inside the networks turns the
    losses non-finite; they are complete by then, so reading them costs no extra synchronisation);
  * `MavaHipError` names the offender and the way out (`system.matmul_mode=f32` runs the exact-f32 kernels everywhere).
"""
from __future__ import annotations

from typing import Iterable

import torch

from ._lib import MavaHipError

F16_LIMIT = 6.0e4  # below f16's 65504, with room for the rounding of the high term


def check_f16_range(params: torch.Tensor, obs_tensors: Iterable[torch.Tensor], prev_metrics, who: str) -> None:
    items = [("parameters", params)] + [(f"observation leaf {i}", t) for i, t in enumerate(obs_tensors) if t is not None and t.numel()]
    # max |.| per item; NaN propagates through amax, inf is caught by the comparison
    vals = torch.stack([t.detach().abs().amax().to(torch.float32) for _, t in items])
    bad_metrics = None
    if prev_metrics is not None and prev_metrics.numel():
        bad_metrics = (~torch.isfinite(prev_metrics)).any().to(torch.float32)
        vals = torch.cat([vals, bad_metrics.view(1)])
    host = vals.cpu().tolist()
    for (name, _), v in zip(items, host):
        if not (v < F16_LIMIT):  # also true for NaN
            raise MavaHipError(
                f"{who}: max |{name}| = {v:g} is outside the range of the f16x2 arithmetic (|x| < {F16_LIMIT:g}); "
                "normalise the input or run the exact-f32 kernels (system.matmul_mode=f32 / MAVA_MATMUL=f32)")
    if bad_metrics is not None and host[-1] != 0.0:
        raise MavaHipError(
            f"{who}: the previous learn() call produced non-finite train metrics - an activation left the range of the "
            "f16x2 arithmetic (or the run diverged); run system.matmul_mode=f32 to tell the two apart")
