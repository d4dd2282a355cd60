"""Host-side view of the policy distribution for the evaluator seam.

The reference's acting contract is `actor_network.apply(params, observation) -> distribution` with
`.mode()`, `.sample(seed=key)`, `.log_prob(a)`, `.entropy()` (mava/evaluator.py:175-207,
mava/distributions.py:146-165 IdentityTransformation over tfd.Categorical).  The logits come from
the HIP forward kernel; this small class only exposes them through the same method names for
callers OFF the hot path (evaluation, tests).  The training loop never touches it - sampling and
log-probs there are fused into mava_policy_step_f32.
"""
from __future__ import annotations

from typing import Optional

import torch

F32_MIN = torch.finfo(torch.float32).min


class Categorical:
    def __init__(self, logits: torch.Tensor, action_mask: Optional[torch.Tensor] = None):
        if action_mask is not None:
            # mava/networks.py:116-120
            logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, F32_MIN))
        self.logits = logits

    def mode(self) -> torch.Tensor:
        return self.logits.argmax(-1).to(torch.int32)

    def sample(self, seed: Optional[torch.Generator] = None) -> torch.Tensor:
        u = torch.rand(self.logits.shape, generator=seed, device=self.logits.device).clamp_(1e-7, 1.0 - 1e-7)
        return (self.logits - torch.log(-torch.log(u))).argmax(-1).to(torch.int32)

    def log_prob(self, action: torch.Tensor) -> torch.Tensor:
        lsm = torch.log_softmax(self.logits, -1)
        return lsm.gather(-1, action.long().unsqueeze(-1)).squeeze(-1)

    def entropy(self, seed=None) -> torch.Tensor:
        lsm = torch.log_softmax(self.logits, -1)
        p = lsm.exp()
        return -(torch.where(p > 0, p * lsm, torch.zeros_like(p))).sum(-1)
