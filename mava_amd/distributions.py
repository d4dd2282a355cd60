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


class TanhNormal:
    """Independent(TanhTransformedDistribution(Normal(loc, softplus(log_std) + 1e-3)), 1): mava/networks.py:127-169,
    mava/distributions.py:24-91.  Host view for the evaluator seam like Categorical above; the rollout and the loss use
    the fused kernels (mava_policy_step_continuous_f32, mava_ppo_actor_grad_continuous_f32)."""

    THRESH = 0.999

    def __init__(self, loc: torch.Tensor, log_std: torch.Tensor, min_scale: float = 1e-3):
        self.loc = loc
        self.scale = torch.nn.functional.softplus(log_std) + float(min_scale)

    def mode(self) -> torch.Tensor:
        return torch.tanh(self.loc)  # distributions.py:75-77

    def sample(self, seed: Optional[torch.Generator] = None) -> torch.Tensor:
        eps = torch.randn(self.loc.shape, generator=seed, device=self.loc.device)
        return torch.tanh(self.loc + self.scale * eps)

    def log_prob(self, action: torch.Tensor) -> torch.Tensor:
        import math

        n = torch.distributions.Normal(self.loc, self.scale.expand_as(self.loc))
        th = self.THRESH
        ath, log_eps = math.atanh(th), math.log(1.0 - th)
        yc = action.clamp(-th, th)
        x = torch.atanh(yc)
        inner = n.log_prob(x) - 2.0 * (math.log(2.0) - x - torch.nn.functional.softplus(-2.0 * x))
        sn = torch.distributions.Normal(0.0, 1.0)
        left = torch.log(sn.cdf((-ath - self.loc) / self.scale)) - log_eps
        right = torch.log(sn.cdf((self.loc - ath) / self.scale)) - log_eps
        return torch.where(yc <= -th, left, torch.where(yc >= th, right, inner)).sum(-1)

    def entropy(self, seed: Optional[torch.Generator] = None) -> torch.Tensor:
        import math

        eps = torch.randn(self.loc.shape, generator=seed, device=self.loc.device)
        x = self.loc + self.scale * eps
        fldj = 2.0 * (math.log(2.0) - x - torch.nn.functional.softplus(-2.0 * x))
        return (0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(self.scale) + fldj).sum(-1)
