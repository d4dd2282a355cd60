"""Network definitions behind the reference's module names (mava/networks.py:39-58 MLPTorso,
:88-124 DiscreteActionHead, :127-169 ContinuousActionHead, :172-183 FeedForwardActor, :186-207 FeedForwardValueNet).

A network here is (a) a flat float32 parameter vector in the kernel layout
[W1(din,128) | b1 | W2(128,128) | b2 | W3(128,n_out) | b3] and (b) a Flax-shaped tree of VIEWS into
that vector, so `Params.actor_params["params"]["torso"]["Dense_0"]["kernel"]` is what a Mava user
expects while the kernels see one contiguous buffer.  `apply` runs the HIP forward kernel.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Sequence, Tuple

import torch

from . import ops
from .distributions import Categorical, TanhNormal
from .types import Observation, ObservationGlobalState

HIDDEN = 128


class MLPTorso:
    """mava/networks.py:39-58.  Only the configuration the fused kernels implement is accepted."""

    def __init__(self, layer_sizes: Sequence[int] = (128, 128), activation: str = "relu", use_layer_norm: bool = False,
                 **_: Any):
        self.layer_sizes = list(layer_sizes)
        self.activation = activation
        self.use_layer_norm = use_layer_norm
        if self.layer_sizes not in ([HIDDEN, HIDDEN], [HIDDEN]) or activation != "relu" or use_layer_norm:
            raise NotImplementedError(
                "mava_amd's kernels implement layer_sizes=[128,128] (network/mlp.yaml) or [128] (network/rnn.yaml), "
                f"activation=relu, use_layer_norm=False; got {self.layer_sizes}, {activation}, layer_norm={use_layer_norm}"
            )

    def require(self, layer_sizes) -> None:
        if self.layer_sizes != list(layer_sizes):
            raise NotImplementedError(f"this network needs a torso with layer_sizes={list(layer_sizes)}, got {self.layer_sizes}")


class DiscreteActionHead:
    """mava/networks.py:88-124."""

    def __init__(self, action_dim: int, **_: Any):
        self.action_dim = int(action_dim)


class ContinuousActionHead:
    """mava/networks.py:127-169: loc = Dense(action_dim), scale = softplus(log_std) + min_scale; log_std is an
    observation-independent parameter vector (independent_std=True, the fused kernels) or Dense(action_dim) of the
    embedding (independent_std=False: the general network path, mava_amd/generic_networks.py); actions in [-1, 1]
    through a tanh bijector."""

    def __init__(self, action_dim: int, min_scale: float = 1e-3, independent_std: bool = True, **_: Any):
        self.action_dim = int(action_dim)
        self.independent_std = bool(independent_std)
        self.min_scale = float(min_scale)  # scale = softplus(log_std) + min_scale (mava/networks.py:134,162); a kernel argument
        if not (self.min_scale >= 0.0):
            raise ValueError(f"ContinuousActionHead.min_scale must be >= 0, got {min_scale}")
        if self.action_dim > 16:
            raise NotImplementedError(f"continuous action heads are instantiated up to 16 dimensions, got {self.action_dim}")


def make_action_head(cfg: Any, action_dim: int):
    """hydra.utils.instantiate(config.network.action_head, action_dim=env.action_dim), ff_mappo.py:348-350."""
    kw = {k: v for k, v in dict(cfg).items() if k != "_target_"} if cfg is not None else {}
    target = str(dict(cfg).get("_target_", "mava.networks.DiscreteActionHead")) if cfg is not None else ""
    if target.endswith("ContinuousActionHead"):
        return ContinuousActionHead(action_dim, **kw)
    if target == "" or target.endswith("DiscreteActionHead"):
        return DiscreteActionHead(action_dim, **kw)
    raise NotImplementedError(f"unknown action head {target}")


def _orthogonal_(w: torch.Tensor, scale: float, gen: torch.Generator) -> None:
    """flax.linen.initializers.orthogonal(scale) semantics (QR of a normal matrix, sign-fixed) on the
    host; the bit stream differs from JAX's PRNG (init parity is not a goal)."""
    rows, cols = w.shape
    a = torch.randn((max(rows, cols), min(rows, cols)), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.T
    w.copy_((scale * q[:rows, :cols]).to(w.dtype))


def mlp_segments(din: int, n_out: int):
    """[(name path, shape, offset)] in kernel order."""
    segs, off = [], 0
    for path, shape in (
        (("Dense_0", "kernel"), (din, HIDDEN)),
        (("Dense_0", "bias"), (HIDDEN,)),
        (("Dense_1", "kernel"), (HIDDEN, HIDDEN)),
        (("Dense_1", "bias"), (HIDDEN,)),
        (("head", "kernel"), (HIDDEN, n_out)),
        (("head", "bias"), (n_out,)),
    ):
        n = math.prod(shape)
        segs.append((path, shape, off))
        off += n
    return segs, off


class _FeedForwardNet:
    head_scale = 1.0
    head_parent = None  # name of the sub-module holding the head Dense

    def first_leaf(self, tree: Dict[str, Any]) -> torch.Tensor:
        return tree["params"]["torso"]["Dense_0"]["kernel"]

    def __init__(self, din: int, n_out: int):
        self.din, self.n_out = int(din), int(n_out)
        self.segments, self.num_params = mlp_segments(self.din, self.n_out)

    # -- parameters -------------------------------------------------------------------------
    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        """networks.py:54 orthogonal(sqrt 2) torso kernels, zero biases; head scale per subclass."""
        gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        flat = torch.zeros(self.num_params, dtype=torch.float32)
        for path, shape, off in self.segments:
            if path[1] == "kernel":
                scale = self.head_scale if path[0] == "head" else math.sqrt(2.0)
                _orthogonal_(flat[off : off + math.prod(shape)].view(shape), scale, gen)
        return flat.to(device) if device is not None else flat

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        """Flax-shaped tree of views into `flat` (optionally with broadcast leading dims)."""
        def view(shape, off):
            v = flat[off : off + math.prod(shape)].view(shape)
            return v.expand(*lead, *shape) if lead else v

        by = {path: view(shape, off) for path, shape, off in self.segments}
        torso = {
            "Dense_0": {"kernel": by[("Dense_0", "kernel")], "bias": by[("Dense_0", "bias")]},
            "Dense_1": {"kernel": by[("Dense_1", "kernel")], "bias": by[("Dense_1", "bias")]},
        }
        head = {"kernel": by[("head", "kernel")], "bias": by[("head", "bias")]}
        return {"params": self._assemble(torso, head)}

    def flat_from_tree(self, tree: Dict[str, Any], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Inverse of `tree` (accepts leaves with extra leading broadcast dims: takes index 0)."""
        torso, head = self._disassemble(tree["params"])
        leaves = {
            ("Dense_0", "kernel"): torso["Dense_0"]["kernel"], ("Dense_0", "bias"): torso["Dense_0"]["bias"],
            ("Dense_1", "kernel"): torso["Dense_1"]["kernel"], ("Dense_1", "bias"): torso["Dense_1"]["bias"],
            ("head", "kernel"): head["kernel"], ("head", "bias"): head["bias"],
        }
        if out is None:
            any_leaf = leaves[("head", "bias")]
            out = torch.empty(self.num_params, dtype=torch.float32, device=any_leaf.device)
        for path, shape, off in self.segments:
            leaf = leaves[path]
            while leaf.dim() > len(shape):
                leaf = leaf[0]
            out[off : off + math.prod(shape)].view(shape).copy_(leaf)
        return out


class FeedForwardActor(_FeedForwardNet):
    """mava/networks.py:172-183 with DiscreteActionHead (:88-124) or ContinuousActionHead (:127-169): head Dense
    init orthogonal(0.01); the continuous head appends its log_std vector (zeros) to the flat parameters."""

    head_scale = 0.01

    def __init__(self, torso: MLPTorso, action_head, obs_dim: int):
        super().__init__(obs_dim, action_head.action_dim)
        torso.require([HIDDEN, HIDDEN])
        self.torso, self.action_head = torso, action_head
        self.continuous = isinstance(action_head, ContinuousActionHead)
        self.num_mlp_params = self.num_params
        if self.continuous:
            self.num_params += self.n_out

    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        flat = torch.zeros(self.num_params, dtype=torch.float32)  # log_std: nn.initializers.zeros (networks.py:141)
        flat[: self.num_mlp_params].copy_(super().init_flat(seed)[: self.num_mlp_params])
        return flat.to(device) if device is not None else flat

    def _log_std(self, flat: torch.Tensor) -> torch.Tensor:
        return flat[self.num_mlp_params : self.num_mlp_params + self.n_out]

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        out = super().tree(flat, lead)
        if self.continuous:
            ls = self._log_std(flat)
            out["params"]["action_head"]["log_std"] = ls.expand(*lead, self.n_out) if lead else ls
        return out

    def flat_from_tree(self, tree: Dict[str, Any], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            leaf = tree["params"]["torso"]["Dense_0"]["bias"]
            out = torch.empty(self.num_params, dtype=torch.float32, device=leaf.device)
        super().flat_from_tree(tree, out)
        if self.continuous:
            ls = tree["params"]["action_head"]["log_std"]
            while ls.dim() > 1:
                ls = ls[0]
            self._log_std(out).copy_(ls)
        return out

    def _assemble(self, torso, head):
        return {"torso": torso, "action_head": {("mean" if self.continuous else "Dense_0"): head}}

    def _disassemble(self, p):
        return p["torso"], p["action_head"]["mean" if self.continuous else "Dense_0"]

    def apply(self, params: Any, observation):
        """actor_network.apply(params, observation) -> distribution (mava/evaluator.py:182-183)."""
        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        av = observation.agents_view
        lead = av.shape[:-1]
        x = av.reshape(-1, av.shape[-1]).contiguous().float()
        flat = flat.contiguous()
        logits = ops.mlp_forward(flat[: self.num_mlp_params], self.din, self.n_out, x)
        if self.continuous:
            return TanhNormal(logits.view(*lead, self.n_out), self._log_std(flat), self.action_head.min_scale)
        mask = observation.action_mask
        return Categorical(logits.view(*lead, self.n_out), None if mask is None else mask.reshape(*lead, self.n_out))


class FeedForwardValueNet(_FeedForwardNet):
    """mava/networks.py:186-207: head Dense(1) init orthogonal(1.0); centralised => global_state."""

    head_scale = 1.0

    def __init__(self, torso: MLPTorso, centralised_critic: bool, input_dim: int):
        super().__init__(input_dim, 1)
        torso.require([HIDDEN, HIDDEN])
        self.torso, self.centralised_critic = torso, centralised_critic

    def _assemble(self, torso, head):
        return {"torso": torso, "Dense_0": head}

    def _disassemble(self, p):
        return p["torso"], p["Dense_0"]

    def apply(self, params: Any, observation) -> torch.Tensor:
        if self.centralised_critic:
            if not isinstance(observation, ObservationGlobalState):
                # mava/networks.py:196-197
                raise ValueError("Global state must be provided to the centralised critic.")
            x = observation.global_state
        else:
            x = observation.agents_view
        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        lead = x.shape[:-1]
        v = ops.mlp_forward(flat.contiguous(), self.din, 1, x.reshape(-1, x.shape[-1]).contiguous().float())
        return v.view(*lead)
