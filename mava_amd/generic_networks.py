"""The GENERAL network path: torsos the fused kernels do not instantiate.

Reference: mava/networks.py:39-58 (MLPTorso: any layer_sizes, activation relu | tanh, use_layer_norm), :61-85 (CNNTorso:
nn.Conv 'SAME' -> [LayerNorm(use_scale=False)] -> activation per layer, then the last three axes collapsed), :88-169 (the
action heads, incl. ContinuousActionHead(independent_std=False): log_std = Dense(action_dim)(embedding)), :172-207
(FeedForwardActor / FeedForwardValueNet).

The default configuration (network/mlp.yaml: [128, 128], relu, no layer norm) runs on the fused gradient / rollout kernels
(mava_amd/networks.py); everything else runs here, LAYER-WISE on T32 matrices with the recurrent path's product kernels
(mava_rec_dense_f32, mava_rec_xty_f32: exact f32 or f16x2 per system.matmul_mode) and csrc/generic_layers.hip between them.
Wide layers are column-blocked (x_ld / y_ld): K in blocks of 384, N in blocks of 128.  Row counts must be multiples of 32.

Flat parameter layout: per torso layer [kernel (K x N; a conv kernel (k, k, Cin, Cout) flattened to (k*k*Cin, Cout)) | bias (N)
| layer-norm bias (N) if use_layer_norm], then per head [kernel | bias], then the raw log_std vector of an
observation-independent continuous head.  tree() exposes Flax-shaped views of it.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import check, ctx_ptr, launch, lib, ptr, stream_ptr
from .networks import _orthogonal_

ACT = {"relu": 1, "tanh": 2}


class CNNTorso:
    """mava/networks.py:61-85 (configuration object; the kernels are driven by GenericNet)."""

    def __init__(self, channel_sizes: Sequence[int], kernel_sizes: Sequence[int], strides: Sequence[int], activation: str = "relu",
                 use_layer_norm: bool = False, **_: Any):
        self.channel_sizes, self.kernel_sizes, self.strides = list(channel_sizes), list(kernel_sizes), list(strides)
        self.activation, self.use_layer_norm = activation, bool(use_layer_norm)
        if not (len(self.channel_sizes) == len(self.kernel_sizes) == len(self.strides)) or not self.channel_sizes:
            raise ValueError("CNNTorso: channel_sizes, kernel_sizes and strides must have the same non-zero length")
        if activation not in ACT:
            raise NotImplementedError(f"activation {activation!r}: the reference knows relu and tanh (networks.py:334-340)")


class GenericMLPTorso:
    """mava/networks.py:39-58 without the restrictions of the fused kernels."""

    def __init__(self, layer_sizes: Sequence[int], activation: str = "relu", use_layer_norm: bool = False, **_: Any):
        self.layer_sizes, self.activation, self.use_layer_norm = [int(v) for v in layer_sizes], activation, bool(use_layer_norm)
        if activation not in ACT:
            raise NotImplementedError(f"activation {activation!r}: the reference knows relu and tanh (networks.py:334-340)")
        if not self.layer_sizes or any(v < 1 for v in self.layer_sizes):
            raise ValueError(f"MLPTorso: bad layer_sizes {self.layer_sizes}")


def torso_from_config(cfg: Any):
    kw = {k: v for k, v in dict(cfg).items() if k != "_target_"}
    if str(dict(cfg).get("_target_", "")).endswith("CNNTorso"):
        return CNNTorso(**kw)
    return GenericMLPTorso(**kw)


def is_default_mlp(cfg: Any) -> bool:
    d = dict(cfg)
    return (not str(d.get("_target_", "MLPTorso")).endswith("CNNTorso") and list(d.get("layer_sizes", [128, 128])) == [128, 128]
            and d.get("activation", "relu") == "relu" and not d.get("use_layer_norm", False))


def _same_out(n: int, stride: int) -> int:
    return -(-n // stride)


class _Layer:
    __slots__ = ("kind", "K", "N", "ln", "act", "w", "b", "lnb", "name", "geo", "rows_mul", "src_flat")


class GenericNet:
    """torso (MLP or CNN) + Dense heads, forward / backward on T32 matrices."""

    ctx = None  # the owning learner's context handle (_lib.Ctx: arithmetic of the products); None = exact f32

    def __init__(self, torso, din: int, heads: List[Tuple[str, int, float]], obs_shape: Optional[Tuple[int, int, int]] = None,
                 raw_tail: int = 0):
        self.torso, self.din, self.heads_spec, self.raw_tail = torso, int(din), heads, int(raw_tail)
        self.layers: List[_Layer] = []
        off = 0
        act, ln = ACT[torso.activation], torso.use_layer_norm
        self.is_cnn = isinstance(torso, CNNTorso)
        if self.is_cnn:
            if obs_shape is None or math.prod(obs_shape) != self.din:
                raise ValueError(f"CNNTorso needs the observation shape (H, W, C) with H*W*C = {self.din}, got {obs_shape}")
            H, W, C = (int(v) for v in obs_shape)
            mul = 1
            for i, (co, k, st) in enumerate(zip(torso.channel_sizes, torso.kernel_sizes, torso.strides)):
                L = _Layer()
                L.kind, L.K, L.N, L.ln, L.act, L.name = "conv", k * k * C, int(co), ln, act, f"Conv_{i}"
                L.geo, L.src_flat = (H, W, C, int(k), int(st)), int(i == 0)
                H, W = _same_out(H, st), _same_out(W, st)
                mul = H * W
                L.rows_mul = mul
                L.w, off = off, off + L.K * L.N
                L.b, off = off, off + L.N
                L.lnb = off if ln else -1
                off += L.N if ln else 0
                C = int(co)
                self.layers.append(L)
            self.P_last, self.C_last = H * W, C
            self.feat = H * W * C
        else:
            K = self.din
            for i, n in enumerate(torso.layer_sizes):
                L = _Layer()
                L.kind, L.K, L.N, L.ln, L.act, L.name, L.rows_mul, L.geo, L.src_flat = "dense", K, int(n), ln, act, f"Dense_{i}", 1, None, 0
                L.w, off = off, off + L.K * L.N
                L.b, off = off, off + L.N
                L.lnb = off if ln else -1
                off += L.N if ln else 0
                K = int(n)
                self.layers.append(L)
            self.feat = K
        self.heads: List[_Layer] = []
        for name, n_out, _scale in heads:
            L = _Layer()
            L.kind, L.K, L.N, L.ln, L.act, L.name, L.rows_mul, L.geo, L.src_flat = "head", self.feat, int(n_out), False, 0, name, 1, None, 0
            L.w, off = off, off + L.K * L.N
            L.b, off = off, off + L.N
            L.lnb = -1
            self.heads.append(L)
        self.num_net_params = off
        self.num_params = off + self.raw_tail
        self.max_n = max([L.N for L in self.layers + self.heads] + [1])

    # ----------------------------------------------------------------------------------- parameters
    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        """orthogonal(sqrt 2) dense torso kernels (networks.py:54), flax nn.Conv default lecun-normal kernels, zero biases,
        head kernels orthogonal(head scale)."""
        gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        flat = torch.zeros(self.num_params, dtype=torch.float32)
        for L in self.layers:
            w = flat[L.w : L.w + L.K * L.N].view(L.K, L.N)
            if L.kind == "conv":
                w.copy_(torch.randn((L.K, L.N), generator=gen) / math.sqrt(L.K))
            else:
                _orthogonal_(w, math.sqrt(2.0), gen)
        for L, (_n, _o, scale) in zip(self.heads, self.heads_spec):
            _orthogonal_(flat[L.w : L.w + L.K * L.N].view(L.K, L.N), scale, gen)
        return flat.to(device) if device is not None else flat

    def _leaf_specs(self):
        """[(tree path, shape, offset)]"""
        out = []
        for i, L in enumerate(self.layers):
            shape = (L.geo[3], L.geo[3], L.geo[2], L.N) if L.kind == "conv" else (L.K, L.N)
            out.append((("torso", L.name, "kernel"), shape, L.w))
            out.append((("torso", L.name, "bias"), (L.N,), L.b))
            if L.ln:
                out.append((("torso", f"LayerNorm_{i}", "bias"), (L.N,), L.lnb))
        return out

    def torso_tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        tree: Dict[str, Any] = {}
        for path, shape, off in self._leaf_specs():
            v = flat[off : off + math.prod(shape)].view(shape)
            tree.setdefault(path[1], {})[path[2]] = v.expand(*lead, *shape) if lead else v
        return tree

    def head_leaf(self, flat: torch.Tensor, h: int, lead: Tuple[int, ...] = ()) -> Dict[str, torch.Tensor]:
        L = self.heads[h]
        k, b = flat[L.w : L.w + L.K * L.N].view(L.K, L.N), flat[L.b : L.b + L.N]
        return {"kernel": k.expand(*lead, L.K, L.N) if lead else k, "bias": b.expand(*lead, L.N) if lead else b}

    def load_torso_tree(self, torso: Dict[str, Any], out: torch.Tensor) -> None:
        for path, shape, off in self._leaf_specs():
            leaf = torch.as_tensor(torso[path[1]][path[2]])
            while leaf.dim() > len(shape):
                leaf = leaf[0]
            out[off : off + math.prod(shape)].view(shape).copy_(leaf)

    def load_head_leaf(self, leaf: Dict[str, Any], h: int, out: torch.Tensor) -> None:
        L = self.heads[h]
        for key, shape, off in (("kernel", (L.K, L.N), L.w), ("bias", (L.N,), L.b)):
            v = torch.as_tensor(leaf[key])
            while v.dim() > len(shape):
                v = v[0]
            out[off : off + math.prod(shape)].view(shape).copy_(v)

    # ----------------------------------------------------------------------------------- products
    def _dense(self, x_ptr: int, x_ld: int, K: int, w: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor, N: int, rows: int,
               accumulate: bool = False, what: str = "gen_dense") -> None:
        """y (T32 rows x N) (+)= x (T32, x_ld features per tile) @ w (K x N) [+ bias], column-blocked."""
        L, s = lib(), stream_ptr()
        for n0 in range(0, N, 128):
            nb = min(128, N - n0)
            for k0 in range(0, K, 384):
                kb = min(384, K - k0)
                launch(what, L.mava_rec_dense_f32, ctx_ptr(self.ctx), x_ptr + 4 * 32 * k0, 0, None, 0, 0, 0, 1, x_ld, int(accumulate or k0 > 0),
                       w.data_ptr() + 4 * (k0 * N + n0), N, (bias.data_ptr() + 4 * n0) if (bias is not None and k0 == 0) else None, None,
                       y.data_ptr() + 4 * 32 * n0, N, kb, nb, rows, 0, s)

    def _xty(self, x_ptr: int, x_ld: int, K: int, y: torch.Tensor, N: int, rows: int, slabs: torch.Tensor, gw: torch.Tensor,
             gb: Optional[torch.Tensor], scale: float, accumulate: bool, gb_accumulate: Optional[bool] = None,
             y_tail: Optional[torch.Tensor] = None, y_split: int = 0) -> None:
        """gw (K x N) (+)= scale * x^T y ; gb (N) (+)= scale * colsum(y).  y_tail: T32 (rows x (N - y_split)) holding the
        features of y from y_split on (a multiple of 128: the column blocks below never straddle it)."""
        from . import ops

        L, s = lib(), stream_ptr()
        for n0 in range(0, N, 128):
            nb = min(128, N - n0)
            for k0 in range(0, K, 128):
                kb = min(128, K - k0)
                want_b = int(gb is not None and k0 == 0)
                if y_tail is not None and n0 >= y_split:
                    y_ptr, y_ld = y_tail.data_ptr() + 4 * 32 * (n0 - y_split), N - y_split
                else:
                    y_ptr, y_ld = y.data_ptr() + 4 * 32 * n0, N
                launch("gen_xty", L.mava_rec_xty_f32, ctx_ptr(self.ctx), x_ptr + 4 * 32 * k0, 0, None, 0, 0, 0, 1, x_ld, y_ptr, y_ld, None, 0, 0,
                       kb, nb, rows, want_b, scale, ptr(slabs), slabs.shape[1], slabs.shape[0], s)
                blk = torch.empty(kb * nb, device=y.device)
                ops.slab_reduce(slabs, kb * nb, blk)
                dst = gw.view(K, N)[k0 : k0 + kb, n0 : n0 + nb]
                if accumulate:
                    dst.add_(blk.view(kb, nb))
                else:
                    dst.copy_(blk.view(kb, nb))
                if want_b:
                    tail = slabs[:, kb * nb : kb * nb + nb].contiguous()
                    ops.slab_reduce(tail, nb, gb[n0 : n0 + nb], accumulate=accumulate if gb_accumulate is None else gb_accumulate)

    # ----------------------------------------------------------------------------------- forward / backward
    def workspace(self, rows: int, device, training: bool) -> "GenericWorkspace":
        return GenericWorkspace(self, rows, device, training)

    def forward(self, flat: torch.Tensor, ws: "GenericWorkspace", x_ext: Optional[torch.Tensor], x_share: int, idx, Rm: int, E: int,
                A: int, T: int = 1, x_t32: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """x_ext: external row-major (T, E, A / x_share, din) source, gathered like mava_rec_dense_f32's row-major input
        (batch row q = t*Rm + m, m = local_env*A + agent, env = idx[local_env] or identity); or x_t32: the input already as a
        T32 matrix of din features (the recurrent post torso reads the hidden states).  Returns the heads' T32 outputs, or
        [features] for a torso without heads."""
        L, s = lib(), stream_ptr()
        rows = T * Rm
        assert rows == ws.rows
        if x_t32 is not None:
            ws.x_src, ws.x_src_ld = x_t32, self.din
        else:
            check(L.mava_rec_gather_t32_f32(ptr(x_ext), ptr(idx), Rm, E, A, x_share, self.din, self.din, rows, ws.kp, ptr(ws.xin), s),
                  "gather")
            ws.x_src, ws.x_src_ld = ws.xin, ws.kp
        cur, cur_ld, cur_rows = ws.x_src, ws.x_src_ld, rows
        for i, ly in enumerate(self.layers):
            w = flat[ly.w : ly.w + ly.K * ly.N]
            b = flat[ly.b : ly.b + ly.N]
            if ly.kind == "conv":
                H, W, C, k, st = ly.geo
                check(L.mava_t32_im2col_f32(ptr(cur), ly.src_flat, rows, H, W, C, k, st, ptr(ws.col[i]), s), "im2col")
                x_ptr, x_ld, lrows = ws.col[i].data_ptr(), ly.K, rows * ly.rows_mul
            else:
                x_ptr, x_ld, lrows = cur.data_ptr(), cur_ld, cur_rows
            self._dense(x_ptr, x_ld, ly.K, w, b, ws.z[i], ly.N, lrows)
            check(L.mava_t32_norm_act_f32(ptr(ws.z[i]), ly.N, lrows, int(ly.ln), ptr(flat[ly.lnb : ly.lnb + ly.N]) if ly.ln else None,
                                          ly.act, ptr(ws.y[i]), ptr(ws.xhat[i]) if ly.ln else None, ptr(ws.rstd[i]) if ly.ln else None, s),
                  "norm_act")
            cur, cur_ld, cur_rows = ws.y[i], ly.N, lrows
        if self.is_cnn:
            check(L.mava_t32_flatten_f32(ptr(cur), rows, self.P_last, self.C_last, 1, ptr(ws.feat), s), "flatten")
            cur, cur_ld = ws.feat, self.feat
        ws.feat_in = cur
        if not self.heads:
            return [cur]
        outs = []
        for h, hd in enumerate(self.heads):
            self._dense(cur.data_ptr(), cur_ld, hd.K, flat[hd.w : hd.w + hd.K * hd.N], flat[hd.b : hd.b + hd.N], ws.out[h], hd.N, rows)
            outs.append(ws.out[h])
        return outs

    def backward(self, flat: torch.Tensor, ws: "GenericWorkspace", d_outs: List[torch.Tensor], grad_out: torch.Tensor, accumulate: bool,
                 grad_scale: float = 1.0, d_feat: Optional[torch.Tensor] = None, dx_out: Optional[torch.Tensor] = None) -> None:
        """d_outs[h]: T32 gradient w.r.t. head h's output (in units of grad_scale); a torso without heads takes d_feat, the
        gradient w.r.t. its features, instead.  grad_out: flat gradient (true units).  dx_out: receives the gradient w.r.t.
        the (T32) input when given."""
        L, s = lib(), stream_ptr()
        rows = ws.rows
        inv = 1.0 / grad_scale
        feat, feat_ld = ws.feat_in, self.feat
        if d_feat is not None:
            ws.dfeat.copy_(d_feat[: ws.dfeat.numel()])
        # heads
        for h, hd in enumerate(self.heads):
            self._xty(feat.data_ptr(), feat_ld, hd.K, d_outs[h], hd.N, rows, ws.slabs, grad_out[hd.w : hd.w + hd.K * hd.N],
                      grad_out[hd.b : hd.b + hd.N], inv, accumulate)
            wt = flat[hd.w : hd.w + hd.K * hd.N].view(hd.K, hd.N).t().contiguous()
            self._dense(d_outs[h].data_ptr(), hd.N, hd.N, wt, None, ws.dfeat, hd.K, rows, accumulate=h > 0)
        d_cur = ws.dfeat
        if self.is_cnn:
            check(L.mava_t32_flatten_f32(ptr(ws.dfeat), rows, self.P_last, self.C_last, 0, ptr(ws.dy[len(self.layers) - 1]), s), "unflatten")
            d_cur = ws.dy[len(self.layers) - 1]
        for i in range(len(self.layers) - 1, -1, -1):
            ly = self.layers[i]
            lrows = rows * ly.rows_mul
            check(L.mava_t32_norm_act_bwd_f32(ptr(d_cur), ptr(ws.y[i]), ly.N, lrows, int(ly.ln), ptr(ws.xhat[i]) if ly.ln else None,
                                              ptr(ws.rstd[i]) if ly.ln else None, ly.act, ptr(ws.dz[i]), ptr(ws.dzin[i]), s), "norm_act_bwd")
            if ly.ln:  # layer-norm bias: column sum of dz
                from . import ops

                check(L.mava_t32_colsum_f32(ptr(ws.dz[i]), ly.N, lrows, inv, ptr(ws.slabs), ws.slabs.shape[1], ws.slabs.shape[0], s), "colsum")
                ops.slab_reduce(ws.slabs, ly.N, grad_out[ly.lnb : ly.lnb + ly.N], accumulate=accumulate)
            if ly.kind == "conv":
                x_ptr, x_ld = ws.col[i].data_ptr(), ly.K
            elif i == 0:
                x_ptr, x_ld = ws.x_src.data_ptr(), ws.x_src_ld
            else:
                x_ptr, x_ld = ws.y[i - 1].data_ptr(), self.layers[i - 1].N
            self._xty(x_ptr, x_ld, ly.K, ws.dzin[i], ly.N, lrows, ws.slabs, grad_out[ly.w : ly.w + ly.K * ly.N],
                      grad_out[ly.b : ly.b + ly.N], inv, accumulate)
            if i == 0 and dx_out is None:
                break
            wt = flat[ly.w : ly.w + ly.K * ly.N].view(ly.K, ly.N).t().contiguous()
            if i == 0:  # gradient w.r.t. the T32 input (dense first layer only)
                assert ly.kind == "dense", "dx_out is implemented for dense first layers"
                self._dense(ws.dzin[0].data_ptr(), ly.N, ly.N, wt, None, dx_out, ly.K, lrows)
                break
            if ly.kind == "conv":
                self._dense(ws.dzin[i].data_ptr(), ly.N, ly.N, wt, None, ws.dcol[i], ly.K, lrows)
                H, W, C, k, st = ly.geo
                check(L.mava_t32_col2im_f32(ptr(ws.dcol[i]), ly.src_flat, rows, H, W, C, k, st, ptr(ws.dy[i - 1]), s), "col2im")
            else:
                self._dense(ws.dzin[i].data_ptr(), ly.N, ly.N, wt, None, ws.dy[i - 1], ly.K, lrows)
            d_cur = ws.dy[i - 1]


class GenericWorkspace:
    def __init__(self, net: GenericNet, rows: int, device, training: bool):
        if rows % 32:
            raise ValueError(f"the general network path needs row counts that are multiples of 32, got {rows}")
        self.rows = rows
        f = lambda n: torch.empty(int(n), device=device)
        self.kp = net.din if net.is_cnn else -(-net.din // 32) * 32  # (im2col reads the gathered tiles at their true width)
        self.xin = f(rows * self.kp)
        self.z, self.y, self.xhat, self.rstd, self.col = [], [], [], [], []
        self.dz, self.dzin, self.dy, self.dcol = [], [], [], []
        for ly in net.layers:
            lr = rows * ly.rows_mul
            self.z.append(f(lr * ly.N))
            self.y.append(f(lr * ly.N))
            self.xhat.append(f(lr * ly.N) if ly.ln else None)
            self.rstd.append(f(lr) if ly.ln else None)
            self.col.append(f(lr * ly.K) if ly.kind == "conv" else None)
            if training:
                self.dz.append(f(lr * ly.N))
                self.dzin.append(f(lr * ly.N) if ly.ln else self.dz[-1])
                self.dy.append(f(lr * ly.N))
                self.dcol.append(f(lr * ly.K) if ly.kind == "conv" else None)
        self.feat = f(rows * net.feat) if net.is_cnn else None
        self.feat_in = None
        self.out = [f(rows * hd.N) for hd in net.heads]
        if training:
            self.dfeat = f(rows * net.feat)
            self.dout = [f(rows * hd.N) for hd in net.heads]
            max_rows = rows * max([ly.rows_mul for ly in net.layers] + [1])
            n_slab = max(1, min(256, max_rows // 32))
            min_tiles = rows // 32  # every X^T Y block must own a tile of the smallest product
            self.slabs = torch.zeros((max(1, min(n_slab, min_tiles)), 128 * 128 + 128 + 8), device=device)
            self.loss_partials = torch.zeros((1024, 2), device=device)


class _GenericFF:
    """Shared part of the general feed-forward actor / critic: Flax-shaped trees over GenericNet's flat layout."""

    net: GenericNet

    @property
    def ctx(self):
        return self.net.ctx

    @ctx.setter
    def ctx(self, value) -> None:  # FFLearner hands its context handle to the networks it drives
        self.net.ctx = value

    @property
    def num_params(self) -> int:
        return self.net.num_params

    def first_leaf(self, tree: Dict[str, Any]) -> torch.Tensor:
        return tree["params"]["torso"][self.net.layers[0].name]["kernel"]

    def _apply_rows(self, flat: torch.Tensor, x: torch.Tensor) -> List[torch.Tensor]:
        """Head outputs [(rows, n_out)] for row-major x (rows, din); rows are padded to the 32-row tiles.  Off the timed
        path (evaluator, tests)."""
        from .rec_networks import t32_to_rows

        R = int(x.shape[0])
        Rp = -(-R // 32) * 32
        xp = torch.zeros((1, Rp, 1, self.net.din), device=x.device)
        xp[0, :R, 0] = x.float()
        ws = self.net.workspace(Rp, x.device, training=False)
        outs = self.net.forward(flat.contiguous().float(), ws, xp, 1, None, Rp, Rp, 1)
        return [t32_to_rows(o, hd.N, Rp)[:R] for o, hd in zip(outs, self.net.heads)]


class GenericActor(_GenericFF):
    """mava/networks.py:172-183 (FeedForwardActor) over any torso and action head."""

    def __init__(self, torso, action_head, obs_dim: int, obs_shape=None):
        from .networks import ContinuousActionHead

        self.action_head = action_head
        self.continuous = isinstance(action_head, ContinuousActionHead)
        self.independent_std = bool(getattr(action_head, "independent_std", True))
        self.n_out = int(action_head.action_dim)
        heads = [("mean" if self.continuous else "Dense_0", self.n_out, 0.01)]
        if self.continuous and not self.independent_std:
            heads.append(("log_std", self.n_out, 0.01))
        self.net = GenericNet(torso, obs_dim, heads, obs_shape, raw_tail=self.n_out if (self.continuous and self.independent_std) else 0)
        self.din = int(obs_dim)

    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        return self.net.init_flat(seed, device)  # (the raw log_std tail stays zero: networks.py:141)

    def log_std(self, flat: torch.Tensor) -> torch.Tensor:
        return flat[self.net.num_net_params : self.net.num_net_params + self.n_out]

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        head: Dict[str, Any] = {self.net.heads[0].name: self.net.head_leaf(flat, 0, lead)}
        if self.continuous:
            if self.independent_std:
                ls = self.log_std(flat)
                head["log_std"] = ls.expand(*lead, self.n_out) if lead else ls
            else:
                head["log_std"] = self.net.head_leaf(flat, 1, lead)
        return {"params": {"torso": self.net.torso_tree(flat, lead), "action_head": head}}

    def flat_from_tree(self, tree: Dict[str, Any], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = tree["params"]
        if out is None:
            out = torch.empty(self.num_params, dtype=torch.float32, device=self.first_leaf(tree).device)
        self.net.load_torso_tree(p["torso"], out)
        self.net.load_head_leaf(p["action_head"][self.net.heads[0].name], 0, out)
        if self.continuous:
            if self.independent_std:
                ls = torch.as_tensor(p["action_head"]["log_std"])
                while ls.dim() > 1:
                    ls = ls[0]
                self.log_std(out).copy_(ls)
            else:
                self.net.load_head_leaf(p["action_head"]["log_std"], 1, out)
        return out

    def apply(self, params: Any, observation):
        """actor_network.apply(params, observation) -> distribution (mava/evaluator.py:182-183)."""
        from .distributions import Categorical, TanhNormal

        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        av = observation.agents_view
        lead = av.shape[: av.dim() - (3 if (self.net.is_cnn and av.shape[-1] != self.din) else 1)]
        outs = self._apply_rows(flat, av.reshape(-1, self.din))
        if self.continuous:
            ls = self.log_std(flat.float()) if self.independent_std else outs[1].view(*lead, self.n_out)
            return TanhNormal(outs[0].view(*lead, self.n_out), ls, self.action_head.min_scale)
        mask = observation.action_mask
        return Categorical(outs[0].view(*lead, self.n_out), None if mask is None else mask.reshape(*lead, self.n_out))


class GenericCritic(_GenericFF):
    """mava/networks.py:186-207 (FeedForwardValueNet) over any torso."""

    def __init__(self, torso, centralised_critic: bool, input_dim: int, obs_shape=None):
        self.centralised_critic = centralised_critic
        self.net = GenericNet(torso, input_dim, [("Dense_0", 1, 1.0)], obs_shape)
        self.din = int(input_dim)

    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        return self.net.init_flat(seed, device)

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        return {"params": {"torso": self.net.torso_tree(flat, lead), "Dense_0": self.net.head_leaf(flat, 0, lead)}}

    def flat_from_tree(self, tree: Dict[str, Any], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = tree["params"]
        if out is None:
            out = torch.empty(self.num_params, dtype=torch.float32, device=self.first_leaf(tree).device)
        self.net.load_torso_tree(p["torso"], out)
        self.net.load_head_leaf(p["Dense_0"], 0, out)
        return out

    def apply(self, params: Any, observation) -> torch.Tensor:
        from .types import ObservationGlobalState

        if self.centralised_critic:
            if not isinstance(observation, ObservationGlobalState):
                raise ValueError("Global state must be provided to the centralised critic.")  # mava/networks.py:196-197
            x = observation.global_state
        else:
            x = observation.agents_view
        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        lead = x.shape[: x.dim() - (3 if (self.net.is_cnn and x.shape[-1] != self.din) else 1)]
        return self._apply_rows(flat, x.reshape(-1, self.din))[0].view(*lead)
