"""Recurrent networks behind the reference's module names (mava/networks.py:238-331: ScannedRNN,
RecurrentActor, RecurrentValueNet) and their sequence forward / backward as a chain of HIP kernels.

Flat parameter layout (float32, one contiguous vector per network):
    [Wpre (din,128) | bpre | Wi (128,384 = ir|iz|in) | bi (384) | Wh (128,384 = hr|hz|hn) | bhn (128)
     | Wpost (128,128) | bpost | Whead (128,n_out) | bhead]
The Flax-shaped tree exposed to callers holds VIEWS of that vector:
    params/pre_torso/Dense_0, params/ScannedRNN_0/GRUCell_0/{ir,iz,in,hr,hz,hn}, params/post_torso/Dense_0,
    params/action_head/Dense_0 (actor) or params/Dense_0 (critic).

Sequence batch convention (rec_mappo.py:334-365: minibatches are env slices, all T steps): rows are
time-major, row = t*Rm + m with m = local_env*A + agent; Rm must be a multiple of 32.  Activations live in
the kernels' T32 tile layout inside a RecWorkspace.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import torch

from . import ops
from ._lib import check, ctx_ptr, launch, lib, ptr, stream_ptr
from .distributions import Categorical
from .networks import ContinuousActionHead, DiscreteActionHead, MLPTorso, _orthogonal_

H = 128
G3 = 3 * H


def rec_segments(din: int, n_out: int):
    segs, off = [], 0
    for name, shape in (("Wpre", (din, H)), ("bpre", (H,)), ("Wi", (H, G3)), ("bi", (G3,)), ("Wh", (H, G3)), ("bhn", (H,)),
                        ("Wpost", (H, H)), ("bpost", (H,)), ("Whead", (H, n_out)), ("bhead", (n_out,))):
        segs.append((name, shape, off))
        off += math.prod(shape)
    return segs, off


class RecWorkspace:
    """Activation buffers (T32) for one sequence batch of `rows` = T*Rm row-steps; shared by actor and critic."""

    def __init__(self, rows: int, n_out_max: int, device, training: bool = True, din_max: int = 0):
        f = lambda n: torch.empty(rows * n, device=device)
        self.rows = rows
        # f16x2 arithmetic: the minibatch's gathered observations as a T32 matrix (features padded to 32), written once
        # per forward pass and read by the pre-torso product and its weight-gradient product
        self.xin = f(-(-din_max // 32) * 32) if (training and din_max > 0) else None
        self.xpre, self.gi, self.hs, self.post, self.y = f(H), f(G3), f(H), f(H), f(n_out_max)
        if training:
            self.hprev, self.saved = f(H), f(4 * H)
            self.dy, self.dpost, self.dh_out, self.dgi, self.dgh, self.dxpre = f(n_out_max), f(H), f(H), f(G3), f(H), f(H)  # (dgh: n third)
            self.loss_partials = torch.zeros((1024, 2), device=device)  # one partial per loss-kernel block (4 blocks per CU)
        self.y2 = self.dy2 = None  # second head (ContinuousActionHead(independent_std=False): the log_std layer), on first use
        self._n_out_max = n_out_max

    def second_head(self, training: bool):
        if self.y2 is None:
            self.y2 = torch.empty(self.rows * self._n_out_max, device=self.hs.device)
        if training and self.dy2 is None:
            self.dy2 = torch.empty(self.rows * self._n_out_max, device=self.hs.device)
        return self.y2, self.dy2


class _RecurrentNet:
    head_scale = 1.0
    _ctx = None  # the owning learner's context handle (_lib.Ctx: arithmetic of the products / scans); None = exact f32

    @property
    def ctx(self):
        return self._ctx

    @ctx.setter
    def ctx(self, value) -> None:
        self._ctx = value
        if getattr(self, "generic", False):  # the general pre / post torsos run their products on the same handle
            self.pre.ctx = value
            self.post.ctx = value

    def __init__(self, din: int, n_out: int, hidden_state_dim: int = 128, pre_torso=None, post_torso=None, obs_shape=None,
                 heads=None):
        # hidden_state_dim = 128 (network/rnn.yaml): the register-resident scan kernels.  Any other width runs the cell one time
        # step at a time on the general layer kernels (_gru_fwd_steps / _gru_bwd_steps: a T32 dense launch + a gate kernel per
        # step) and needs the general torsos around it (RecLearner builds them for it).
        self.Hd = int(hidden_state_dim)
        if self.Hd < 1:
            raise ValueError(f"hidden_state_dim must be positive, got {hidden_state_dim}")
        self.din, self.n_out = int(din), int(n_out)
        # Torsos other than network/rnn.yaml's [128] relu (any layer sizes, tanh, layer norm: mava/networks.py:39-58) run on
        # the general layer kernels (mava_amd/generic_networks.py) around the same GRU scans; the fused acting step and the
        # fused output path are for the default torsos only.
        from .generic_networks import CNNTorso, GenericMLPTorso, GenericNet

        self.generic = isinstance(pre_torso, (GenericMLPTorso, CNNTorso)) or isinstance(post_torso, GenericMLPTorso)
        if self.Hd != H and not self.generic:
            raise ValueError("hidden_state_dim != 128 runs on the general torsos (GenericMLPTorso / CNNTorso)")
        if self.generic:
            if isinstance(post_torso, CNNTorso):
                raise ValueError("post_torso consumes the 128 hidden features: it cannot be a CNNTorso (network/rcnn.yaml: MLPTorso)")
            # network/rcnn.yaml: CNNTorso pre-torso on (H, W, C) observations (im2col + the same products, section 3.11)
            self.pre = GenericNet(pre_torso, self.din, [], obs_shape=obs_shape if isinstance(pre_torso, CNNTorso) else None)
            # heads of the post-torso: one Dense(n_out), or ContinuousActionHead(independent_std=False)'s mean and log_std layers
            Hd = self.Hd
            self.post = GenericNet(post_torso, Hd, heads or [("head", self.n_out, self.head_scale)])
            Np, off = self.pre.feat, self.pre.num_params
            segs = []
            for name, shape in (("Wi", (Np, 3 * Hd)), ("bi", (3 * Hd,)), ("Wh", (Hd, 3 * Hd)), ("bhn", (Hd,))):
                segs.append((name, shape, off))
                off += math.prod(shape)
            self.post_off = off
            self.segments, self.num_params = segs, off + self.post.num_params
            self.Np = Np
        else:
            if heads is not None and len(heads) > 1:
                raise ValueError("several heads run on the general torsos (GenericMLPTorso)")
            self.segments, self.num_params = rec_segments(self.din, self.n_out)
        self.num_net_params = self.num_params  # the continuous actor appends log_std(n_out) behind the network
        self.off = {n: (o, s) for n, s, o in self.segments}

    def seg(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o, s = self.off[name]
        return flat[o : o + math.prod(s)].view(s)

    def init_flat(self, seed: int, device=None) -> torch.Tensor:
        """orthogonal(sqrt 2) torsos (networks.py:54), flax GRUCell defaults (lecun-normal input kernels,
        orthogonal recurrent kernels, zero biases), orthogonal(head_scale) head."""
        gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        flat = torch.zeros(self.num_params, dtype=torch.float32)
        if self.generic:
            flat[: self.pre.num_params].copy_(self.pre.init_flat(seed))
            flat[self.post_off : self.post_off + self.post.num_params].copy_(self.post.init_flat(seed + 1))
            Hd = self.Hd
            self.seg(flat, "Wi").copy_(torch.randn((self.Np, 3 * Hd), generator=gen) / math.sqrt(self.Np))
            for g in range(3):
                blk = torch.empty(Hd, Hd)
                _orthogonal_(blk, 1.0, gen)
                self.seg(flat, "Wh")[:, g * Hd : (g + 1) * Hd].copy_(blk)
            return flat.to(device) if device is not None else flat
        _orthogonal_(self.seg(flat, "Wpre"), math.sqrt(2.0), gen)
        self.seg(flat, "Wi").copy_(torch.randn((H, G3), generator=gen) / math.sqrt(H))
        for g in range(3):
            blk = torch.empty(H, H)
            _orthogonal_(blk, 1.0, gen)
            self.seg(flat, "Wh")[:, g * H : (g + 1) * H].copy_(blk)
        _orthogonal_(self.seg(flat, "Wpost"), math.sqrt(2.0), gen)
        _orthogonal_(self.seg(flat, "Whead"), self.head_scale, gen)
        return flat.to(device) if device is not None else flat

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        ex = (lambda v: v.expand(*lead, *v.shape)) if lead else (lambda v: v)
        Wi, bi, Wh = self.seg(flat, "Wi"), self.seg(flat, "bi"), self.seg(flat, "Wh")
        if self.generic:
            D = self.Hd
            cell = {
                "ir": {"kernel": ex(Wi[:, :D]), "bias": ex(bi[:D])}, "iz": {"kernel": ex(Wi[:, D : 2 * D]), "bias": ex(bi[D : 2 * D])},
                "in": {"kernel": ex(Wi[:, 2 * D :]), "bias": ex(bi[2 * D :])}, "hr": {"kernel": ex(Wh[:, :D])},
                "hz": {"kernel": ex(Wh[:, D : 2 * D])}, "hn": {"kernel": ex(Wh[:, 2 * D :]), "bias": ex(self.seg(flat, "bhn"))},
            }
            fpost = flat[self.post_off : self.post_off + self.post.num_params]
            tree = {"pre_torso": self.pre.torso_tree(flat[: self.pre.num_params], lead), "ScannedRNN_0": {"GRUCell_0": cell},
                    "post_torso": self.post.torso_tree(fpost, lead)}
            tree.update(self._head_tree(self.post.head_leaf(fpost, 0, lead)))
            return {"params": tree}
        cell = {
            "ir": {"kernel": ex(Wi[:, :H]), "bias": ex(bi[:H])},
            "iz": {"kernel": ex(Wi[:, H : 2 * H]), "bias": ex(bi[H : 2 * H])},
            "in": {"kernel": ex(Wi[:, 2 * H :]), "bias": ex(bi[2 * H :])},
            "hr": {"kernel": ex(Wh[:, :H])},
            "hz": {"kernel": ex(Wh[:, H : 2 * H])},
            "hn": {"kernel": ex(Wh[:, 2 * H :]), "bias": ex(self.seg(flat, "bhn"))},
        }
        tree = {
            "pre_torso": {"Dense_0": {"kernel": ex(self.seg(flat, "Wpre")), "bias": ex(self.seg(flat, "bpre"))}},
            "ScannedRNN_0": {"GRUCell_0": cell},
            "post_torso": {"Dense_0": {"kernel": ex(self.seg(flat, "Wpost")), "bias": ex(self.seg(flat, "bpost"))}},
        }
        head = {"kernel": ex(self.seg(flat, "Whead")), "bias": ex(self.seg(flat, "bhead"))}
        tree.update(self._head_tree(head))
        return {"params": tree}

    def flat_from_tree(self, tree: Dict[str, Any], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Inverse of tree(): gathers a Flax-shaped parameter tree (leading replica dims allowed: replica 0 is
        taken, like unreplicate_n_dims, mava/utils/jax_utils.py:52-59) into the flat layout the kernels read."""
        p = tree["params"]
        cell = p["ScannedRNN_0"]["GRUCell_0"]
        if self.generic:
            return self._flat_from_tree_generic(p, cell, out)
        if "action_head" in p:  # discrete: Dense_0 (networks.py:106); continuous: mean + log_std (networks.py:138-141)
            head = p["action_head"]["mean"] if "mean" in p["action_head"] else p["action_head"]["Dense_0"]
        else:
            head = p["Dense_0"]

        def leaf(v, shape):
            v = torch.as_tensor(v)
            while v.dim() > len(shape):
                v = v[0]
            return v.reshape(shape)

        dev = leaf(p["pre_torso"]["Dense_0"]["kernel"], (self.din, H)).device
        flat = out if out is not None else torch.empty(self.num_params, dtype=torch.float32, device=dev)
        self.seg(flat, "Wpre").copy_(leaf(p["pre_torso"]["Dense_0"]["kernel"], (self.din, H)))
        self.seg(flat, "bpre").copy_(leaf(p["pre_torso"]["Dense_0"]["bias"], (H,)))
        for g, (ik, hk) in enumerate((("ir", "hr"), ("iz", "hz"), ("in", "hn"))):
            self.seg(flat, "Wi")[:, g * H : (g + 1) * H].copy_(leaf(cell[ik]["kernel"], (H, H)))
            self.seg(flat, "bi")[g * H : (g + 1) * H].copy_(leaf(cell[ik]["bias"], (H,)))
            self.seg(flat, "Wh")[:, g * H : (g + 1) * H].copy_(leaf(cell[hk]["kernel"], (H, H)))
        self.seg(flat, "bhn").copy_(leaf(cell["hn"]["bias"], (H,)))
        self.seg(flat, "Wpost").copy_(leaf(p["post_torso"]["Dense_0"]["kernel"], (H, H)))
        self.seg(flat, "bpost").copy_(leaf(p["post_torso"]["Dense_0"]["bias"], (H,)))
        self.seg(flat, "Whead").copy_(leaf(head["kernel"], (H, self.n_out)))
        self.seg(flat, "bhead").copy_(leaf(head["bias"], (self.n_out,)))
        if "action_head" in p and "log_std" in p["action_head"]:
            flat[self.num_net_params : self.num_net_params + self.n_out].copy_(leaf(p["action_head"]["log_std"], (self.n_out,)))
        return flat

    def _flat_from_tree_generic(self, p, cell, out):
        def leaf(v, shape):
            v = torch.as_tensor(v)
            while v.dim() > len(shape):
                v = v[0]
            return v.reshape(shape)

        if "action_head" in p:
            head = p["action_head"]["mean"] if "mean" in p["action_head"] else p["action_head"]["Dense_0"]
        else:
            head = p["Dense_0"]
        dev = torch.as_tensor(cell["hr"]["kernel"]).device
        flat = out if out is not None else torch.empty(self.num_params, dtype=torch.float32, device=dev)
        self.pre.load_torso_tree(p["pre_torso"], flat[: self.pre.num_params])
        fpost = flat[self.post_off : self.post_off + self.post.num_params]
        self.post.load_torso_tree(p["post_torso"], fpost)
        self.post.load_head_leaf(head, 0, fpost)
        dep_std = "action_head" in p and isinstance(p["action_head"].get("log_std", None), dict)  # a Dense layer's {kernel, bias}
        if dep_std:
            self.post.load_head_leaf(p["action_head"]["log_std"], 1, fpost)
        D = self.Hd
        for g, (ik, hk) in enumerate((("ir", "hr"), ("iz", "hz"), ("in", "hn"))):
            self.seg(flat, "Wi")[:, g * D : (g + 1) * D].copy_(leaf(cell[ik]["kernel"], (self.Np, D)))
            self.seg(flat, "bi")[g * D : (g + 1) * D].copy_(leaf(cell[ik]["bias"], (D,)))
            self.seg(flat, "Wh")[:, g * D : (g + 1) * D].copy_(leaf(cell[hk]["kernel"], (D, D)))
        self.seg(flat, "bhn").copy_(leaf(cell["hn"]["bias"], (D,)))
        if "action_head" in p and "log_std" in p["action_head"] and not dep_std:
            flat[self.num_net_params : self.num_net_params + self.n_out].copy_(leaf(p["action_head"]["log_std"], (self.n_out,)))
        return flat

    def _gen_ws(self, ws: RecWorkspace, training: bool):
        """General-path workspaces of this network inside a RecWorkspace (created on first use)."""
        key = (id(self), training)
        if not hasattr(ws, "gen"):
            ws.gen = {}
        if key not in ws.gen:
            dev = ws.hs.device
            ws.gen[key] = (self.pre.workspace(ws.rows, dev, training), self.post.workspace(ws.rows, dev, training),
                           torch.empty(ws.rows * self.Np, device=dev) if training else None)
        return ws.gen[key]

    def _bufs(self, ws: RecWorkspace, training: bool):
        """The GRU's activation buffers: the workspace's own (128 wide) or, for another hidden width, this network's (same names)."""
        if self.Hd == H:
            return ws
        key = (id(self), training, "gru")
        if not hasattr(ws, "gen"):
            ws.gen = {}
        if key not in ws.gen:
            from types import SimpleNamespace

            D, dev = self.Hd, ws.hs.device
            f = lambda n: torch.empty(ws.rows * n, device=dev)
            b = SimpleNamespace(gi=f(3 * D), hs=f(D), hprev=f(D), saved=f(4 * D) if training else None)
            if training:
                b.dh_out, b.dgi, b.dgh = f(D), f(3 * D), f(3 * D)
            ws.gen[key] = b
        return ws.gen[key]

    def last_hidden(self, ws: RecWorkspace, T: int, Rm: int, training: bool = False) -> torch.Tensor:
        """T32 (Rm x hidden) state after the last step of the sequence batch forward_sequence has just run."""
        D = self.Hd
        return self._bufs(ws, training).hs[(T - 1) * Rm * D : T * Rm * D]

    # ---- the cell one step at a time (hidden_state_dim != 128): csrc/generic_layers.hip mava_t32_gru_*
    def _gru_fwd_steps(self, flat, b, done_ext, h0, h0_t32, idx, T, Rm, E, A, training: bool) -> None:
        D, L, s = self.Hd, lib(), stream_ptr()
        dev = b.hs.device
        if h0_t32:
            h_in = h0
        else:  # external (E, A, D) state of the selected envs -> T32 rows (off the hot path: this whole route is the slow one)
            hv = h0.reshape(E, A, D)
            hv = hv if idx is None else hv.index_select(0, idx.long())
            h_in = rows_to_t32(hv.reshape(Rm, D).float().contiguous())
        gh = torch.empty(Rm * 3 * D, device=dev)
        Wh, bhn = self.seg(flat, "Wh"), self.seg(flat, "bhn")
        EA = E * A
        dptr = lambda t: done_ext.data_ptr() + t * EA
        check(L.mava_t32_gru_mask_f32(ptr(h_in), dptr(0), ptr(idx), E, A, D, Rm, ptr(b.hprev), s), "mava_t32_gru_mask_f32")
        f4 = lambda buf, n, t: buf.data_ptr() + 4 * t * Rm * n
        for t in range(T):
            self.pre._dense(f4(b.hprev, D, t), D, D, Wh, None, gh, 3 * D, Rm, what="gru_step(gh)")
            nxt = t + 1 < T
            check(L.mava_t32_gru_gates_f32(f4(b.gi, 3 * D, t), ptr(gh), ptr(bhn), f4(b.hprev, D, t), D, Rm, f4(b.hs, D, t),
                                           f4(b.saved, 4 * D, t) if (training and b.saved is not None) else None,
                                           f4(b.hprev, D, t + 1) if nxt else None, dptr(t + 1) if nxt else None, ptr(idx), E, A, s),
                  "mava_t32_gru_gates_f32")

    def _gru_bwd_steps(self, flat, b, done_ext, idx, T, Rm, E, A) -> None:
        D, L, s = self.Hd, lib(), stream_ptr()
        dev = b.hs.device
        WhT = self.seg(flat, "Wh").t().contiguous()
        acc = [torch.empty(Rm * D, device=dev) for _ in range(2)]
        dhp = [torch.empty(Rm * D, device=dev) for _ in range(2)]
        EA = E * A
        dptr = lambda t: done_ext.data_ptr() + t * EA
        f4 = lambda buf, n, t: buf.data_ptr() + 4 * t * Rm * n
        for t in range(T - 1, -1, -1):
            nxt = t + 1 < T
            check(L.mava_t32_gru_gates_bwd_f32(f4(b.saved, 4 * D, t), f4(b.hprev, D, t), f4(b.dh_out, D, t),
                                               ptr(acc[(t + 1) & 1]) if nxt else None, ptr(dhp[(t + 1) & 1]) if nxt else None,
                                               dptr(t + 1) if nxt else None, ptr(idx), E, A, D, Rm, f4(b.dgi, 3 * D, t),
                                               f4(b.dgh, 3 * D, t), ptr(dhp[t & 1]), s), "mava_t32_gru_gates_bwd_f32")
            if t > 0:  # gradient into the state entering step t (the reset of step t is applied where it is consumed)
                self.pre._dense(f4(b.dgh, 3 * D, t), 3 * D, 3 * D, WhT, None, acc[t & 1], D, Rm, what="gru_step(dh)")

    def _apply_sequence(self, params: Any, hstate: torch.Tensor, x: torch.Tensor, done: torch.Tensor):
        """Shared body of RecurrentActor.apply / RecurrentValueNet.apply: x (T, E, A, din), done (T, E, A),
        hstate (E, A, 128) -> (new hstate (E, A, 128), outputs (T, E, A, n_out)).  Rows are padded to a multiple
        of 32 for the tile kernels; off the timed path (evaluator, tests)."""
        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        flat = flat.contiguous().float()
        T, E, A = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        R = E * A
        Rp = -(-R // 32) * 32
        dev = x.device
        xp = torch.zeros((T, Rp, self.din), device=dev)
        xp[:, :R] = x.reshape(T, R, self.din).float()
        dp = torch.zeros((T, Rp), dtype=torch.uint8, device=dev)
        dp[:, :R] = done.reshape(T, R).to(torch.uint8)
        D = self.Hd
        hp = torch.zeros((Rp, D), device=dev)
        hp[:R] = hstate.reshape(R, D).float()
        ws = RecWorkspace(T * Rp, max(self.n_out, 1), dev, training=False)
        # the padded batch is presented as Rp single-agent "envs", identity permutation
        y = self.forward_sequence(flat, ws, xp, 1, dp, hp, False, None, T, Rp, Rp, 1, training=False)
        out = t32_to_rows(y, self.n_out, T * Rp).view(T, Rp, self.n_out)[:, :R].reshape(T, E, A, self.n_out)
        self._applied_second = (t32_to_rows(ws.y2, self.n_out, T * Rp).view(T, Rp, self.n_out)[:, :R].reshape(T, E, A, self.n_out)
                                if ws.y2 is not None else None)
        h_last = t32_to_rows(self.last_hidden(ws, T, Rp), D, Rp)[:R].reshape(E, A, D)
        return h_last, out

    # ------------------------------------------------------------------------------------- kernels
    def forward_sequence(self, flat, ws: RecWorkspace, x_ext, x_share, done_ext, h0, h0_t32, idx, T, Rm, E, A,
                         training: bool, y_out: Optional[torch.Tensor] = None, stop_after_scan: bool = False) -> torch.Tensor:
        """Runs the network over a time-major sequence batch; returns T32 outputs (ws.y).  x_ext is the external
        row-major (T, E, A/x_share.., din) tensor, done_ext (T, E, A) u8 the flags entering each step."""
        rows = T * Rm
        L = lib()
        s = stream_ptr()
        W = lambda n: ptr(self.seg(flat, n))
        if self.generic:
            from .generic_networks import GenericNet

            wpre, wpost, _ = self._gen_ws(ws, training)
            b = self._bufs(ws, training)
            feat = self.pre.forward(flat[: self.pre.num_params], wpre, x_ext, x_share, idx, Rm, E, A, T=T)[0]
            self.pre._dense(feat.data_ptr(), self.Np, self.Np, self.seg(flat, "Wi"), self.seg(flat, "bi"), b.gi, 3 * self.Hd, rows,
                            what="rec_dense(gi)")
            if self.Hd == H:
                launch(f"gru_scan_fwd:{Rm}", L.mava_gru_scan_fwd_f32, ctx_ptr(self.ctx), T, Rm, E, A, ptr(idx), ptr(done_ext), ptr(h0), int(h0_t32), W("Wh"), W("bhn"),
                       ptr(ws.gi), ptr(ws.hs), ptr(ws.hprev) if training else None, ptr(ws.saved) if training else None, s)
            else:
                self._gru_fwd_steps(flat, b, done_ext, h0, h0_t32, idx, T, Rm, E, A, training)
            if stop_after_scan:
                raise NotImplementedError("the fused output path serves the default torsos only")
            outs = self.post.forward(flat[self.post_off : self.post_off + self.post.num_params], wpost, None, 1, None, rows, rows, 1,
                                     x_t32=b.hs)
            y = outs[0]
            dst = ws.y if y_out is None else y_out  # (the losses / sampling kernels read the workspace's output buffer)
            dst.view(-1)[: y.numel()].copy_(y)
            if len(outs) > 1:  # the log_std layer's rows (networks.py:161)
                ws.second_head(training)[0][: outs[1].numel()].copy_(outs[1])
            return dst
        # pre-torso: inputs wider than 384 are consumed in column blocks (accumulating products; the weight
        # slice of one block stays register-resident), the ReLU rides on the last block
        din, Wpre = self.din, self.seg(flat, "Wpre")
        t32_in = training and ws.xin is not None and (self.ctx is not None and self.ctx.matmul_mode == "f16x2")
        kp = -(-din // 32) * 32
        if t32_in:
            launch("rec_gather", L.mava_rec_gather_t32_f32, ptr(x_ext), ptr(idx), Rm, E, A, x_share, din, din, rows, kp, ptr(ws.xin), s)
        k0 = 0
        while k0 < din:
            kc = min(384, din - k0)
            last = k0 + kc >= din
            if t32_in:  # column block k0 of a T32 tile starts k0 * 32 floats into the tile
                launch("rec_dense(pre)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), ws.xin.data_ptr() + 4 * 32 * k0, 0, None, 0, 0, 0, 1, kp, int(k0 > 0),
                       Wpre.data_ptr() + 4 * k0 * H, H, W("bpre") if k0 == 0 else None, None, ptr(ws.xpre), 0, kc, H, rows, int(last), s)
            else:
                launch("rec_dense(pre)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), x_ext.data_ptr() + 4 * k0, 1, ptr(idx), Rm, E, A, x_share, din,
                       int(k0 > 0), Wpre.data_ptr() + 4 * k0 * H, H, W("bpre") if k0 == 0 else None, None, ptr(ws.xpre), 0, kc, H, rows,
                       int(last), s)
            k0 += kc
        launch("rec_dense(gi)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), ptr(ws.xpre), 0, None, 0, 0, 0, 1, H, 0, W("Wi"), G3, W("bi"), None, ptr(ws.gi), 0, H,
               G3, rows, 0, s)
        launch(f"gru_scan_fwd:{Rm}", L.mava_gru_scan_fwd_f32, ctx_ptr(self.ctx), T, Rm, E, A, ptr(idx), ptr(done_ext), ptr(h0), int(h0_t32), W("Wh"), W("bhn"),
               ptr(ws.gi), ptr(ws.hs), ptr(ws.hprev) if training else None, ptr(ws.saved) if training else None, s)
        if stop_after_scan:  # the output path runs fused (fused_output)
            return ws.hs
        launch("rec_dense(post)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), ptr(ws.hs), 0, None, 0, 0, 0, 1, H, 0, W("Wpost"), H, W("bpost"), None,
               ptr(ws.post), 0, H, H, rows, 1, s)
        y = ws.y if y_out is None else y_out
        launch("rec_dense(head)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), ptr(ws.post), 0, None, 0, 0, 0, 1, H, 0, W("Whead"), self.n_out, W("bhead"),
               None, ptr(y), 0, H, self.n_out, rows, 0, s)
        return y

    def fused_output(self, flat, ws: RecWorkspace, idx, T, Rm, E, A, agents_per_row, is_actor: bool, mask, action, f0, f1, stats,
                     clip_eps: float, coef: float, slabs, grad_out, loss_out, accumulate: bool, grad_scale: float) -> bool:
        """post_torso -> head -> loss -> backward to ws.dh_out plus the gradients of both layers in ONE launch
        (mava_rec_out_f32, csrc/rec_out_h2.hip).  False when the library does not instantiate the shape."""
        n_out = self.n_out
        o = self.off["Wpost"][0]
        n_main = H * H + H + H * n_out + n_out
        slabs = slabs[: max(1, min(slabs.shape[0], (T * Rm) // 32))]  # every block owns a tile (the once-per-env critic has fewer)
        rc = launch("rec_out", lib().mava_rec_out_f32, T, Rm, E, A, n_out, agents_per_row, ptr(idx), ptr(ws.hs), flat.data_ptr() + 4 * o,
                    ptr(mask), ptr(action), ptr(f0), ptr(f1), ptr(stats), 0 if stats is None else stats.shape[0], clip_eps, coef,
                    grad_scale, int(is_actor), ptr(ws.dh_out), ptr(slabs), slabs.shape[1], slabs.shape[0], stream_ptr(), ok=(0, 1))
        if rc == 1:
            return False
        ops.slab_reduce2(slabs, n_main, grad_out[o : o + n_main], loss_out.numel(), loss_out, accumulate=accumulate)
        return True

    def backward_sequence(self, flat, ws: RecWorkspace, x_ext, x_share, done_ext, idx, T, Rm, E, A, slabs, grad_out,
                          accumulate: bool, grad_scale: float = 1.0, from_scan: bool = False) -> None:
        """BPTT from ws.dy (T32 d loss / d outputs, in units of `grad_scale`: a power of two, see
        mava_seq_actor_loss_f32) to the flat gradient `grad_out` (same layout as `flat`, true units)."""
        rows = T * Rm
        L = lib()
        s = stream_ptr()
        n_out = self.n_out
        if self.generic:
            from .generic_networks import GenericNet

            wpre, wpost, dfeat = self._gen_ws(ws, True)
            b = self._bufs(ws, True)
            D = self.Hd
            fpost = flat[self.post_off : self.post_off + self.post.num_params]
            gpost = grad_out[self.post_off : self.post_off + self.post.num_params]
            self.post.backward(fpost, wpost, [ws.dy] + ([ws.dy2] if len(self.post.heads) > 1 else []), gpost, accumulate, grad_scale,
                               dx_out=b.dh_out)
            if D == H:
                launch(f"gru_scan_bwd:{Rm}", L.mava_gru_scan_bwd_f32, ctx_ptr(self.ctx), T, Rm, E, A, ptr(idx), ptr(done_ext), ptr(self.seg(flat, "Wh")),
                       ptr(ws.saved), ptr(ws.hprev), ptr(ws.dh_out), ptr(ws.dgi), ptr(ws.dgh), 1, s)  # (dgh: the n third alone)
            else:
                self._gru_bwd_steps(flat, b, done_ext, idx, T, Rm, E, A)
            o = lambda n: self.off[n][0]
            Np = self.Np
            self.pre._xty(wpre.feat_in.data_ptr(), Np, Np, b.dgi, 3 * D, rows, wpre.slabs, grad_out[o("Wi") : o("Wi") + Np * 3 * D],
                            grad_out[o("bi") : o("bi") + 3 * D], 1.0 / grad_scale, accumulate)
            # dW_h, and db_hn = the n part of colsum(dgh) (taken from a scratch vector: r and z have no hidden-side bias)
            tmp_b = torch.empty(3 * D, device=flat.device)
            if D == H:  # dgh = [dgi's r and z thirds | ws.dgh]
                self.pre._xty(ws.hprev.data_ptr(), H, H, ws.dgi, G3, rows, wpre.slabs, grad_out[o("Wh") : o("Wh") + H * G3], tmp_b,
                                1.0 / grad_scale, accumulate, gb_accumulate=False, y_tail=ws.dgh, y_split=2 * H)
            else:
                self.pre._xty(b.hprev.data_ptr(), D, D, b.dgh, 3 * D, rows, wpre.slabs, grad_out[o("Wh") : o("Wh") + D * 3 * D], tmp_b,
                                1.0 / grad_scale, accumulate, gb_accumulate=False)
            gbhn = grad_out[o("bhn") : o("bhn") + D]
            if accumulate:
                gbhn.add_(tmp_b[2 * D :])
            else:
                gbhn.copy_(tmp_b[2 * D :])
            WiT = self.seg(flat, "Wi").t().contiguous()
            self.pre._dense(b.dgi.data_ptr(), 3 * D, 3 * D, WiT, None, dfeat, Np, rows, what="rec_dense(bwd)")
            self.pre.backward(flat[: self.pre.num_params], wpre, [], grad_out[: self.pre.num_params], accumulate, grad_scale, d_feat=dfeat)
            return
        # transposed weights for the dX = dY W^T products (tiny, re-materialised per call)
        WheadT = None if from_scan else self.seg(flat, "Whead").t().contiguous()  # (the fused output path needs neither)
        WpostT = None if from_scan else self.seg(flat, "Wpost").t().contiguous()
        WiT = self.seg(flat, "Wi").t().contiguous()
        d = lambda k, N, x, w, ldw, gate, y: launch(
            "rec_dense(bwd)", L.mava_rec_dense_f32, ctx_ptr(self.ctx), ptr(x), 0, None, 0, 0, 0, 1, k, 0, ptr(w), ldw, None, ptr(gate), ptr(y), 0, k, N,
            rows, 0, s)
        if not from_scan:  # (from_scan: fused_output already left ws.dh_out and the output path's gradients)
            d(n_out, H, ws.dy, WheadT, H, ws.post, ws.dpost)        # d post pre-activation (relu mask = post > 0)
            d(H, H, ws.dpost, WpostT, H, None, ws.dh_out)           # gradient reaching h_t from the output path
        launch(f"gru_scan_bwd:{Rm}", L.mava_gru_scan_bwd_f32, ctx_ptr(self.ctx), T, Rm, E, A, ptr(idx), ptr(done_ext), ptr(self.seg(flat, "Wh")), ptr(ws.saved),
               ptr(ws.hprev), ptr(ws.dh_out), ptr(ws.dgi), ptr(ws.dgh), 1, s)  # (dgh: the n third alone, r and z thirds are dgi's)
        d(G3, H, ws.dgi, WiT, H, ws.xpre, ws.dxpre)             # d pre-torso pre-activation

        def xty(x_ptr, x_rowmajor, x_ld, K, N, y, w_off, b_off, nb, xs=1, bias_slice=0, y_tail=None, y_split=0):
            """grad[w_off : w_off + K*N] (+)= X^T Y ; grad[b_off : b_off + nb] (+)= colsum(Y)[bias_slice : bias_slice + nb];
            Y's features from y_split on are those of y_tail (T32, N - y_split features per tile) when given"""
            launch("rec_xty", L.mava_rec_xty_f32, ctx_ptr(self.ctx), x_ptr, x_rowmajor, ptr(idx) if x_rowmajor else None, Rm, E, A, xs, x_ld, ptr(y), 0,
                   ptr(y_tail), y_split, N - y_split, K, N, rows, 1, 1.0 / grad_scale, ptr(slabs), slabs.shape[1], slabs.shape[0], s)
            if b_off is not None and bias_slice == 0:  # weights and the adjacent bias sums in one launch
                ops.slab_reduce2(slabs, K * N, grad_out[w_off : w_off + K * N], nb, grad_out[b_off : b_off + nb], accumulate=accumulate)
            else:
                ops.slab_reduce(slabs, K * N, grad_out[w_off : w_off + K * N], accumulate=accumulate)
                if b_off is not None:  # a window of the column sums (db_hn: the n third), summed where it lies
                    ops.slab_reduce_cols(slabs, K * N + bias_slice, nb, grad_out[b_off : b_off + nb], accumulate=accumulate)

        o = lambda n: self.off[n][0]
        if not from_scan:
            xty(ptr(ws.post), 0, H, H, n_out, ws.dy, o("Whead"), o("bhead"), n_out)
            xty(ptr(ws.hs), 0, H, H, H, ws.dpost, o("Wpost"), o("bpost"), H)
        xty(ptr(ws.xpre), 0, H, H, G3, ws.dgi, o("Wi"), o("bi"), G3)
        # dgh = [dgi's r and z thirds | ws.dgh]; db_hn: n-part of its column sums
        xty(ptr(ws.hprev), 0, H, H, G3, ws.dgi, o("Wh"), o("bhn"), H, bias_slice=2 * H, y_tail=ws.dgh, y_split=2 * H)
        t32_in = ws.xin is not None and (self.ctx is not None and self.ctx.matmul_mode == "f16x2")  # the forward pass left the gathered input in ws.xin
        kp = -(-self.din // 32) * 32
        k0 = 0
        while k0 < self.din:  # column blocks of wide inputs (each block's rows of W_pre are contiguous)
            kc = min(384, self.din - k0)
            if t32_in:
                xty(ws.xin.data_ptr() + 4 * 32 * k0, 0, kp, kc, H, ws.dxpre, o("Wpre") + k0 * H, o("bpre") if k0 == 0 else None, H)
            else:
                xty(x_ext.data_ptr() + 4 * k0, 1, self.din, kc, H, ws.dxpre, o("Wpre") + k0 * H, o("bpre") if k0 == 0 else None, H,
                    xs=x_share)
            k0 += kc


class RecurrentActor(_RecurrentNet):
    """mava/networks.py:269-294 with DiscreteActionHead or ContinuousActionHead (head init orthogonal(0.01); the
    continuous head's log_std vector, zeros, follows the network in the flat parameters)."""

    head_scale = 0.01

    def __init__(self, pre_torso: MLPTorso, post_torso: MLPTorso, action_head, obs_dim: int,
                 hidden_state_dim: int = 128, obs_shape=None):
        self.continuous = isinstance(action_head, ContinuousActionHead)
        # networks.py:137-141: log_std is a parameter vector, or (independent_std=False) a second Dense layer on the embedding
        self.independent_std = bool(getattr(action_head, "independent_std", True)) or not self.continuous
        n = int(action_head.action_dim)
        heads = None if self.independent_std else [("mean", n, self.head_scale), ("log_std", n, self.head_scale)]
        super().__init__(obs_dim, n, hidden_state_dim, pre_torso, post_torso, obs_shape, heads=heads)
        self.action_head = action_head
        if self.continuous and self.independent_std:
            self.num_params += self.n_out

    def log_std(self, flat: torch.Tensor) -> torch.Tensor:
        return flat[self.num_net_params : self.num_net_params + self.n_out]

    def tree(self, flat: torch.Tensor, lead: Tuple[int, ...] = ()) -> Dict[str, Any]:
        out = super().tree(flat, lead)
        if self.continuous and self.independent_std:
            ls = self.log_std(flat)
            out["params"]["action_head"]["log_std"] = ls.expand(*lead, self.n_out) if lead else ls
        elif self.continuous:
            fpost = flat[self.post_off : self.post_off + self.post.num_params]
            out["params"]["action_head"]["log_std"] = self.post.head_leaf(fpost, 1, lead)
        return out

    def _head_tree(self, head):
        return {"action_head": {("mean" if self.continuous else "Dense_0"): head}}

    def apply(self, params: Any, hstate: torch.Tensor, observation_done) -> Tuple[torch.Tensor, Any]:
        """actor_network.apply(params, hstate, (observation, done)) -> (hstate, distribution) with a leading
        time axis on observation and done (mava/networks.py:277-294; call site mava/evaluator.py:198-207)."""
        from .distributions import Categorical, TanhNormal

        observation, done = observation_done
        flat = params if isinstance(params, torch.Tensor) else self.flat_from_tree(params)
        h, logits = self._apply_sequence(flat, hstate, observation.agents_view, done)
        if self.continuous:
            ls = self.log_std(flat.float()) if self.independent_std else self._applied_second
            return h, TanhNormal(logits, ls, self.action_head.min_scale)
        return h, Categorical(logits, observation.action_mask)


class RecurrentValueNet(_RecurrentNet):
    """mava/networks.py:297-331 (head Dense(1) init orthogonal(1.0))."""

    head_scale = 1.0

    def __init__(self, pre_torso: MLPTorso, post_torso: MLPTorso, centralised_critic: bool, input_dim: int,
                 hidden_state_dim: int = 128, obs_shape=None):
        super().__init__(input_dim, 1, hidden_state_dim, pre_torso, post_torso, obs_shape)
        self.centralised_critic = centralised_critic

    def _head_tree(self, head):
        return {"Dense_0": head}

    def apply(self, params: Any, hstate: torch.Tensor, observation_done) -> Tuple[torch.Tensor, torch.Tensor]:
        """critic_network.apply(params, hstate, (observation, done)) -> (hstate, value (T, E, A))
        (mava/networks.py:306-331)."""
        observation, done = observation_done
        if self.centralised_critic:
            if not hasattr(observation, "global_state") or observation.global_state is None:
                raise ValueError("Global state must be provided to the centralised critic.")  # networks.py:196-197 analogue
            x = observation.global_state
        else:
            x = observation.agents_view
        h, v = self._apply_sequence(params, hstate, x, done)
        return h, v.squeeze(-1)


def t32_to_rows(src: torch.Tensor, N: int, rows: int) -> torch.Tensor:
    out = torch.empty((rows, N), device=src.device)
    check(lib().mava_t32_convert_f32(ptr(src), N, rows, 0, ptr(out), stream_ptr()), "t32_convert")
    return out


def rows_to_t32(src: torch.Tensor) -> torch.Tensor:
    rows, N = src.shape
    out = torch.empty(rows * N, device=src.device)
    check(lib().mava_t32_convert_f32(ptr(src.contiguous()), N, rows, 1, ptr(out), stream_ptr()), "t32_convert")
    return out
