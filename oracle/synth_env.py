"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

NumPy restatement of the synthetic RWARE-shaped environment (mava_amd/csrc/synth_rware.hip),
bit for bit on the same Philox4x32-10 stream.  The wrapper semantics follow
mava/wrappers/observation.py:41-53 (agent one-hot id on agents_view only),
mava/wrappers/jumanji.py:53-59,128-143 (global_state = concatenated raw views, team reward
repeated per agent), mava/wrappers/auto_reset_wrapper.py:88-101 (terminal step returns the reset
observation) and mava/wrappers/episode_metrics.py:78-111 (episode_return / episode_length /
is_terminal_step).  RWARE itself (Jumanji) is not available: observations are synthetic.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .philox import ENV_STREAM, philox4x32_10, u01_open


class SynthRware:
    def __init__(self, E: int, A: int, O: int = 66, n_actions: int = 5, time_limit: int = 500, seed: int = 42,
                 env_offset: int = 0, gs_tiles: int = 1, state_dim: int = 0, reward_mode: str = "random"):
        self.E, self.A, self.O, self.nA = E, A, O, n_actions
        self.time_limit, self.seed, self.env_offset, self.gs_tiles = time_limit, seed, env_offset, gs_tiles
        self.state_dim = state_dim
        self.reward_mode = reward_mode  # "random" | "match" (action-dependent team reward, see synth_rware.hip)
        self.step_count = np.zeros((E, A), np.int32)
        self.run_return = np.zeros(E, np.float32)
        self.run_length = np.zeros(E, np.int32)
        self.ep_return = np.zeros(E, np.float32)
        self.ep_length = np.zeros(E, np.int32)

    def _observe(self, t: int) -> Dict[str, np.ndarray]:
        E, A, O = self.E, self.A, self.O
        slo, shi = self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF
        env_id = (np.arange(E, dtype=np.uint64) + np.uint64(self.env_offset)).astype(np.uint32)
        ent = (env_id[:, None].astype(np.uint64) * np.uint64(A) + np.arange(A, dtype=np.uint64)[None, :]).astype(np.uint32)
        av = np.zeros((E, A, A + O), np.float32)
        av[:, np.arange(A), np.arange(A)] = 1.0
        cm = philox4x32_10(ent, t, 0xFFFF, ENV_STREAM, slo, shi)
        av[:, :, A + 0] = (cm[0] % np.uint32(10)).astype(np.float32)
        av[:, :, A + 1] = (cm[1] % np.uint32(10)).astype(np.float32)
        for c in range((O - 2 + 15) // 16):
            w = philox4x32_10(ent, t, c, ENV_STREAM, slo, shi)
            for q in range(16):
                f = 2 + 16 * c + q
                if f < O:
                    byte = (w[q >> 2] >> np.uint32(8 * (q & 3))) & np.uint32(0xFF)
                    av[:, :, A + f] = (byte < 51).astype(np.float32)
        raw = av[:, :, A:]
        if self.state_dim > 0:  # independent state vector per env: entity id = 0x80000000 | env
            S = self.state_dim
            sent = (np.uint32(0x80000000) | env_id).astype(np.uint32)
            st = np.zeros((E, S), np.float32)
            cs = philox4x32_10(sent, t, 0xFFFF, ENV_STREAM, slo, shi)
            st[:, 0] = (cs[0] % np.uint32(10)).astype(np.float32)
            if S > 1:
                st[:, 1] = (cs[1] % np.uint32(10)).astype(np.float32)
            for c in range((S - 2 + 15) // 16):
                w = philox4x32_10(sent, t, c, ENV_STREAM, slo, shi)
                for q in range(16):
                    f = 2 + 16 * c + q
                    if f < S:
                        st[:, f] = (((w[q >> 2] >> np.uint32(8 * (q & 3))) & np.uint32(0xFF)) < 51).astype(np.float32)
            gs = st.reshape(E, 1, S).repeat(self.gs_tiles, 1)
        else:
            gs = raw.reshape(E, 1, A * O).repeat(self.gs_tiles, 1)
        mask = np.ones((E, A, self.nA), bool)
        if self.nA > 1:
            mask[:, :, 1] = ~((cm[2] & np.uint32(0xFF)) < 51)
        return {"agents_view": av, "global_state": gs, "action_mask": mask}

    def reset(self, t: int):
        obs = self._observe(t)
        self.step_count[:] = 0
        self.run_return[:] = 0
        self.run_length[:] = 0
        self.ep_return[:] = 0
        self.ep_length[:] = 0
        obs["step_count"] = self.step_count.copy()
        return obs

    def step(self, t: int, action=None):
        """Returns (obs, reward (E,A), done (E,A), info).  reward_mode "random": actions do not influence the stream;
        "match": team reward = fraction of agents whose action equals (first coordinate of the observation generated at
        step t - 1, the one they acted on) mod n_actions."""
        E, A = self.E, self.A
        slo, shi = self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF
        env_id = (np.arange(E, dtype=np.uint64) + np.uint64(self.env_offset)).astype(np.uint32)
        ev = philox4x32_10(env_id, t, 0, ENV_STREAM ^ 1, slo, shi)
        rew = (u01_open(ev[0]) < np.float32(0.02)).astype(np.float32)
        if self.reward_mode == "match":
            ent = (env_id[:, None].astype(np.uint64) * np.uint64(A) + np.arange(A, dtype=np.uint64)[None, :]).astype(np.uint32)
            pc = philox4x32_10(ent, t - 1, 0xFFFF, ENV_STREAM, slo, shi)
            target = ((pc[0] % np.uint32(10)) % np.uint32(self.nA)).astype(np.int64)
            hits = (np.asarray(action).reshape(E, A).astype(np.int64) == target).sum(1)
            rew = hits.astype(np.float32) / np.float32(A)
        sc_new = self.step_count[:, 0] + 1
        term = (sc_new >= self.time_limit) | (u01_open(ev[1]) < np.float32(0.002))
        obs = self._observe(t)
        self.step_count[:] = np.where(term, 0, sc_new)[:, None]
        obs["step_count"] = self.step_count.copy()
        new_ret = self.run_return + rew
        new_len = self.run_length + 1
        ret_info = np.where(term, new_ret, self.ep_return).astype(np.float32)
        len_info = np.where(term, new_len, self.ep_length).astype(np.int32)
        self.run_return = np.where(term, 0, new_ret).astype(np.float32)
        self.run_length = np.where(term, 0, new_len).astype(np.int32)
        self.ep_return, self.ep_length = ret_info, len_info
        info = {"episode_return": ret_info.copy(), "episode_length": len_info.copy(), "is_terminal_step": term.copy()}
        return obs, np.repeat(rew[:, None], A, 1), np.repeat(term[:, None], A, 1), info
