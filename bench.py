#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the ff_mappo hot path (rollout -> GAE -> minibatch PPO)
on synthetic RWARE tiny-4ag-shaped inputs, 4096 envs per GPU, rollout_length 128, 4 epochs x 2
minibatches (BASELINE.json configs[1]).  One "step" = one full PPO update.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  env-steps are counted exactly like the reference
(mava/systems/ppo/ff_mappo.py:468-474,496-504): D * updates * T * U * E over the wall time of the
updates with device syncs on both sides; `value` is the MEDIAN of --repeats such learn(state) calls of --steps updates.
Kernel durations for the roofline objects are measured live with HIP events on the launch stream in a separate
instrumented pass of the same updates (outside every timed call).  The default single-GPU run then times the secondary
claims (exact-f32 arithmetic, the reference's U = 2 replica split, ff_ippo, BASELINE config 4) the same way and attaches
them under "secondary".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
# f16x2 arithmetic: every f32-equivalent product costs three f16 MFMAs, so the algorithmic (f32-equivalent) FLOP rate
# is priced against a third of the dense f16 peak
F16X2_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0
HBM_PEAK_GBS = 8000.0         # HBM3E spec (6290 GB/s measured float4 copy)


def _ev_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def usable_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  os.cpu_count() reports the
    whole host and oversubscribes a quota-limited container by an order of magnitude."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 64))


def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: N child processes of this script with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set (what torch.distributed.run would export), rank 0's stdout (the JSON line) relayed.
    Returns non-zero if any rank fails.  The parent never initialises the GPU."""
    import subprocess

    n_dev = torch.cuda.device_count()  # counts devices without initialising HIP on this image
    if n_dev < n and os.environ.get("MAVA_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {n} but only {n_dev} GPU(s) visible; RCCL needs one GPU per rank "
              f"(MAVA_DIST_BACKEND=gloo rehearses several ranks on one card)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; the parent polls EVERY child: if any rank dies (e.g. on RCCL init) the others
    # would wait in their next collective forever, so the first non-zero exit (or the overall timeout) stops the rest
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("MAVA_BENCH_TIMEOUT", "1500"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc == 0 for rc in rcs):
            failed = bad or None
            break
        if time.monotonic() > deadline:
            failed = [("timeout", -1)]
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(5)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if failed:
        print(f"bench.py: ranks failed: {failed}", file=sys.stderr)
        return 1
    return 0


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _sync_barrier(world: int) -> None:
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        torch.cuda.synchronize()


def build_learner(system: str, env_name: str, scenario: str, envs_per_gpu: int, U: int, matmul: str, steps: int, warmup: int, dev,
                  continuous=None):
    """learner_setup of one workload through the drop-in boundary: (learn, state, learner, cfg)."""
    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_ippo, ff_mappo, rec_ippo, rec_mappo

    E = envs_per_gpu // U
    cfg = compose(f"default_{system}", [f"env={env_name}", f"env/scenario={scenario}", f"arch.num_envs={E}",
                                         f"system.update_batch_size={U}"])
    if continuous is not None:
        cfg.network.action_head = {"_target_": "mava.networks.ContinuousActionHead"}
        cfg.env.scenario.task_config.num_agents = continuous["agents"]
        cfg.env.synthetic = {"obs_dim": continuous["obs_dim"], "num_actions": continuous["action_dim"]}
    # the timed call is learn(state) itself, num_updates_per_eval = steps updates per call, timed like
    # run_experiment does (mava/systems/ppo/ff_mappo.py:496-504: wall time of learn + block_until_ready)
    cfg.system.num_updates_per_eval = max(steps, 1)
    cfg.system.num_updates = max(4 * steps + warmup, 1)
    cfg.system.matmul_mode = matmul
    central = system.endswith("mappo")
    mod = {"ff_mappo": ff_mappo, "ff_ippo": ff_ippo, "rec_mappo": rec_mappo, "rec_ippo": rec_ippo}[system]
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, _actor_network, state = mod.learner_setup(env, (42, 43, 44), cfg, device=dev)
    return learn, state, learn.learner, cfg


def time_learn(learn, state, L, steps: int, warmup: int, repeats: int, world: int):
    """`warmup` untimed updates, then `repeats` timed learn(state) calls of exactly `steps` updates each, every call
    bracketed by a barrier + device synchronisation on both sides (the reference's own window, ff_mappo.py:496-504).
    Returns the per-call wall times (seconds, this rank) and the last output."""
    for i in range(warmup):
        L.update(0)
        torch.cuda.synchronize()
    # untimed learn() calls on top of the warm-up updates until the device has been busy for >= 0.5 s (at least one call):
    # the first calls through the boundary pay one-off costs (permutation / statistics buffers, the caching allocator's
    # growth for the stacked output leaves) and a fresh box holds low clocks for the first few hundred milliseconds of
    # load - both showed as 1.5 - 3x slower first calls next to identical later ones
    t_w = time.perf_counter()
    while True:
        state = learn(state).learner_state
        torch.cuda.synchronize()
        if time.perf_counter() - t_w >= 0.5:
            break
    times, out_state = [], None
    for _ in range(repeats):
        _sync_barrier(world)
        t0 = time.perf_counter()
        out_state = learn(state)
        _sync_barrier(world)
        times.append(time.perf_counter() - t0)
        state = out_state.learner_state
        assert tuple(out_state.train_metrics["total_loss"].shape[:2]) == (1, max(steps, 1))
        assert bool(torch.isfinite(out_state.train_metrics["total_loss"]).all()), "non-finite loss in the timed updates"
    return times, out_state


def _median(xs):
    ys = sorted(xs)
    return ys[len(ys) // 2]


def _graph_time_us(fn_of_set, n_sets: int, dev, n_launch: int = 20, reps: int = 5) -> float:
    """Median device time per launch (us) of `n_launch` launches captured in one HIP graph (a Python launch takes ~10 us,
    longer than the kernels timed here); launch i works on buffer set i % n_sets."""
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for i in range(3):
            fn_of_set(0)
        with torch.cuda.graph(graph, stream=side):
            for i in range(n_launch):
                fn_of_set(i % n_sets)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0_.record()
        graph.replay()
        e1_.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0_.elapsed_time(e1_) / n_launch)
    return _median(ts)


def _load_traffic(names):
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["kernels"], f"profiles/{name} (committed rocprofv3 --pmc pass, not measured in this run)"
        except (OSError, ValueError, KeyError):
            continue
    return {}, None


FF_TRAFFIC = ("r03_v3_pmc_traffic.json", "r03_v2_pmc_traffic.json", "r03_v1_pmc_traffic.json", "r02_v5_pmc_traffic.json")
REC_TRAFFIC = ("r03_rec2_pmc_traffic.json", "r03_rec1_pmc_traffic.json", "r02_rec3_pmc_traffic.json", "r02_rec2_pmc_traffic.json")


def ff_rooflines(L, matmul: str, timers: dict, timer_steps: int, default_shape: bool, aggregated_off: bool, continuous: bool, dev,
                 with_gae: bool = True) -> dict:
    """roofline objects of a feed-forward workload from the HIP-event pairs of the instrumented pass."""
    out = {}
    T, A, E = L.T, L.A, L.E
    # HBM traffic per launch: counters cannot be read from inside this process, so `traffic` is NOT measured in this
    # run - it is attached from the committed rocprofv3 PMC pass of the same workload named in `traffic_source`
    # (regenerated per round with tools/profile_round.sh; stale once a kernel changes after that pass).
    traffic, traffic_source = _load_traffic(FF_TRAFFIC) if default_shape else ({}, None)
    tms = {k: _ev_ms(v) for k, v in timers.items()}
    avg = {k: sum(v) / len(v) for k, v in tms.items() if v}
    rows = L.Rb * A  # agent rows per minibatch launch
    # With a centralised critic on a shared global state (x_share == A) the critic kernel evaluates each (t,e)
    # row once and back-propagates the sum of its agents' loss gradients (MAVA_CTX_CRITIC_AGGREGATION,
    # default on): its EXECUTED matrix work is 1/A of the reference's A identical passes.  Rooflines are
    # priced on executed FLOPs.
    aggregated = bool(getattr(L, "critic_share", 1) == A and 1 < A <= 8 and not aggregated_off)
    rows_c = L.Rb if aggregated else rows
    # algorithmic FLOPs per row (2*MAC; bwd = 2*fwd - first-layer dX)
    fwd_c = 2 * (L.Oc * 128 + 128 * 128 + 128)
    flop_c = (3 * fwd_c - 2 * L.Oc * 128) * rows_c
    fwd_a = 2 * (L.Oa * 128 + 128 * 128 + 128 * L.nA)
    flop_a = (3 * fwd_a - 2 * L.Oa * 128) * rows
    tf_c = flop_c / (avg["critic_grad"] * 1e-3) / 1e12
    tf_a = flop_a / (avg["actor_grad"] * 1e-3) / 1e12
    # which kernels ran: the split-f16 ones (ppo_train_h2.hip) for the widths they instantiate, else exact f32
    h2_a = matmul == "f16x2" and L.Oa <= 287 and not continuous
    h2_c = matmul == "f16x2" and L.Oc <= 287
    peak_a = F16X2_PEAK_TFLOPS if h2_a else F32_MFMA_PEAK_TFLOPS
    peak_c = F16X2_PEAK_TFLOPS if h2_c else F32_MFMA_PEAK_TFLOPS
    # the discrete actor up to 127 inputs / 16 actions runs on the eight-wave kernel (ppo_train_w8.hip) unless the handle's
    # MAVA_CTX_TRAIN_VARIANT is 1
    w8_a = h2_a and L.Oa + 1 <= 128 and L.nA <= 16 and (L.ctx.get(L.ctx.TRAIN_VARIANT) & 1) == 0
    # ... and so does the value network on inputs up to 127 wide when no agents are aggregated into a row (ff_ippo's critic)
    w8_c = h2_c and L.Oc + 1 <= 128 and not aggregated and (L.ctx.get(L.ctx.TRAIN_VARIANT) & 1) == 0

    def kname(h2, net):
        if (net == "actor" and w8_a) or (net == "critic" and w8_c):
            return f"ppo_train_w8_kernel<{net}> (fused fwd+loss+bwd+dW, eight waves, 3 f16 MFMAs per product; peak = dense f16 peak / 3)"
        if h2:
            return f"ppo_train_h2_kernel<{net}> (fused fwd+loss+bwd+dW, 3 f16 MFMAs per product; peak = dense f16 peak / 3)"
        return f"ppo_train_kernel<{net}> (fused fwd+loss+bwd+dW, exact-f32 MFMA)"

    def tkey(h2, net):
        if net == "actor" and w8_a:
            return next((k for k in traffic if k.startswith("ppo_train_w8_kernel")), None)
        if net == "critic" and w8_c:
            return None  # (no committed counter pass for this secondary shape)
        return next((k for k in traffic if k.startswith("ppo_train_h2_kernel" if h2 else "ppo_train_kernel")
                     and (("true" in k.split(",")[2]) == (net == "actor") if h2 else net in k)), None)
    roof_c = {"kernel": kname(h2_c, "critic") + (", agents of a row aggregated" if aggregated else ""),
              "bound": "mfma", "achieved": tf_c, "peak": peak_c, "unit": "TFLOP/s", "frac": tf_c / peak_c,
              "traffic": traffic.get(tkey(h2_c, "critic"), {}).get("hbm_bytes_corrected"),
              "mfma_busy_pmc": traffic.get(tkey(h2_c, "critic"), {}).get("mfma_util"),
              "traffic_source": traffic_source,
              "avg_launch_ms": avg["critic_grad"], "flop_per_launch": flop_c, "rows_per_launch": rows_c}
    roof_a = {"kernel": kname(h2_a, "actor"), "bound": "mfma", "achieved": tf_a, "peak": peak_a, "unit": "TFLOP/s",
              "frac": tf_a / peak_a,
              "traffic": traffic.get(tkey(h2_a, "actor"), {}).get("hbm_bytes_corrected"),
              "mfma_busy_pmc": traffic.get(tkey(h2_a, "actor"), {}).get("mfma_util"),
              "traffic_source": traffic_source,
              "avg_launch_ms": avg["actor_grad"], "flop_per_launch": flop_a, "rows_per_launch": rows}
    # "roofline" = the dominant kernel of the update (most time per update)
    if avg["actor_grad"] >= avg["critic_grad"]:
        out["roofline"], out["roofline_critic"] = roof_a, roof_c
    else:
        out["roofline"], out["roofline_actor"] = roof_c, roof_a
    if with_gae:
        # GAE (target: >= 60 % of the 8 TB/s HBM roofline).  With the fused rollout (the default) the scan runs in the
        # rollout kernel's tail on data each workgroup has just written - there is no GAE launch in the loop.  The standalone
        # kernel (mava_gae_f32: recurrent systems, per-step rollout, other callers) is measured here on the rollout's own
        # outputs, outside the timed region, through captured HIP graphs of 20 launches.  The headline figure is the COLD
        # one - 10 buffer sets (357 MB, more than the 256 MB Infinity Cache) cycled, operands from HBM, as in a training
        # loop; the cache-resident relaunch figure is reported beside it, labelled as such; and a device-to-device copy
        # of the same byte count through the same harness is the ceiling this footprint allows (`copy_ceiling_us`).
        from mava_amd import ops as _ops

        rep0 = L.reps[0]
        EA_ = E * A
        NSET = 10
        mk = lambda: ((torch.randn(T, EA_, device=dev), torch.randn(T, EA_, device=dev),
                       (torch.rand(T, EA_, device=dev) < 0.002).to(torch.uint8), torch.randn(EA_, device=dev), 0.99, 0.95),
                      (torch.empty(T, EA_, device=dev), torch.empty(T, EA_, device=dev)))
        sets = [((rep0.reward.view(T, EA_), rep0.value.view(T, EA_), rep0.done.view(T, EA_), rep0.last_val.view(EA_), 0.99, 0.95),
                 (torch.empty(T, EA_, device=dev), torch.empty(T, EA_, device=dev)))] + [mk() for _ in range(NSET - 1)]
        run = lambda i: _ops.gae(*sets[i][0], out=sets[i][1])
        warm_us, cold_us = _graph_time_us(run, 1, dev), _graph_time_us(run, NSET, dev)
        gae_bytes = 17 * T * E * A + 4 * E * A
        # the copy moves gae_bytes in total: half of them read, half written, like the kernel's 9 read + 8 written of 17
        srcs = [torch.empty(gae_bytes // 8, device=dev).normal_() for _ in range(NSET)]
        dsts = [torch.empty_like(x) for x in srcs]
        copy_cold_us = _graph_time_us(lambda i: dsts[i].copy_(srcs[i]), NSET, dev)
        copy_warm_us = _graph_time_us(lambda i: dsts[i].copy_(srcs[i]), 1, dev)
        del sets, srcs, dsts
        gbs_cold, gbs_warm = gae_bytes / cold_us / 1e3, gae_bytes / warm_us / 1e3
        out["roofline_gae"] = {
            "kernel": "gae_kernel (standalone mava_gae_f32; the fused rollout runs GAE in its own tail, no launch in the loop)",
            "bound": "hbm", "achieved": gbs_cold, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_cold / HBM_PEAK_GBS,
            "avg_launch_us": cold_us, "bytes_per_launch": gae_bytes,
            "state": "operands from HBM: 20 launches per captured HIP graph cycling through 10 buffer sets (357 MB > the 256 MB Infinity Cache), incl. the ~1.5 us launch boundary each",
            "copy_ceiling_us": copy_cold_us, "copy_ceiling_gbs": gae_bytes / copy_cold_us / 1e3,
            "frac_of_copy_ceiling": copy_cold_us / cold_us,
            "copy_ceiling": "torch device-to-device copy of the same byte count (half read, half written), same harness, same buffer cycling",
            "cache_resident": {"achieved": gbs_warm, "frac": gbs_warm / HBM_PEAK_GBS, "avg_launch_us": warm_us,
                               "copy_us": copy_warm_us,
                               "state": "relaunch on ONE buffer set: operands resident in the Infinity Cache - cache bandwidth, not HBM"},
            "traffic": traffic.get(next((k for k in traffic if k.startswith("gae_kernel")), ""), {}).get("hbm_bytes_corrected"),
        }
    if "clip_adam" in avg:
        adam_bytes = 28 * L.P
        out["roofline_adam"] = {"kernel": "clip_adam_kernel", "bound": "hbm (launch-bound at 77K params)",
                                "achieved": adam_bytes / (avg["clip_adam"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "avg_launch_us": avg["clip_adam"] * 1e3}
    out["kernel_ms_per_step"] = {k: round(sum(v) / timer_steps, 4) for k, v in tms.items()}
    return out


def rec_rooflines(L, matmul: str, rec_timers: dict, timer_steps: int) -> dict:
    """Recurrent systems: the kernel with the most time per update is the roofline object.  The GRU scans
    are priced on BOTH rooflines - algorithmic FLOPs (h.W_h: 2*128*384 per row-step) and algorithmic HBM bytes per
    row-step (forward: read gi 1536 + done, write h 512 + saved gates 2048 + h_prev 512; BPTT: read saved 2048 + h_prev 512
    + dh_out 512, write dgi 1536 + the n third of dgh 512 - since round 3 the r and z thirds, equal to dgi's, are not
    stored) - the binding one is `bound`."""
    out = {}
    T = L.T
    tms = {k: _ev_ms(v) for k, v in rec_timers.items()}
    per_update = {k: sum(v) / timer_steps for k, v in tms.items()}
    avg = {k: sum(v) / len(v) for k, v in tms.items() if v}
    out["kernel_ms_per_step"] = {k: round(v, 4) for k, v in per_update.items()}
    # launches are keyed "<kernel>:<sequences per step>" (actor: Rm = envs x agents of a minibatch; the centralised
    # critic on a shared state: one sequence per env)
    scans = [k for k in per_update if k.startswith("gru_scan") or k.startswith("rollout_rec")]
    if not scans:
        return out
    dom_key = max((k for k in per_update if k.startswith("gru_scan")), key=lambda k: per_update[k])
    dom, seqs = dom_key.split(":")[0], int(dom_key.split(":")[1])
    rows_avg = T * seqs                   # row-steps per launch
    flop = 2.0 * 128 * 384 * rows_avg
    by = {"gru_scan_fwd": 1536 + 1 + 512 + 2048 + 512, "gru_scan_bwd": 2048 + 512 + 512 + 1536 + 512}[dom] * rows_avg
    t_s = avg[dom_key] * 1e-3
    tf, gbs = flop / t_s / 1e12, by / t_s / 1e9
    # the matrix peak of the arithmetic the scan runs in: exact-f32 MFMAs, or three f16 MFMAs per product (f16x2)
    mfma_peak = F16X2_PEAK_TFLOPS if matmul == "f16x2" else F32_MFMA_PEAK_TFLOPS
    hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tf / mfma_peak
    rec_traffic, rec_src = None, None
    kname = f"{dom}_h2_kernel" if matmul == "f16x2" else f"{dom}_kernel"
    if seqs == 8192 and T == 128:
        tr, src = _load_traffic(REC_TRAFFIC)
        grid = f" grid={min(256, seqs // 32)}"
        rec_traffic = (tr.get(kname + grid) or tr.get(kname + "<true>" + grid) or {}).get("hbm_bytes_corrected")
        rec_src = src if rec_traffic else None
    if hbm_frac >= mfma_frac:
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": hbm_frac, "traffic": rec_traffic, "mfma_frac": mfma_frac}
    else:
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": tf, "peak": mfma_peak,
                           "unit": "TFLOP/s", "frac": mfma_frac, "traffic": rec_traffic, "hbm_frac": hbm_frac}
    out["roofline"].update({"avg_launch_ms": avg[dom_key], "row_steps_per_launch": rows_avg, "traffic_source": rec_src})
    return out


def instrumented_pass(L, n_updates: int, world: int):
    """The same updates again with a HIP event pair around every kernel launch on the launch stream (NOT part of any
    `value`: ~430 event pairs per update cost ~1.5 ms).  Returns (ff timers | None, rec timers | None, updates run)."""
    if hasattr(L, "_timed"):
        L.timers = {}
        for _ in range(n_updates):
            L.update(0)
        torch.cuda.synchronize()
        tm, L.timers = L.timers, None
        _sync_barrier(world)
        return tm, None, n_updates
    from mava_amd import _lib as _mava_lib_mod

    _mava_lib_mod.TIMERS = rec_timers = {}
    for _ in range(n_updates):
        L.update(0)
    torch.cuda.synchronize()
    _mava_lib_mod.TIMERS = None
    _sync_barrier(world)
    return None, rec_timers, n_updates


def secondary_workload(name: str, dev, *, system, env_name, scenario, envs_per_gpu, U, matmul, steps, warmup, repeats=3,
                       roofline=False) -> dict:
    """One secondary claim, driver-timed like the primary (median of `repeats` learn() calls of `steps` updates)."""
    t_start = time.perf_counter()
    learn, state, L, cfg = build_learner(system, env_name, scenario, envs_per_gpu, U, matmul, steps, warmup, dev)
    times, _ = time_learn(learn, state, L, steps, warmup, repeats, 1)
    el = _median(times)
    env_steps = steps * L.T * L.U * L.E
    res = {"workload": name, "value": env_steps / el, "unit": "env-steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
           "warmup": warmup, "repeats": repeats, "ms_per_step_min": 1e3 * min(times) / steps,
           "ms_per_step_max": 1e3 * max(times) / steps, "matmul_mode": matmul,
           "config": f"{system} {env_name} {scenario}, {envs_per_gpu} envs/GPU (update_batch_size={U} x num_envs={L.E}), "
                     f"rollout_length={L.T}, ppo_epochs={L.K}, num_minibatches={L.M}, agents={L.A}, obs={L.Oa}/{L.Oc}, actions={L.nA}"}
    if roofline:
        ff_t, rec_t, n = instrumented_pass(L, 2, 1)
        if rec_t:
            res.update(rec_rooflines(L, matmul, rec_t, n))
        elif ff_t:
            res.update(ff_rooflines(L, matmul, ff_t, n, False, False, False, dev, with_gae=False))
    del learn, state, L
    torch.cuda.empty_cache()
    log(f"secondary [{name}]: {res['value']:,.0f} env-steps/s, {res['ms_per_step']:.2f} ms per update ({time.perf_counter() - t_start:.1f} s incl. setup)")
    return res


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed learn() calls of --steps updates each; value / ms_per_step are those of the MEDIAN call")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (arch.num_envs * update_batch_size)")
    ap.add_argument("--update-batch-size", type=int, default=1)
    ap.add_argument("--scenario", default="tiny-4ag")
    ap.add_argument("--system", default="ff_mappo", choices=["ff_mappo", "ff_ippo", "rec_mappo", "rec_ippo"])
    ap.add_argument("--env", default="rware", choices=["rware", "smax"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads that the default single-GPU run times after the primary line's region")
    ap.add_argument("--no-critic-aggregation", action="store_true",
                    help="centralised critic: one network pass per AGENT row (the reference's A identical passes) instead of one per (t,e) row")
    ap.add_argument("--action-head", default="discrete", choices=["discrete", "continuous"],
                    help="continuous: network.action_head=ContinuousActionHead on a MaBrax-shaped synthetic env "
                         "(--agents, --obs-dim, --action-dim; SURVEY 8f N4) - a secondary workload, not the headline metric")
    ap.add_argument("--matmul", default="f16x2", choices=["f32", "f16x2"],
                    help="arithmetic of the fused PPO gradient kernels: exact-f32 MFMA, or split-f16 operands (3 f16 MFMAs per "
                         "product, f32 accumulate; MAVA_CTX_MATMUL_MODE of the learner's context handle)")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--obs-dim", type=int, default=27)
    ap.add_argument("--action-dim", type=int, default=2)
    args = ap.parse_args()
    continuous = args.action_head == "continuous"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Invoked without a launcher (`python bench.py --gpus N`): start the N ranks ourselves, one fresh child process
        # per GPU, BEFORE anything in this process touches the GPU (no exec of a process that initialised HIP).
        raise SystemExit(_spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU; the modulo only matters when several ranks are rehearsed on a one-GPU box (gloo)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MAVA_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()  # n_gpus of the JSON line = the ranks the process group really has
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but the job has {world} rank(s)", file=sys.stderr)

    U = args.update_batch_size
    cont = dict(agents=args.agents, obs_dim=args.obs_dim, action_dim=args.action_dim) if continuous else None
    learn, state, L, cfg = build_learner(args.system, args.env, args.scenario, args.envs, U, args.matmul, args.steps, args.warmup,
                                         dev, continuous=cont)
    E, T, A, K, M = L.E, L.T, L.A, L.K, L.M
    central = args.system.endswith("mappo")
    if args.no_critic_aggregation:
        L.ctx.set(L.ctx.CRITIC_AGGREGATION, 0)
    log(f"setup done: {world} rank(s), E={E} U={U} T={T} A={A} Oa={L.Oa} Oc={L.Oc} matmul={args.matmul}")
    # ---- the timed region: `repeats` learn(state) calls of exactly args.steps updates each through the drop-in boundary
    # (adopt + updates + the returned state's leaf stacking), no instrumentation inside; the line reports the MEDIAN call
    repeats = max(1, args.repeats)
    times, _out_state = time_learn(learn, state, L, max(args.steps, 1), args.warmup, repeats, world)
    rank_times = None
    if world > 1:
        import torch.distributed as dist

        # per call: the slowest rank's time (MAX over ranks); per-rank values of the median call are kept for the line
        te = torch.tensor(times, dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(gathered, te)
        allt = torch.stack(gathered, 0)           # (rank, call)
        times = [float(x) for x in allt.max(0).values.tolist()]
        med_call = sorted(range(len(times)), key=lambda i: times[i])[len(times) // 2]
        rank_times = [float(x) for x in allt[:, med_call].tolist()]
    elapsed = _median(times)
    env_steps = world * args.steps * T * U * E  # ff_mappo.py:468-474
    value = env_steps / elapsed
    log(f"timed region: {repeats} x {args.steps} updates, median call {elapsed:.3f} s -> {value:,.0f} env-steps/s "
        f"(calls: {', '.join(f'{t:.3f}' for t in times)} s)")

    out = {
        "metric": ("env-steps/sec (whole node), ff_mappo RWARE tiny-4ag"
                   if args.system == "ff_mappo" and args.env == "rware" and not continuous
                   else f"env-steps/sec (whole node), {args.system} continuous-action synthetic (MaBrax-shaped)" if continuous
                   else f"env-steps/sec (whole node), {args.system} {args.env} {args.scenario}"),
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.matmul == "f32" else "f32 via 2 x f16 split operands on the MFMA (3 products, f32 accumulate)",
        "matmul_mode": args.matmul,
        "data": "synthetic",
        "repeats": repeats,
        "timing": f"median of {repeats} learn(state) calls of {args.steps} updates each (every call bracketed by barrier + device sync; max over ranks per call)",
        "ms_per_step_min": 1e3 * min(times) / max(args.steps, 1),
        "ms_per_step_max": 1e3 * max(times) / max(args.steps, 1),
        "config": {"workload": (f"{args.system} ContinuousActionHead, MaBrax-shaped synthetic obs, " if continuous else
                                f"{args.system} {args.env.upper()} {args.scenario}-shaped synthetic obs, ") + f"{E * U} envs/GPU "
                               f"(update_batch_size={U} x num_envs={E}), rollout_length={T}, ppo_epochs={K}, "
                               f"num_minibatches={M}, agents={A}, obs={L.Oa}/{L.Oc}, actions={L.nA}, "
                               f"one step = one PPO update = {T * U * E} env-steps per GPU",
                   "parallelism": f"dp{world}"},
    }
    if world > 1:
        # evidence that the collective really spanned N ranks: the process group's own size and backend, and the per-rank
        # wall times of the reported (median) call
        import torch.distributed as dist

        out["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                              "rank_elapsed_s_min": min(rank_times), "rank_elapsed_s_max": max(rank_times),
                              "rank_elapsed_s": rank_times, "rccl_cus_reserved": getattr(L, "rccl_cus", None),
                              "gradient_slabs": getattr(L, "n_slab", None)}

    default_shape = (args.envs == 4096 and args.update_batch_size == 1 and args.scenario == "tiny-4ag"
                     and args.system == "ff_mappo" and not continuous)
    # ---- instrumented pass (NOT part of `value`): per-kernel HIP-event timings for the roofline objects
    if not args.no_kernel_timers:
        ff_t, rec_t, n_t = instrumented_pass(L, max(1, min(args.steps, 5 if hasattr(L, "_timed") else 2)), world)
        if rank == 0 and ff_t:
            out.update(ff_rooflines(L, args.matmul, ff_t, n_t, default_shape, args.no_critic_aggregation, continuous, dev))
        if rank == 0 and rec_t:
            out.update(rec_rooflines(L, args.matmul, rec_t, n_t))

    Oa, Oc, nA = L.Oa, L.Oc, L.nA
    # ---- secondary claims, driver-timed in the same run (single-GPU default run only; each a fresh learner with its own
    # context handle, after the primary learner's buffers are released)
    if rank == 0 and world == 1 and default_shape and not args.no_secondary:
        del learn, state, L, _out_state
        torch.cuda.empty_cache()
        sec = {}
        try:
            sec["exact_f32"] = secondary_workload(
                "the headline workload on the exact-f32 kernels everywhere (system.matmul_mode=f32: per-step rollout, f32 MFMA)", dev,
                system="ff_mappo", env_name="rware", scenario="tiny-4ag", envs_per_gpu=4096, U=1, matmul="f32", steps=10, warmup=2)
            sec["reference_default_replicas"] = secondary_workload(
                "the reference's default replica split (system/ppo/ff_mappo.yaml:13 update_batch_size=2): U=2 x E=2048", dev,
                system="ff_mappo", env_name="rware", scenario="tiny-4ag", envs_per_gpu=4096, U=2, matmul=args.matmul, steps=10, warmup=2)
            sec["ff_ippo"] = secondary_workload(
                "ff_ippo (decentralised critic), same shape", dev,
                system="ff_ippo", env_name="rware", scenario="tiny-4ag", envs_per_gpu=4096, U=1, matmul=args.matmul, steps=10, warmup=2)
            sec["config4_rec_mappo_smax_3s5z"] = secondary_workload(
                "BASELINE config 4: rec_mappo, SMAX 3s5z shape, 2048 envs, seq_len 128", dev,
                system="rec_mappo", env_name="smax", scenario="3s5z", envs_per_gpu=2048, U=1, matmul=args.matmul, steps=4, warmup=2,
                roofline=True)
        except Exception as ex:  # a secondary must never cost the primary line
            sec["error"] = f"{type(ex).__name__}: {ex}"
            log(f"secondary workloads stopped: {sec['error']}")
        out["secondary"] = sec

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.system.startswith("rec") and not continuous:
        from oracle import rec_cpu_loop

        cores = usable_cores()
        cpu_E = max(2 * M, min(args.cpu_envs, 32))
        log(f"cpu baseline (recurrent) on {cores} usable cores, {cpu_E} envs ...")
        res = rec_cpu_loop.run(E=cpu_E, A=A, Oa=Oa, Oc=Oc, nA=nA, T=T, K=K, M=M, updates=8, warmup=0, threads=cores,
                               max_seconds=args.cpu_seconds, shared_state=central)
        out["cpu_baseline"] = {
            "value": res["env_steps_per_sec"], "unit": "env-steps/s", "cores": res["threads"], "kind": "port",
            "sample": f"oracle/rec_cpu_loop.py (torch-CPU f32 restatement of the same recurrent PPO update loop) at {cpu_E} envs x "
                      f"{T} steps, {res['env_steps']} env-steps in {res['seconds']:.1f} s; substitute for Mava's JAX CPU path",
        }
        log(f"cpu baseline: {res['env_steps_per_sec']:,.0f} env-steps/s")

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.system.startswith("ff") and not continuous:
        from oracle import cpu_loop

        cores = usable_cores()
        log(f"cpu baseline on {cores} usable cores (os.cpu_count()={os.cpu_count()}) ...")
        res = cpu_loop.run(E=args.cpu_envs, A=A, O=Oa - A, nA=nA, T=T, K=K, M=M, updates=64, warmup=1,
                           threads=cores, max_seconds=args.cpu_seconds)
        out["cpu_baseline"] = {
            "value": res["env_steps_per_sec"], "unit": "env-steps/s", "cores": res["threads"], "kind": "port",
            "sample": f"oracle/cpu_loop.py (torch-CPU f32 restatement of the same ff_mappo update loop, same synthetic "
                      f"input distributions) at {args.cpu_envs} envs x {T} steps, {res['env_steps']} env-steps in "
                      f"{res['seconds']:.1f} s; substitute for Mava's JAX CPU path, which is not installable here",
        }
        log(f"cpu baseline: {res['env_steps_per_sec']:,.0f} env-steps/s")

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
