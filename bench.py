#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the ff_mappo hot path (rollout -> GAE -> minibatch PPO)
on synthetic RWARE tiny-4ag-shaped inputs, 4096 envs per GPU, rollout_length 128, 4 epochs x 2
minibatches (BASELINE.json configs[1]).  One "step" = one full PPO update.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  env-steps are counted exactly like the reference
(mava/systems/ppo/ff_mappo.py:468-474,496-504): D * updates * T * U * E over the wall time of the
updates with device syncs on both sides.  Kernel durations for the roofline objects are measured
live with HIP events on the launch stream inside the timed region.
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
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed: {bad}", file=sys.stderr)
        return 1
    return 0


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (arch.num_envs * update_batch_size)")
    ap.add_argument("--update-batch-size", type=int, default=1)
    ap.add_argument("--scenario", default="tiny-4ag")
    ap.add_argument("--system", default="ff_mappo", choices=["ff_mappo", "ff_ippo", "rec_mappo", "rec_ippo"])
    ap.add_argument("--env", default="rware", choices=["rware", "smax"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-kernel-timers", action="store_true")
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

    from mava_amd import envs
    from mava_amd.config import compose
    from mava_amd.systems.ppo import ff_ippo, ff_mappo, rec_ippo, rec_mappo

    U = args.update_batch_size
    E = args.envs // U
    cfg = compose(f"default_{args.system}", [f"env={args.env}", f"env/scenario={args.scenario}", f"arch.num_envs={E}",
                                              f"system.update_batch_size={U}"])
    if continuous:
        cfg.network.action_head = {"_target_": "mava.networks.ContinuousActionHead"}
        cfg.env.scenario.task_config.num_agents = args.agents
        cfg.env.synthetic = {"obs_dim": args.obs_dim, "num_actions": args.action_dim}
    # the timed call is learn(state) itself, num_updates_per_eval = steps updates per call, timed like
    # run_experiment does (mava/systems/ppo/ff_mappo.py:496-504: wall time of learn + block_until_ready)
    cfg.system.num_updates_per_eval = max(args.steps, 1)
    cfg.system.num_updates = max(args.steps + args.warmup, 1)
    cfg.system.matmul_mode = args.matmul
    central = args.system.endswith("mappo")
    mod = {"ff_mappo": ff_mappo, "ff_ippo": ff_ippo, "rec_mappo": rec_mappo, "rec_ippo": rec_ippo}[args.system]
    env, _ = envs.make(cfg, add_global_state=central, device=dev)
    learn, actor_network, state = mod.learner_setup(env, (42, 43, 44), cfg, device=dev)
    L = learn.learner
    T, A, K, M = L.T, L.A, L.K, L.M

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    if args.no_critic_aggregation:
        L.ctx.set(L.ctx.CRITIC_AGGREGATION, 0)
    log(f"setup done: {world} rank(s), E={E} U={U} T={T} A={A} Oa={L.Oa} Oc={L.Oc} matmul={args.matmul}")
    for i in range(args.warmup):
        L.update(0)
        torch.cuda.synchronize()
        log(f"warmup update {i} done")
    # ---- the timed region: ONE learn(state) call = exactly args.steps updates through the drop-in boundary
    # (adopt + updates + the returned state's leaf stacking), no instrumentation inside
    barrier()
    t0 = time.perf_counter()
    out_state = learn(state)
    barrier()
    elapsed = time.perf_counter() - t0
    assert tuple(out_state.train_metrics["total_loss"].shape[:2]) == (1, max(args.steps, 1))
    assert bool(torch.isfinite(out_state.train_metrics["total_loss"]).all()), "non-finite loss in the timed updates"
    # ---- instrumented pass (NOT part of `value`): the same updates again with a HIP event pair around every
    # kernel launch on the launch stream, for the per-kernel roofline figures.  ~430 event pairs per update cost
    # ~1.5 ms, which is why they stay out of the timed region.
    timer_steps = 0
    rec_timers = None
    if not args.no_kernel_timers and hasattr(L, "_timed"):
        L.timers = {}
        timer_steps = max(1, min(args.steps, 5))
        for _ in range(timer_steps):
            L.update(0)
        torch.cuda.synchronize()
        barrier()
    elif not args.no_kernel_timers:
        from mava_amd import _lib as _mava_lib_mod

        _mava_lib_mod.TIMERS = rec_timers = {}
        timer_steps = max(1, min(args.steps, 2))
        for _ in range(timer_steps):
            L.update(0)
        torch.cuda.synchronize()
        _mava_lib_mod.TIMERS = None
        barrier()
    if world > 1:
        import torch.distributed as dist

        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    env_steps = world * args.steps * T * U * E  # ff_mappo.py:468-474
    value = env_steps / elapsed
    log(f"timed region: {args.steps} updates in {elapsed:.3f} s -> {value:,.0f} env-steps/s")

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
        "config": {"workload": (f"{args.system} ContinuousActionHead, MaBrax-shaped synthetic obs, " if continuous else
                                f"{args.system} {args.env.upper()} {args.scenario}-shaped synthetic obs, ") + f"{E * U} envs/GPU "
                               f"(update_batch_size={U} x num_envs={E}), rollout_length={T}, ppo_epochs={K}, "
                               f"num_minibatches={M}, agents={A}, obs={L.Oa}/{L.Oc}, actions={L.nA}, "
                               f"one step = one PPO update = {T * U * E} env-steps per GPU",
                   "parallelism": f"dp{world}"},
    }

    # HBM traffic per launch: counters cannot be read from inside this process, so `traffic` is NOT measured in this
    # run - it is attached from the committed rocprofv3 PMC pass of the same workload named in `traffic_source`
    # (regenerated per round with tools/pmc_traffic.sh; stale once a kernel changes after that pass).
    traffic, traffic_source = {}, None
    for name in ("r02_v5_pmc_traffic.json", "r02_v4_pmc_traffic.json", "r02_v3_pmc_traffic.json", "r02_v2_pmc_traffic.json", "r02_v1_pmc_traffic.json", "r01_v5_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = json.load(f)["kernels"]
            traffic_source = f"profiles/{name} (committed rocprofv3 --pmc pass, not measured in this run)"
            break
        except (OSError, ValueError, KeyError):
            continue
    default_shape = (args.envs == 4096 and args.update_batch_size == 1 and args.scenario == "tiny-4ag"
                     and args.system == "ff_mappo" and not continuous)

    if rank == 0 and getattr(L, "timers", None):
        timers = {k: _ev_ms(v) for k, v in L.timers.items()}
        avg = {k: sum(v) / len(v) for k, v in timers.items() if v}
        rows = L.Rb * A  # agent rows per minibatch launch
        # With a centralised critic on a shared global state (x_share == A) the critic kernel evaluates each (t,e)
        # row once and back-propagates the sum of its agents' loss gradients (MAVA_CTX_CRITIC_AGGREGATION,
        # default on): its EXECUTED matrix work is 1/A of the reference's A identical passes.  Rooflines are
        # priced on executed FLOPs.
        aggregated = bool(getattr(L, "critic_share", 1) == A and 1 < A <= 8 and not args.no_critic_aggregation)
        rows_c = L.Rb if aggregated else rows
        # algorithmic FLOPs per row (2*MAC; bwd = 2*fwd - first-layer dX)
        fwd_c = 2 * (L.Oc * 128 + 128 * 128 + 128)
        flop_c = (3 * fwd_c - 2 * L.Oc * 128) * rows_c
        fwd_a = 2 * (L.Oa * 128 + 128 * 128 + 128 * L.nA)
        flop_a = (3 * fwd_a - 2 * L.Oa * 128) * rows
        tf_c = flop_c / (avg["critic_grad"] * 1e-3) / 1e12
        tf_a = flop_a / (avg["actor_grad"] * 1e-3) / 1e12
        # which kernels ran: the split-f16 ones (ppo_train_h2.hip) for the widths they instantiate, else exact f32
        h2_a = args.matmul == "f16x2" and L.Oa <= 287 and not continuous
        h2_c = args.matmul == "f16x2" and L.Oc <= 287
        peak_a = F16X2_PEAK_TFLOPS if h2_a else F32_MFMA_PEAK_TFLOPS
        peak_c = F16X2_PEAK_TFLOPS if h2_c else F32_MFMA_PEAK_TFLOPS
        kname = lambda h2, net: (f"ppo_train_h2_kernel<{net}> (fused fwd+loss+bwd+dW, 3 f16 MFMAs per product; peak = dense f16 peak / 3)"
                                 if h2 else f"ppo_train_kernel<{net}> (fused fwd+loss+bwd+dW, exact-f32 MFMA)")
        tkey = lambda h2, net: next((k for k in traffic if k.startswith("ppo_train_h2_kernel" if h2 else "ppo_train_kernel")
                                     and (("true" in k.split(",")[2]) == (net == "actor") if h2 else net in k)), None)
        roof_c = {"kernel": kname(h2_c, "critic") + (", agents of a row aggregated" if aggregated else ""),
                  "bound": "mfma", "achieved": tf_c, "peak": peak_c, "unit": "TFLOP/s",
                  "frac": tf_c / peak_c,
                  "traffic": (traffic.get(tkey(h2_c, "critic"), {}).get("hbm_bytes_corrected") if default_shape else None),
                  "mfma_busy_pmc": (traffic.get(tkey(h2_c, "critic"), {}).get("mfma_util") if default_shape else None),
                  "traffic_source": traffic_source if default_shape else None,
                  "avg_launch_ms": avg["critic_grad"], "flop_per_launch": flop_c, "rows_per_launch": rows_c}
        roof_a = {"kernel": kname(h2_a, "actor"), "bound": "mfma", "achieved": tf_a,
                  "peak": peak_a, "unit": "TFLOP/s", "frac": tf_a / peak_a,
                  "traffic": (traffic.get(tkey(h2_a, "actor"), {}).get("hbm_bytes_corrected") if default_shape else None),
                  "mfma_busy_pmc": (traffic.get(tkey(h2_a, "actor"), {}).get("mfma_util") if default_shape else None),
                  "traffic_source": traffic_source if default_shape else None,
                  "avg_launch_ms": avg["actor_grad"], "flop_per_launch": flop_a, "rows_per_launch": rows}
        # "roofline" = the dominant kernel of the update (most time per update)
        if avg["actor_grad"] >= avg["critic_grad"]:
            out["roofline"], out["roofline_critic"] = roof_a, roof_c
        else:
            out["roofline"], out["roofline_actor"] = roof_c, roof_a
        # GAE.  With the fused rollout (the default) the scan runs in the rollout kernel's tail on data each workgroup
        # has just written - there is no GAE launch in the loop.  The standalone kernel (mava_gae_f32: recurrent systems,
        # per-step rollout, other callers) is measured here on the rollout's own outputs, outside the timed region:
        # back-to-back launches (its 35.7 MB of inputs and outputs resident in the 256 MB Infinity Cache) and behind a
        # 1 GiB fill (inputs from HBM: the rate of a device-to-device copy of the same bytes).
        from mava_amd import ops as _ops

        rep0 = L.reps[0]
        EA_ = E * A
        g_args = (rep0.reward.view(T, EA_), rep0.value.view(T, EA_), rep0.done.view(T, EA_), rep0.last_val.view(EA_), 0.99, 0.95)
        g_out = (torch.empty(T, EA_, device=dev), torch.empty(T, EA_, device=dev))
        # timed through captured HIP graphs of 20 launches (a Python launch takes ~10 us, longer than the kernel):
        # "warm" relaunches on the same buffers; "cold" cycles through 10 buffer sets (357 MB > the 256 MB cache)
        NSET, NL = 10, 20
        sets = [(g_args, g_out)] + [((torch.randn(T, EA_, device=dev), torch.randn(T, EA_, device=dev),
                                      (torch.rand(T, EA_, device=dev) < 0.002).to(torch.uint8), torch.randn(EA_, device=dev), 0.99, 0.95),
                                     (torch.empty(T, EA_, device=dev), torch.empty(T, EA_, device=dev))) for _ in range(NSET - 1)]

        def _time_gae(cold: bool):
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for i in range(3):
                    _ops.gae(*sets[0][0], out=sets[0][1])
                with torch.cuda.graph(graph, stream=side):
                    for i in range(NL):
                        ga, go = sets[i % NSET] if cold else sets[0]
                        _ops.gae(*ga, out=go)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0_.record()
                graph.replay()
                e1_.record()
                torch.cuda.synchronize()
                ts.append(e0_.elapsed_time(e1_) / NL)
            ts.sort()
            return ts[len(ts) // 2]

        warm_ms, cold_ms = _time_gae(False), _time_gae(True)
        del sets
        gae_bytes = 17 * T * E * A + 4 * E * A
        gbs = gae_bytes / (warm_ms * 1e-3) / 1e9
        gbs_cold = gae_bytes / (cold_ms * 1e-3) / 1e9
        out["roofline_gae"] = {"kernel": "gae_kernel (standalone mava_gae_f32; the fused rollout runs GAE in its own tail, no launch in the loop)",
                               "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "state": "back-to-back launches (captured HIP graph of 20, incl. ~1.5 us launch boundary each): operands resident in the Infinity Cache",
                               "frac_of_measured_copy_peak": gbs / 6290.0,
                               "cold": {"achieved": gbs_cold, "frac": gbs_cold / HBM_PEAK_GBS, "avg_launch_us": cold_ms * 1e3,
                                        "state": "20 launches cycling through 10 buffer sets (357 MB): operands from HBM"},
                               "traffic": (traffic.get(next((k for k in traffic if k.startswith("gae_kernel")), ""), {}).get("hbm_bytes_corrected")
                                           if default_shape else None),
                               "avg_launch_us": warm_ms * 1e3, "bytes_per_launch": gae_bytes}
        adam_bytes = 28 * L.P
        out["roofline_adam"] = {"kernel": "clip_adam_kernel", "bound": "hbm (launch-bound at 77K params)",
                                "achieved": adam_bytes / (avg["clip_adam"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "avg_launch_us": avg["clip_adam"] * 1e3}
        per_update = {k: sum(v) / timer_steps for k, v in timers.items()}
        out["kernel_ms_per_step"] = {k: round(v, 4) for k, v in per_update.items()}

    if rank == 0 and rec_timers:
        # Recurrent systems: the kernel with the most time per update is the roofline object.  The GRU scans
        # are priced on BOTH rooflines - algorithmic FLOPs (h.W_h: 2*128*384 per row-step, exact-f32 MFMA peak) and
        # algorithmic HBM bytes per row-step (forward: read gi 1536 + done, write h 512 + saved gates 2048 + h_prev 512;
        # BPTT: read saved 2048 + h_prev 512 + dh_out 512, write dgi 1536 + dgh 1536) - the binding one is `bound`.
        timers = {k: _ev_ms(v) for k, v in rec_timers.items()}
        per_update = {k: sum(v) / timer_steps for k, v in timers.items()}
        avg = {k: sum(v) / len(v) for k, v in timers.items() if v}
        out["kernel_ms_per_step"] = {k: round(v, 4) for k, v in per_update.items()}
        # launches are keyed "<kernel>:<sequences per step>" (actor: Rm = envs x agents of a minibatch; the centralised
        # critic on a shared state: one sequence per env)
        dom_key = max((k for k in per_update if k.startswith("gru_scan")), key=lambda k: per_update[k])
        dom, seqs = dom_key.split(":")[0], int(dom_key.split(":")[1])
        rows_avg = T * seqs                   # row-steps per launch
        flop = 2.0 * 128 * 384 * rows_avg
        by = {"gru_scan_fwd": 1536 + 1 + 512 + 2048 + 512, "gru_scan_bwd": 2048 + 512 + 512 + 1536 + 1536}[dom] * rows_avg
        t_s = avg[dom_key] * 1e-3
        tf, gbs = flop / t_s / 1e12, by / t_s / 1e9
        # the matrix peak of the arithmetic the scan runs in: exact-f32 MFMAs, or three f16 MFMAs per product (f16x2)
        mfma_peak = F16X2_PEAK_TFLOPS if args.matmul == "f16x2" else F32_MFMA_PEAK_TFLOPS
        hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tf / mfma_peak
        rec_traffic, rec_src = None, None
        kname = f"{dom}_h2_kernel" if args.matmul == "f16x2" else f"{dom}_kernel"
        for fname in ("r02_rec3_pmc_traffic.json", "r02_rec2_pmc_traffic.json", "r02_rec_pmc_traffic.json", "r01_rec_v5_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    rec_traffic = (json.load(f).get("kernels", {}).get(f"{kname} grid={min(256, seqs // 32)}", {})
                                   .get("hbm_bytes_corrected")) if (seqs == 8192 and T == 128) else None
            except (OSError, ValueError, KeyError):
                rec_traffic = None
            if rec_traffic:
                rec_src = f"profiles/{fname} (committed PMC pass, not measured in this run)"
                break
        if hbm_frac >= mfma_frac:
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": hbm_frac, "traffic": rec_traffic, "mfma_frac": mfma_frac}
        else:
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": tf, "peak": mfma_peak,
                               "unit": "TFLOP/s", "frac": mfma_frac, "traffic": rec_traffic, "hbm_frac": hbm_frac}
        out["roofline"].update({"avg_launch_ms": avg[dom_key], "row_steps_per_launch": rows_avg,
                                "traffic_source": rec_src})

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.system.startswith("rec") and not continuous:
        from oracle import rec_cpu_loop

        cores = usable_cores()
        cpu_E = max(2 * M, min(args.cpu_envs, 32))
        log(f"cpu baseline (recurrent) on {cores} usable cores, {cpu_E} envs ...")
        res = rec_cpu_loop.run(E=cpu_E, A=A, Oa=L.Oa, Oc=L.Oc, nA=L.nA, T=T, K=K, M=M, updates=8, warmup=0, threads=cores,
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
        res = cpu_loop.run(E=args.cpu_envs, A=A, O=L.Oa - A, nA=L.nA, T=T, K=K, M=M, updates=64, warmup=1,
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
