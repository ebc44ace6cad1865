#!/usr/bin/env python3
"""bench.py — env-steps/sec/GPU on the BASELINE.json workload (3 jammers / 4 radars, batch_envs=4096).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode env|rollout|train] [--sweep]

One "step" = one pass of the hot path over one batch of E environments per GPU.  Modes:
  env      the HIP env-step kernel alone (Philox uniforms in-kernel, synthetic actions resident in HBM)
  rollout  agent forward + MP-DQN multi-pass Q + eps-greedy + env-step kernel + replay write
  train    rollout + QMixLearner.train at the reference cadence (default when available)
For N>1 launch through torch.distributed.run (one rank per GPU, RCCL); envs shard across ranks with
no data-path collective in env/rollout mode ("weak" scaling); train mode all-reduces the gradients.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (env-step kernel, HBM bound,
timed with HIP events on the launch stream) and `cpu_baseline` (the C oracle timed on host cores).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def per_env_table_bytes(J, R):
    """Per-env scenario mode: each env-step also streams its env's SoA table column —
    6R + 3J + JR float64 rows and JR flag bytes (include/macjd.h, macjd_step_io.pe_tables)."""
    return 8 * (6 * R + 3 * J + J * R) + J * R


def algorithmic_bytes_per_env_step(J, R, uniforms_supplied, info_outputs=True):
    """DESIGN.md section 'Algorithmic bytes': what one env-step must move through HBM.
    reads : T 4J + P 4J + step 4 + (uniforms, counted as SURVEY.md 8(d) does, 4(R+J), when supplied | episode index 4)
    writes: reward 4 + (r_d,r_p,r_j) 12 + terminated 1 + track R + step 4 (+ pd 4R + snr_with 4R info)
    The FSM's next state does not depend on the previous one (core/radar.py:102-117), so `track` is
    written, never read."""
    b = 4 * J + 4 * J + 4 + 4 + 12 + 1 + R + 4
    if uniforms_supplied:
        b += 4 * (R + J)
    else:
        b += 4   # the env's episode index (int32), part of the in-kernel generator's counter
    if info_outputs:
        b += 8 * R
    return b


def dist_setup(n_gpus):
    import macjd_amd  # noqa: F401  (first: reserves the graph-launch hardware queue before the HIP runtime starts)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and os.environ.get("MACJD_BENCH_ONE_GPU_REHEARSAL") == "1":
        # rehearsal of the N > 1 code path on a one-GPU box: every rank on cuda:0, gloo carrying the HIP tensors.
        # Exercises sharding, broadcast, the two update graphs around the all-reduce, barriers and the max-over-ranks
        # timing; the JSON line says so (config.rehearsal) and its rate means nothing.
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, world, local


def cpu_baseline(sc, E, seed, budget_s=12.0):
    """Times the CPU oracle (oracle/libmacjd_oracle.so, a C port of the reference's env.step; kind="port") on a bounded
    sample of the same workload, INSIDE C on pre-allocated buffers (oracle/macjd_oracle_mt.c, macjd_oracle_bench: one
    OpenMP region for the whole run, no per-step Python).  Single thread: E envs (the reference itself is
    single-threaded Python).  All cores: E envs PER THREAD, so every thread steps a full 4096-env batch.  Each leg is
    sized from a short calibration run to take ~budget_s / 2 seconds."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _harness import oracle_bench, oracle_lib
    out = {}
    cores_all = max(1, min(oracle_lib().macjd_oracle_max_threads(), os.cpu_count() or 1))
    try:   # threads the affinity mask / cgroup CPU quota actually give this process
        cores_all = max(1, min(cores_all, len(os.sched_getaffinity(0))))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores_all = max(1, min(cores_all, int(float(quota) / float(period) + 0.5)))
    except (AttributeError, OSError, ValueError):
        pass
    for label, nt in (("1", 1), ("all", cores_all)):
        n_env = E * nt
        rate, _ = oracle_bench(sc, n_env, nt, 20, seed=seed)                 # calibration (also warms the threads)
        n_steps = int(min(20000, max(50, rate * (budget_s / 2) / n_env)))
        rate, sec = oracle_bench(sc, n_env, nt, n_steps, seed=seed)
        out[label] = (rate, nt, n_steps, n_env, sec)
    return out


def pmc_traffic(J, R, E, per_env=False, many=False):
    """HBM bytes per env_step launch from the committed rocprofv3 PMC summary (profiles/), collected in
    separate --pmc passes and corrected as MI355X_MICROARCH.md prescribes; None when no matching run.
    ``many``: the many-step launch (key "many_<E>" of the summary)."""
    import glob
    key = f"many_{E}" if many else str(E)
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_env_step*_pmc.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel", "").endswith(f"<{J},{R},per-env>" if per_env else f"<{J},{R}>") and key in d.get("runs", {}):
            return d["runs"][key]["traffic_bytes_per_launch"], os.path.basename(f)
    return None, None


def profiled_kernel_time(kernel_substr, per_env=False, pattern=None):
    """(average us, min us, calls, file) of a kernel in the latest committed rocprofv3 --kernel-trace --stats summary of
    the env-roofline replay (scripts/collect_profiles.sh: `bench.py --mode env --steps 1 --warmup 0`, i.e. the 200-launch
    graph of macjd_env_step_timed, warm + timed); None when no such file."""
    import csv
    import glob
    pat = pattern or ("r*_bench_env_per_env_kernel_stats.csv" if per_env else "r*_bench_env_roofline_kernel_stats.csv")
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", pat)), reverse=True):
        try:
            for r in csv.DictReader(open(f)):
                if kernel_substr in r["Name"]:
                    return float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"]), os.path.basename(f)
        except Exception:
            continue
    return None


def time_graph_replay(fn, dev, calls=20, replays=10):
    """Average microseconds per call of ``fn`` (a launch sequence on the current stream), replayed from a HIP graph and
    bracketed by events on the replay stream."""
    import torch
    with torch.no_grad():
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(calls):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (calls * replays) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--mode", default="auto", choices=["auto", "env", "rollout", "train"])
    ap.add_argument("--batch-envs", type=int, default=4096)
    ap.add_argument("--jammers", type=int, default=3)
    ap.add_argument("--radars", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--mixer-dtype", dest="mixer_dtype", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: BASELINE.json config 5's hyper-network option (library GEMMs with bf16 inputs; outside the 1e-5 bar)")
    ap.add_argument("--sweep", action="store_true", help="also print an E-sweep of the env kernel (stderr)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", dest="no_other_modes", action="store_true",
                    help="measure only --mode (default: the other two modes of SURVEY.md 8(d) are timed briefly as well)")
    ap.add_argument("--per-env-scenarios", dest="per_env", action="store_true",
                    help="every env gets its own randomised scenario (positions / threat / powers): tables streamed from HBM")
    ap.add_argument("--no-graphs", dest="no_graphs", action="store_true", help="eager launches (A/B against HIP graphs)")
    ap.add_argument("--no-gemm-tuning", dest="no_gemm_tuning", action="store_true",
                    help="library GEMMs with default heuristics (A/B against TunableOp)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    rank, world, local = dist_setup(args.gpus)
    if rank == 0:
        entry.build()
    if world > 1:
        torch.distributed.barrier()
    # what ran where (the driver reads this to see that RCCL really had N ranks on N devices)
    dist_info = {"world_size": world, "backend": torch.distributed.get_backend() if world > 1 else None,
                 "devices": [torch.cuda.current_device()]}
    if world > 1:
        devs = [None] * world
        torch.distributed.all_gather_object(devs, {"rank": rank, "local_rank": local, "device": torch.cuda.current_device(),
                                                   "name": torch.cuda.get_device_name(torch.cuda.current_device())})
        dist_info["devices"] = devs
    from macjd_amd.scenario import Scenario, ring_scenario_dict
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment

    if os.environ.get("MACJD_PE_TILED", "1") == "0":   # A/B hook: plain SoA per-env tables
        BatchedElectromagneticEnvironment.pe_tiled = False
    J, R, E = args.jammers, args.radars, args.batch_envs
    sc = Scenario.from_dict(ring_scenario_dict(J, R))
    dev = torch.device("cuda", local)
    mode = args.mode
    bench_mod = None
    if mode in ("auto", "rollout", "train"):
        try:
            from macjd_amd import bench_rollout as bench_mod  # provided once the agent path exists
        except ImportError:
            if mode != "auto":
                raise
        mode = ("train" if bench_mod is not None else "env") if mode == "auto" else mode

    batch = None
    if args.per_env:
        from macjd_amd.scenario import ScenarioBatch
        batch = ScenarioBatch.randomized(ring_scenario_dict(J, R), E, seed=42, env_offset=rank * E)

    def make_env():
        if args.per_env:
            return BatchedElectromagneticEnvironment(scenario_batch=batch, device=dev, seed=42, env_offset=rank * E)
        return BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=dev, seed=42, env_offset=rank * E)

    env = make_env()
    env.reset()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    # synthetic actions, agent-major storage [J, E] (T ~ U{0..2R}, P ~ U[0,1))
    T_am = torch.randint(0, 2 * R + 1, (J, E), generator=g, device=dev, dtype=torch.int32)
    P_am = torch.rand((J, E), generator=g, device=dev)
    T, P = T_am.t(), P_am.t()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def build_step(which, env_, cli):
        if which == "env":
            def fn(i):
                if i % sc.episode_limit == 0:
                    env_.reset()
                env_.step(T, P)
            return fn, {}
        if which == "env_many_step":
            # the env kernel as the rollout launches it: all steps of an episode batch in ONE launch (macjd_env_step_many),
            # synthetic actions of every step resident in HBM; episodes aligned to the timed regions like the rollout's
            Tn = sc.episode_limit
            Tm_ = torch.randint(0, 2 * R + 1, (Tn, E, J), generator=g, device=dev, dtype=torch.int32)
            Pm_ = torch.rand((Tn, E, J), generator=g, device=dev)
            rew_, ter_ = torch.zeros((Tn, E), device=dev), torch.zeros((Tn, E), dtype=torch.uint8, device=dev)
            rd_ = torch.zeros((Tn, E, 3), device=dev)

            def fn(i):
                j, region = (i, cli.warmup) if i < cli.warmup else (i - cli.warmup, cli.steps)
                if j % Tn == 0:
                    n_ = min(Tn, region - j)
                    env_.reset()
                    env_.step_many(Tm_[:n_], Pm_[:n_], rew_[:n_], ter_[:n_], rd_[:n_])
            return fn, {}
        return bench_mod.make_step(cli, sc, env_, dev, rank, world, which)   # aligns its episodes to cli.warmup / cli.steps

    def timed(fn, warmup, steps):
        for i in range(warmup):
            fn(i)
        barrier()
        t0_ = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        barrier()
        d = time.perf_counter() - t0_
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d = float(tt.item())
        return d

    step_fn, extra = build_step(mode, env, args)
    dt = timed(step_fn, args.warmup, args.steps)

    # ---- the other two modes of SURVEY.md 8(d) in the same run (single process; shorter regions) ----
    modes = {mode: round(dt / args.steps * 1e3, 4)}
    if world == 1 and bench_mod is not None and not args.no_other_modes:
        for which in ("env", "env_many_step", "rollout", "train"):
            if which in modes or (which == "env_many_step" and (args.per_env or E * sc.episode_limit > (1 << 24))):
                continue
            e2 = make_env()
            e2.reset()
            cli2 = argparse.Namespace(**vars(args))
            cli2.steps, cli2.warmup = (300, 100) if which in ("rollout", "train") else (1000, 100)
            fn2, _ = build_step(which, e2, cli2)
            modes[which] = round(timed(fn2, cli2.warmup, cli2.steps) / cli2.steps * 1e3, 5 if which == "env_many_step" else 4)
            del fn2
            torch.cuda.synchronize()

    # ---- roofline of the env-step kernel: HIP events on the launch stream ----
    # (1) the launch the rollout / train modes actually issue: ONE launch for the T = episode_limit steps of all E envs
    #     (macjd_env_step_many: T x E independent work items in the streaming lane kernel; no pd / snr_with outputs,
    #     `track` and the counters written once per env).  (2) the single-step launch of the step-by-step API at the
    #     same E (what `mode: env` issues; launch-latency bound).  (3) below: the 2^22-env streaming point.
    ms = env.time_step_kernel(T, P, iters=200)
    B_step = algorithmic_bytes_per_env_step(J, R, uniforms_supplied=False) + (per_env_table_bytes(J, R) if args.per_env else 0)
    achieved = E * B_step / (ms * 1e-3) / 1e9
    kname = (f"env_step_kernel<{J},{R},per-env tables>" if args.per_env else
             f"env_step_slots_kernel<{J},{R}>" if E < (1 << 16) else f"env_step_kernel<{J},{R}>")
    single = {"kernel": kname, "us_per_launch": round(ms * 1e3, 3), "achieved": round(achieved, 3),
              "frac": round(achieved / HBM_PEAK_GBS, 6), "bytes_per_env_step": B_step, "envs_per_launch": E,
              "traffic": pmc_traffic(J, R, E, args.per_env)[0], "traffic_source": pmc_traffic(J, R, E, args.per_env)[1]}
    prof = profiled_kernel_time("env_step_kernel<" if (args.per_env or E >= (1 << 16)) else "env_step_slots_kernel<", args.per_env)
    if prof is not None and E == 4096:
        # the committed rocprofv3 summary of the same replay: its per-dispatch duration (start -> end timestamps of ONE
        # dispatch, taken with the dispatches serialised by the profiler) is ~0.7 us above the back-to-back figure the
        # HIP events give for a launch this short; both are stated, `frac` uses the live HIP-event time
        single["profile"] = {"file": prof[3], "avg_us": round(prof[0], 3), "min_us": round(prof[1], 3), "calls": prof[2],
                             "frac_from_avg": round(E * B_step / (prof[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 6)}
    Tn = sc.episode_limit
    Tm = torch.randint(0, 2 * R + 1, (Tn, E, J), generator=g, device=dev, dtype=torch.int32)
    Pm = torch.rand((Tn, E, J), generator=g, device=dev)
    rew_m = torch.zeros((Tn, E), device=dev)
    ter_m = torch.zeros((Tn, E), dtype=torch.uint8, device=dev)
    rd_m = torch.zeros((Tn, E, 3), device=dev)
    env.time_step_many_kernel(Tm, Pm, rew_m, ter_m, rd_m, iters=3)
    ms_m = env.time_step_many_kernel(Tm, Pm, rew_m, ter_m, rd_m, iters=50)
    # algorithmic (compulsory) bytes per env-step of THIS launch: T 4J + P 4J read, reward 4 + (r_d,r_p,r_j) 12 +
    # terminated 1 written by every work item; the env's step counter 4 + episode index 4 are read by each of its T work
    # items but are ONE HBM read per env (the other T - 1 hit L2: PMC traffic = 1.01x this count), and track R is written
    # once per env: (8 + R) / T per env-step; per-env tables likewise — an env's table column is read by each of its T work
    # items but comes from HBM once
    B_many = 8 * J + 4 + 12 + 1 + (8 + R + (per_env_table_bytes(J, R) if args.per_env else 0)) / Tn
    ach_m = Tn * E * B_many / (ms_m * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(ach_m, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_m / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(J, R, E, args.per_env, many=True)[0],
                "traffic_source": pmc_traffic(J, R, E, args.per_env, many=True)[1],
                "kernel": f"env_step_kernel<{J},{R}" + (",per-env tables" if args.per_env else "") +
                          f"> as launched by the rollout: {Tn} steps x {E} envs = {Tn * E} work items per launch (macjd_env_step_many)",
                "us_per_launch": round(ms_m * 1e3, 3),
                "timing": "HIP events around 50 launches replayed from one HIP graph on their own stream (macjd_env_step_many_timed); "
                          "the committed rocprofv3 kernel-trace summary of this launch alone is quoted under 'profile'",
                "traffic_measured_in_run": False,
                "limiter": "VALU issue, not HBM: ~735 VALU instructions per env-step at 3j/4r (~145 float64 for the power / received-power "
                           "quotients, 2 Philox blocks = 40 quarter-rate 32 x 32 multiplies, float32 SNR / detection probabilities "
                           "with a float64 fallback for compares within 4e-6) for 41 bytes; traffic = 1.01x algorithmic (DESIGN.md 4.1)",
                "bytes_per_env_step": round(B_many, 2), "env_steps_per_launch": Tn * E,
                "single_step_launch": single}
    prof_m = profiled_kernel_time("env_step_kernel<", args.per_env, pattern=f"r*_env_many_step_{J}j{R}r_kernel_stats.csv")
    if prof_m is not None and E == 4096 and not args.per_env:
        roofline["profile"] = {"file": prof_m[3], "avg_us": round(prof_m[0], 3), "min_us": round(prof_m[1], 3), "calls": prof_m[2],
                               "note": "rocprofv3 --kernel-trace --stats of scripts/replay_many_step.py: this launch alone"}
    del Tm, Pm, rew_m, ter_m, rd_m
    # large-batch point of the same kernel family (2^22 envs): the HBM-bound asymptote, measured every run so the
    # launch-bound fraction at the benchmark's E is not mistaken for the kernel's streaming rate
    if rank == 0:
        Eb = 1 << 22
        if args.per_env:   # the 4096 compiled scenarios, cycled
            eb = BatchedElectromagneticEnvironment(scenario_batch=batch.tile(Eb), device=dev, seed=1)
        else:
            eb = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=Eb, device=dev, seed=1)
        Tb = torch.randint(0, 2 * R + 1, (J, Eb), generator=g, device=dev, dtype=torch.int32).t()
        Pb = torch.rand((J, Eb), generator=g, device=dev).t()
        eb.time_step_kernel(Tb, Pb, iters=3)
        mb = eb.time_step_kernel(Tb, Pb, iters=20)
        gb = Eb * B_step / (mb * 1e-3) / 1e9
        roofline["large_batch"] = {"envs_per_launch": Eb, "us_per_launch": round(mb * 1e3, 2), "achieved": round(gb, 1),
                                   "frac": round(gb / HBM_PEAK_GBS, 4),
                                   "kernel": f"env_step_kernel<{J},{R}" + (",per-env tables>" if args.per_env else ">"),
                                   "traffic": pmc_traffic(J, R, Eb, args.per_env)[0]}
        eb.close()
        del eb, Tb, Pb
    # ---- MFMA-bound kernel of the path: the actor's fused dense chain (exact-f32 v_mfma_f32_16x16x4_f32) ----
    roofline_mfma = None
    if rank == 0:
        try:
            from macjd_amd import ops
            A_, S_, N_rows = 2 * R + 1, sc.state_dim, E * J
            dims = (S_, 128, 128, A_)
            if ops.mlp_supported(list(dims)):
                layers = [(torch.randn(dims[l + 1], dims[l], device=dev) / dims[l] ** 0.5,
                           torch.randn(dims[l + 1], device=dev) * 0.1, (1, 1, 2)[l]) for l in range(3)]

                def time_chain(n_rows, calls=20, replays=10):
                    xs = torch.randn(n_rows, S_, device=dev)
                    with torch.no_grad():
                        side = torch.cuda.Stream(device=dev)
                        side.wait_stream(torch.cuda.current_stream(dev))
                        with torch.cuda.stream(side):
                            for _ in range(3):
                                ops.mlp_forward(xs, layers)
                        torch.cuda.current_stream(dev).wait_stream(side)
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr):
                            for _ in range(calls):
                                ops.mlp_forward(xs, layers)
                        gr.replay()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(replays):
                            gr.replay()
                        e1.record()
                        torch.cuda.synchronize()
                    us_ = e0.elapsed_time(e1) / (calls * replays) * 1e3
                    fl = 2.0 * n_rows * (dims[0] * dims[1] + dims[1] * dims[2] + dims[2] * dims[3])
                    return us_, fl, fl / (us_ * 1e-6) / 1e12

                us, flops, tf = time_chain(N_rows)
                roofline_mfma = {"bound": "mfma", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": round(tf / 157.3, 4), "traffic": None,
                                 "kernel": "mlp_forward_kernel (actor %d-%d-%d-%d, %d rows)" % (*dims, N_rows),
                                 "us_per_call": round(us, 2), "flops_per_call": int(flops), "dtype": "f32 (exact-f32 MFMA)"}
                # streaming-size point of the same kernel (16x the rows: every workgroup stages the weights once and
                # walks 12 tiles), so the launch / staging-bound fraction at the benchmark's size is not mistaken for
                # the kernel's matrix-core rate
                Nb = 16 * N_rows
                usb, flb, tfb = time_chain(Nb, calls=5, replays=5)
                roofline_mfma["large_batch"] = {"rows": Nb, "us_per_call": round(usb, 2), "achieved": round(tfb, 2),
                                                "frac": round(tfb / 157.3, 4)}
        except Exception as ex:  # the HBM roofline above is the contract item; this one is additional
            roofline_mfma = {"error": str(ex)[:200]}
    # ---- MFMA fraction of the mixer's hyper_w_1 network (2 M (S 128 + 128 J 64) flop over the learner's
    # M = batch x (T + 1) rows) as the learner evaluates it ----
    roofline_mixer = None
    if rank == 0 and bench_mod is not None:
        try:
            roofline_mixer = bench_mod.mixer_hyper_w1_roofline(sc, args.hidden, dev, 32 * (sc.episode_limit + 1), time_graph_replay)
        except Exception as ex:
            roofline_mixer = {"error": str(ex)[:200]}
    sweep = None
    if args.sweep and rank == 0:
        sweep = []
        for logE in (12, 14, 16, 18, 20, 22):
            Es = 1 << logE
            e2 = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=Es, device=dev, seed=1)
            Ts = torch.randint(0, 2 * R + 1, (J, Es), generator=g, device=dev, dtype=torch.int32).t()
            Ps = torch.rand((J, Es), generator=g, device=dev).t()
            from macjd_amd import _native
            ms_k = {}
            for label, flag in (("auto", 0), ("slot", _native.STEP_SLOT_KERNEL), ("lane", _native.STEP_LANE_KERNEL)):
                e2.kernel_flags = flag   # A/B of the two kernel variants on the same inputs, same process
                e2.time_step_kernel(Ts, Ps, iters=5)
                ms_k[label] = e2.time_step_kernel(Ts, Ps, iters=50)
            m = ms_k["auto"]
            gbs = Es * B_step / (m * 1e-3) / 1e9
            sweep.append({"E": Es, "us": round(m * 1e3, 2), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                          "us_slot_kernel": round(ms_k["slot"] * 1e3, 2), "us_lane_kernel": round(ms_k["lane"] * 1e3, 2)})
            print(f"[sweep] E=2^{logE} {m*1e3:9.2f} us/launch  {gbs:8.1f} GB/s  frac {gbs/HBM_PEAK_GBS:.4f}"
                  f"   (slot {ms_k['slot']*1e3:.2f} us, lane {ms_k['lane']*1e3:.2f} us)", file=sys.stderr, flush=True)
            e2.close()

    if rank == 0:
        res = {
            "metric": "env-steps/sec (all GPUs; batch_envs=%d/GPU, %dj/%dr)" % (E, J, R),
            "value": round(E * world * args.steps / dt, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{J} jammers / {R} radars, batch_envs={E} per GPU, GRU hidden={args.hidden}"
                                   + (", per-env randomised scenarios" if args.per_env else ""),
                       "mode": mode, "per_gpu_env_steps_per_s": round(E * args.steps / dt, 1), **extra},
            "roofline": roofline,
            # ms per batched step of each mode of SURVEY.md 8(d): (i) env kernel only — `env`: one launch per step through
            # the single-step API (host-enqueue bound), `env_many_step`: the launch the rollout issues, all steps of an
            # episode batch at once — (ii) rollout, (iii) rollout + training
            "modes_ms_per_step": modes,
        }
        if "train" in modes:
            upd = extra.get("train_calls_per_step", 1) if mode == "train" else 1
            res["updates_per_s"] = round(world * upd / (modes["train"] * 1e-3), 1)
            res["train_env_ratio"] = ("1 learner update (32 whole episodes) per BATCHED env step = 1 update / %d env-steps per GPU; "
                                      "the reference does 1 update / 1 env-step of its single env (main.py:212-216) — at that ratio "
                                      "the rate is bounded by 1 / update time = updates_per_s env-steps/s per GPU" % E)
        res["distributed"] = dist_info
        if world > 1 and os.environ.get("MACJD_BENCH_ONE_GPU_REHEARSAL") == "1":
            res["config"]["rehearsal"] = "all ranks on ONE GPU over gloo: code-path check only, not a measurement"
        if roofline_mfma is not None:
            res["roofline_mfma"] = roofline_mfma
        if roofline_mixer is not None:
            res["roofline_mfma_mixer"] = roofline_mixer
        if sweep:
            res["env_kernel_sweep"] = sweep
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N=1 only
            cb = cpu_baseline(sc, E, seed=42)
            res["cpu_baseline"] = {
                "value": round(cb["all"][0], 1), "unit": "env-steps/s", "cores": cb["all"][1], "kind": "port",
                "sample": (f"C port (oracle) of reference env.step only, Philox uniforms, timed inside C on pre-allocated "
                           f"buffers: {cb['all'][3]} envs x {cb['all'][2]} steps on {cb['all'][1]} OpenMP threads "
                           f"({cb['all'][4]:.1f} s); single thread: {cb['1'][3]} envs x {cb['1'][2]} steps ({cb['1'][4]:.1f} s)"),
                "single_core_value": round(cb["1"][0], 1),
                "thread_scaling": round(cb["all"][0] / cb["1"][0], 2),
                "note": "covers env.step only; the Python reference measured in the build container (BASELINE.md): "
                        "8725 env.step/s, 626 env-steps/s full rollout, ~1 env-step/s at its 1-train-per-step cadence"}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
