#!/usr/bin/env python3
"""
bench.py -- env.step()/s of the vectorised CounterTraffic band-assignment env on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE env.step() of every environment of the batch: one launch of the HIP step
kernel over 65 536 envs x 4 devices per GPU (BASELINE.json configs[1]).  Actions are
synthetic (seeded uniform device/duration), generated on the GPU before the timed region;
every env is reset() at step 0 and every 64 steps so the data-carrying phase stays in play
(SURVEY.md 8d).  Rank 0 prints ONE JSON line.

  value        whole-job env-steps/s: N_gpus * envs_per_gpu * K / max-over-ranks wall time
  roofline     HBM bound: algorithmic bytes per launch / average launch duration (HIP events on the
               launch stream around the timed region, / K), against 8 TB/s.  Algorithmic bytes per env-step:
               B(D) = 17 + 2*(12 + 20*D) + 4*(k_app + k_pop), k_app/k_pop counted by the kernel.
  cpu_baseline the C oracle (scalar restatement of the reference algorithm, oracle/ct_oracle.c)
               timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
RESET_EVERY = 64


def algorithmic_bytes(D, env_steps, appended, popped):
    return env_steps * (17 + 2 * (12 + 20 * D)) + 4 * (appended + popped)


def state_bytes(D):
    """Bytes the default kernel actually loads + stores per env-step."""
    rb = 16 * ((2 * D + 1 + 15) // 16)
    return (16 + 16 + 16 + rb + 32 + 8) + (16 + 4 + 8 + rb + 32 + 9)


def measured_traffic(D, N):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 for the gfx950
    wide-load under-count + WRITE_SIZE, both in KB), if a profile of this exact shape is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            for row in json.load(fh)["rows"]:
                if row["devices"] == D and row["envs"] == N:
                    return (2 * row["fetch_kb"] + row["write_kb"]) * 1024.0
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(D, seconds_target=12.0, single_thread_seconds=3.0):
    """The oracle on the host cores, same action distribution, same reset cadence: all cores of the box's
    CPU share for one GPU (the headline `value`) and one thread beside it."""
    import numpy as np
    from oracle.ct_oracle import CtOracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))           # the GPU box's CPU share for one GPU

    def timed(nthreads, seconds):
        n_env, K = 256 * nthreads, 64
        rng = np.random.default_rng(1234)
        dev = rng.integers(0, D, (K, n_env), dtype=np.int32)
        dur = rng.integers(0, 20, (K, n_env), dtype=np.int32)
        orc = CtOracle(n_env, D, nthreads=nthreads)
        orc.reset()
        for k in range(8):
            orc.step(dev[k], dur[k])             # warm-up
        done_steps, t0 = 0, time.perf_counter()
        while True:
            orc.reset()
            for k in range(K):
                orc.step(dev[k], dur[k])
            done_steps += K * n_env
            el = time.perf_counter() - t0
            if el >= seconds:
                return done_steps / el, n_env, K, el

    v, n_env, K, el = timed(cores, seconds_target)
    v1, _, _, el1 = timed(1, single_thread_seconds)
    return {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps per pass (D=%d, reset every pass), repeated for %.1f s on %d threads (OpenMP)"
                      % (n_env, K, D, el, cores),
            "single_thread_value": v1, "single_thread_sample": "256 envs x %d steps per pass for %.1f s on 1 thread" % (K, el1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)      # 64 warm-up + 1 024 timed steps
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--devices", type=int, default=4, help="senders per env (D)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the CPU-baseline sample")
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: skip the end-of-step observation gather (the only exchange of the path: one byte per "
                         "env-step, one RCCL all-gather per 64 steps, overlapped with stepping)")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--no-graph", action="store_true", help="skip the secondary hipGraph-replay measurement")
    ap.add_argument("--no-steady", action="store_true", help="skip the secondary no-reset steady-state measurement")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL; must precede HIP initialisation
    import torch
    import torch.distributed as dist
    import gymwipe_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # GW_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # host-side collectives); the driver's runs use the default, RCCL with one GPU per rank.
    backend = os.environ.get("GW_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend != "nccl" or local >= ndev:       # rehearsal, or a launcher that shows each rank only its own GPU
        local = local % ndev
    torch.cuda.set_device(local)
    dev_t = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev_t)
        else:
            dist.init_process_group(backend)

    N, D, K, W = args.envs, args.devices, args.steps, args.warmup
    env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D, device=dev_t)

    # outputs as three views of ONE packed record buffer so that the end-of-step observation
    # gather is a single RCCL all-gather without a packing kernel (gymwipe_amd/sharding.py)
    from gymwipe_amd.sharding import ChunkedFeedbackGather, StepRecord
    pipe = None
    if world > 1 and not args.no_gather:
        if backend == "nccl":
            pipe = ChunkedFeedbackGather(N, dev_t, env.pack_feedback, world, chunk=RESET_EVERY)
        else:                                                    # rehearsal: pack on the GPU, gather on the host
            stage = torch.empty((RESET_EVERY, N), dtype=torch.uint8, device=dev_t)

            def pack_to_host(o, r, d, out):
                env.pack_feedback(o, r, d, stage[:o.shape[0]])
                out.copy_(stage[:o.shape[0]])
            pipe = ChunkedFeedbackGather(N, dev_t, pack_to_host, world, chunk=RESET_EVERY)
            pipe.packed = [torch.zeros((RESET_EVERY, N), dtype=torch.uint8) for _ in range(pipe.depth)]
            pipe.gathered = [torch.zeros((world, RESET_EVERY, N), dtype=torch.uint8) for _ in range(pipe.depth)]
    rec = StepRecord(N, dev_t)
    env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done

    g = torch.Generator(device=dev_t)
    g.manual_seed(1234 + rank)
    a_dev = torch.randint(0, D, (W + K, N), dtype=torch.int32, device=dev_t, generator=g)
    a_dur = torch.randint(0, 20, (W + K, N), dtype=torch.int32, device=dev_t, generator=g)
    acts = [{"device": a_dev[i], "duration": a_dur[i]} for i in range(W + K)]

    def one(i):
        if pipe is not None:                      # this step's outputs go into the current chunk record
            env._obs, env._rew, env._done = pipe.slot()
        if i % RESET_EVERY == 0:
            env.reset()                           # (its observation lands in the slot the step then overwrites)
        env.step(acts[i])
        if pipe is not None:
            pipe.stepped()                        # every 64th step: pack to bytes + async all-gather over RCCL

    for i in range(W):
        one(i)
    torch.cuda.synchronize()
    s0 = env.stats()

    # ---- timed region: exactly K steps -------------------------------------------------------
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(W, W + K):
        one(i)
    ev[1].record()
    if pipe is not None:
        pipe.drain()                              # the job is done when the last gather has landed
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    s1 = env.stats()
    env.check()

    # the job's time is the slowest rank's; taken now so that the secondaries below contain no collective
    t = torch.tensor([wall], dtype=torch.float64, device=dev_t if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    # ---- kernel duration --------------------------------------------------------------------------
    # The K launches of the timed region run back-to-back on one stream (the GPU is the bottleneck),
    # so (HIP event at the end - HIP event at the start) / K is the average launch duration including
    # inter-kernel gaps and the reset kernel every 64 steps: an UPPER bound on the kernel's own time.
    # (Event pairs around every launch were tried first: each pair adds ~2.5 us of its own.)
    stream_s = ev[0].elapsed_time(ev[1]) * 1e-3
    kern_avg_s = stream_s / K

    # ---- secondaries (no collectives inside; a failure is reported in the JSON, it does not cost the headline) ----
    def guarded(fn):
        try:
            return fn()
        except Exception as exc:
            return {"error": repr(exc)}

    def secondary_rollout():
        """The same K steps through gw_rollout: one persistent launch per 64 pre-staged steps."""
        r_obs = torch.empty((RESET_EVERY, N), dtype=torch.int32, device=dev_t)
        r_rew = torch.empty((RESET_EVERY, N), dtype=torch.float32, device=dev_t)
        r_done = torch.empty((RESET_EVERY, N), dtype=torch.uint8, device=dev_t)
        chunks = [(i, min(i + RESET_EVERY, W + K)) for i in range(W, W + K, RESET_EVERY)]

        def run_rollouts():
            for lo, hi in chunks:
                if lo % RESET_EVERY == 0:
                    env.reset()
                env.rollout(a_dev[lo:hi], a_dur[lo:hi], out=(r_obs[:hi - lo], r_rew[:hi - lo], r_done[:hi - lo]))
        run_rollouts()                                # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_rollouts()
        torch.cuda.synchronize()
        roll_wall = time.perf_counter() - t1
        env.check()
        return {"env_steps_per_s_this_rank": N * K / roll_wall, "ms_per_step": roll_wall / K * 1e3,
                "what": "gw_rollout: one persistent launch per %d pre-staged steps (ct_rollout_sfx.hip), same K steps, "
                        "same outputs; not the headline because env.step() is one call per step" % RESET_EVERY}

    def secondary_graph():
        """The same K gw_step launches replayed from a hipGraph of reset + 64 steps."""
        G = RESET_EVERY
        g_dev = torch.zeros((G, N), dtype=torch.int32, device=dev_t)
        g_dur = torch.zeros((G, N), dtype=torch.int32, device=dev_t)
        env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done
        side = torch.cuda.Stream(device=dev_t)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            env.reset()
            for j in range(G):
                env.step({"device": g_dev[j], "duration": g_dur[j]})

        def run_graphs():
            for lo in range(W, W + K, G):
                g_dev.copy_(a_dev[lo:lo + G])
                g_dur.copy_(a_dur[lo:lo + G])
                graph.replay()
        run_graphs()                                  # warm-up
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        run_graphs()
        torch.cuda.synchronize()
        g_wall = time.perf_counter() - t2
        env.check()
        return {"env_steps_per_s_this_rank": N * K / g_wall, "ms_per_step": g_wall / K * 1e3,
                "what": "the same K gw_step launches replayed from a hipGraph of reset + %d steps (launch-bound "
                        "host loop removed; includes the copy of each chunk's actions into the graph's input buffers)" % G}

    def secondary_steady():
        """Steady state without resets (SURVEY 8d asks for it separately): after ~0.2 s of simulated time the
        packets have outgrown every window and steps carry no data any more."""
        env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done
        n_ss = min(256, K)
        for i in range(W, W + min(64, K)):
            env.step(acts[i])
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(W, W + n_ss):
            env.step(acts[i])
        torch.cuda.synchronize()
        ss_wall = time.perf_counter() - t3
        return {"env_steps_per_s_this_rank": N * n_ss / ss_wall, "ms_per_step": ss_wall / n_ss * 1e3, "steps": n_ss,
                "what": "no reset for >= 64 steps before and during the timed steps: queues hold only packets too long "
                        "for any window, so a step is the announcement plus counter ticks"}

    roll = guarded(secondary_rollout) if not args.no_rollout else None
    # graph replay at N = 1 only: stream capture next to a live RCCL communicator (whose watchdog thread queries
    # events) is a needless risk for a secondary figure
    graph_sec = (guarded(secondary_graph)
                 if (not args.no_graph and world == 1 and K % RESET_EVERY == 0 and W % RESET_EVERY == 0) else None)
    steady = guarded(secondary_steady) if not args.no_steady else None

    if rank == 0:
        env_steps = s1["steps"] - s0["steps"]
        bytes_launch = algorithmic_bytes(D, env_steps, s1["appended"] - s0["appended"],
                                         s1["popped"] - s0["popped"]) / K
        achieved = bytes_launch / kern_avg_s
        value = world * N * K / wall_max
        out = {
            "metric": "env.step()/s at 65 536 parallel envs, counter-traffic band-assign",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_max / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "CounterTrafficEnv, %d devices, %d vectorised envs per GPU, reset every %d steps"
                                   % (D, N, RESET_EVERY),
                       "envs_per_gpu": N, "devices": D, "global_envs": world * N,
                       "obs_gather": pipe is not None,
                       "parallelism": "independent env shards, one process per GPU; only exchange: end-of-step feedback "
                                      "gather, 1 byte per env-step, one RCCL all-gather per %d steps overlapped with stepping"
                                      % RESET_EVERY,
                       "launches_per_step": 1, "stream_ms_per_step": stream_s / K * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": measured_traffic(D, N),
                         "kernel": "ct_step_sfx_kernel", "kernel_avg_us": kern_avg_s * 1e6,
                         "how": "HIP events around the K timed launches on the launch stream / K (upper bound: includes gaps)",
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "algorithmic_bytes_per_env_step": bytes_launch / N,
                         "state_bytes_per_env_step": state_bytes(D),
                         "note": "achieved uses SURVEY 8d's algorithmic bytes (4 B per queue entry appended/popped); "
                                 "the suffix queue encoding never materialises those entries, so measured HBM traffic "
                                 "(traffic, rocprofv3 PMC, profiles/) is BELOW the algorithmic bytes; by bytes actually moved the "
                                 "launch runs at ~25% of HBM peak and is bound by latency: one wave per SIMD, every wave in the same "
                                 "phase (load burst, serial per-env walk, store burst), 63% of wave cycles in s_waitcnt "
                                 "(profiles/r1_final/SUMMARY.txt)"},
        }
        if roll is not None:
            out["fused_rollout"] = roll
        if graph_sec is not None:
            out["graph_replay"] = graph_sec
        if steady is not None:
            out["steady_state_no_reset"] = steady
        if not args.no_cpu_baseline and world == 1:           # reported at N = 1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(D, args.cpu_seconds, min(3.0, args.cpu_seconds / 4))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
