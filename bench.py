#!/usr/bin/env python3
"""
bench.py -- env.step()/s of the vectorised CounterTraffic band-assignment env on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either under an external launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or plain `python bench.py --gpus N ...`:
    then THIS process -- before it has imported torch or touched HIP -- starts N fresh rank processes of itself
    (launch_ranks), relays rank 0's one JSON line and exits non-zero if any rank does.

A "step" is ONE env.step() of every environment of the batch: one launch of the HIP step kernel over
65 536 envs x 4 devices per GPU (BASELINE.json configs[1]).  Actions are synthetic: a counter-based
generator (gymwipe_amd/actions.py: a pure function of seed, global env id and step index) evaluated on the
GPU before the timed region and, identically, by the CPU baseline.

One WINDOW is  reset() -> W untimed warm-up steps -> K timed steps  (reset again at every 64th step of a
window, SURVEY.md 8d, so that the data-carrying phase stays in play).  The timed K steps are bracketed by
barrier + torch.cuda.synchronize() on both sides.  Because K steps of a 7 us kernel are far too short to
time once (the driver's --steps 20 is 0.2 ms), the window is REPEATED until >= 0.25 s of timed stream time
have accumulated; every repeat starts from a reset, so every repeat times the same step indices after a
reset (`timed_step_indices_after_reset`).  `value` = all env-steps of all timed regions / the sum over
windows of the slowest rank's wall time.  Rank 0 prints ONE JSON line.

At N > 1 every env.step() of the timed region is followed by the end-of-step observation gather the north star names:
ONE RCCL all-gather of the rank's 9*N-byte (obs, reward, done) record, ordered on the step's stream so that step k+1 is
launched behind it -- every rank holds every env's feedback of step k before step k+1 runs, as the reference's caller
acts on every observation (agents/dqn_counter_traffic.py:63-70).  `value` at N > 1 is THAT form (`config.obs_gather` =
"per-step").  Two relaxed forms are reported beside it as named secondaries with the lateness they trade for speed:
`pipelined_gather` (the gather of step k overlaps step k+1: observations one step late) and `chunked_gather` (one byte
per env-step, one all-gather per 16 / 64 steps: up to a chunk late).  The line's `ranks` object is the evidence that RCCL
met N distinct GPUs: world size as torch.distributed reports it, the PCI bus id and device name of every rank
(all-gathered), the backend and its version.

  roofline     HBM bound.  `achieved` = SURVEY 8d's ALGORITHMIC bytes per launch / average launch duration (HIP events
               on the launch stream around each timed region, / K), `frac` = achieved / 8 TB/s.  Algorithmic bytes per
               env-step: B(D) = 17 + 2*(12 + 20*D) + 4*(k_app + k_pop), k_app/k_pop counted by the kernel.  Because the
               kernel's queue encoding never materialises queue entries, the bytes it MOVES are fewer:
               `frac_moved` = PMC traffic per launch (profiles/traffic.json, rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this
               shape) / the same duration / peak, and `min_bytes_per_env_step` is the encoding's own read+write minimum.
  cpu_baseline the C oracle (scalar restatement of the reference algorithm, oracle/ct_oracle.c) timed on this box's host
               cores on a bounded sample of the same windows with the same action stream
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
RESET_EVERY = 64
SEED = 1234
MIN_TIMED_S = 0.25         # timed stream time to accumulate over the repeated windows
MAX_REPEATS = 20000
MAX_WALL_S = 30.0          # bound on the whole repeated measurement


def algorithmic_bytes(D, env_steps, appended, popped):
    return env_steps * (17 + 2 * (12 + 20 * D)) + 4 * (appended + popped)


def state_bytes(D):
    """Bytes the default kernel itself must load + store per env-step: the minimum of its encoding."""
    rb = 16 * ((2 * D + 1 + 15) // 16)
    return (16 + 16 + 16 + rb + 8) + (16 + 4 + 8 + rb + 9)       # (+ 4-byte counter atomics only where a packet was popped / delivered)


def measured_traffic(D, N):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 for the gfx950
    wide-load under-count + WRITE_SIZE, both in KB) -- only if a profile of this exact shape is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            for row in json.load(fh)["rows"]:
                if row["devices"] == D and row["envs"] == N and row.get("kernel", "ct_step_sfx_kernel") == "ct_step_sfx_kernel":
                    return (2 * row["fetch_kb"] + row["write_kb"]) * 1024.0, row.get("source", "profiles/traffic.json")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, cores)


def cpu_baseline(D, W, K, seconds_target=10.0, single_thread_seconds=2.5, other_devices=(2, 4, 16), other_seconds=2.0):
    """The oracle on the host cores over the SAME windows (reset -> W + K steps, reset every 64) with the SAME counter-based
    action stream (envs 0 .. n-1 of rank 0): all threads of this process's CPU share, one thread beside it, and short
    samples for the other device counts of BASELINE's configs."""
    from gymwipe_amd.actions import actions_numpy
    from oracle.ct_oracle import CtOracle
    nproc = host_cores()
    cores = max(1, min(nproc, 16))           # the GPU box's CPU share for one GPU

    def timed(dd, nthreads, seconds):
        n_env = 256 * nthreads
        dev, dur = actions_numpy(SEED, 0, n_env, 0, W + K, dd)
        orc = CtOracle(n_env, dd, nthreads=nthreads)
        orc.reset()
        for k in range(min(8, W + K)):
            orc.step(dev[k], dur[k])             # warm-up
        done_steps, t0 = 0, time.perf_counter()
        while True:
            for k in range(W + K):
                if k % RESET_EVERY == 0:
                    orc.reset()
                orc.step(dev[k], dur[k])
            done_steps += (W + K) * n_env
            el = time.perf_counter() - t0
            if el >= seconds:
                return done_steps / el, n_env, el

    v, n_env, el = timed(D, cores, seconds_target)
    v1, _, el1 = timed(D, 1, single_thread_seconds)
    out = {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port", "nproc": nproc,
           "sample": "envs 0..%d of the GPU's action stream, windows of reset + %d steps (reset every %d), D=%d, repeated for %.1f s "
                     "on %d threads (OpenMP)" % (n_env - 1, W + K, RESET_EVERY, D, el, cores),
           "single_thread_value": v1,
           "single_thread_sample": "envs 0..255, same windows, %.1f s on 1 thread" % el1}
    by = {}
    for dd in other_devices:
        if dd == D:
            by[str(dd)] = {"value": v, "cores": cores}
        elif other_seconds > 0:
            vv, _, _ = timed(dd, cores, other_seconds)
            by[str(dd)] = {"value": vv, "cores": cores}
    out["by_devices"] = by
    return out


_REAL_STDOUT = None


def emit(line):
    """The ONE line this program prints: written to the stdout the process was started with.  Everything else that lands on
    fd 1 (RCCL prints a five-line version banner there when its communicator is created) is sent to stderr instead."""
    data = (line + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def _free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch_ranks(n, child_argv, timeout=None, env=None, grace=10.0):
    """Start `n` fresh rank processes of `child_argv` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in
    their environment), relay what rank 0 writes to its stdout, and return the job's exit code: 0 if every rank exited 0,
    else the first non-zero code seen (the other ranks are then terminated -- by their exact PIDs -- after `grace` seconds,
    since a rank whose peer died would wait in its next collective for ever).  The ranks are CHILD processes, never an exec
    of this one; the caller must not have initialised HIP (bench.py calls this before importing torch).  The other ranks'
    stdout goes to this process's stderr, so that stdout carries rank 0's single JSON line and nothing else."""
    import subprocess
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in base:
        base["MASTER_PORT"] = str(_free_port())
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["GW_BENCH_SELF_LAUNCHED"] = "1"
    procs = []
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
            procs.append(subprocess.Popen(list(child_argv), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(),
                                          stderr=None, cwd=os.getcwd()))
        # rank 0's stdout is read by a thread so that a full pipe can never stall it
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = None if not timeout else time.monotonic() + float(timeout)
        rc, failed_at = 0, None
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad and rc == 0:
                rc, failed_at = bad[0], time.monotonic()
                sys.stderr.write("bench.py: rank(s) %s exited with %s; stopping the job\n"
                                 % ([i for i, c in enumerate(codes) if c not in (None, 0)], bad))
            if all(c is not None for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline and rc == 0:
                rc, failed_at = 124, time.monotonic() - grace
                sys.stderr.write("bench.py: ranks still running after %.0f s; stopping the job\n" % float(timeout))
            if failed_at is not None:
                late = time.monotonic() - failed_at
                for p_ in procs:
                    if p_.poll() is None:
                        if late > 2 * grace:
                            p_.kill()
                        elif late > grace:
                            p_.terminate()
            time.sleep(0.05)
        reader.join(timeout=5.0)
        out = b"".join(c for c in chunks if c)
        if out:
            os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, out)
        return rc
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
        for p_ in procs:
            try:
                p_.wait(timeout=5.0)
            except Exception:
                pass


def device_identity(torch, index):
    """What tells two GPUs apart: the PCI bus id HIP reports for the device this rank computes on, its name and (when torch
    exposes it) its UUID."""
    import ctypes
    ident = {"index": int(index), "name": torch.cuda.get_device_name(index), "pci_bus_id": None}
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            ident["pci_bus_id"] = buf.value.decode()
    except Exception as exc:                                   # evidence only: never costs the measurement
        ident["pci_bus_id_error"] = repr(exc)
    props = torch.cuda.get_device_properties(index)
    if ident["pci_bus_id"] is None and hasattr(props, "pci_bus_id"):
        ident["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    if hasattr(props, "uuid"):
        ident["uuid"] = str(props.uuid)
    ident["pid"] = os.getpid()
    ident["visible"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    return ident


def main():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)      # 64 warm-up + 1 024 timed steps
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65 536; 32 768 for --config 4)")
    ap.add_argument("--devices", type=int, default=4, help="senders per env (D)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4),
                    help="BASELINE.json config: 2 = CounterTraffic D=4 (the headline), 3 = D=16, 4 = pendulum env (builder-defined plant)")
    ap.add_argument("--repeats", type=int, default=0, help="windows to time (0 = as many as give %.2f s of timed stream time)" % MIN_TIMED_S)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="length of the CPU-baseline sample")
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: skip the end-of-step observation gather (the only exchange of the path)")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--no-graph", action="store_true", help="skip the secondary hipGraph-replay measurement")
    ap.add_argument("--no-steady", action="store_true", help="skip the secondary no-reset steady-state measurement")
    ap.add_argument("--no-secondaries", action="store_true", help="headline only (profiling runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher gave us ranks, so start them -- from a process that has not
        # imported torch or made a HIP call -- and relay rank 0's line
        rc = launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                          timeout=float(os.environ.get("GW_BENCH_LAUNCH_TIMEOUT", "1700")))
        sys.exit(rc)
    if args.no_secondaries:
        args.no_rollout = args.no_graph = args.no_steady = True
    if args.config == 3:
        args.devices = 16
    if args.config == 4:
        import bench_pendulum                                    # tools-level module next to this file
        return bench_pendulum.main(args, emit)
    if args.envs is None:
        args.envs = 65536

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL; must precede HIP initialisation
    import torch
    import torch.distributed as dist
    import gymwipe_amd
    from gymwipe_amd.actions import actions_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with --nproc-per-node %d, or with no "
                         "launcher at all (it then starts its own ranks)" % (args.gpus, world, args.gpus))
    # GW_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # host-side collectives); the driver's runs use the default, RCCL with one GPU per rank.
    backend = os.environ.get("GW_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend != "nccl" or local >= ndev:       # rehearsal, or a launcher that shows each rank only its own GPU
        local = local % ndev
    torch.cuda.set_device(local)
    dev_t = torch.device("cuda", local)
    # GW_BENCH_FORCE_GATHER=1 (tests): run the N > 1 code path -- process group, chunked gather over RCCL, per-step gather --
    # with a single rank, so that a one-GPU box exercises every collective call the multi-GPU runs make
    multi = world > 1 or bool(os.environ.get("GW_BENCH_FORCE_GATHER"))
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev_t)
        else:
            dist.init_process_group(backend)
    red_dev = dev_t if backend == "nccl" else "cpu"

    N, D, K, W = args.envs, args.devices, args.steps, args.warmup
    env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D, device=dev_t)

    # ---- who is in the job (the line's `ranks` object): world size as the process group reports it and every rank's GPU ----
    ranks_info = None
    if multi:
        mine = device_identity(torch, local)
        mine["rank"] = rank
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        bus = [e.get("pci_bus_id") for e in everyone]
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "devices": bus, "distinct_devices": len(set(b for b in bus if b)),
                      "device_names": sorted(set(e.get("name") for e in everyone)),
                      "per_rank": everyone,
                      "launcher": "bench.py launch_ranks (self-started child processes)" if os.environ.get("GW_BENCH_SELF_LAUNCHED")
                                  else ("external (RANK/WORLD_SIZE from the environment)" if "RANK" in os.environ and world > 1 else "single process")}
        if backend == "nccl":
            try:
                ranks_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:
                ranks_info["rccl_version"] = repr(exc)
        if backend == "nccl" and ranks_info["distinct_devices"] != ranks_info["world_size"] and world > 1:
            raise SystemExit("RCCL job of %d ranks sees %d distinct GPUs (%s): one GPU per rank is the contract"
                             % (world, ranks_info["distinct_devices"], bus))

    # outputs as three views of ONE packed record buffer so that the per-step observation
    # gather is a single RCCL all-gather without a packing kernel (gymwipe_amd/sharding.py)
    from gymwipe_amd.sharding import ChunkedFeedbackGather, ObservationGather, PipelinedGather, StepRecord
    rec = StepRecord(N, dev_t)
    env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done
    gathering = multi and not args.no_gather
    # steps per all-gather of the CHUNKED secondary: 64 for long timed regions; a short region (the driver's 20 steps) gathers
    # every 16 steps, so that most of its feedback travels while the region is still stepping
    GATHER_EVERY = RESET_EVERY if args.steps >= 2 * RESET_EVERY else max(1, min(16, args.steps))

    # the headline's exchange at N > 1: ONE all-gather of this rank's 9*N-byte record after EVERY step, ordered on the step's stream
    step_gather = None
    gather_entry = None
    if gathering:
        if backend == "nccl":
            og = ObservationGather(rec, world)
            step_gather, gather_entry = og.step_done, og.entry
        else:                                                    # rehearsal: host-side collective (ranks share a GPU)
            host_rec = torch.zeros(rec.nbytes, dtype=torch.uint8)
            host_out = torch.zeros(world * rec.nbytes, dtype=torch.uint8)
            gather_entry = "all_gather_into_tensor on host copies (%s rehearsal)" % backend

            def step_gather():
                host_rec.copy_(rec.buf)
                dist.all_gather_into_tensor(host_out, host_rec)

    # this rank's shard of the global action stream: envs [rank*N, (rank+1)*N), steps [0, W+K) of a window
    a_dev, a_dur = actions_torch(SEED, rank * N, (rank + 1) * N, 0, W + K, D, device=dev_t)
    acts = [{"device": a_dev[i], "duration": a_dur[i]} for i in range(W + K)]

    from gymwipe_amd import StepOutputs
    own = StepOutputs(rec.obs, rec.reward, rec.done)     # every step writes the same record

    if step_gather is None:
        def one(i):
            if i % RESET_EVERY == 0:
                env.reset()
            env.step(acts[i], own)
    else:
        def one(i):
            if i % RESET_EVERY == 0:
                env.reset()
            env.step(acts[i], own)
            step_gather()                             # step i+1 is launched behind the gather of step i's record

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def window(with_stats):
        """reset -> W warm-up steps -> K timed steps; returns (wall s, stream s, stats delta or None)."""
        for i in range(W):
            one(i)
        s0 = env.stats() if with_stats else None   # (synchronises)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        ev[0].record()                              # (the stream-time marker: enqueued, not waited for, before the clock starts, so
        t0 = time.perf_counter()                    #  that its 2 us of host time are not billed to the K steps it brackets)
        for i in range(W, W + K):
            one(i)
        ev[1].record()
        torch.cuda.synchronize()                      # (N > 1: the stream is ordered behind the last step's gather)
        wall = time.perf_counter() - t0
        if multi:
            dist.barrier()
        stream_s = ev[0].elapsed_time(ev[1]) * 1e-3
        delta = None
        if with_stats:
            s1 = env.stats()
            delta = {k: s1[k] - s0[k] for k in ("steps", "appended", "popped", "transmissions")}
        return wall, stream_s, delta

    # ---- calibration window (also the process warm-up): how many repeats give MIN_TIMED_S of timed stream time ----
    window(False)
    cal_wall, cal_stream, _ = window(False)
    cal = torch.tensor([cal_stream, cal_wall], dtype=torch.float64, device=red_dev)
    if multi:
        dist.all_reduce(cal, op=dist.ReduceOp.MAX)
    cal_stream, cal_wall = float(cal[0]), float(cal[1])
    if args.repeats > 0:
        R = args.repeats
    else:
        R = int(math.ceil(MIN_TIMED_S / max(cal_stream, 1e-6)))
        per_window = cal_wall * (W + K) / max(K, 1) + 1e-3
        R = max(1, min(R, MAX_REPEATS, int(MAX_WALL_S / per_window) or 1))
    stat_stride = max(1, R // 32)

    # ---- the measurement: R windows --------------------------------------------------------------------
    walls, streams, deltas = [], [], []
    for r in range(R):
        wall, stream_s, delta = window(r % stat_stride == 0)
        walls.append(wall)
        streams.append(stream_s)
        if delta is not None:
            deltas.append(delta)
    env.check()
    env._fb = None                                   # (the secondaries below step without the gather's byte rows)

    # the job's time per window is the slowest rank's
    t = torch.tensor(walls, dtype=torch.float64, device=red_dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    walls_max = t.cpu().tolist()
    wall_total = sum(walls_max)
    stream_total = sum(streams)
    kern_avg_s = stream_total / (K * R)
    kern_how = ("HIP events on the launch stream around each window's K timed launches, summed over the %d windows, / (K * windows) "
                "(upper bound on the kernel's own duration: includes inter-launch gaps and the reset kernel every %d steps)"
                % (R, RESET_EVERY))
    if step_gather is not None:
        # with a gather behind every step the stream time between the two events is step + collective; the KERNEL's average
        # duration (what the roofline prices) comes from extra windows of the same K steps without the gather
        saved, step_gather_saved = one, step_gather

        def one(i):                                   # noqa: F811 -- window() looks `one` up at call time
            if i % RESET_EVERY == 0:
                env.reset()
            env.step(acts[i], own)
        ks, kn = 0.0, 0
        for _ in range(max(3, min(R, int(0.1 / max(cal_stream, 1e-6)) or 1))):
            _, st_s, _ = window(False)
            ks, kn = ks + st_s, kn + 1
        one = saved
        kk = torch.tensor([ks / (K * kn)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        kern_avg_s = float(kk.item())
        kern_how = ("HIP events around %d extra windows of the same K timed launches WITHOUT the per-step gather (the headline's "
                    "stream time per step, %.2f us, is step + collective), slowest rank" % (kn, stream_total / (K * R) * 1e6))
    # The K launches of a timed region run back-to-back on one stream, so (HIP event at the end - HIP event at the
    # start) / K is the average launch duration including inter-kernel gaps and the reset kernel every 64 steps: an
    # UPPER bound on the kernel's own time (event pairs around every launch add ~2.5 us each).

    # ---- secondaries (a failure is reported in the JSON, it does not cost the headline) ----
    def guarded(fn):
        try:
            return fn()
        except Exception as exc:
            return {"error": repr(exc)}

    def best_of(fn, n=5):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(n):
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            best = el if best is None or el < best else best
        return best

    def secondary_rollout():
        """The same window through gw_rollout: one persistent launch per <= 64 pre-staged steps."""
        r_obs = torch.empty((RESET_EVERY, N), dtype=torch.int32, device=dev_t)
        r_rew = torch.empty((RESET_EVERY, N), dtype=torch.float32, device=dev_t)
        r_done = torch.empty((RESET_EVERY, N), dtype=torch.uint8, device=dev_t)
        chunks = [(i, min(i + RESET_EVERY, W + K)) for i in range(0, W + K, RESET_EVERY)]

        def run_rollouts():
            for lo, hi in chunks:
                env.reset()
                env.rollout(a_dev[lo:hi], a_dur[lo:hi], out=(r_obs[:hi - lo], r_rew[:hi - lo], r_done[:hi - lo]))
        roll_wall = best_of(run_rollouts)
        env.check()
        return {"env_steps_per_s_this_rank": N * (W + K) / roll_wall, "ms_per_step": roll_wall / (W + K) * 1e3,
                "what": "gw_rollout: one persistent launch per <= %d pre-staged steps (ct_rollout_sfx.hip), the window's %d steps "
                        "(warm-up included), same outputs; not the headline because env.step() is one call per step" % (RESET_EVERY, W + K)}

    def secondary_graph():
        """The window's launches replayed from a hipGraph of reset + 64 steps."""
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
            for lo in range(0, W + K, G):
                g_dev.copy_(a_dev[lo:lo + G])
                g_dur.copy_(a_dur[lo:lo + G])
                graph.replay()
        g_wall = best_of(run_graphs)
        env.check()
        return {"env_steps_per_s_this_rank": N * (W + K) / g_wall, "ms_per_step": g_wall / (W + K) * 1e3,
                "what": "the window's gw_step launches replayed from a hipGraph of reset + %d steps (host launch loop removed; "
                        "includes the copy of each chunk's actions into the graph's input buffers)" % G}

    def secondary_steady():
        """Steady state without resets (SURVEY 8d asks for it separately): after ~0.2 s of simulated time the
        packets have outgrown every window and steps carry no data any more."""
        env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done
        n_ss = max(64, min(256, W + K))
        for j in range(64):
            env.step(acts[j % (W + K)], own)

        def run():
            for j in range(n_ss):
                env.step(acts[j % (W + K)], own)
        ss_wall = best_of(run, 3)
        return {"env_steps_per_s_this_rank": N * n_ss / ss_wall, "ms_per_step": ss_wall / n_ss * 1e3, "steps": n_ss,
                "what": "no reset for >= 64 steps before and during the timed steps: queues hold only packets too long "
                        "for any window, so a step is the announcement plus counter ticks"}

    def timed_job(run):
        """One untimed pass, then one pass of the W+K-step window timed between barriers; the slowest rank's wall time."""
        run()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=red_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def secondary_no_gather():
        """The same ranks stepping their shards with NO exchange at all (what a learner co-located with each shard would see):
        the difference to `value` is what the per-step gather costs."""
        def run():
            for i in range(W + K):
                if i % RESET_EVERY == 0:
                    env.reset()
                env.step(acts[i], own)
        el = timed_job(run)
        return {"env_steps_per_s": world * N * (W + K) / el, "ms_per_step": el / (W + K) * 1e3, "observations_late_by_steps": None,
                "what": "no observation gather: every rank keeps its shard's feedback to itself"}

    def secondary_pipelined_gather():
        """The per-step record gather, double-buffered: the all-gather of step k runs on RCCL's stream while step k+1's
        kernel runs, and is waited for only before step k+2 overwrites its record."""
        pg2 = PipelinedGather(N, dev_t, world)
        outs = [StepOutputs(r.obs, r.reward, r.done) for r in pg2.records]
        if backend != "nccl":                                  # rehearsal: host-side collective on copies of the records
            host = [(torch.zeros(r.nbytes, dtype=torch.uint8), torch.zeros(world * r.nbytes, dtype=torch.uint8)) for r in pg2.records]

        def run():
            for i in range(W + K):
                if i % RESET_EVERY == 0:
                    env.reset()
                j = pg2.k % pg2.depth
                pg2.current()                                  # waits for the gather that last used record j
                env.step(acts[i], outs[j])
                if backend == "nccl":
                    pg2.submit()
                else:
                    host[j][0].copy_(pg2.records[j].buf)
                    pg2.pending[j] = dist.all_gather_into_tensor(host[j][1], host[j][0], async_op=True)
                    pg2.k += 1
            pg2.drain()
        el = timed_job(run)
        return {"env_steps_per_s": world * N * (W + K) / el, "ms_per_step": el / (W + K) * 1e3, "observations_late_by_steps": 1,
                "bytes_per_rank_per_step": rec.nbytes,
                "what": "the 9*N-byte record of step k all-gathered while step k+1 runs (two records, alternating): every rank "
                        "holds step k's observations when step k+2 is launched, i.e. an agent acts on observations one step old"}

    def secondary_chunked_gather():
        """One byte per env-step, one all-gather per GATHER_EVERY steps, overlapped with the next chunk's stepping."""
        if backend == "nccl":
            # pack=None: the step kernel writes each step's one-byte row itself (gw_step_fb), no packing launch per chunk
            pipe = ChunkedFeedbackGather(N, dev_t, None, world, chunk=GATHER_EVERY)
        else:                                                    # rehearsal: pack on the GPU, gather on the host
            stage = torch.empty((GATHER_EVERY, N), dtype=torch.uint8, device=dev_t)

            def pack_to_host(o, r, d, out):
                env.pack_feedback(o, r, d, stage[:o.shape[0]])
                out.copy_(stage[:o.shape[0]])
            pipe = ChunkedFeedbackGather(N, dev_t, pack_to_host, world, chunk=GATHER_EVERY)
            pipe.packed = [torch.zeros((GATHER_EVERY, N), dtype=torch.uint8) for _ in range(pipe.depth)]
            pipe.gathered = [torch.zeros((world, GATHER_EVERY, N), dtype=torch.uint8) for _ in range(pipe.depth)]
        fused_rows = pipe._pack is None

        def run():
            v = pipe.begin()
            for i in range(W + K):
                if i % RESET_EVERY == 0:
                    env.reset()
                if fused_rows:                        # this step's outputs go into the current chunk record, its one-byte feedback
                    env.step(acts[i], v)              # row included (gw_step_fb): v is the chunk slot's StepOutputs
                else:                                 # rehearsal: typed outputs into the chunk record, packing kernel per chunk
                    env._obs, env._rew, env._done = v[0], v[1], v[2]
                    env.step(acts[i])
                v = pipe.advance()                    # chunk full: async all-gather
            pipe.drain()                              # the job is done when the last gather has landed
        el = timed_job(run)
        env._obs, env._rew, env._done = rec.obs, rec.reward, rec.done
        env._fb = None
        return {"env_steps_per_s": world * N * (W + K) / el, "ms_per_step": el / (W + K) * 1e3,
                "observations_late_by_steps": GATHER_EVERY, "bytes_per_rank_per_step": N, "steps_per_all_gather": GATHER_EVERY,
                "entry": pipe.entry,
                "what": "feedback compressed to ONE byte per env-step (lossless for the built-in interpreter; written by the step "
                        "kernel itself, gw_step_fb) and all-gathered once per %d steps while the next chunk is stepped: an "
                        "observation reaches the other ranks up to %d steps after its step" % (GATHER_EVERY, GATHER_EVERY)}

    no_gather = pipelined = chunked = None
    if gathering and not args.no_secondaries:
        no_gather = guarded(secondary_no_gather)
        pipelined = guarded(secondary_pipelined_gather)
        chunked = guarded(secondary_chunked_gather)
    roll = guarded(secondary_rollout) if not args.no_rollout else None
    # graph replay at N = 1 only: stream capture next to a live RCCL communicator (whose watchdog thread queries
    # events) is a needless risk for a secondary figure
    graph_sec = guarded(secondary_graph) if (not args.no_graph and not multi and (W + K) % RESET_EVERY == 0) else None
    steady = guarded(secondary_steady) if not args.no_steady else None

    if rank == 0:
        n_st = max(1, len(deltas))
        env_steps = sum(d["steps"] for d in deltas) / n_st
        app = sum(d["appended"] for d in deltas) / n_st
        pop = sum(d["popped"] for d in deltas) / n_st
        txs = sum(d["transmissions"] for d in deltas) / n_st
        bytes_launch = algorithmic_bytes(D, env_steps, app, pop) / K
        achieved = bytes_launch / kern_avg_s
        value = world * N * K * R / wall_total
        traffic, traffic_src = measured_traffic(D, N)
        us = [w / K * 1e6 for w in walls_max]
        us_stream = [s / K * 1e6 for s in streams]
        lo, hi = W % RESET_EVERY, (W % RESET_EVERY) + K - 1
        roof = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic,
                "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this shape; FETCH_SIZE x 2 for "
                                   "the gfx950 wide-load under-count; not measured in this run)") if traffic else None,
                "frac_moved": (traffic / kern_avg_s / HBM_PEAK) if traffic else None,
                "min_bytes_per_env_step": state_bytes(D),
                "frac_min_bytes": state_bytes(D) * N / kern_avg_s / HBM_PEAK,
                "kernel": "ct_step_sfx_kernel", "kernel_avg_us": kern_avg_s * 1e6,
                "how": kern_how,
                "algorithmic_bytes_per_launch": bytes_launch,
                "algorithmic_bytes_per_env_step": bytes_launch / N,
                "transmissions_per_env_step": txs / max(env_steps, 1),
                "note": "frac prices the launch at SURVEY 8d's algorithmic bytes (4 B per queue entry appended or popped). The kernel's "
                        "suffix queue encoding never materialises those entries: it loads and stores min_bytes_per_env_step per env-step "
                        "(frac_min_bytes), and frac_moved is the PMC-measured HBM traffic over the same duration. By bytes actually moved "
                        "the launch is latency-bound, not bandwidth-bound (one wave per SIMD at 65 536 envs; profiles/)."}
        if roof["frac"] > 1.0:
            roof["frac_above_one_because"] = ("algorithmic bytes (%.0f B per env-step, mostly 4 B x %.0f queue appends) exceed what the suffix "
                                              "encoding moves (%d B per env-step): 8d's byte model does not bound this implementation at "
                                              "D = %d; read frac_moved / frac_min_bytes" % (bytes_launch / N, app / max(env_steps, 1), state_bytes(D), D))
        out = {
            "metric": "env.step()/s at 65 536 parallel envs, counter-traffic band-assign",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_total / (K * R) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": R,
            "timed_step_indices_after_reset": [lo, hi] if hi < RESET_EVERY else [lo, hi, "reset again every %d steps" % RESET_EVERY],
            "us_per_step": {"min": min(us), "mean": sum(us) / len(us), "max": max(us),
                            "stream_min": min(us_stream), "stream_mean": sum(us_stream) / len(us_stream), "stream_max": max(us_stream)},
            "timed_stream_s": stream_total, "timed_wall_s": wall_total,
            "config": {"workload": "CounterTrafficEnv, %d devices, %d vectorised envs per GPU, reset every %d steps"
                                   % (D, N, RESET_EVERY),
                       "envs_per_gpu": N, "devices": D, "global_envs": world * N,
                       "window": "reset -> %d warm-up steps -> %d timed steps, repeated %d times; each timed region bracketed by "
                                 "barrier + synchronize; value = env-steps of all timed regions / sum of their wall times. The opening "
                                 "HIP-event marker of a timed region is enqueued (not waited for) just before the wall clock starts, so "
                                 "its ~2 us of host time (~0.1 us per step at K = 20) are outside the wall time" % (W, K, R),
                       "actions": "counter-based generator, seed %d (gymwipe_amd/actions.py), identical for the CPU baseline" % SEED,
                       "obs_gather": ("per-step" if gathering else False) if multi else False,
                       "parallelism": ("independent env shards, one process per GPU, no data-path collective; only exchange: the "
                                       "end-of-step observation gather -- one %s all-gather of each rank's %d-byte (obs, reward, done) "
                                       "record after EVERY env.step(), on the step's stream, so that step k+1 is launched behind the "
                                       "gather of step k (entry: %s)" % ("RCCL" if backend == "nccl" else backend, rec.nbytes, gather_entry)) if gathering
                                      else "independent env shards, one process per GPU, no exchange",
                       "launches_per_step": 1, "stream_ms_per_step": kern_avg_s * 1e3},
            "roofline": roof,
        }
        if ranks_info is not None:
            out["ranks"] = ranks_info
        if multi:
            out["gather_forms"] = {"headline": "per-step (value): observations of step k on every rank before step k+1 is launched",
                                   "no_gather": no_gather, "pipelined_gather": pipelined, "chunked_gather": chunked}
        if roll is not None:
            out["fused_rollout"] = roll
        if graph_sec is not None:
            out["graph_replay"] = graph_sec
        if steady is not None:
            out["steady_state_no_reset"] = steady
        if not args.no_cpu_baseline and world == 1:           # reported at N = 1 only (the other ranks would idle)
            cs = args.cpu_seconds
            out["cpu_baseline"] = cpu_baseline(D, W, K, cs, min(2.5, cs / 4), other_seconds=min(2.0, cs / 5))
        emit(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
