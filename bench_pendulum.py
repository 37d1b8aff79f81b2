#!/usr/bin/env python3
"""
bench_pendulum.py -- BASELINE config 4 (`python bench.py --config 4`): env.step()/s of the pendulum band-assignment env,
32 768 envs on one MI355X.  One env.step() = ONE kernel launch (gw_pendulum_step): the band-assignment step of the env's
network (angle sensor + silent controller + RRM, as the reference ships it) + the linear plant x <- A x + B u advanced to
the env's new clock on the f64 matrix cores (v_mfma_f64_16x16x4_f64) + InvertedPendulumInterpreter's feedback.

The plant is BUILDER-DEFINED (the reference's is an ODE rigid-body world in an env that cannot be constructed), so this
line carries no reference-parity claim; same window scheme, contract keys, roofline and cpu_baseline objects as bench.py.
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12
F64_MATRIX_PEAK = 78.6e12      # AMD's MI355X spec: FP64 matrix = FP64 vector rate (MI355X_MICROARCH.md lists no f64 MFMA row)
RESET_EVERY = 64
SEED = 1234
MIN_TIMED_S = 0.25
D = 2


def algorithmic_bytes(env_steps, appended, popped, plant_updates):
    """SURVEY 8d's network bytes for D = 2 + the plant: state x (32 B) and last-update time (8 B) read and written, input u
    (8 B) read, for every env whose clock advanced; I/O: actions 8 B in, obs int32 + reward f32 + angle f64 = 16 B out."""
    return env_steps * (8 + 16 + 2 * (12 + 20 * D)) + 4 * (appended + popped) + plant_updates * (2 * 40 + 8)


def pmc_profile(N):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            for row in json.load(fh)["rows"]:
                if row.get("kernel") == "pend_step_kernel" and row["envs"] == N:
                    return row
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(N_gpu, W, K, seconds):
    import numpy as np
    from gymwipe_amd import _native as nat
    from gymwipe_amd.actions import actions_numpy
    from oracle.ct_oracle import CtOracle, default_config
    from oracle.plant_oracle import PlantOracle
    import ctypes as C
    try:
        nproc = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc = os.cpu_count() or 1
    cores = max(1, min(nproc, 16))
    n_env = 256 * cores
    pc = nat.PlantConfig()
    nat.check(nat.lib().gw_plant_config_default(C.byref(pc), n_env))
    cfg = default_config(2, positions=[(0.0, 0.0), (0.0, -1.0)], rrm_pos=(0.0, 1.0), mult=[1, 0], dest=[1, 0])
    net = CtOracle(n_env, 2, config=cfg, nthreads=cores)
    plant = PlantOracle(n_env, list(pc.A), list(pc.B), pc.dt, list(pc.x0), pc.u0)
    dev, dur = actions_numpy(SEED, 0, n_env, 0, W + K, 2)
    done_steps, t0 = 0, time.perf_counter()
    while True:
        for k in range(W + K):
            if k % RESET_EVERY == 0:
                net.reset()
            net.step(dev[k], dur[k])
            plant.update(net.get("now"))
        done_steps += (W + K) * n_env
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": done_steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port", "nproc": nproc,
            "sample": "envs 0..%d of the GPU's action stream, windows of reset + %d steps: network oracle on %d OpenMP threads + scalar "
                      "plant recurrence (oracle/plant_oracle.c, one thread), %.1f s" % (n_env - 1, W + K, cores, el)}


def main(args, emit=print):
    import torch
    from gymwipe_amd import VecInvertedPendulumEnv
    from gymwipe_amd.actions import actions_torch

    if args.gpus != 1:
        raise SystemExit("--config 4 is a one-GPU configuration (BASELINE.json configs[3])")
    N, K, W = (args.envs or 32768), args.steps, args.warmup
    torch.cuda.set_device(0)
    penv = VecInvertedPendulumEnv(N)
    net, plant = penv.network, penv.plant
    a_dev, a_dur = actions_torch(SEED, 0, N, 0, W + K, D, device="cuda")
    acts = [{"device": a_dev[i], "duration": a_dur[i]} for i in range(W + K)]

    def one(i):
        if i % RESET_EVERY == 0:
            net.reset()
        penv.step(acts[i])

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def window(with_stats):
        for i in range(W):
            one(i)
        s0 = (net.stats(), int(plant.get_state("substeps").sum())) if with_stats else None
        torch.cuda.synchronize()
        ev[0].record()                              # (stream-time marker, enqueued before the wall clock starts: see bench.py)
        t0 = time.perf_counter()
        for i in range(W, W + K):
            one(i)
        ev[1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        delta = None
        if with_stats:
            s1 = net.stats()
            delta = {k: s1[k] - s0[0][k] for k in ("steps", "appended", "popped", "transmissions")}
            delta["substeps"] = int(plant.get_state("substeps").sum()) - s0[1]
        return wall, ev[0].elapsed_time(ev[1]) * 1e-3, delta

    window(False)
    cal_wall, cal_stream, _ = window(False)
    R = args.repeats if args.repeats > 0 else max(1, min(20000, int(math.ceil(MIN_TIMED_S / max(cal_stream, 1e-6))),
                                                          int(30.0 / (cal_wall * (W + K) / max(K, 1) + 1e-3)) or 1))
    stride = max(1, R // 32)
    walls, streams, deltas = [], [], []
    for r in range(R):
        wall, stream_s, delta = window(r % stride == 0)
        walls.append(wall)
        streams.append(stream_s)
        if delta:
            deltas.append(delta)
    net.check()
    wall_total, stream_total = sum(walls), sum(streams)
    kern_avg_s = stream_total / (K * R)
    n_st = max(1, len(deltas))
    mean = lambda key: sum(d[key] for d in deltas) / n_st
    env_steps, substeps = mean("steps"), mean("substeps")
    bytes_launch = algorithmic_bytes(env_steps, mean("appended"), mean("popped"), env_steps) / K
    achieved = bytes_launch / kern_avg_s
    # matrix-core work per launch: every wave runs 4 rounds x ceil(max substeps of the round / 4) groups x 2 MFMAs of
    # 2*16*16*4 flop; useful flops are 40 per substep (2*4*4 + 2*4)
    useful_flops = 40.0 * substeps / K
    prof = pmc_profile(N)
    roof = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
            "traffic": ((2 * prof["fetch_kb"] + prof["write_kb"]) * 1024.0) if prof else None,
            "traffic_source": (prof.get("source") if prof else None),
            "kernel": "pend_step_kernel", "kernel_avg_us": kern_avg_s * 1e6, "launches_per_step": 1,
            "how": "HIP events on the launch stream around each window's K timed launches, summed over %d windows, / (K * windows)" % R,
            "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_env_step": bytes_launch / N,
            "mfma": {"instruction": "v_mfma_f64_16x16x4_f64", "useful_gflops": useful_flops / kern_avg_s / 1e9,
                     "plant_substeps_per_env_step": substeps / max(env_steps, 1),
                     "peak_tflops": F64_MATRIX_PEAK / 1e12,
                     "busy_frac": (prof.get("mfma_busy_frac") if prof else None),
                     "note": "the plant update is 40 useful flop per 1 ms substep: the matrix cores are idle almost all of the "
                             "launch by construction; the bound is the network walk's latency and HBM, not MFMA"}}
    if prof:
        roof["frac_moved"] = roof["traffic"] / kern_avg_s / HBM_PEAK
    us = [w / K * 1e6 for w in walls]
    out = {"metric": "env.step()/s, InvertedPendulum band-assign env (linear plant via MFMA), 32 768 envs",
           "value": N * K * R / wall_total, "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": W,
           "ms_per_step": wall_total / (K * R) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "repeats": R,
           "us_per_step": {"min": min(us), "mean": sum(us) / len(us), "max": max(us)},
           "config": {"workload": "InvertedPendulum sensor+controller band-assign env (builder-defined linear plant x<-Ax+Bu via MFMA), "
                                  "%d envs, network reset every %d steps" % (N, RESET_EVERY),
                      "envs_per_gpu": N, "devices": D,
                      "window": "reset -> %d warm-up steps -> %d timed steps, repeated %d times" % (W, K, R),
                      "parity": "unpinned against the reference (its env cannot be constructed); kernel vs own oracles only"},
           "roofline": roof}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, W, K, max(2.0, args.cpu_seconds))
    emit(json.dumps(out))


if __name__ == "__main__":
    raise SystemExit("run as: python bench.py --config 4")
