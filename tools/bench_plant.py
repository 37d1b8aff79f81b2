#!/usr/bin/env python3
"""BASELINE config 4 (builder-defined): 32 768 envs, each env.step() = band-assignment step (2 devices)
+ linear-plant advance to the env's new simulated time on the f64 matrix cores.  Prints one JSON line."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd import _native as nat

# W: past the process's first ~300 launches -- this round's builds show a one-off ~40 ms HOST stall of the HIP runtime around
# the 280th launch of a process (count-based: it moves with the call path, not with the data; no kernel in the trace is long);
# bench.py's calibration windows absorb it, a 32-step warm-up did not
N, K, W = int(os.environ.get("N", 32768)), 256, 352
penv = gymwipe_amd.VecInvertedPendulumEnv(N)             # band-assignment step + plant advance + interpreter feedback
env, plant = penv.network, penv.plant
g = torch.Generator(device="cuda"); g.manual_seed(7)
dev = torch.randint(0, 2, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
dur = torch.randint(0, 20, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
acts = [{"device": dev[i], "duration": dur[i]} for i in range(W + K)]
env.reset()
for i in range(W):
    penv.step(acts[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for i in range(W, W + K):
    if i % 64 == 0:
        env.reset()
    penv.step(acts[i])
e1.record(); torch.cuda.synchronize()
wall = time.perf_counter() - t0
sub = int(plant.get_state("substeps").sum())
# useful flops: 40 per plant substep (2*4*4 + 2*4); issued MFMA flops: 2 MFMAs x 2*16*16*4 per 16 envs per candidate group
print(json.dumps({"workload": "pendulum band-assign env (builder-defined): %d envs, VecInvertedPendulumEnv.step = band-assignment step (sensor + silent controller) + linear plant advance + interpreter feedback" % N,
                  "env_steps_per_s": N * K / wall, "ms_per_step": wall / K * 1e3, "stream_ms_per_step": e0.elapsed_time(e1) / K,
                  "plant_substeps_total": sub, "useful_plant_gflops": 40.0 * sub / (W + K) * K / wall / 1e9}))
