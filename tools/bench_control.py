#!/usr/bin/env python3
"""The closed control loop (VecControlLoopEnv) at BASELINE config 4's size: env-steps/s.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd

N, K, W = int(os.environ.get("N", 32768)), 256, 32
env = gymwipe_amd.VecControlLoopEnv(N)
g = torch.Generator(device="cuda"); g.manual_seed(5)
dev = torch.randint(0, 2, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
dur = torch.randint(0, 20, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
for i in range(W):
    env.step({"device": dev[i], "duration": dur[i]})
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(W, W + K):
    env.step({"device": dev[i], "duration": dur[i]})
torch.cuda.synchronize()
wall = time.perf_counter() - t0
rec = env.get_state("received")
print(json.dumps({"workload": "closed control loop (builder-defined): %d envs, sensor + controller + actuator + RRM, plant in the step kernel" % N,
                  "env_steps_per_s": N * K / wall, "ms_per_step": wall / K * 1e3,
                  "sensor_packets_at_controller": int(rec[:, 0].sum()), "commands_at_actuator": int(rec[:, 1].sum()),
                  "plant_substeps": int(env.get_state("substeps").astype("int64").sum()),
                  "flags_or": int(env.get_state("flags").max())}))
