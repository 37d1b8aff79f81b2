#!/usr/bin/env python3
"""The live-PHY kernels at the BASELINE shape (4 devices x 65 536 envs, reset every 64 steps): per-env geometry in the default
queue mode (ct_step_dyn.hip) and with explicit queues (the generic kernel's live-PHY instantiation).  One JSON line per mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gymwipe_amd
from gymwipe_amd.actions import actions_torch

N, D, K, W = int(os.environ.get("N", 65536)), int(os.environ.get("D", 4)), 256, 64
MODES = os.environ.get("GW_BENCH_MODES", "suffix,explicit").split(",")
dev, dur = actions_torch(7, 0, N, 0, W + K, D, device="cuda")
rng = np.random.default_rng(3)
pos = np.zeros((N, D + 1, 2))
ang, rad = rng.uniform(0, 2 * np.pi, (N, D)), rng.uniform(1.0, 3.0, (N, D))
pos[:, :D, 0], pos[:, :D, 1] = rad * np.cos(ang), rad * np.sin(ang)
for name, kw in (("live PHY, per-env geometry, suffix queues", {}), ("live PHY, per-env geometry, explicit queues", {"explicit_queue": True})):
    if ("explicit" if kw else "suffix") not in MODES:
        continue
    env = gymwipe_amd.VecCounterTrafficEnv(N, D, per_env_geometry=True, **kw)
    env.set_positions(pos)
    def run(lo, hi):
        for i in range(lo, hi):
            if i % 64 == 0:
                env.reset()
            env.step({"device": dev[i], "duration": dur[i]})
    run(0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, W + K)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = env.check()
    print(json.dumps({"workload": "%s, %d devices x %d envs, reset every 64 steps" % (name, D, N),
                      "env_steps_per_s": N * K / wall, "ms_per_step": wall / K * 1e3, "delivered": st["delivered"]}))
    env.close()
