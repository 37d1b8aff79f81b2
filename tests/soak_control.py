#!/usr/bin/env python3
"""The closed control loop against its event-driven model over many random configurations (geometry, control start
and period, initial plant state).   python tests/soak_control.py [configs] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from gymwipe_amd import VecControlLoopEnv
from oracle import des_model as dm

COUNT = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bits = lambda a: np.asarray(a, np.float64).view(np.uint64)
t0, deliveries, ties = time.time(), 0, 0
for it in range(COUNT):
    N, K = 4, int(rng.integers(40, 110))
    pos = [(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5))) for _ in range(4)]   # sensor, controller, actuator, RRM
    if min(abs(a[0] - b[0]) + abs(a[1] - b[1]) for i, a in enumerate(pos) for b in pos[i + 1:]) < 0.05:
        continue                                                     # (nearly) co-located radios: skip
    start, period = int(rng.integers(0, 60)), int(rng.integers(1, 15))
    x0 = (0.0, 0.0, float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.5, 0.5)))
    env = VecControlLoopEnv(N, ctrl_start_tick=start, ctrl_period_ticks=period, positions=pos, x0=x0)
    models = [dm.ControlLoopModel(positions=pos[:3], rrm_pos=pos[3], start=start, period=period, x0=x0) for _ in range(N)]
    for k in range(K):
        dev = rng.integers(0, 2, N).astype(np.int32)
        dur = rng.integers(0, 20, N).astype(np.int32)
        obs, rew, done, info = env.step({"device": torch.from_numpy(dev), "duration": torch.from_numpy(dur)})
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        st = {f: env.get_state(f) for f in ("now", "x", "u", "angle_deg", "qlen", "received", "n_tx", "commands", "rx_power")}
        for e, m in enumerate(models):
            o, r, d, i = m.step(int(dev[e]), int(dur[e]))
            s = m.snapshot()
            w = (it, k, e)
            assert obs[e] == o and rew[e] == np.float32(r), w
            assert bits(st["now"][e]) == bits(s["now"]) and (bits(st["x"][e]) == bits(s["x"])).all(), w
            assert bits(st["u"][e]) == bits(s["u"]) and bits(st["angle_deg"][e]) == bits(s["angle_deg"]), w
            assert st["qlen"][e].tolist() == s["qlen"][:2] and st["received"][e].tolist() == s["received"][1:], w
            assert int(st["n_tx"][e]) == s["n_tx"] and int(st["commands"][e]) == s["commands"], w
            assert (bits(st["rx_power"][e]) == bits(s["rx_power"])).all(), w
    deliveries += int(env.get_state("received").sum())
    ties += int((env.get_state("flags") & 4).any())
    env.close()
print("control-loop soak ok: %d configurations, %d packets handed up, %d with exact time ties, %.0f s"
      % (COUNT, deliveries, ties, time.time() - t0))
