#!/usr/bin/env python3
"""Receive-mode MACs + enqueues + counter traffic against the event-driven model over longer runs and random layouts
(the Python model is the slow side).   python tests/soak_receive.py [configs] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from test_receive_mode import _snapshot_equal
from gymwipe_amd import VecCounterTrafficEnv
from oracle import des_model as dm

COUNT = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
t0, delivered = time.time(), 0
for it in range(COUNT):
    D = int(rng.integers(2, 5))
    N, K = 4, 60
    ang, rad = rng.uniform(0, 2 * np.pi, D), rng.uniform(0.3, 2.5, D)          # close enough for peers to decode each other
    pos = [(float(r * np.cos(a)), float(r * np.sin(a))) for r, a in zip(rad, ang)]
    traffic = bool(rng.integers(0, 2))
    fl = bool(rng.integers(0, 2))
    dest = [int((i + 1 + rng.integers(0, D - 1)) % D) for i in range(D)]
    env = VecCounterTrafficEnv(N, D, explicit_queue=True, peer_receive=True, positions=pos, dest=dest,
                               counter_traffic=traffic, float_duration=fl)
    models = [dm.CounterTrafficModel(D, peer_receive=True, positions=pos, dest=dest, traffic=traffic, float_duration=fl)
              for _ in range(N)]
    for k in range(K):
        if k % 2 == 0:
            i = int(rng.integers(0, D))
            pb = rng.integers(-1, 60, size=N).astype(np.int32)
            env.enqueue(i, torch.from_numpy(pb).cuda())
            for e in range(N):
                if pb[e] >= 0:
                    models[e].enqueue(i, int(pb[e]))
        dev = rng.integers(0, D, size=N).astype(np.int32)
        dur = rng.integers(0, 20, size=N).astype(np.int32)
        obs, rew, done, _ = env.step({"device": torch.from_numpy(dev).cuda(), "duration": torch.from_numpy(dur).cuda()})
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for e in range(N):
            o, r, d, _ = models[e].step(int(dev[e]), int(dur[e]))
            assert obs[e] == o and rew[e] == r, (it, k, e)
        _snapshot_equal(env, models, "config %d step %d" % (it, k))
    delivered += int(env.get_state("peer_received").sum())
    env.close()
print("receive-mode soak ok: %d configurations, %d packets handed up by peer MACs, %.0f s" % (COUNT, delivered, time.time() - t0))
