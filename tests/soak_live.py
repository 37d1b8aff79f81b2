#!/usr/bin/env python3
"""Long-run soak of the live-PHY step kernel with per-env geometry, outside the pytest tiers: N envs over L layouts (one oracle
handle per layout), K steps, resets every 64 steps, a Position.set on a random radio every 50 steps.  Outputs compared every
step; integers, flags and clocks bit for bit and received powers within 1e-5 every 128 steps (link powers come from the
device libm).    python tests/soak_live.py [D] [N] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from util import action_stream
from gymwipe_amd import VecCounterTrafficEnv
from oracle.ct_oracle import CtOracle, default_config

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
L, R = 32, D + 1
INT_FIELDS = ("counter", "qlen", "received", "latest_diff", "last_abs", "flags", "n_delivered", "n_popped")
rng = np.random.default_rng(77 + D)
lay = np.zeros((L, R, 2))
ang, rad = rng.uniform(0, 2 * np.pi, (L, D)), rng.uniform(0.6, 3.2, (L, D))
lay[:, :D, 0], lay[:, :D, 1] = rad * np.cos(ang), rad * np.sin(ang)
lay[:, D] = rng.uniform(-0.3, 0.3, (L, 2))
grp = np.arange(N) % L
env = VecCounterTrafficEnv(N, num_devices=D, per_env_geometry=True)
env.set_positions(lay[grp])
orcs = [CtOracle(N // L, D, config=default_config(D, positions=[tuple(p) for p in lay[l, :D]], rrm_pos=tuple(lay[l, D])), nthreads=8)
        for l in range(L)]
t0 = time.time()
env.reset()
for o in orcs:
    o.reset()
CH = 256
for k0 in range(0, K, CH):
    dev, dur = action_stream(1000 + k0, min(CH, K - k0), N, D)
    for kk in range(dev.shape[0]):
        k = k0 + kk
        if k and k % 64 == 0:
            a = env.reset().cpu().numpy()
            for l, o in enumerate(orcs):
                assert (a[l::L] == o.reset()).all(), ("reset", k, l)
        if k % 50 == 25:
            r = int(rng.integers(0, R))
            nx, ny = rng.uniform(-3, 3, L), rng.uniform(-3, 3, L)
            env.set_position(r, nx[grp], ny[grp])
            for l, o in enumerate(orcs):
                o.set_position(r, nx[l], ny[l])
        o_, r_, d_, _ = env.step({"device": torch.from_numpy(dev[kk]), "duration": torch.from_numpy(dur[kk])})
        o_, r_, d_ = o_.cpu().numpy(), r_.cpu().numpy(), d_.cpu().numpy()
        for l, o in enumerate(orcs):
            wo, wr, wd = o.step(dev[kk][l::L], dur[kk][l::L])
            assert (o_[l::L] == wo).all() and (r_[l::L] == wr).all() and (d_[l::L] == wd).all(), (k, l)
        if (k + 1) % 128 == 0 or k == K - 1:
            for f in INT_FIELDS + ("now", "wake"):
                a = env.get_state(f)
                for l, o in enumerate(orcs):
                    assert (np.ascontiguousarray(a[l::L]).view(np.uint8) == o.get(f).view(np.uint8)).all(), (f, k, l)
            a = env.get_state("rx_power")
            for l, o in enumerate(orcs):
                b = o.get("rx_power")
                assert np.max(np.abs(a[l::L] - b) / b) < 1e-5, (k, l)
            print("step %d ok, %.0f s" % (k + 1, time.time() - t0), flush=True)
st = env.check()
print("live-PHY soak ok: %d env-steps (D=%d, %d layouts, moves every 50 steps), delivered %d" % (N * K, D, L, st["delivered"]))
