#!/usr/bin/env python3
"""Diagnostic: cycles of the pendulum step kernel's plant epilogue (stamp build, see tools/stamps.py).
stamps 12 = end of the network walk, 13 = substep count known, 14 = states transposed / shuffled, 15 = MFMA loop done."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd import _native as nat

N = 32768
penv = gymwipe_amd.VecInvertedPendulumEnv(N)
env = penv.network
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = []
env.reset()
for k in range(48):
    a = {"device": torch.randint(0, 2, (N,), dtype=torch.int32, device="cuda", generator=g),
         "duration": torch.randint(0, 20, (N,), dtype=torch.int32, device="cuda", generator=g)}
    penv.step(a)
    torch.cuda.synchronize()
    if k >= 8:
        n_slots = (N + 15) // 16 + 1
        out = np.empty((n_slots, 16), np.uint64)
        nat.check(env._L.gw_get_state(env._h, b"stamps", out.ctypes.data, out.nbytes))
        rows.append(np.diff(out[: N // 64].astype(np.int64), axis=1))
d = np.concatenate(rows)
names = ["0..11 network walk"] * 12 + ["12 end of walk -> substep count", "13 LDS transpose + shuffles + max", "14 MFMA loop"]
print("walk total (stamps 0-12)      median %8.0f  p90 %8.0f" % (np.median(d[:, :12].sum(axis=1)), np.percentile(d[:, :12].sum(axis=1), 90)))
for i in (12, 13, 14):
    print("%-30s median %8.0f  p90 %8.0f  mean %8.0f" % (names[i], np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].mean()))
