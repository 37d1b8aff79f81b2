#!/usr/bin/env python3
"""Diagnostic: where does a wave of the step kernel spend its cycles?  Needs the stamp build:
   make -C gymwipe_amd/csrc stamps && GW_LIB=$PWD/gymwipe_amd/lib/libgymwipe_amd_stamps.so python tools/stamps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd import _native as nat

N, D = 65536, 4
env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D)
g = torch.Generator(device="cuda"); g.manual_seed(1)
names = ["0 issue table+state loads, write LDS", "1 barrier", "2 state landed (touch ip)", "3 unpack, LDS lookups, consts",
         "4 announcement tx_times", "5 announcement decode, t_end", "6 window loop", "7 tail ticks_to(t_end)",
         "8 other senders + pack", "9 qb store", "10 feedback + stores", "11 counters store"]
rows = []
env.reset()
for k in range(48):
    a = {"device": torch.randint(0, D, (N,), dtype=torch.int32, device="cuda", generator=g),
         "duration": torch.randint(0, 20, (N,), dtype=torch.int32, device="cuda", generator=g)}
    env.step(a)
    torch.cuda.synchronize()
    if k >= 8:
        n_slots = (N + 15) // 16
        out = np.empty((n_slots, 16), np.uint64)
        nat.check(env._L.gw_get_state(env._h, b"stamps", out.ctypes.data, out.nbytes))
        w = out[: N // 64].astype(np.int64)
        rows.append(np.diff(w[:, :13], axis=1))
d = np.concatenate(rows)
print("cycles per wave (s_memtime ticks), median / p90 / mean over %d waves x %d launches" % (N // 64, len(rows)))
for i, n in enumerate(names):
    print("  %-48s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].mean()))
tot = d.sum(axis=1)
print("  %-48s %8.0f %8.0f %8.0f" % ("total in-kernel", np.median(tot), np.percentile(tot, 90), tot.mean()))
# (wave start/end offsets across the launch are not reported: s_memtime counters of different CUs are not comparable)
