#!/usr/bin/env python3
"""Diagnostic: where does a wave of the step kernel spend its cycles?  Needs the stamp build:
   make -C gymwipe_amd/csrc stamps && GW_LIB=$PWD/gymwipe_amd/lib/libgymwipe_amd_stamps.so python tools/stamps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd import _native as nat

from gymwipe_amd.actions import actions_torch
N, D = 65536, int(os.environ.get("STAMPS_D", "4"))
# the situation bench.py measures: clock well past the early binades (bench.py's clock keeps growing over thousands of
# windows; right after t = 0 the tick jump declines at every binade end and the plain loop shows up instead), the shared
# counter-based action stream, a reset every W + K steps, stamps taken on the timed indices only
W, K = int(os.environ.get("STAMPS_W", "5")), int(os.environ.get("STAMPS_K", "20"))
EXPLICIT = os.environ.get("STAMPS_EXPLICIT", "0") == "1"        # the generic kernel (ct_step.hip) instead of the default one
LIVE = os.environ.get("STAMPS_LIVE", "0") == "1"                # the live-PHY kernel with per-env geometry (ct_step_dyn.hip)
env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D, start_time=float(os.environ.get("STAMPS_T0", "300.0")),
                                       explicit_queue=EXPLICIT, per_env_stats=EXPLICIT, per_env_geometry=LIVE)
if LIVE:
    rng = np.random.default_rng(3)
    pos = np.zeros((N, D + 1, 2))
    ang, rad = rng.uniform(0, 2 * np.pi, (N, D)), rng.uniform(1.0, 3.0, (N, D))
    pos[:, :D, 0], pos[:, :D, 1] = rad * np.cos(ang), rad * np.sin(ang)
    env.set_positions(pos)
a_dev, a_dur = actions_torch(1234, 0, N, 0, W + K, D, device="cuda")
names = ["0 issue table+state loads, write LDS", "1 barrier", "2 state landed (touch ip)", "3 unpack, LDS lookups, consts",
         "4 announcement tx_times", "5 announcement decode, t_end", "6 window loop", "7 tail ticks_to(t_end)",
         "8 other senders + pack", "9 qb store", "10 feedback + stores", "11 counters store"]
if EXPLICIT:
    names = ["0 issue table+state loads, write LDS, barrier", "1 bad-action test, constants", "2 announcement", "3 queue record, lookups, prefetch",
             "4 window loop", "5 tail ticks", "6 other senders' queues", "7 noise states", "8 feedback values", "9 totals", "10 all stores"]
if LIVE:
    names = ["0 issue loads, write LDS, barrier", "1 action-dependent loads, scalars", "2 announcement", "3 window loop", "4 tail ticks, queue lengths, qb pack",
             "5 rows in registers / walker's part", "6 lane groups' all-pairs pass", "7 walker's stores + counters"]
NS = len(names) + 1
rows = []
rows_loop = []
pops_rows = []
prev_pop = env.get_state("n_popped").astype(np.int64)
for k in range(3 * (W + K)):
    if k % (W + K) == 0:
        env.reset()
    a = {"device": a_dev[k % (W + K)], "duration": a_dur[k % (W + K)]}
    env.step(a)
    torch.cuda.synchronize()
    if k % (W + K) >= W:
        n_slots = (N + 15) // 16 + 1
        out = np.empty((n_slots, 16), np.uint64)
        nat.check(env._L.gw_get_state(env._h, b"stamps", out.ctypes.data, out.nbytes))
        w = out[: N // 64].astype(np.int64)
        if EXPLICIT and not os.environ.get("GW_NO_SPLIT"):      # two waves per 64 envs: even = the walker, odd = the helper
            w = out[: 2 * (N // 64)].astype(np.int64)[0::2]
        rows.append(np.diff(w[:, :NS], axis=1))
        rows_loop.append(w[:, 12:16].copy())
    cur_pop = env.get_state("n_popped").astype(np.int64)
    if k % (W + K) >= W:
        pops_rows.append((cur_pop - prev_pop).reshape(-1, 64).max(axis=1))      # data packets of the wave's busiest lane
    prev_pop = cur_pop
d = np.concatenate(rows)
print("cycles per wave (s_memtime ticks; each stamp itself costs ~100), median / p90 / mean / max over %d waves x %d launches (steps %d..%d after a reset)" % (N // 64, len(rows), W, W + K - 1))
for i, n in enumerate(names):
    print("  %-48s %8.0f %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].mean(), d[:, i].max()))
if EXPLICIT:
    # phases of the window loop's LAST iteration (stamps 12..15 are overwritten by every iteration)
    ph = np.concatenate(rows_loop)
    # (lane 0 of the wave takes the stamps; its last visit to the loop top usually ends at the fit test, so 12 is newer than 13)
    for i, n in enumerate(["12->13 head size, fit test, pop, tx_times", "13->14 decode (+ peer)", "14->15 ticks up to t_e"]):
        sel = (ph[:, i + 1] > ph[:, i]) & (ph[:, i] > 0)
        x = (ph[sel, i + 1] - ph[sel, i])
        if x.size:
            print("  loop phase %-44s median %6.0f mean %6.0f  (%d samples)" % (n, np.median(x), x.mean(), x.size))
tot = d.sum(axis=1)
print("  %-48s %8.0f %8.0f %8.0f %8.0f" % ("total in-kernel", np.median(tot), np.percentile(tot, 90), tot.mean(), tot.max()))
per_launch_max = np.array([r.sum(axis=1).max() for r in rows])
print("  slowest wave of a launch: mean %.0f  (the launch lasts at least this long)" % per_launch_max.mean())
# window-loop cycles against the packet count of the wave's busiest lane: slope = cycles per packet iteration
mp = np.concatenate(pops_rows)
loop = d[:, 3 if LIVE else (4 if EXPLICIT else 6)]
A = np.stack([mp, np.ones_like(mp)], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, loop.astype(np.float64), rcond=None)
print("  window loop ~= %.0f + %.0f x (packets of the busiest lane); busiest lane: median %d, p90 %d, max %d packets" % (coef[1], coef[0], np.median(mp), np.percentile(mp, 90), mp.max()))
for q in sorted(set(mp.tolist())):
    sel = loop[mp == q]
    print("    %2d packets: %6d waves, loop median %6.0f" % (q, sel.size, np.median(sel)))
# (wave start/end offsets across the launch are not reported: s_memtime counters of different CUs are not comparable)
