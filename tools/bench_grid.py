#!/usr/bin/env python3
"""The reference's benchmark_simulation_grid (tests/test_benchmark.py:87-88: wall time of 1 simulated second of an
n-device grid), as N replicas on one MI355X.  Prints one JSON line per n.  The CPU figure beside it in
profiles/r1_grid_phy (the event-driven Python model, one core) was taken once with the test infrastructure; it is
not run from here."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gymwipe_amd

N = int(os.environ.get("N", 4096))
SIM = float(os.environ.get("SIM", 1.0))
MOBILE = bool(int(os.environ.get("MOBILE", "0")))
for n in (4, 16, 20):
    rng = np.random.default_rng(n)
    delays = rng.uniform(0, 1e-2, (N, n))
    grid = gymwipe_amd.VecPhyGrid(N, n, delays, mobile=MOBILE, seed=99)
    grid.runSimulation(0.05)                       # warm-up (also past the start-up transient)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grid.runSimulation(SIM)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev = grid.get_state("events").astype(np.int64)
    out = {"workload": "PHY grid, %d devices, %d replicas, %.2f s simulated each (%s)" % (n, N, SIM, "mobile" if MOBILE else "static"),
           "replica_seconds_per_s": N * SIM / wall, "wall_s": wall, "events_total": int(ev.sum()),
           "events_per_s": float(ev.sum()) / (wall * (SIM + 0.05) / SIM), "n_tx_mean": float(grid.get_state("n_tx").mean()),
           "hdr_ok": int(grid.get_state("hdr_ok").sum()), "hdr_fail": int(grid.get_state("hdr_fail").sum()),
           "flags_or": int(np.bitwise_or.reduce(grid.get_state("flags")))}
    print(json.dumps(out))
    grid.close()
