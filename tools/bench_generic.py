#!/usr/bin/env python3
"""The generic step kernel (explicit MAC queues, GW_CFG_EXPLICIT_QUEUE) with and without receive-mode MACs:
env-steps/s at the BASELINE shape, for the record next to the default kernel's figure.  One JSON line per mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd

N, D, K, W = int(os.environ.get("N", 65536)), 4, 256, 64
g = torch.Generator(device="cuda"); g.manual_seed(11)
dev = torch.randint(0, D, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
dur = torch.randint(0, 20, (W + K, N), dtype=torch.int32, device="cuda", generator=g)
MODES = (("explicit queues", {}), ("explicit queues + receive-mode MACs", {"peer_receive": True}))
if os.environ.get("GW_BENCH_MODES") == "plain":      # profiling runs: one mode only
    MODES = MODES[:1]
for name, kw in MODES:
    env = gymwipe_amd.VecCounterTrafficEnv(N, D, explicit_queue=True, **kw)
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
    env.check()
    out = {"workload": "generic step kernel, %s, %d devices x %d envs, reset every 64 steps" % (name, D, N),
           "env_steps_per_s": N * K / wall, "ms_per_step": wall / K * 1e3}
    if kw:
        out["peer_packets_delivered"] = int(env.get_state("peer_received").sum())
    print(json.dumps(out))
    env.close()
