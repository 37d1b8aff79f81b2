#!/usr/bin/env python3
"""Long-run soak, outside the pytest tiers (minutes of host time for the oracle): N envs x K steps, outputs
compared every step, full state every 512 steps, resets every 64 steps plus random per-env resets.
   python tests/soak_gpu.py [D] [N] [K] [step|rollout|explicit]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from util import assert_state_equal, STATE_FIELDS, STAT_FIELDS
from gymwipe_amd import VecCounterTrafficEnv
from oracle.ct_oracle import CtOracle, default_config

D = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
MODE = sys.argv[4] if len(sys.argv) > 4 else "step"
env = VecCounterTrafficEnv(N, D, explicit_queue=(MODE == "explicit"), per_env_stats=True)
orc = CtOracle(N, D, config=default_config(D), nthreads=min(16, os.cpu_count() or 1))
rng = np.random.default_rng(2024)
fields = tuple(f for f in STATE_FIELDS + STAT_FIELDS if f != "queue")     # queue contents only at the very end (size)
t0 = time.time()
assert (env.reset().cpu().numpy() == orc.reset()).all()
def resets(k):
    if k and k % 64 == 0:
        assert (env.reset().cpu().numpy() == orc.reset()).all()
    elif k % 37 == 5 and MODE != "rollout":
        mask = (rng.random(N) < 0.05).astype(np.uint8)
        assert (env.reset(torch.from_numpy(mask)).cpu().numpy() == orc.reset(mask)).all()


def progress(k):
    assert_state_equal(env, orc, fields, where="after step %d" % k)
    print("step %d ok, %.0f s, simulated time %.1f s, flags %d" % (k + 1, time.time() - t0, float(orc.get("now").max()),
                                                                  int(orc.get("flags").max())), flush=True)


if MODE == "rollout":                                   # 64 pre-staged steps per fused launch
    for k0 in range(0, K, 64):
        resets(k0)
        dev = rng.integers(0, D, (64, N), dtype=np.int32)
        dur = rng.integers(0, 20, (64, N), dtype=np.int32)
        fo, fr, fd = env.rollout(torch.from_numpy(dev).cuda(), torch.from_numpy(dur).cuda())
        fo, fr, fd = fo.cpu().numpy(), fr.cpu().numpy(), fd.cpu().numpy()
        for j in range(64):
            oo, orr, od = orc.step(dev[j], dur[j])
            assert (fo[j] == oo).all() and (fr[j] == orr).all() and (fd[j] == od).all(), k0 + j
        if (k0 + 64) % 512 == 0:
            progress(k0 + 63)
else:
    for k in range(K):
        resets(k)
        dev = rng.integers(0, D, N, dtype=np.int32)
        dur = rng.integers(0, 20, N, dtype=np.int32)
        o, r, d, _ = env.step({"device": torch.from_numpy(dev), "duration": torch.from_numpy(dur)})
        oo, orr, od = orc.step(dev, dur)
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all() and (d.cpu().numpy() == od).all(), k
        if k % 512 == 511:
            progress(k)
assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="at the end")
print("soak ok (%s): %d env-steps bit-exact (D=%d), flags OR = %d" % (MODE, N * K, D, int(np.bitwise_or.reduce(orc.get("flags")))))
