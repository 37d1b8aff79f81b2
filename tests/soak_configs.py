#!/usr/bin/env python3
"""Many random configurations, each for a short run (outside the pytest tiers): device count, geometry, RRM position,
multiplicities (incl. silent senders), custom attenuation, counter bound, start time; step kernel in both queue modes
and the fused rollout against the oracle.   python tests/soak_configs.py [count] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from util import assert_state_equal, STATE_FIELDS, STAT_FIELDS
from gymwipe_amd import VecCounterTrafficEnv
from oracle.ct_oracle import CtOracle, default_config

COUNT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
N, K = 256, 48
done, skipped, t0 = 0, 0, time.time()
for it in range(COUNT):
    D = int(rng.choice([2, 2, 3, 4, 4, 5, 6, 7, 8, 11, 16, 32]))
    ang, rad = rng.uniform(0, 2 * np.pi, D), rng.uniform(0.4, 6.5, D)
    pos = [(float(r * np.cos(a)), float(r * np.sin(a))) for r, a in zip(rad, ang)]
    rrm = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)))
    mult = [int(m) for m in rng.choice([0, 1, 1, 2, 3, 5, 9, 15, 16, 37, 64, 100], D)]
    extra = {}
    for _ in range(int(rng.integers(0, 4))):
        a, b = sorted(int(x) for x in rng.choice(D + 1, 2, replace=False))
        extra[(a, b)] = float(rng.choice([0.5, 1.0, 3.0, 4.65, 6.0, 12.0]))
    bound = int(rng.choice([65536, 65536, 300, 40]))
    t_start = float(rng.choice([0.0, 0.0, 0.0, 17.25, 4096.0, 9.9e5, 1.2e6]))
    explicit = bool(rng.integers(0, 2))
    if explicit and rng.integers(0, 3) == 0:            # the generic kernel takes any multiplicity up to the deque's capacity
        mult = [int(m) for m in rng.choice([0, 1, 7, 16, 37, 64, 100], D)]
    kw = dict(positions=pos, rrm_position=rrm, multiplicity=mult, extra_attenuation=extra or None,
              counter_bound=bound, start_time=t_start)
    env = VecCounterTrafficEnv(N, D, explicit_queue=explicit, per_env_stats=True, **kw)   # any layout, either queue mode
    cfg = default_config(D, positions=pos, mult=mult, rrm_pos=rrm, extra_att=extra or None, start_time=t_start)
    cfg.counter_bound = bound
    orc = CtOracle(N, D, config=cfg, nthreads=8)
    dev = rng.integers(0, D, (K, N), dtype=np.int32)
    dur = rng.integers(0, 20, (K, N), dtype=np.int32)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    use_rollout = (not explicit) and bool(rng.integers(0, 2))
    if use_rollout:
        fo, fr, fd = env.rollout(torch.from_numpy(dev).cuda(), torch.from_numpy(dur).cuda())
        fo, fr, fd = fo.cpu().numpy(), fr.cpu().numpy(), fd.cpu().numpy()
    for k in range(K):
        if not use_rollout:
            if k in (13, 14, 30):
                mask = (rng.random(N) < 0.5).astype(np.uint8)
                assert (env.reset(torch.from_numpy(mask)).cpu().numpy() == orc.reset(mask)).all()
            o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
            o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        else:
            o, r, d = fo[k], fr[k], fd[k]
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o == oo).all() and (r == orr).all() and (d == od).all(), (it, k, D, mult, explicit, use_rollout)
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="config %d (D=%d mult=%s explicit=%s rollout=%s t0=%g)"
                       % (it, D, mult, explicit, use_rollout, t_start))
    env.close()
    done += 1
print("config soak ok: %d configurations compared, %d refused (noise-state closure), %.0f s" % (done, skipped, time.time() - t0))
