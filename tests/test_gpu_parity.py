"""
GPU parity tests proper: the HIP path, called through the C-ABI, against the C oracle
(oracle/ct_oracle.c) on the same seeded action streams.  Bit-exact for every integer
(observations, rewards, counters, queue contents, event counts) AND for every f64
(simulated time, tick time, rx power): the kernels perform the reference's individual
IEEE operations, so no tolerance is needed (north_star allows 1e-5 relative).
"""
import numpy as np
import pytest

from util import action_stream, assert_state_equal, STATE_FIELDS, STAT_FIELDS

pytestmark = pytest.mark.gpu


def _mk(num_envs, D, **kw):
    from gymwipe_amd import VecCounterTrafficEnv
    from oracle.ct_oracle import CtOracle
    env = VecCounterTrafficEnv(num_envs, num_devices=D, per_env_stats=True, **kw)
    orc = CtOracle(num_envs, D, nthreads=8)
    return env, orc


def _run(env, orc, dev, dur, reset_every=None, check_every=16, reset_first=True):
    import torch
    K = dev.shape[0]
    if reset_first:
        assert (env.reset().cpu().numpy() == orc.reset()).all()
    for k in range(K):
        if reset_every and k and k % reset_every == 0:
            assert (env.reset().cpu().numpy() == orc.reset()).all()
        o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o.cpu().numpy() == oo).all(), "obs differ at step %d" % k
        assert (r.cpu().numpy() == orr).all(), "reward differs at step %d" % k
        assert (d.cpu().numpy() == od).all(), "done differs at step %d" % k
        if (k + 1) % check_every == 0 or k == K - 1:
            assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after step %d" % k)
    assert int(orc.get("flags").max()) & 3 == 0          # horizon closed everywhere
    env.check()


@pytest.mark.parametrize("D,N,K", [(2, 4096, 96), (4, 4096, 96), (16, 1024, 64)])
def test_parity_with_resets(D, N, K):
    env, orc = _mk(N, D)
    dev, dur = action_stream(100 + D, K, N, D)
    _run(env, orc, dev, dur, reset_every=32)


def test_parity_fresh_env_no_reset():
    """The reference's own test never calls reset(): counters start at 1."""
    env, orc = _mk(2048, 2)
    dev, dur = action_stream(7, 48, 2048, 2)
    _run(env, orc, dev, dur, reset_first=False)


def test_reference_known_answer_through_c_abi():
    """tests/envs/test_counter_traffic.py:25-34 of the reference, via the drop-in env."""
    import gymwipe_amd
    env = gymwipe_amd.make('CounterTraffic-v0')
    center = env.COUNTER_BOUND
    obs, reward, _, info = env.step({"device": 0, "duration": 3})
    assert obs - center == 2 and reward == -2
    obs, reward, _, info = env.step({"device": 1, "duration": 12})
    assert obs - center == 0 and reward == 2
    assert info == {"Latest received values": "[2, 2]"}
    assert env.get_state("now")[0] == 0.017804000036000002
