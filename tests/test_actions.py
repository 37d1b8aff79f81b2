"""The counter-based synthetic action stream (gymwipe_amd/actions.py): the GPU leg of bench.py (torch, int64 wrapping
arithmetic) and the CPU baseline (numpy uint64) must draw the SAME actions, and a rank's shard must be a slice of the
global stream."""
import numpy as np


def test_numpy_and_torch_streams_are_identical_and_shardable():
    from gymwipe_amd.actions import actions_numpy, actions_torch
    for D in (2, 4, 16, 7):
        a_dev, a_dur = actions_numpy(1234, 0, 3000, 0, 40, D)
        t_dev, t_dur = actions_torch(1234, 0, 3000, 0, 40, D)
        assert (a_dev == t_dev.numpy()).all() and (a_dur == t_dur.numpy()).all()
        assert a_dev.dtype == np.int32 and a_dev.min() == 0 and a_dev.max() == D - 1
        assert a_dur.min() == 0 and a_dur.max() == 19
        # a shard of envs / a later window of steps is a slice of the global stream (a pure function of seed, env, step)
        s_dev, s_dur = actions_numpy(1234, 1000, 2000, 10, 25, D)
        assert (s_dev == a_dev[10:25, 1000:2000]).all() and (s_dur == a_dur[10:25, 1000:2000]).all()
        # roughly uniform
        assert abs(a_dev.mean() - (D - 1) / 2) < 0.05 * D and abs(a_dur.mean() - 9.5) < 0.2
    b_dev, _ = actions_numpy(99, 0, 3000, 0, 40, 4)
    assert (b_dev != actions_numpy(1234, 0, 3000, 0, 40, 4)[0]).any()


def test_scalar_definition():
    """The documented formula, evaluated with Python integers."""
    from gymwipe_amd.actions import actions_numpy
    M = (1 << 64) - 1

    def ref(seed, e, k, D):
        z = (seed ^ (e * 0x9E3779B97F4A7C15) ^ (k * 0xD1B54A32D192ED03)) & M
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return (z & 0xffffffff) % D, (z >> 32) % 20
    dev, dur = actions_numpy(1234, 65530, 65540, 60, 70, 4)
    for e in range(65530, 65540):
        for k in range(60, 70):
            assert (dev[k - 60, e - 65530], dur[k - 60, e - 65530]) == ref(1234, e, k, 4)
