"""Shared helpers for the parity tests."""
import numpy as np


def action_stream(seed, steps, num_envs, num_devices, max_duration=20):
    """Seeded uniform actions: device in [0,D), duration in [0,20) -- the env itself is
    deterministic, so actions are the only random input."""
    rng = np.random.default_rng(seed)
    dev = rng.integers(0, num_devices, size=(steps, num_envs), dtype=np.int32)
    dur = rng.integers(0, max_duration, size=(steps, num_envs), dtype=np.int32)
    return dev, dur


STATE_FIELDS = ("now", "wake", "counter", "qlen", "queue", "received", "latest_diff",
                "last_abs", "rx_power", "flags")
STAT_FIELDS = ("n_tx", "n_delivered", "n_appended", "n_popped", "n_dropped")


def assert_state_equal(gpu_env, oracle, fields=STATE_FIELDS, where=""):
    for f in fields:
        a = gpu_env.get_state(f)
        b = oracle.get(f)
        if a.dtype.kind == "f":
            same = a.view(np.uint64) == b.view(np.uint64)      # bit-exact, not approx
        else:
            same = a == b
        if not same.all():
            bad = np.argwhere(~same)[:5]
            raise AssertionError("state field %r differs %s at %s: gpu=%r oracle=%r"
                                 % (f, where, bad.tolist(), a[tuple(bad[0])], b[tuple(bad[0])]))
