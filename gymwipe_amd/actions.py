"""
Counter-based synthetic action streams (SURVEY.md 8d "synthetic inputs").

The env is deterministic (the reference never uses its seeded ``np_random``, envs/core.py:46-52), so
"synthetic traffic" means action streams only.  The action of env ``e`` at step ``k`` is a pure function
of ``(seed, e, k)``:

    h        = splitmix64(seed ^ (e * 0x9E3779B97F4A7C15) ^ (k * 0xD1B54A32D192ED03))
    device   = (h & 0xffffffff) % num_devices
    duration = (h >> 32)        % max_duration

so the GPU (torch int64 arithmetic, wrapping) and the CPU baseline (numpy uint64) draw IDENTICAL streams
without storing or exchanging them, and a rank generates the actions of its own shard from global env ids.
"""
import numpy as np

_M64 = (1 << 64) - 1
_G1, _G2 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03
_C1, _C2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _i64(x):
    """Python int (mod 2^64) -> the int64 with the same bit pattern."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def actions_numpy(seed, env_lo, env_hi, step_lo, step_hi, num_devices, max_duration=20):
    """(device, duration) int32[steps][envs] for envs [env_lo, env_hi) and steps [step_lo, step_hi)."""
    e = np.arange(env_lo, env_hi, dtype=np.uint64)[None, :]
    k = np.arange(step_lo, step_hi, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) ^ (e * np.uint64(_G1)) ^ (k * np.uint64(_G2))
        z = z + np.uint64(_G1)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        z = z ^ (z >> np.uint64(31))
    dev = ((z & np.uint64(0xffffffff)) % np.uint64(num_devices)).astype(np.int32)
    dur = ((z >> np.uint64(32)) % np.uint64(max_duration)).astype(np.int32)
    return dev, dur


def actions_torch(seed, env_lo, env_hi, step_lo, step_hi, num_devices, max_duration=20, device="cpu"):
    """Same stream as :func:`actions_numpy`, evaluated with torch on ``device`` (int64, wrapping)."""
    import torch

    def lsr(x, s):                                    # logical shift right on int64
        return (x >> s) & ((1 << (64 - s)) - 1)

    e = torch.arange(env_lo, env_hi, dtype=torch.int64, device=device)[None, :]
    k = torch.arange(step_lo, step_hi, dtype=torch.int64, device=device)[:, None]
    z = (e * _i64(_G1)) ^ (k * _i64(_G2)) ^ _i64(seed)
    z = z + _i64(_G1)
    z = (z ^ lsr(z, 30)) * _i64(_C1)
    z = (z ^ lsr(z, 27)) * _i64(_C2)
    z = z ^ lsr(z, 31)
    dev = ((z & 0xffffffff) % num_devices).to(torch.int32)
    dur = (lsr(z, 32) % max_duration).to(torch.int32)
    return dev.contiguous(), dur.contiguous()
