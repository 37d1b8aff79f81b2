"""The closed control loop (second half) -- builder-defined, so its oracle is the event-driven
ControlLoopModel of the same rules (oracle/des_model.py); the HIP kernel must match it bit for bit: integers, the f64
clock, the plant state, the controller's angle, the motor velocity, queue lengths, deliveries, received powers."""
import numpy as np
import pytest

from oracle import des_model as dm


def test_control_loop_model_closes_the_loop():
    """CPU tier: in the model, sensor packets reach the controller, commands reach the actuator and change the plant's
    input; an open loop (nobody assigned) leaves it untouched."""
    m = dm.ControlLoopModel()
    for k in range(30):
        m.step(k % 2, 12)
    s = m.snapshot()
    assert s["received"][1] > 0 and s["received"][2] > 0 and s["commands"] >= s["received"][2]
    assert s["u"] != 0.1 and s["angle_deg"] != 0.0
    assert s["substeps"] == m.tick                       # one plant substep per sensor tick
    idle = dm.ControlLoopModel()
    for k in range(10):
        idle.step(0, 0)                                   # zero-length windows: nothing is ever transmitted but announcements
    assert idle.snapshot()["received"] == [0, 0, 0] and idle.snapshot()["u"] == 0.1


def _bits(a):
    return np.asarray(a, np.float64).view(np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,start,period", [(1, 20, 10), (2, 0, 3), (3, 50, 1)])
def test_control_loop_kernel_matches_event_driven_model(seed, start, period):
    import torch
    from gymwipe_amd import VecControlLoopEnv
    N, K = 6, 70
    rng = np.random.default_rng(seed)
    env = VecControlLoopEnv(N, ctrl_start_tick=start, ctrl_period_ticks=period)
    models = [dm.ControlLoopModel(start=start, period=period) for _ in range(N)]
    for k in range(K):
        dev = rng.integers(0, 2, N).astype(np.int32)
        dur = rng.integers(0, 20, N).astype(np.int32)
        obs, rew, done, info = env.step({"device": torch.from_numpy(dev), "duration": torch.from_numpy(dur)})
        obs, rew, ang = obs.cpu().numpy(), rew.cpu().numpy(), info["Sensor angle"].cpu().numpy()
        st = {f: env.get_state(f) for f in ("now", "x", "u", "angle_deg", "qlen", "received", "n_tx", "commands", "substeps",
                                            "rx_power", "flags")}
        for e, m in enumerate(models):
            o, r, d, i = m.step(int(dev[e]), int(dur[e]))
            s = m.snapshot()
            where = (seed, k, e)
            assert obs[e] == o and rew[e] == np.float32(r) and _bits(ang[e]) == _bits(i["Sensor angle"]), where
            assert _bits(st["now"][e]) == _bits(s["now"]), where
            assert (_bits(st["x"][e]) == _bits(s["x"])).all(), where
            assert _bits(st["u"][e]) == _bits(s["u"]) and _bits(st["angle_deg"][e]) == _bits(s["angle_deg"]), where
            assert st["qlen"][e].tolist() == s["qlen"][:2] and s["qlen"][2] == 0, where
            assert st["received"][e].tolist() == s["received"][1:], where
            assert int(st["n_tx"][e]) == s["n_tx"] and int(st["commands"][e]) == s["commands"], where
            assert int(st["substeps"][e]) == s["substeps"], where
            assert (_bits(st["rx_power"][e]) == _bits(s["rx_power"])).all(), where
            assert int(st["flags"][e]) & 3 == 0, where
    assert env.get_state("received").sum() > 0


@pytest.mark.gpu
def test_control_loop_env_surface():
    import torch
    import gymwipe_amd
    env = gymwipe_amd.make("VecControlLoop-v0", num_envs=1024)
    assert env.action_space.contains({"device": 1, "duration": 19}) and env.observation_space.n == 180
    z = torch.zeros(1024, dtype=torch.int32, device="cuda")
    assert (env.reset() == 2).all()                                    # int(degrees(0.05 rad)), nothing is reset
    obs, rew, done, info = env.step({"device": z, "duration": z + 7})
    assert env.reset() is obs
    assert obs.shape == (1024,) and rew.dtype == torch.float32 and not done.any()
    env.step({"device": z + 2, "duration": z})                         # the actuator is not assignable: flagged, env untouched
    assert (env.get_state("flags") & 8).all()
    env.close()
