"""
Linear plant (BASELINE config 4): the f64 matrix-core kernel against its scalar
oracle.  Builder-defined model, PARITY UNPINNED against the reference (its plant is an ODE world in
an env that cannot be constructed); tolerance 1e-5 relative as the north star states for
floating-point plant state -- the observed difference is ~1e-13 (A^n precomputed vs n applications).
"""
import ctypes as C

import numpy as np
import pytest


def test_plant_oracle_matches_numpy_recurrence():
    from gymwipe_amd import _native as nat
    from oracle.plant_oracle import PlantOracle
    cfg = nat.PlantConfig()
    nat.check(nat.lib().gw_plant_config_default(C.byref(cfg), 8))
    A = np.array(list(cfg.A)).reshape(4, 4)
    B = np.array(list(cfg.B))
    assert cfg.dt == 1e-3 and cfg.u0 == 0.1                       # sliding_pendulum.py:52, inverted_pendulum.py:81
    orc = PlantOracle(8, A, B, cfg.dt, list(cfg.x0), cfg.u0)
    now = np.array([0.0, 0.001, 0.0029, 0.01, 0.0204, 0.05, 0.1, 1.0])
    orc.update(now)
    assert orc.substeps.tolist() == [0, 1, 3, 10, 20, 50, 100, 1000]
    for e in range(8):
        x = np.array(list(cfg.x0))
        for _ in range(int(orc.substeps[e])):
            x = A @ x + B * cfg.u0
        assert np.allclose(orc.x[e], x, rtol=1e-12, atol=1e-15)
    assert (np.abs(orc.x) < 10).all()                              # the default model is stable


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 37, 4096])
def test_plant_kernel_matches_oracle(N):
    import torch
    import gymwipe_amd
    from oracle.plant_oracle import PlantOracle
    plant = gymwipe_amd.VecLinearPlant(N)
    cfg = plant.config
    orc = PlantOracle(N, list(cfg.A), list(cfg.B), cfg.dt, list(cfg.x0), cfg.u0)
    rng = np.random.default_rng(5)
    now = np.zeros(N)
    worst = 0.0
    for it in range(30):
        # steps of 0 .. 45 ms (more than GW_PLANT_KMAX = 32 substeps in one call) and some envs standing still
        now = now + rng.integers(0, 46, N) * 1e-3 * (rng.random(N) > 0.1) + rng.random(N) * 2e-4
        plant.updateState(torch.from_numpy(now))
        orc.update(now)
        if it % 7 == 3:
            u = rng.normal(0, 0.3, N)
            mask = rng.random(N) > 0.5
            plant.setMotorVelocity(torch.from_numpy(u), torch.from_numpy(mask))
            orc.set_input(u, mask)
        x = plant.state()
        assert (plant.get_state("substeps") == orc.substeps).all()
        assert (plant.get_state("t_last") == orc.t_last).all()
        scale = np.maximum(np.abs(orc.x), 1e-6)
        err = np.abs(x - orc.x) / scale
        worst = max(worst, float(err.max()))
        assert err.max() < 1e-5, "relative error %g at iteration %d" % (err.max(), it)
    assert worst < 1e-9                                            # in practice ~1e-13
    assert plant.getAngle().shape == (N,) and plant.getWagonPos().shape == (N,)


@pytest.mark.gpu
def test_plant_follows_the_env_clock():
    """The pendulum-env coupling: step the band-assignment env, then advance every plant to its
    env's simulated time (read in place through gw_now_ptr)."""
    import torch
    import gymwipe_amd
    from gymwipe_amd import _native as nat
    N = 512
    env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=2)
    plant = gymwipe_amd.VecLinearPlant(N)
    base, stride = C.c_void_p(), C.c_int64()
    nat.check(env._L.gw_now_ptr(env._h, C.byref(base), C.byref(stride)))
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    env.reset()
    for _ in range(10):
        env.step({"device": torch.randint(0, 2, (N,), dtype=torch.int32, device="cuda", generator=g),
                  "duration": torch.randint(0, 20, (N,), dtype=torch.int32, device="cuda", generator=g)})
        plant.updateState((base.value, stride.value))
    now = env.get_state("now")
    assert (plant.get_state("t_last") == now).all()
    sub = plant.get_state("substeps").astype(np.int64)
    assert (np.abs(sub - now / 1e-3) <= 10).all()                  # one rounding per update, 10 updates


@pytest.mark.gpu
def test_inverted_pendulum_env_surface_and_interpreter_formulas():
    """envs/inverted_pendulum.py: gym surface, and observation / reward exactly as InvertedPendulumInterpreter
    computes them from the plant (:27-57) -- int(degrees(angle)), float(abs(180 - degrees(angle))) -- while the plant
    follows the env clock.  The network is the env as shipped: the sensor's queue fills and drains, the controller
    never has anything to send."""
    import math
    import torch
    from gymwipe_amd import VecInvertedPendulumEnv, InvertedPendulumEnv, spaces
    N = 512
    env = VecInvertedPendulumEnv(N)
    assert env.action_space.contains({"device": 1, "duration": 19}) and isinstance(env.observation_space, spaces.Discrete)
    assert env.observation_space.n == 180
    obs0 = env.reset()
    x0 = env.plant.state()
    assert (obs0.cpu().numpy() == np.array([int(math.degrees(a)) for a in x0[:, 2]])).all()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for k in range(40):
        act = {"device": torch.randint(0, 2, (N,), dtype=torch.int32, device="cuda", generator=g),
               "duration": torch.randint(0, 20, (N,), dtype=torch.int32, device="cuda", generator=g)}
        obs, rew, done, info = env.step(act)
    now = env.network.get_state("now")
    x = env.plant.state()
    np.testing.assert_allclose(env.plant.get_state("t_last"), np.floor(now / env.plant.config.dt + 1e-9) * env.plant.config.dt,
                               rtol=0, atol=env.plant.config.dt)            # the plant has been advanced to the env clock
    deg = np.array([math.degrees(a) for a in x[:, 2]])
    assert (obs.cpu().numpy() == deg.astype(np.int64)).all()                 # int(): truncation towards zero
    np.testing.assert_allclose(rew.cpu().numpy(), np.abs(180.0 - deg).astype(np.float32), rtol=1e-7)
    np.testing.assert_allclose(info["Sensor angle"].cpu().numpy(), deg, rtol=1e-15)
    assert not done.any()
    q = env.network.get_state("qlen")
    assert (q[:, 1] == 0).all() and q[:, 0].max() > 0                        # silent controller, busy sensor
    assert int(env.network.get_state("flags").max()) & 3 == 0
    one = InvertedPendulumEnv()                                              # the scalar surface of the reference
    o, r, d, i = one.step({"device": 0, "duration": 5})
    assert isinstance(o, int) and isinstance(r, float) and d is False and isinstance(i["Sensor angle"], float)
    with pytest.raises(AssertionError):
        one.step({"device": 2, "duration": 5})


@pytest.mark.gpu
def test_pendulum_env_one_launch_at_config4_size():
    """BASELINE config 4 at its stated size, 32 768 envs: env.step() as ONE launch (gw_pendulum_step) against
    (a) the same step issued as two launches -- bit-identical plant state, clocks and feedback;
    (b) the network's oracle (oracle/ct_oracle.c: clocks, queues, bit-exact) and the plant's oracle (oracle/plant_oracle.c,
        advanced to the oracle's clocks: <= 1e-5 relative, the north star's bound; observed ~1e-13);
    (c) the interpreter's formulas evaluated with Python's math on the plant state."""
    import math
    import torch
    from gymwipe_amd import VecInvertedPendulumEnv
    from gymwipe_amd.actions import actions_torch
    from oracle.ct_oracle import CtOracle, default_config
    from oracle.plant_oracle import PlantOracle
    N, K = 32768, 48
    fused, split = VecInvertedPendulumEnv(N), VecInvertedPendulumEnv(N)
    cfg = default_config(2, positions=[(0.0, 0.0), (0.0, -1.0)], rrm_pos=(0.0, 1.0), mult=[1, 0], dest=[1, 0])
    net = CtOracle(N, 2, config=cfg, nthreads=8)
    pc = fused.plant.config
    porc = PlantOracle(N, list(pc.A), list(pc.B), pc.dt, list(pc.x0), pc.u0)
    a_dev, a_dur = actions_torch(77, 0, N, 0, K, 2, device="cuda")
    h_dev, h_dur = a_dev.cpu().numpy(), a_dur.cpu().numpy()
    for k in range(K):
        act = {"device": a_dev[k], "duration": a_dur[k]}
        o1, r1, d1, i1 = fused.step(act)
        o2, r2, d2, i2 = split.step(act, fused=False)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(i1["Sensor angle"], i2["Sensor angle"]), k
        net.step(h_dev[k], h_dur[k])
        porc.update(net.get("now"))
        if k % 16 == 15 or k == K - 1:
            for f in ("now", "wake", "counter", "qlen", "rx_power"):
                a, b = fused.network.get_state(f), net.get(f)
                assert (a.view(np.uint8) == b.view(np.uint8)).all(), (f, k)
            x = fused.plant.state()
            assert (x.view(np.uint8) == split.plant.state().view(np.uint8)).all()
            assert (fused.plant.get_state("substeps") == porc.substeps).all()
            assert (fused.plant.get_state("t_last") == porc.t_last).all()
            err = np.abs(x - porc.x) / np.maximum(np.abs(porc.x), 1e-6)
            assert err.max() < 1e-5, err.max()
            deg = np.array([math.degrees(a) for a in x[:, 2]])
            assert (o1.cpu().numpy() == deg.astype(np.int64)).all()
            np.testing.assert_allclose(r1.cpu().numpy(), np.abs(180.0 - deg).astype(np.float32), rtol=1e-7)
    assert int(fused.network.get_state("flags").max()) & 3 == 0
    st = fused.network.stats()
    assert st["steps"] == N * K and st["transmissions"] > N * K          # announcements + the sensor's packets


@pytest.mark.gpu
@pytest.mark.parametrize("N", [33000, 100])
def test_pendulum_one_launch_both_wave_mappings(N):
    """gw_pendulum_step maps 32 envs to a wave up to 32 768 envs (so that every SIMD gets one) and 64 beyond: both must equal
    the two-launch form bit for bit (N = 100: ragged last wave of the 32-env mapping; 33 000: ragged last wave of the 64-env one)."""
    import torch
    from gymwipe_amd import VecInvertedPendulumEnv
    from gymwipe_amd.actions import actions_torch
    fused, split = VecInvertedPendulumEnv(N), VecInvertedPendulumEnv(N)
    a_dev, a_dur = actions_torch(5, 0, N, 0, 24, 2, device="cuda")
    for k in range(24):
        act = {"device": a_dev[k], "duration": a_dur[k]}
        o1, r1, d1, i1 = fused.step(act)
        o2, r2, d2, i2 = split.step(act, fused=False)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(i1["Sensor angle"], i2["Sensor angle"]), k
    assert (fused.plant.state().view(np.uint8) == split.plant.state().view(np.uint8)).all()
    assert (fused.plant.get_state("substeps") == split.plant.get_state("substeps")).all()
    for f in ("now", "wake", "qlen", "rx_power", "n_tx", "n_popped"):
        assert (fused.network.get_state(f).view(np.uint8) == split.network.get_state(f).view(np.uint8)).all(), f
