"""
PHY grid (the reference's benchmark scenario tests/test_benchmark.py:20-91): the HIP
event-queue kernel against the event-driven oracle (oracle/des_model.py scenario_grid) on the same initial
delays.  Integer outcomes (packets sent, transmissions, header/payload decisions per radio) bit-exact;
floating-point state (received power, simulated time) to 1e-9 relative -- the kernel evaluates the BER
with the device libm.  PARITY vs the live reference: unpinned (the reference asserts nothing here; it
is a timing benchmark) -- the oracle's PHY is pinned by tests/networking/test_stack.py:65-132.
"""
import numpy as np
import pytest


def test_grid_oracle_scenario_is_deterministic_and_collides():
    from oracle import des_model as dm
    rng = np.random.default_rng(3)
    d = rng.uniform(0, 1e-2, 9).tolist()
    a, b = dm.scenario_grid(9, d, 0.12), dm.scenario_grid(9, d, 0.12)
    assert a == b
    assert sum(a["n_sent"]) >= 9 * 10 and a["n_tx"] <= sum(a["n_sent"])
    assert sum(a["hdr_fail"]) > 0                      # uncoordinated senders do collide
    assert dm.grid_positions(4) == [(0.0, 0.0), (0.5, 1.0), (1.0, 0.0), (1.5, 1.0)]   # (i / cols, i % cols)


@pytest.mark.gpu
@pytest.mark.parametrize("n,N,T", [(2, 6, 0.25), (4, 6, 0.2), (9, 5, 0.15), (16, 4, 0.12), (20, 2, 0.08), (33, 2, 0.05), (64, 2, 0.04)])
def test_grid_kernel_matches_event_driven_oracle(n, N, T):
    import gymwipe_amd
    from oracle import des_model as dm
    rng = np.random.default_rng(100 + n)
    delays = rng.uniform(0, 1e-2, (N, n))
    grid = gymwipe_amd.VecPhyGrid(N, n, delays)
    grid.runSimulation(T * 0.4)                         # two runs: the second stop event gets a later id
    grid.runSimulation(T * 0.6)
    got = {f: grid.get_state(f) for f in ("now", "n_tx", "n_sent", "hdr_ok", "hdr_fail", "pay_ok", "pay_fail", "rx_power", "flags")}
    for e in range(N):
        w = dm.World()
        devs = [dm.GridDevice(w, i, *dm.grid_positions(n)[i], dm.GRID_SEND_INTERVAL, float(delays[e, i])) for i in range(n)]
        w.sim.run(T * 0.4)
        w.sim.run(T * 0.6)
        assert got["now"][e] == w.sim.now
        assert got["n_sent"][e].tolist() == [d.n_sent for d in devs], "env %d" % e
        assert int(got["n_tx"][e]) == len(w.band.log), "env %d" % e
        for key, what, ok in (("hdr_ok", "hdr", True), ("hdr_fail", "hdr", False), ("pay_ok", "pay", True), ("pay_fail", "pay", False)):
            want = [sum(1 for k in d.phy.decisions if k[0] == what and k[1] is ok) for d in devs]
            assert got[key][e].tolist() == want, "%s env %d: %s vs %s" % (key, e, got[key][e].tolist(), want)
        rx = np.array([d.phy.rx_power for d in devs])
        assert np.allclose(got["rx_power"][e], rx, rtol=1e-9, atol=0.0)
        assert int(got["flags"][e]) == 0


def test_grid_rng_is_shared_with_the_oracle():
    """The mobile walk is a counter-based hash both sides evaluate: check a few values of the oracle's
    version (the kernel's copy is compared through positions in the GPU test below)."""
    from oracle import des_model as dm
    assert dm.splitmix64(0) == 0xE220A8397B1DCDAF
    u = [dm.grid_uniform(7, 3, d, k, w) for d in (0, 5) for k in (0, 9) for w in (0, 1, 2)]
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) == len(u)


@pytest.mark.gpu
@pytest.mark.parametrize("n,N,T", [(2, 4, 0.2), (4, 4, 0.15), (9, 3, 0.1), (16, 2, 0.07), (40, 1, 0.03)])
def test_mobile_grid_kernel_matches_event_driven_oracle(n, N, T):
    """mobile_device_grid (tests/test_benchmark.py:73-85): positions change every 1 ms, so attenuation,
    received power and BER move WHILE packets are being received (simple_stack.py:119-128,223-231)."""
    import gymwipe_amd
    from oracle import des_model as dm
    rng = np.random.default_rng(500 + n)
    delays = rng.uniform(0, 1e-2, (N, n))
    seed = 1234
    grid = gymwipe_amd.VecPhyGrid(N, n, delays, mobile=True, seed=seed)
    grid.runSimulation(T * 0.5)
    grid.runSimulation(T * 0.5)
    got = {f: grid.get_state(f) for f in ("now", "n_tx", "n_sent", "hdr_ok", "hdr_fail", "pay_ok", "pay_fail", "rx_power", "pos", "flags")}
    for e in range(N):
        want = dm.scenario_mobile_grid(n, delays[e].tolist(), T, seed=seed, replica=e, runs=[T * 0.5, T * 0.5])
        assert got["now"][e] == want["now"]
        assert np.array_equal(got["pos"][e], np.array(want["pos"])), "the random walk itself must be bit-identical"
        assert got["n_sent"][e].tolist() == want["n_sent"] and int(got["n_tx"][e]) == want["n_tx"], "env %d" % e
        for key in ("hdr_ok", "hdr_fail", "pay_ok", "pay_fail"):
            assert got[key][e].tolist() == want[key], "%s env %d: %s vs %s" % (key, e, got[key][e].tolist(), want[key])
        # received power: the device evaluates FSPL with its own log10/pow, so link powers (~0.2 mW at 1 m,
        # 40 dBm) differ from the host's in the last bits, and so does the residue left in an idle radio after
        # +p/-p pairs: compare to a few ulps of the LARGEST power, plus 1e-7 relative
        assert np.allclose(got["rx_power"][e], np.array(want["rx_power"]), rtol=1e-7, atol=1e-13)
        assert int(got["flags"][e]) == 0


@pytest.mark.gpu
def test_reference_simple_phy_known_answer_on_the_gpu():
    """tests/networking/test_stack.py:43-132 (test_simple_phy) replayed through the C-ABI: one 0 dBm packet
    (8 B header + 128 B payload) between (0,0) and (1,1); the receiver is moved to x = 2 while the header is on
    air.  The reference asserts: one active transmission during, received power lower after the move, band
    empty afterwards, packet handed to the MAC.  Powers are also compared with the event-driven model."""
    from gymwipe_amd.grid import VecPhyGrid
    from oracle import des_model as dm
    want = dm.scenario_simple_phy()
    dr = dm.BpskMcs().data_rate
    N = 3
    delays = np.tile(np.array([0.0, 1e9]), (N, 1))           # device 2 never sends
    g = VecPhyGrid(N, 2, delays, positions=[(0.0, 0.0), (1.0, 1.0)], mobile=True, tx_power_dbm=0.0,
                   header_bytes=8, payload_bytes=128, send_interval=0.5, move_interval=1e12)
    g.runSimulation(0.25)
    assert (g.get_state("on_air") == want["idle_before"]).all()
    g.runSimulation(0.25 + 8 / dr)                                     # the SEND command is issued at t = 0.5
    assert (g.get_state("on_air") == want["active_during"]).all()
    before = g.get_state("rx_power")[:, 1].copy()
    np.testing.assert_allclose(before, want["power_before"], rtol=1e-12)
    g.runSimulation(64 / dr)
    g.setPosition(1, 2.0, 1.0)
    g.runSimulation(16 / dr)
    after = g.get_state("rx_power")[:, 1]
    assert (after < before).all()
    np.testing.assert_allclose(after, want["power_after"], rtol=1e-12)
    np.testing.assert_allclose(g.get_state("pos")[:, 1], [[2.0, 1.0]] * N)
    g.runSimulation(0.2)
    assert (g.get_state("on_air") == want["active_after"]).all()
    assert (g.get_state("hdr_ok")[:, 1] == 1).all() and (g.get_state("pay_ok")[:, 1] == 1).all()
    assert (g.get_state("hdr_fail") == 0).all() and (g.get_state("pay_fail") == 0).all()
    assert [d[1] for d in want["decisions"]] == [True, True] and want["delivered_last_is_packet"]
    assert (g.get_state("flags") == 0).all()
