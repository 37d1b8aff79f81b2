"""Receive-mode MACs and scripted traffic on the GPU (first half) -- `-m gpu`.

The reference pins this behaviour in tests/networking/test_stack.py:134-235 (test_simple_mac): two SimpleMac
devices and an RRM; ten packets queued on each; ten alternating 10000-slot assignments; receivers count what
their MACs hand up: 4, 4, 8, 8 after rounds 1-4 and 10, 10 at the end.  Here that scenario runs through the
C-ABI (generic kernel + GW_CFG_NO_COUNTER_TRAFFIC | PEER_RECEIVE | FLOAT_DURATION + gw_enqueue) and is
compared (a) with the reference's asserted numbers and (b) step by step with the event-driven model.
"""
import json
import os

import numpy as np
import pytest

from oracle import des_model as dm

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def _snapshot_equal(env, models, where):
    now, qlen = env.get_state("now"), env.get_state("qlen")
    peer, rxp = env.get_state("peer_received"), env.get_state("rx_power")
    rec = env.get_state("received")
    for e, m in enumerate(models):
        s = m.snapshot()
        assert _bits(now[e]) == _bits(s["now"]), (where, e, now[e], s["now"])
        assert qlen[e].tolist() == s["qlen"], (where, e)
        assert peer[e].tolist() == s["peer_received"], (where, e, peer[e].tolist(), s["peer_received"])
        assert (_bits(rxp[e]) == _bits(s["rx_power"])).all(), (where, e)
        assert rec[e].tolist() == s["received"], (where, e)


def test_reference_simple_mac_known_answer_on_the_gpu():
    import torch
    from gymwipe_amd import VecCounterTrafficEnv
    with open(GOLDEN) as fh:
        r = json.load(fh)["simple_mac"]["delivered_after_rounds"]
    want = [r["round1_device2"], r["round2_device1"], r["round3_device2"], r["round4_device1"],
            r["final_device1"], r["final_device2"]]
    assert dm.scenario_simple_mac() == want == [4, 4, 8, 8, 10, 10]
    N = 5
    kw = dict(positions=[(0, 0), (1, 1)], rrm_position=(2, 2))
    env = VecCounterTrafficEnv(N, 2, explicit_queue=True, counter_traffic=False, peer_receive=True,
                               float_duration=True, **kw)
    models = [dm.CounterTrafficModel(2, positions=[(0, 0), (1, 1)], rrm_pos=(2, 2), traffic=False,
                                     peer_receive=True, float_duration=True)]
    for i in range(10):                                    # Transmittable(i): 1 byte for 0..9, 2 for 10..19
        env.enqueue(0, 1)
        env.enqueue(1, 2)
        models[0].enqueue(0, 1)
        models[0].enqueue(1, 2)
    assert (env.get_state("qlen") == 10).all()
    got = []
    for k in range(10):
        dev = torch.full((N,), k % 2, dtype=torch.int32, device="cuda")
        dur = torch.full((N,), 10, dtype=torch.int32, device="cuda")           # 10 * 1000 slots = 0.01 s
        obs, rew, done, _ = env.step({"device": dev, "duration": dur})
        o, r, d, _ = models[0].step(k % 2, 10)
        assert (obs.cpu().numpy() == o).all() and (rew.cpu().numpy() == r).all()
        peer = env.get_state("peer_received")
        assert (peer == peer[0]).all()
        got.append(peer[0].tolist())
        _snapshot_equal(env, models * 1, "round %d" % (k + 1))
    # rounds 1..4: receivedPackets2, 1, 2, 1; then both after round 10 (test_stack.py:219-235)
    assert [got[0][1], got[1][0], got[2][1], got[3][0], got[9][0], got[9][1]] == want
    assert (env.get_state("flags") == 0).all()


CLOSE = {2: [(1.0, 0.0), (0.0, 1.0)], 3: [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0)]}


@pytest.mark.parametrize("D,seed,close", [(2, 1, False), (2, 4, True), (3, 2, True), (4, 3, False)])
def test_receive_mode_with_counter_traffic_and_enqueues_matches_model(D, seed, close):
    """Counter traffic ON, receive-mode MACs ON, extra packets enqueued between steps, random actions:
    every env compared with its own event-driven model after every step."""
    import torch
    from gymwipe_amd import VecCounterTrafficEnv
    N, K = 4, 24
    rng = np.random.default_rng(seed)
    pos = CLOSE[D] if close else None          # on the default layouts the senders are too far apart to decode each other
    env = VecCounterTrafficEnv(N, D, explicit_queue=True, peer_receive=True, positions=pos)
    models = [dm.CounterTrafficModel(D, peer_receive=True, positions=pos) for _ in range(N)]
    for k in range(K):
        if k % 3 == 1:
            i = int(rng.integers(0, D))
            pb = rng.integers(-1, 40, size=N).astype(np.int32)
            env.enqueue(i, torch.from_numpy(pb).cuda())
            for e in range(N):
                if pb[e] >= 0:
                    models[e].enqueue(i, int(pb[e]))
        dev = rng.integers(0, D, size=N).astype(np.int32)
        dur = rng.integers(0, 20, size=N).astype(np.int32)
        obs, rew, done, _ = env.step({"device": torch.from_numpy(dev).cuda(), "duration": torch.from_numpy(dur).cuda()})
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for e in range(N):
            o, r, d, _ = models[e].step(int(dev[e]), int(dur[e]))
            assert obs[e] == o and rew[e] == r, (k, e)
        _snapshot_equal(env, models, "step %d" % k)
    if close:
        assert env.get_state("peer_received").sum() > 0


def test_new_modes_need_the_generic_kernel():
    from gymwipe_amd import VecCounterTrafficEnv
    with pytest.raises(Exception):
        VecCounterTrafficEnv(4, 2, peer_receive=True)               # default (suffix) mode: unsupported
    env = VecCounterTrafficEnv(4, 2)
    with pytest.raises(Exception):
        env.enqueue(0, 3)
