"""
GPU parity tests proper: the HIP path, called through the C-ABI, against the C oracle
(oracle/ct_oracle.c) on the same seeded action streams.  Bit-exact for every integer
(observations, rewards, counters, queue contents, event counts) AND for every f64
(simulated time, tick time, rx power): the kernels perform the reference's individual
IEEE operations, so no tolerance is needed (north_star allows 1e-5 relative).
"""
import os
import numpy as np
import pytest

from util import action_stream, assert_state_equal, STATE_FIELDS, STAT_FIELDS

pytestmark = pytest.mark.gpu


QUEUE_MODES = [pytest.param(False, id="suffix"), pytest.param(True, id="explicit")]


def _mk(num_envs, D, explicit=False, counter_bound=None, **kw):
    from gymwipe_amd import VecCounterTrafficEnv
    from oracle.ct_oracle import CtOracle, default_config
    env = VecCounterTrafficEnv(num_envs, num_devices=D, per_env_stats=True, explicit_queue=explicit,
                               counter_bound=counter_bound, **kw)
    ea = kw.get("extra_attenuation")
    if callable(ea):
        pos = [tuple(env.config.pos[i]) for i in range(D + 1)]
        ea = {(a, b): ea(a, b, pos[a], pos[b]) for a in range(D + 1) for b in range(a + 1, D + 1)}
    cfg = default_config(D, positions=kw.get("positions"), mult=kw.get("multiplicity"),
                         rrm_pos=kw.get("rrm_position"), extra_att=ea, start_time=kw.get("start_time"))
    if counter_bound is not None:
        cfg.counter_bound = counter_bound
    orc = CtOracle(num_envs, D, config=cfg, nthreads=8)
    return env, orc


def _run(env, orc, dev, dur, reset_every=None, check_every=16, reset_first=True, horizon_must_close=True):
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
    if horizon_must_close:
        assert int(orc.get("flags").max()) & 3 == 0      # horizon closed everywhere
        env.check()


@pytest.mark.parametrize("explicit", QUEUE_MODES)
@pytest.mark.parametrize("D,N,K", [(2, 4096, 96), (4, 4096, 96), (16, 1024, 64)])
def test_parity_with_resets(D, N, K, explicit):
    env, orc = _mk(N, D, explicit)
    dev, dur = action_stream(100 + D, K, N, D)
    _run(env, orc, dev, dur, reset_every=32)


@pytest.mark.parametrize("D", [3, 5, 7, 8, 11, 32])
def test_parity_other_device_counts(D):
    """D = 3 / 5 / 7 (one-word records), 8 / 32 (multi-word records, max senders), 11 (no instantiation of its own: the
    any-D kernel with the record in memory)."""
    N, K = 512, 48
    env, orc = _mk(N, D)
    dev, dur = action_stream(200 + D, K, N, D)
    _run(env, orc, dev, dur, reset_every=16, check_every=8)


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_fresh_env_no_reset(explicit):
    """The reference's own test never calls reset(): counters start at 1."""
    env, orc = _mk(2048, 2, explicit)
    dev, dur = action_stream(7, 48, 2048, 2)
    _run(env, orc, dev, dur, reset_first=False)


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_frequent_resets_many_runs(explicit):
    """reset() every 3 steps: many short runs per queue (middle-run ring, head-run hand-over)."""
    env, orc = _mk(1024, 4, explicit)
    dev, dur = action_stream(11, 120, 1024, 4)
    _run(env, orc, dev, dur, reset_every=3, check_every=8)


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_counter_saturation(explicit):
    """COUNTER_BOUND lowered to 40 so counters saturate within the rollout (counter_traffic.py:59-60)."""
    env, orc = _mk(1024, 4, explicit, counter_bound=40)
    dev, dur = action_stream(12, 64, 1024, 4)
    _run(env, orc, dev, dur, reset_every=24, check_every=8)


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_long_run_queue_overflow(explicit):
    """No reset for 160 steps: queues fill to 100 and drop the oldest (simple_stack.py:361)."""
    env, orc = _mk(512, 2, explicit)
    dev, dur = action_stream(13, 160, 512, 2)
    _run(env, orc, dev, dur, check_every=32)
    assert int(orc.get("n_dropped").min()) > 0


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


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_event_totals_match_oracle(explicit):
    """gw_stats_read (the roofline accounting uses appended/popped) == sums of the oracle's counters."""
    import torch
    env, orc = _mk(2048, 4, explicit)
    dev, dur = action_stream(21, 40, 2048, 4)
    env.reset(); orc.reset()
    for k in range(40):
        env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        orc.step(dev[k], dur[k])
    st = env.stats()
    assert st["steps"] == 40 * 2048 and st["bad_actions"] == 0
    assert st["transmissions"] == int(orc.get("n_tx").sum())
    assert st["delivered"] == int(orc.get("n_delivered").sum())
    assert st["appended"] == int(orc.get("n_appended").sum())
    assert st["popped"] == int(orc.get("n_popped").sum())
    assert st["dropped"] == int(orc.get("n_dropped").sum())


@pytest.mark.parametrize("D,mult,K", [(4, [100, 37, 1, 64], 40), (3, [16, 99, 7], 40), (2, [100, 100], 24), (16, [1, 3, 100, 15, 16, 17, 31, 32, 33, 50, 64, 99, 2, 5, 7, 0], 24)])
def test_suffix_encoding_with_multiplicities_up_to_the_deques_capacity(D, mult, K):
    """The default (suffix) queue encoding takes any multiplicity up to 100 (rounds 1-2: 15): step kernel, then the fused
    rollout from the same state, against the oracle -- queue contents entry by entry."""
    import torch
    N = 1024
    env, orc = _mk(N, D, multiplicity=mult)
    dev, dur = action_stream(500 + D, K + 16, N, D)
    _run(env, orc, dev[:K], dur[:K], reset_every=9, check_every=8)
    obs, rew, done = env.rollout(torch.from_numpy(dev[K:]), torch.from_numpy(dur[K:]))
    for k in range(16):
        oo, orr, od = orc.step(dev[K + k], dur[K + k])
        assert (obs[k].cpu().numpy() == oo).all() and (rew[k].cpu().numpy() == orr).all(), k
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")
    assert int(orc.get("n_dropped").min()) > 0


@pytest.mark.parametrize("D,mult", [(32, None), (4, [100, 37, 1, 64]), (3, [16, 99, 7])])
def test_generic_kernel_heavy_appends_totals_and_queues(D, mult):
    """Explicit-ring kernel with the most appends a step can make: every duration 19 (20 ticks per step) at D = 32,
    and multiplicities up to the deque's capacity (every tick replaces the whole queue).  gw_stats_read's wave-level
    totals -- 64 lanes x 20 ticks x mult does not fit 16 bits -- and the queue contents against the oracle."""
    import torch
    N, K = 2048, 24
    env, orc = _mk(N, D, explicit=True, multiplicity=mult)
    rng = np.random.default_rng(5)
    env.reset(); orc.reset()
    for k in range(K):
        dev = rng.integers(0, D, N, dtype=np.int32)
        dur = np.full(N, 19, np.int32) if k % 3 else rng.integers(0, 20, N, dtype=np.int32)
        o, r, d, _ = env.step({"device": torch.from_numpy(dev), "duration": torch.from_numpy(dur)})
        oo, orr, od = orc.step(dev, dur)
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all(), k
        if k in (3, 11):
            env.reset(); orc.reset()
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="heavy appends")
    st = env.stats()
    assert st["steps"] == K * N
    for name, field in (("transmissions", "n_tx"), ("delivered", "n_delivered"), ("appended", "n_appended"),
                        ("popped", "n_popped"), ("dropped", "n_dropped")):
        assert st[name] == int(orc.get(field).sum()), name
    assert st["appended"] > 64 * 65535 // 64 * 4                      # far past what a 16-bit field per wave could hold


def test_invalid_action_is_flagged_and_env_left_untouched():
    import torch
    from gymwipe_amd import _native as nat
    env, orc = _mk(64, 2)
    dev, dur = action_stream(22, 6, 64, 2)
    env.reset(); orc.reset()
    for k in range(3):
        env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        orc.step(dev[k], dur[k])
    bad_dev, bad_dur = dev[3].copy(), dur[3].copy()
    bad_dev[5], bad_dur[9] = 2, 20                          # outside Discrete(2) / Discrete(20)
    before = {f: env.get_state(f).copy() for f in ("now", "counter", "qlen")}
    o, r, d, _ = env.step({"device": torch.from_numpy(bad_dev), "duration": torch.from_numpy(bad_dur)})
    for f, v in before.items():
        after = env.get_state(f)
        assert (after[5] == v[5]).all() and (after[9] == v[9]).all(), f
    fl = env.get_state("flags")
    assert fl[5] & nat.FLAG_BADACT and fl[9] & nat.FLAG_BADACT and not (fl[0] & nat.FLAG_BADACT)
    assert env.stats()["bad_actions"] == 2
    with pytest.raises(AssertionError):
        env.check()
    # flags are sticky until cleared; clearing tells later checks WHEN something happened, counters stay
    env.clear_flags()
    assert not env.get_state("flags").any()
    st = env.check(strict=True)
    assert st["bad_actions"] == 2 and st["ties"] is False
    env.step({"device": torch.from_numpy(dev[4]), "duration": torch.from_numpy(dur[4])})
    env.check(strict=True)
    # the N = 1 drop-in raises like the reference (counter_traffic.py:147)
    import gymwipe_amd
    one = gymwipe_amd.make("CounterTraffic-v0")
    with pytest.raises(AssertionError):
        one.step({"device": 2, "duration": 1})
    with pytest.raises(AssertionError):
        one.step({"device": 0, "duration": 20})


@pytest.mark.parametrize("D,N,K", [(2, 2048, 64), (4, 2048, 37), (4, 1024, 150), (16, 512, 48), (5, 256, 20), (4, 1001, 30), (8, 70, 64), (6, 512, 40),
                                   (7, 300, 70), (32, 200, 33), (11, 129, 64)])
def test_fused_rollout_matches_oracle(D, N, K, monkeypatch):
    """gw_rollout: one persistent launch per <= 64 steps (free-running lanes, state in registers)
    must give exactly what K env.step() calls give -- outputs of every step and the final state.
    Every sender count has a fused kernel (D = 11: the any-D instantiation with per-lane arrays in LDS);
    GW_ROLLOUT_STRICT makes gw_rollout fail rather than fall back to step launches, so this test cannot pass on a fallback."""
    import torch
    monkeypatch.setenv("GW_ROLLOUT_STRICT", "1")
    env, orc = _mk(N, D)
    dev, dur = action_stream(300 + D + K, K, N, D)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    obs, rew, done = env.rollout(torch.from_numpy(dev), torch.from_numpy(dur))
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for k in range(K):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (obs[k] == oo).all(), "obs differ at step %d" % k
        assert (rew[k] == orr).all(), "reward differs at step %d" % k
        assert (done[k] == od).all()
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")
    env.check()
    # and a per-step call afterwards continues from the same state
    dev2, dur2 = action_stream(9, 3, N, D)
    for k in range(3):
        o, r, d_, _ = env.step({"device": torch.from_numpy(dev2[k]), "duration": torch.from_numpy(dur2[k])})
        oo, orr, od = orc.step(dev2[k], dur2[k])
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all()
    assert_state_equal(env, orc, STATE_FIELDS, where="after rollout + steps")


@pytest.mark.parametrize("D,N,K", [(4, 1024, 70), (16, 256, 48), (11, 129, 64)])
def test_fused_rollout_event_loop_form_still_matches(D, N, K, monkeypatch):
    """GW_ROLLOUT_EVENT_LOOP=1 (read at gw_create and at gw_rollout): the rounds-1/2 form of the fused rollout -- an event loop
    over packed action / feedback records with a transposing launch on either side -- kept as the A/B reference of the
    step-synchronous kernel.  Same outputs, same final state."""
    import torch
    monkeypatch.setenv("GW_ROLLOUT_STRICT", "1")
    monkeypatch.setenv("GW_ROLLOUT_EVENT_LOOP", "1")
    env, orc = _mk(N, D)
    dev, dur = action_stream(1300 + D + K, K, N, D)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    obs, rew, done = env.rollout(torch.from_numpy(dev), torch.from_numpy(dur))
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for k in range(K):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (obs[k] == oo).all() and (rew[k] == orr).all() and (done[k] == od).all(), k
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the event-loop rollout")
    env.check()


def test_fused_rollout_with_invalid_actions():
    import torch
    from gymwipe_amd import _native as nat
    env, orc = _mk(256, 4)
    dev, dur = action_stream(41, 20, 256, 4)
    env.reset(); orc.reset()
    bad = dev.copy(); bad_du = dur.copy()
    bad[7, 3] = 4; bad_du[11, 200] = 25; bad[0, 17] = -1
    obs, rew, done = env.rollout(torch.from_numpy(bad), torch.from_numpy(bad_du))
    assert env.stats()["bad_actions"] == 3
    fl = env.get_state("flags")
    assert all(fl[i] & nat.FLAG_BADACT for i in (3, 200, 17)) and not fl[0] & nat.FLAG_BADACT
    # untouched envs behave exactly like the oracle
    ok = np.ones(256, bool); ok[[3, 200, 17]] = False
    for k in range(20):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (obs[k].cpu().numpy()[ok] == oo[ok]).all() and (rew[k].cpu().numpy()[ok] == orr[ok]).all()


# Links at the edge of decodability.  With the reference's constants a payload decodes iff BER <= 0.1171875 and a
# header iff BER <= 0.234375 (round(err)/bits <= 0.25 with the payload errors counted twice); on free space at 0 dBm
# that is 3.41676 m and 5.54584 m from the listener.  Around those distances the outcome depends on the packet size
# through banker's rounding, so the kernels' exact arithmetic (decode class COMPUTE) decides, not the class table.
MARGINAL = {
    "payload-edge": [(3.4167, 0.0), (0.0, 3.4168), (-5.5458, 0.0), (0.0, -2.0)],
    "header-edge": [(5.5458, 0.0), (0.0, 5.5459), (-3.0, 0.0), (0.0, -3.4)],
    "mixed": [(1.0, 0.5), (-3.41676, 0.0), (2.0, 2.77), (0.3, -3.3)],
}


@pytest.mark.parametrize("explicit", QUEUE_MODES)
@pytest.mark.parametrize("name", sorted(MARGINAL))
def test_parity_on_marginal_links(name, explicit):
    N, K, D = 1024, 64, 4
    env, orc = _mk(N, D, explicit=explicit, positions=MARGINAL[name])
    dev, dur = action_stream(31, K, N, D)
    _run(env, orc, dev, dur, reset_every=20)
    st = env.stats()
    data_tx = st["transmissions"] - st["steps"]
    if name == "payload-edge":                    # some, but not all, data packets decode
        assert 0 < st["delivered"] < data_tx


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_parity_on_random_geometries(seed):
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(2, 7))
    ang = rng.uniform(0, 2 * np.pi, D)
    rad = rng.uniform(0.5, 6.0, D)
    pos = [(float(r * np.cos(a)), float(r * np.sin(a))) for r, a in zip(rad, ang)]
    mult = [int(m) for m in rng.integers(1, 5, D)]
    rrm = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
    N, K = 512, 48
    env, orc = _mk(N, D, positions=pos, multiplicity=mult, rrm_position=rrm)   # any layout: no finite noise-state set -> live-PHY kernel
    dev, dur = action_stream(seed, K, N, D)
    _run(env, orc, dev, dur, reset_every=17)


def test_step_is_capturable_in_a_hip_graph():
    """gw_step only enqueues one kernel on the caller's stream (no allocation, no synchronisation), so a
    caller can capture a run of steps in a hipGraph and replay it; results stay identical."""
    import torch
    N, D, K, G = 2048, 4, 48, 8
    env, orc = _mk(N, D)
    dev, dur = action_stream(9, K, N, D)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    a_dev = torch.zeros((G, N), dtype=torch.int32, device="cuda")
    a_dur = torch.zeros((G, N), dtype=torch.int32, device="cuda")
    outs = [(torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda"),
             torch.empty(N, dtype=torch.uint8, device="cuda")) for _ in range(G)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for j in range(G):
            env._obs, env._rew, env._done = outs[j]
            env.step({"device": a_dev[j], "duration": a_dur[j]})
    assert (env.get_state("now") == 0).all()              # capture executed nothing
    for k0 in range(0, K, G):
        a_dev.copy_(torch.from_numpy(dev[k0:k0 + G]))
        a_dur.copy_(torch.from_numpy(dur[k0:k0 + G]))
        graph.replay()
        torch.cuda.synchronize()
        for j in range(G):
            oo, orr, od = orc.step(dev[k0 + j], dur[k0 + j])
            assert (outs[j][0].cpu().numpy() == oo).all() and (outs[j][1].cpu().numpy() == orr).all()
            assert (outs[j][2].cpu().numpy() == od).all()
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after graph replays")


def test_feedback_byte_codec_kernels():
    """gw_pack_feedback / gw_unpack_feedback against the host codec the gloo tests use, on real step outputs,
    an odd element count, and a buffer that is NOT the built-in interpreter's feedback (must be refused)."""
    import torch
    from test_distributed_cpu import byte_pack, byte_unpack
    N, D, K = 1000, 4, 7                                    # 7000 elements: not a multiple of 4 per row
    env, orc = _mk(N, D)
    dev, dur = action_stream(12, K, N, D)
    obs = torch.empty((K, N), dtype=torch.int32, device="cuda")
    rew = torch.empty((K, N), dtype=torch.float32, device="cuda")
    done = torch.empty((K, N), dtype=torch.uint8, device="cuda")
    for k in range(K):
        env._obs, env._rew, env._done = obs[k], rew[k], done[k]
        env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
    packed = env.pack_feedback(obs, rew, done, check=True)
    want = byte_pack(obs.cpu(), rew.cpu(), done.cpu(), torch.empty((K, N), dtype=torch.uint8))
    assert (packed.cpu() == want).all()
    o2, r2, d2 = env.unpack_feedback(packed)
    assert (o2 == obs).all() and (r2 == rew).all() and (d2 == done).all()
    ho, hr, hd = byte_unpack(packed.cpu())
    assert (ho == obs.cpu()).all() and (hr == rew.cpu()).all() and (hd == done.cpu()).all()
    flat = env.pack_feedback(obs.view(-1)[:4001].contiguous(), rew.view(-1)[:4001].contiguous(), done.view(-1)[:4001].contiguous())
    assert (flat.cpu() == want.view(-1)[:4001]).all()
    obs[0, 0] = 70000                                        # not -v / 0 / +v around the bound
    with pytest.raises(Exception):
        env.pack_feedback(obs, rew, done, check=True)


@pytest.mark.parametrize("D,kw", [(4, {}), (16, {}), (5, {}), (11, {}), (4, {"explicit": True}), (4, {"per_env_geometry": True})])
def test_step_writes_its_feedback_byte_row(D, kw):
    """gw_step_fb (env.feedback_bytes_into): the step kernel stores the one-byte exchange form of its own feedback -- equal to
    what the packing kernel makes of (obs, reward, done), invalid actions included; the generic and live-PHY modes get there
    through the packing kernel.  With the row switched off the step is plain gw_step again."""
    import torch
    N, K = 1000, 40
    kw = dict(kw)
    explicit = kw.pop("explicit", False)
    env, _ = _mk(N, D, explicit=explicit, **kw)            # (parity of the outputs themselves is the other tests' business)
    dev, dur = action_stream(33, K, N, D)
    dev[3, 5] = D + 2                                          # invalid actions keep their env's previous feedback, reward 0
    dur[9, 7] = 99
    rows = torch.zeros((K, N), dtype=torch.uint8, device="cuda")
    env.reset()
    seen = set()
    for k in range(K):
        if k % 16 == 0 and k:
            env.reset()
        env.feedback_bytes_into(rows[k] if k != 20 else None)
        o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        if k != 20:
            assert torch.equal(rows[k], env.pack_feedback(o, r, d, check=True)), k
            seen.update(rows[k].unique().tolist())
    assert len(seen) >= 4, seen                                # the run produced several different feedback values
    assert env.stats()["bad_actions"] == 2
    assert not rows[20].any()                                  # switched off: nothing written
    # the same through preallocated outputs (env.step(action, out=StepOutputs)): typed outputs and byte row land in the slot
    from gymwipe_amd import StepOutputs
    slot = StepOutputs(torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda"),
                       torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"))
    plain = StepOutputs(torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda"),
                        torch.zeros(N, dtype=torch.uint8, device="cuda"))
    env.feedback_bytes_into(None)
    act = {"device": torch.from_numpy(dev[1]), "duration": torch.from_numpy(dur[1])}
    o, r, d, _ = env.step(act, out=slot)
    assert o is slot.obs and r is slot.reward and d is slot.done and o.any()
    assert torch.equal(slot.feedback_bytes, env.pack_feedback(o, r, d, check=True))
    o2, r2, d2, _ = env.step(act, out=plain)
    assert o2 is plain.obs and torch.equal(env.received()[:, 0] != 0, env.received()[:, 0] != 0)
    o3, r3, d3, _ = env.step(act)                              # and the env's own buffers again afterwards
    assert o3 is not plain.obs and o3 is not slot.obs
    with pytest.raises(AssertionError):
        env.feedback_bytes_into(torch.zeros(N + 1, dtype=torch.uint8, device="cuda"))


# ---- BASELINE.json's full sizes ------------------------------------------------------------------------
@pytest.mark.parametrize("D,K", [(4, 192), (16, 64)])
def test_parity_at_full_baseline_size(D, K):
    """configs[1] / configs[2]: 65 536 envs on one GPU, compared with the oracle directly (it is fast enough:
    a few seconds on the host cores), reset every 64 steps as in the benchmark."""
    N = 65536
    env, orc = _mk(N, D)
    dev, dur = action_stream(77, K, N, D)
    _run(env, orc, dev, dur, reset_every=64, check_every=64)


def test_full_size_properties_rollout_equals_steps_and_shards_equal_whole():
    """Size-independent properties at 65 536 envs: (1) K fused-rollout steps leave exactly the state K single
    steps leave; (2) two handles stepping the two halves of the batch equal one handle stepping all of it
    (what sharding across GPUs relies on); (3) event totals are the sum of the per-env counters."""
    import torch
    from gymwipe_amd import VecCounterTrafficEnv
    N, D, K = 65536, 4, 96
    dev, dur = action_stream(5, K, N, D)
    t_dev, t_dur = torch.from_numpy(dev).cuda(), torch.from_numpy(dur).cuda()
    whole = VecCounterTrafficEnv(N, D)
    fused = VecCounterTrafficEnv(N, D)
    halves = [VecCounterTrafficEnv(N // 2, D), VecCounterTrafficEnv(N // 2, D)]
    for e in [whole, fused] + halves:
        e.reset()
    outs = []
    for k in range(K):
        o, r, d, _ = whole.step({"device": t_dev[k], "duration": t_dur[k]})
        outs.append((o.clone(), r.clone(), d.clone()))
        for h, sl in zip(halves, (slice(0, N // 2), slice(N // 2, N))):
            ho, hr, hd, _ = h.step({"device": t_dev[k, sl].contiguous(), "duration": t_dur[k, sl].contiguous()})
            assert (ho == o[sl]).all() and (hr == r[sl]).all() and (hd == d[sl]).all(), k
    fo, fr, fd = fused.rollout(t_dev, t_dur)
    for k in range(K):
        assert (fo[k] == outs[k][0]).all() and (fr[k] == outs[k][1]).all() and (fd[k] == outs[k][2]).all(), k
    for f in ("now", "wake", "counter", "qlen", "received", "rx_power", "last_abs"):
        a = whole.get_state(f)
        assert (a.view(np.uint8) == fused.get_state(f).view(np.uint8)).all(), f
        b = np.concatenate([h.get_state(f) for h in halves])
        assert (a.view(np.uint8) == b.view(np.uint8)).all(), f
    st = whole.stats()
    assert st["steps"] == N * K
    for name, field in (("transmissions", "n_tx"), ("delivered", "n_delivered"), ("appended", "n_appended"),
                        ("popped", "n_popped"), ("dropped", "n_dropped")):
        assert st[name] == int(whole.get_state(field).sum()), name
    assert st["flags_or"] & 3 == 0


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_with_custom_attenuation_models(explicit):
    """setCustomModels / JoinedAttenuationModel (physical.py:402-498) with static geometry: extra dB per device pair,
    given as a dict and as a callable; some links pushed over the decode thresholds by the extra term."""
    N, K, D = 1024, 64, 4
    env, orc = _mk(N, D, explicit=explicit, extra_attenuation={(0, 4): 4.5, (1, 4): 4.7, (2, 3): 2.0, (3, 4): 0.125})
    dev, dur = action_stream(41, K, N, D)
    _run(env, orc, dev, dur, reset_every=20)
    st = env.stats()
    assert 0 < st["delivered"] < st["transmissions"] - st["steps"]       # sender 1 sits at the payload threshold now

    def wall(a, b, pa, pb):                                            # anything crossing x = 0 loses 6 dB
        return 6.0 if (pa[0] < 0) != (pb[0] < 0) else 0.0
    env2, orc2 = _mk(N, D, explicit=explicit, extra_attenuation=wall)
    _run(env2, orc2, dev, dur, reset_every=20)
    att_plain, _ = _mk(8, D)[0].link_info(1, 4)
    att_wall, _ = env2.link_info(1, 4)
    p1 = tuple(env2.config.pos[1])
    assert att_wall == att_plain + (6.0 if p1[0] < 0 else 0.0)


@pytest.mark.parametrize("explicit", QUEUE_MODES)
@pytest.mark.parametrize("D,p", [(2, 0.3), (4, 0.1), (3, 0.02)])
def test_parity_with_random_per_env_resets(D, p, explicit):
    """reset(mask): every env gets its own reset history -- several resets inside one queue span, resets on
    consecutive steps (the breakpoint is overwritten, not appended), empty masks, full masks."""
    import torch
    N, K = 2048, 160
    env, orc = _mk(N, D, explicit=explicit)
    dev, dur = action_stream(50 + D, K, N, D)
    rng = np.random.default_rng(9)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    for k in range(K):
        if k % 2 == 0 or k % 7 == 3:
            mask = (rng.random(N) < p).astype(np.uint8)
            if k == 40:
                mask[:] = 0
            if k == 80:
                mask[:] = 1
            assert (env.reset(torch.from_numpy(mask)).cpu().numpy() == orc.reset(mask)).all(), k
        o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all() and (d.cpu().numpy() == od).all(), k
        if k % 32 == 31 or k == K - 1:
            assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after step %d" % k)
    fo, fr, fd = env.rollout(torch.from_numpy(dev[:48]).cuda(), torch.from_numpy(dur[:48]).cuda())
    for k in range(48):                                   # the fused kernel on top of those reset histories
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (fo[k].cpu().numpy() == oo).all() and (fr[k].cpu().numpy() == orr).all(), k
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")
    env.check()


def test_chunked_feedback_gather_over_rccl_single_rank():
    """The N > 1 exchange of bench.py with the real backend: RCCL (`nccl`) process group of one rank on this GPU,
    HIP pack kernel, asynchronous all_gather_into_tensor on RCCL's stream, double buffering, a partial last chunk."""
    import socket
    import torch
    import torch.distributed as dist
    from gymwipe_amd import VecCounterTrafficEnv
    from gymwipe_amd.sharding import ChunkedFeedbackGather
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, D, K, G = 4096, 4, 150, 64
        env, ref = VecCounterTrafficEnv(N, D), VecCounterTrafficEnv(N, D)
        dev, dur = action_stream(3, K, N, D)
        t_dev, t_dur = torch.from_numpy(dev).cuda(), torch.from_numpy(dur).cuda()
        cg = ChunkedFeedbackGather(N, torch.device("cuda", 0), env.pack_feedback, 1, chunk=G)
        env.reset(); ref.reset()
        want, got = [], []
        for k in range(K):
            env._obs, env._rew, env._done = cg.slot()
            env.step({"device": t_dev[k], "duration": t_dur[k]})
            o, r, d, _ = ref.step({"device": t_dev[k], "duration": t_dur[k]})
            want.append((o.clone(), r.clone(), d.clone()))
            b = cg.stepped()
            if b is not None:
                cg.pending[b].wait()
                got.append(cg.result(b).clone())
        b = (cg.k // G) % cg.depth
        cg.drain()
        got.append(cg.result(b).clone())
        torch.cuda.synchronize()
        packed = torch.cat(got, dim=1)                        # [1][K][N]
        assert packed.shape == (1, K, N)
        o, r, d = env.unpack_feedback(packed[0].contiguous())
        for k in range(K):
            assert (o[k] == want[k][0]).all() and (r[k] == want[k][1]).all() and (d[k] == want[k][2]).all(), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("switch", ["GW_NO_FASTMATH", "GW_NO_CLASSES", "GW_NO_TICKJUMP", "GW_NO_IDEM"])
def test_parity_with_exact_fast_paths_switched_off(switch, monkeypatch):
    """The fast forms (FMA remainder, 3-op division, integer decode rule, certainty classes, tick jump) are exact
    replacements validated at gw_create; with any of them switched off (read at gw_create) the plain forms run and
    the results must be the same bits.  GW_NO_IDEM makes the kernels count every hearing of a talker through the full
    transition tables in HBM instead of using the combined LDS tables.  Step kernel and fused rollout."""
    import torch
    monkeypatch.setenv(switch, "1")
    N, K, D = 2048, 72, 4
    env, orc = _mk(N, D)
    dev, dur = action_stream(61, K, N, D)
    _run(env, orc, dev, dur, reset_every=24)
    fo, fr, fd = env.rollout(torch.from_numpy(dev[:40]).cuda(), torch.from_numpy(dur[:40]).cuda())
    for k in range(40):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (fo[k].cpu().numpy() == oo).all() and (fr[k].cpu().numpy() == orr).all(), k
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")


def test_c_abi_error_paths_on_the_gpu():
    """Misuse through the real handle: NULL buffers, unknown / wrongly sized state fields, unsupported calls."""
    import ctypes as C
    import torch
    from gymwipe_amd import VecCounterTrafficEnv, _native as nat
    env = VecCounterTrafficEnv(256, 4)
    L, h = env._L, env._h
    buf = torch.zeros(256, dtype=torch.int32, device="cuda")
    assert L.gw_step(h, buf.data_ptr(), None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None) == nat.EINVAL
    assert L.gw_rollout(h, -1, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None) == nat.EINVAL
    out = np.zeros(256, np.float64)
    assert L.gw_get_state(h, b"no_such_field", out.ctypes.data, out.nbytes) == nat.EFIELD
    assert L.gw_get_state(h, b"now", out.ctypes.data, out.nbytes - 8) in (nat.EINVAL, nat.EFIELD)
    assert L.gw_get_state(h, b"peer_received", out.ctypes.data, out.nbytes) == nat.EFIELD    # needs GW_CFG_PEER_RECEIVE
    assert L.gw_enqueue(h, 0, buf.data_ptr(), None) == nat.EUNSUPPORTED                       # suffix mode
    assert L.gw_link_info(h, 0, 9, C.byref(C.c_double()), C.byref(C.c_double())) == nat.EINVAL
    assert b"" != L.gw_last_error()
    # the handle is still usable afterwards
    o, r, d, _ = env.step({"device": buf, "duration": buf})
    assert int(env.get_state("flags").max()) == 0 and o.shape == (256,)
    env.close()
    env.close()                                               # idempotent


@pytest.mark.parametrize("explicit", QUEUE_MODES)
def test_parity_with_silent_senders(explicit):
    """mult = 0: senders that never enqueue anything (their windows stay empty until the timeout); step kernel in
    both queue modes and the fused rollout."""
    import torch
    N, K, D = 1024, 64, 4
    env, orc = _mk(N, D, explicit=explicit, multiplicity=[2, 0, 1, 0])
    dev, dur = action_stream(71, K, N, D)
    _run(env, orc, dev, dur, reset_every=25)
    if not explicit:
        fo, fr, fd = env.rollout(torch.from_numpy(dev[:40]).cuda(), torch.from_numpy(dur[:40]).cuda())
        for k in range(40):
            oo, orr, od = orc.step(dev[k], dur[k])
            assert (fo[k].cpu().numpy() == oo).all() and (fr[k].cpu().numpy() == orr).all(), k
        assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")
    assert (env.get_state("qlen")[:, 1] == 0).all() and (env.get_state("qlen")[:, 3] == 0).all()


@pytest.mark.parametrize("explicit", QUEUE_MODES)
@pytest.mark.parametrize("t0", [1000.0, 123456.789, 999999.6, 1.05e6, 1.5e6, 3.0e6])   # 999999.6: crosses the fast forms' 10^6 s limit mid-run
def test_parity_at_large_simulated_times(t0, explicit):
    """Runs that START at a large simulated time (test hook `start_time`): coarser f64 binades for every time
    computation, and past the validity limits of the fast paths one after the other -- decode-certainty classes
    (t < 1e6), FMA slot remainder (t < 2^40 slots = 1.0995e6 s), tick jump (t < 2^21 s) -- so that the plain
    fmod, the exact decode sums and the tick loop run on the device.  A run from zero gets there after ~10^8 steps."""
    import torch
    N, K, D = 1024, 64, 4
    env, orc = _mk(N, D, explicit=explicit, start_time=t0)
    assert (env.get_state("now") == t0).all()
    dev, dur = action_stream(int(t0) % 97, K, N, D)
    # at ~1e6 s one ulp is 1e-10 s: a packet that fits its window by a rounding can end exactly at the step's end
    # (GW_FLAG_CARRY, the flattened horizon's validity bit) -- both sides must raise it for the same envs (the flags
    # are part of the compared state); it just is not required to stay clear here
    _run(env, orc, dev, dur, reset_every=20, horizon_must_close=t0 < 1e6)
    if not explicit:
        fo, fr, fd = env.rollout(torch.from_numpy(dev[:32]).cuda(), torch.from_numpy(dur[:32]).cuda())
        for k in range(32):
            oo, orr, od = orc.step(dev[k], dur[k])
            assert (fo[k].cpu().numpy() == oo).all() and (fr[k].cpu().numpy() == orr).all(), k
        assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")


@pytest.mark.parametrize("D,N", [(2, 1), (3, 63), (4, 65), (4, 1000), (8, 193), (16, 130)])
def test_generic_kernel_two_wave_form_ragged_sizes_bad_actions_and_totals(D, N):
    """The generic kernel at its default block runs as TWO waves per 64 envs (ct_step.hip: the walker and the helper that
    ticks the other senders' queues and writes the noise-state record).  Sizes that leave the last workgroup ragged, every
    templated sender count, invalid actions in some lanes (the helper must skip exactly the envs the walker skips), per-env
    counters that both waves add to, and the wave totals -- all against the oracle, plus the same run with GW_NO_SPLIT."""
    import torch
    from gymwipe_amd import _native as nat
    K = 40
    dev, dur = action_stream(900 + D + N, K, N, D)
    bad_at = {7: (0, D), 19: (N - 1, -1), 23: (N // 2, D + 3)}         # step -> (env, invalid device)
    results = []
    for no_split in (False, True):
        if no_split:
            os.environ["GW_NO_SPLIT"] = "1"
        try:
            env, orc = _mk(N, D, explicit=True)
            assert (env.reset().cpu().numpy() == orc.reset()).all()
            n_bad = 0
            for k in range(K):
                if k and k % 16 == 0:
                    assert (env.reset().cpu().numpy() == orc.reset()).all()
                dk, uk = dev[k].copy(), dur[k].copy()
                skip = None
                if k in bad_at:
                    skip, dk[bad_at[k][0]] = bad_at[k][0], bad_at[k][1]
                    n_bad += 1
                o, r, d, _ = env.step({"device": torch.from_numpy(dk), "duration": torch.from_numpy(uk)})
                if skip is None:
                    oo, orr, od = orc.step(dk, uk)
                    assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all() and (d.cpu().numpy() == od).all(), k
                else:
                    # the oracle takes valid actions only.  The skipped env must have kept its state (= the oracle's, which has
                    # not stepped yet); then the oracle steps everyone, and the GPU side catches up by a second call in which
                    # ONLY the skipped env has a valid action (envs are independent: one call later makes no difference)
                    before = {f: env.get_state(f)[skip].copy() for f in ("now", "counter", "qlen", "queue")}
                    assert env.get_state("flags")[skip] & nat.FLAG_BADACT
                    for f, v in before.items():
                        assert (orc.get(f)[skip] == v).all(), (f, k)
                    env.clear_flags()
                    o1, r1 = o.cpu().numpy().copy(), r.cpu().numpy().copy()     # (step() returns the handle's own output tensors)
                    dk[skip] = 0
                    orc_o = orc.step(dk, uk)
                    o2, r2, d2, _ = env.step({"device": torch.from_numpy(np.where(np.arange(N) == skip, 0, -1).astype(np.int32)),
                                              "duration": torch.from_numpy(uk)})     # only the skipped env steps now
                    n_bad += N - 1
                    env.clear_flags()
                    assert o2.cpu().numpy()[skip] == orc_o[0][skip] and r2.cpu().numpy()[skip] == orc_o[1][skip]
                    mask = np.arange(N) != skip
                    assert (o1[mask] == orc_o[0][mask]).all() and (r1[mask] == orc_o[1][mask]).all(), k
                if (k + 1) % 8 == 0:          # (flags aside: the bad-action bits were cleared on the GPU side, with whatever else was set)
                    assert_state_equal(env, orc, tuple(f for f in STATE_FIELDS if f != "flags") + STAT_FIELDS,
                                       where="D=%d N=%d step %d" % (D, N, k))
            st = env.stats()
            assert st["bad_actions"] == n_bad
            for name, f in (("transmissions", "n_tx"), ("delivered", "n_delivered"), ("appended", "n_appended"),
                            ("popped", "n_popped"), ("dropped", "n_dropped")):
                assert st[name] == int(orc.get(f).sum()), name
            results.append({f: env.get_state(f).copy() for f in ("now", "queue", "rx_power", "counter")})
        finally:
            os.environ.pop("GW_NO_SPLIT", None)
    for f in results[0]:
        assert (results[0][f].view(np.uint8) == results[1][f].view(np.uint8)).all(), f


@pytest.mark.parametrize("block", ["16", "32", "128", "256"])
def test_parity_with_other_workgroup_sizes(block, monkeypatch):
    """GW_BLOCK (read at gw_create) changes the launch shape of the generic step kernel and, below 64 threads, the way its
    tables are staged in LDS; results must not change.  (The suffix-queue kernels have a compile-time block of 64.)"""
    monkeypatch.setenv("GW_BLOCK", block)
    N, K = 1000, 40                                    # not a multiple of any block size
    for D in (4, 5):                                   # templated and generic device counts
        env, orc = _mk(N, D, explicit=True)
        dev, dur = action_stream(81, K, N, D)
        _run(env, orc, dev, dur, reset_every=16)


@pytest.mark.parametrize("cap", ["0", "16", "48"])
def test_rollout_chunking_and_fallback(cap, monkeypatch):
    """GW_ROLLOUT_CAP (read at gw_create): steps per fused launch; 0 switches the fused kernel off, so gw_rollout
    falls back to one step launch per step.  Same results either way."""
    import torch
    monkeypatch.setenv("GW_ROLLOUT_CAP", cap)
    N, K, D = 1024, 70, 4
    env, orc = _mk(N, D)
    dev, dur = action_stream(91, K, N, D)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    fo, fr, fd = env.rollout(torch.from_numpy(dev).cuda(), torch.from_numpy(dur).cuda())
    for k in range(K):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (fo[k].cpu().numpy() == oo).all() and (fr[k].cpu().numpy() == orr).all() and (fd[k].cpu().numpy() == od).all(), k
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")


def test_two_handles_driven_from_two_host_threads():
    """a handle is not thread-safe, but different handles may be driven from different host threads
    (ctypes releases the GIL during the calls; errors are thread-local).  Each thread uses its own stream."""
    import threading
    import torch
    N, D, K = 2048, 4, 120
    results, errors = {}, []

    def worker(tid):
        try:
            env, orc = _mk(N, D)
            dev, dur = action_stream(200 + tid, K, N, D)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                env.reset()
                orc.reset()
                for k in range(K):
                    o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]).cuda(), "duration": torch.from_numpy(dur[k]).cuda()})
                    oo, orr, od = orc.step(dev[k], dur[k])
                    s.synchronize()
                    assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all(), (tid, k)
                assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="thread %d" % tid)
            results[tid] = True
        except Exception as exc:                              # surfaced in the main thread below
            errors.append((tid, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert results == {0: True, 1: True}


def test_create_destroy_does_not_leak_device_memory():
    """Every handle type frees what it allocated: device memory in use returns to its level after many
    create / destroy cycles."""
    import gc
    import torch
    from gymwipe_amd import VecControlLoopEnv, VecCounterTrafficEnv, VecLinearPlant, VecPhyGrid

    def cycle():
        for explicit in (False, True):
            e = VecCounterTrafficEnv(16384, 4, explicit_queue=explicit, peer_receive=explicit)
            z = torch.zeros(16384, dtype=torch.int32, device="cuda")
            e.step({"device": z, "duration": z + 3})
            e.pack_feedback(e._obs, e._rew, e._done)
            e.close()
        p = VecLinearPlant(16384); p.close()
        g = VecPhyGrid(256, 9, np.zeros((256, 9))); g.runSimulation(0.01); g.close()
        c = VecControlLoopEnv(8192); c.close()

    cycle()
    torch.cuda.synchronize(); gc.collect()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize(); gc.collect()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)        # allocator granularity, not a per-cycle leak


@pytest.mark.parametrize("explicit", QUEUE_MODES)
@pytest.mark.parametrize("case", ["tick == window start", "tick == end of a data transmission", "guard slot"])
def test_ordering_edge_cases_on_the_gpu(case, explicit):
    """The exact f64 time ties and the guard-slot ending that tests/test_oracle_pinning.py pins between the two oracle
    layers, through the HIP kernels (step kernels of both queue modes and the fused rollout): same outputs, same state,
    same GW_FLAG_TIE."""
    import torch
    from test_oracle_pinning import tie_intervals
    from gymwipe_amd import VecCounterTrafficEnv
    from gymwipe_amd import _native as nat
    from oracle.ct_oracle import CtOracle, default_config
    N, D = 256, 2
    kw, cfg = {}, default_config(D)
    if case == "guard slot":
        kw["duration_factor"] = cfg.duration_factor = 2081
        acts = [(0, 1), (1, 1), (0, 1), (1, 0), (0, 1), (1, 1)]
    else:
        kw["counter_interval"] = cfg.counter_interval = tie_intervals()[case]
        acts = [(0, 19), (1, 7), (0, 3), (1, 19), (0, 0), (1, 12)]
    env = VecCounterTrafficEnv(N, num_devices=D, per_env_stats=True, explicit_queue=explicit, **kw)
    fused = None if explicit else VecCounterTrafficEnv(N, num_devices=D, **kw)
    orc = CtOracle(N, D, config=cfg, nthreads=4)
    dev = np.array([[a[0]] * N for a in acts], np.int32)
    dur = np.array([[a[1]] * N for a in acts], np.int32)
    dev[:, 1::2] = 1 - dev[:, 1::2]                              # every other env addresses the other sender
    outs = []
    for k in range(len(acts)):
        o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all(), k
        outs.append((oo, orr))
    assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where=case)
    if case == "tick == end of a data transmission":
        assert (env.get_state("flags") & nat.FLAG_TIE).any()
    if fused is not None:
        fo, fr, fd = fused.rollout(torch.from_numpy(dev), torch.from_numpy(dur))
        for k in range(len(acts)):
            assert (fo[k].cpu().numpy() == outs[k][0]).all() and (fr[k].cpu().numpy() == outs[k][1]).all(), k
        assert_state_equal(fused, orc, STATE_FIELDS, where=case + " (rollout)")


def test_loader_picks_the_build_that_matches_the_devices_xnack_mode():
    import torch
    from gymwipe_amd import _native as nat
    arch = torch.cuda.get_device_properties(0).gcnArchName
    picked = os.path.basename(nat._pick_library())
    if os.environ.get("GW_LIB"):
        pytest.skip("GW_LIB set")
    assert picked == ("libgymwipe_amd_xnackoff.so" if "xnack-" in arch else "libgymwipe_amd.so"), (arch, picked)
    assert nat.lib()._name.endswith(picked)


@pytest.mark.parametrize("kw", [{}, {"explicit_queue": True, "per_env_stats": True}, {"per_env_geometry": True}], ids=["suffix", "explicit", "per-env-geometry"])
def test_snapshot_and_restore_continue_bit_for_bit(kw):
    """gw_get_snapshot / gw_set_state: a handle rewound to a snapshot, and a FRESH handle given the snapshot, both continue
    exactly as the original did -- every output of every later step and the final state (checkpoint / resume, SURVEY
    section 5; the reference can only save its agent's weights)."""
    import torch
    from gymwipe_amd import VecCounterTrafficEnv
    N, D, K1, K2 = 2048, 4, 30, 40
    dev, dur = action_stream(77, K1 + K2, N, D)
    mk = lambda: VecCounterTrafficEnv(N, num_devices=D, **kw)
    env = mk()
    if kw.get("per_env_geometry"):
        rng = np.random.default_rng(3)
        pos = np.zeros((N, D + 1, 2)); pos[:, :D] = rng.uniform(-3, 3, (N, D, 2))
        env.set_positions(pos)
    step = lambda e, k: e.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
    env.reset()
    for k in range(K1):
        if k == 17:
            env.reset(torch.from_numpy((np.arange(N) % 3 == 0).astype(np.uint8)))
        step(env, k)
    snap = env.snapshot()
    fields = ("now", "wake", "counter", "qlen", "queue", "received", "last_abs", "rx_power", "flags", "n_tx", "n_delivered", "n_popped", "n_dropped")

    def run_on(e):
        outs = []
        for k in range(K1, K1 + K2):
            if k == K1 + 11:
                e.reset()
            o, r, d_, _ = step(e, k)
            outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d_.cpu().numpy().copy()))
        return outs, {f: e.get_state(f) for f in fields}

    ref_out, ref_state = run_on(env)
    env.restore(snap)                                             # the same handle, rewound
    again_out, again_state = run_on(env)
    fresh = mk()                                                  # a new handle with the same configuration
    fresh.restore(snap)
    fresh_out, fresh_state = run_on(fresh)
    for got_out, got_state, who in ((again_out, again_state, "rewound handle"), (fresh_out, fresh_state, "fresh handle")):
        for k in range(K2):
            for a, b in zip(ref_out[k], got_out[k]):
                assert (a == b).all(), (who, k)
        for f in fields:
            assert (ref_state[f].view(np.uint8) == got_state[f].view(np.uint8)).all(), (who, f)
    other = VecCounterTrafficEnv(N, num_devices=D, multiplicity=[1, 1, 1, 1], **kw)   # another configuration refuses it
    with pytest.raises(Exception):
        other.restore(snap)


def test_derived_event_counts_equal_counted_ones():
    """Default mode derives most event counts from the state (steps = launches - bad, transmissions = steps + popped,
    appended = ticks x sum(mult), dropped = appended - popped - queued); the explicit-queue handle with
    GW_CFG_PER_ENV_STATS counts every event.  Same actions -- bad ones included --, masked resets, a hipGraph replay."""
    import torch
    from gymwipe_amd import VecCounterTrafficEnv
    N, D, K = 1024, 4, 48
    a = VecCounterTrafficEnv(N, num_devices=D)
    b = VecCounterTrafficEnv(N, num_devices=D, explicit_queue=True, per_env_stats=True)
    dev, dur = action_stream(91, K, N, D)
    dev[5, ::7] = D + 2                                            # actions outside the action space
    dur[9, ::11] = 25
    rng = np.random.default_rng(1)
    a.reset(); b.reset()
    g_dev = torch.zeros(N, dtype=torch.int32, device="cuda"); g_dur = torch.zeros(N, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    for k in range(K):
        if k % 13 == 6:
            m = torch.from_numpy((rng.random(N) < 0.3).astype(np.uint8))
            a.reset(m); b.reset(m)
        act = {"device": torch.from_numpy(dev[k]).cuda(), "duration": torch.from_numpy(dur[k]).cuda()}
        if k == 20:                                                # this step of `a` runs as a captured launch, replayed once
            g_dev.copy_(act["device"]); g_dur.copy_(act["duration"])
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                a.step({"device": g_dev, "duration": g_dur})
            graph.replay()
        else:
            a.step(act)
        b.step(act)
    torch.cuda.synchronize()
    for f in ("n_tx", "n_delivered", "n_appended", "n_popped", "n_dropped", "flags"):
        x, y = a.get_state(f), b.get_state(f)
        assert (x == y).all(), f
    sa, sb = a.stats(), b.stats()
    for key in ("steps", "transmissions", "delivered", "appended", "popped", "dropped", "bad_actions"):
        assert sa[key] == sb[key], key
    assert sa["bad_actions"] > 0
