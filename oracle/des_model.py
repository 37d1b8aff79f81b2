"""
oracle/des_model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Layer 1 of the oracle: a small, general, event-driven CPU restatement of the
Gym-WiPE network model that backs the frequency-band-assignment environments.
Pure-Python loops; meant for SMALL cases only (a few hundred env steps).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/.  The product path (gymwipe_amd/) never does.

What is restated, and from where (file:line relative to the reference tree):

  * the discrete-event core the reference rides on -- SimPy 3.0.11
    (Pipfile.lock:176-182), a third-party dependency that is ABSENT from the
    reference tree and not importable in the build container.  Its published
    algorithm is restated in `Sim`/`Ev`/`Proc`/`AnyOf` below: a binary heap of
    (time, priority, insertion-id) entries, URGENT(0) for process
    initialisation and numeric `until`, NORMAL(1) otherwise; timeouts are
    triggered at creation and queued at now+delay; `succeed()` queues at
    `now`; popping an event marks it processed and runs its callbacks in
    append order; a process is resumed by feeding its generator until it
    yields an event that is not yet processed.   
  * gymwipe/simtools.py:44-53,77-88,103-116 (slot timer, run, timeoutUntil),
    :232-432 (Notifier: prioritised callbacks, blocking/queued process
    admission)
  * gymwipe/devices/core.py:15-123 (Position, Device)
  * gymwipe/networking/physical.py:25-98 (dBm helpers, Eb/N0, Q approximation),
    :160-212 (Varshamov-Gilbert bound, BPSK MCS), :214-290 (Transmission),
    :308-397,500-528 (attenuation models + factory), :576-623 (FrequencyBand)
  * gymwipe/networking/attenuation_models.py:28-36 (FSPL)
  * gymwipe/networking/messages.py:42-75,113-124,143-154,172-180 (byte sizes)
  * gymwipe/networking/simple_stack.py:32-286 (SimplePhy), :289-484
    (SimpleMac), :486-561 (SimpleRrmMac)
  * gymwipe/networking/devices.py:40-111,113-203 (network / RRM devices)
  * gymwipe/networking/construction.py:77-174 (gates, ports), :221-342 (gate listeners)
  * gymwipe/envs/core.py:14-57,142-153 and gymwipe/envs/counter_traffic.py
    (the benchmarked environment, including its quirks: swapped payload
    constructor arguments :57, reset() that does not rewind time :135-144)

PINNING.  The reference itself cannot be run here (simpy, gym, pygame, py3ode
are not installed and stay absent).  This restatement is therefore pinned by
the KNOWN ANSWERS THE REFERENCE'S OWN TESTS HOLD, reproduced scenario by
scenario in `scenario_*` below and asserted in tests/test_oracle_pinning.py:
    tests/envs/test_counter_traffic.py:25-34   (+2/-2.0 then 0/+2.0)
    tests/networking/test_stack.py:219-235     (4, 4, 8, 8, 10/10 deliveries)
    tests/networking/test_stack.py:102-124,128 (PHY transmission properties)
    tests/test_simtools.py:16-43,60-120        (callback priorities, process admission counts)
    tests/networking/test_construction.py:18-40,73-135,137-200
                                               (ports; module ping-pong 19/20 at t=20, ten receptions per port at
                                                t=40; gate-listener admission)
    tests/networking/test_messages.py:6-14     (packet sizes)
-- i.e. every test of the reference that pins a result on this path.
Everything those tests do not cover (long rollouts, D=4/16 compositions) is
"parity unpinned" against the live reference and is only cross-checked
between this restatement and the independently written C restatement
(oracle/ct_oracle.c).
"""
import heapq
import math
from collections import deque
from fractions import Fraction
from math import e as _E, log10, pi, sqrt

# --------------------------------------------------------------------------
# Discrete-event core (restates SimPy 3.0.11 ordering)
# --------------------------------------------------------------------------
URGENT, NORMAL = 0, 1
_PENDING = object()


class _Stop(Exception):
    pass


class Ev:
    """One-shot event.  `cbs is None` <=> processed; `val` set <=> triggered."""
    __slots__ = ("sim", "cbs", "val", "ok")

    def __init__(self, sim):
        self.sim, self.cbs, self.val, self.ok = sim, [], _PENDING, None

    @property
    def triggered(self):
        return self.val is not _PENDING

    @property
    def processed(self):
        return self.cbs is None

    def succeed(self, value=None):
        if self.val is not _PENDING:
            raise RuntimeError("event already triggered")
        self.ok, self.val = True, value
        self.sim._push(self, NORMAL, 0)
        return self

    def __or__(self, other):
        return AnyOf(self.sim, (self, other))


class AnyOf(Ev):
    """Condition that fires as soon as one operand has been processed."""
    __slots__ = ()

    def __init__(self, sim, events):
        Ev.__init__(self, sim)
        for ev in events:
            if ev.cbs is None:
                self._check(ev)
            else:
                ev.cbs.append(self._check)

    def _check(self, _ev):
        if self.val is _PENDING:
            self.succeed()


class Proc(Ev):
    """Generator-backed process; its own event fires when the generator ends."""
    __slots__ = ("gen",)

    def __init__(self, sim, gen):
        Ev.__init__(self, sim)
        self.gen = gen
        init = Ev(sim)                      # "Initialize": URGENT, at now
        init.ok, init.val = True, None
        init.cbs.append(self._resume)
        sim._push(init, URGENT, 0)

    def _resume(self, ev):
        while True:
            try:
                ev = self.gen.send(ev.val)
            except StopIteration as stop:
                self.ok = True
                self.val = stop.args[0] if stop.args else None
                self.sim._push(self, NORMAL, 0)
                return
            if ev.cbs is not None:          # not processed yet: wait for it
                ev.cbs.append(self._resume)
                return


class Sim:
    def __init__(self, initial_time=0):     # simpy.Environment(initial_time=0)
        self.now = initial_time
        self._heap = []
        self._eid = 0
        self.n_popped = 0

    def _push(self, ev, prio, delay):
        heapq.heappush(self._heap, (self.now + delay, prio, self._eid, ev))
        self._eid += 1

    def event(self):
        return Ev(self)

    def timeout(self, delay, value=None):
        if delay < 0:
            raise ValueError("negative delay %r" % (delay,))
        ev = Ev(self)
        ev.ok, ev.val = True, value         # triggered at creation
        self._push(ev, NORMAL, delay)
        return ev

    def process(self, gen):
        return Proc(self, gen)

    # simtools.py:103-116
    def timeout_until(self, when, value=None):
        now = self.now
        if when > now:
            return self.timeout(when - now, value)
        return self.timeout(0, value)

    # simtools.py:44-53 (waits a FULL slot when already aligned)
    def next_slot(self, slot):
        return self.timeout(slot - (self.now % slot))

    def step(self):
        self.now, _, _, ev = heapq.heappop(self._heap)
        self.n_popped += 1
        cbs, ev.cbs = ev.cbs, None
        for cb in cbs:
            cb(ev)

    # simtools.py:77-88 + SimPy Environment.run
    def run(self, until):
        if not isinstance(until, Ev):
            at = float(self.now + until)    # runSimulation: now + duration
            if at <= self.now:
                raise ValueError("until must lie in the future")
            stop = Ev(self)
            stop.ok, stop.val = True, None
            self._push(stop, URGENT, at - self.now)
            until = stop
        elif until.cbs is None:
            return until.val

        def _halt(ev):
            raise _Stop()
        until.cbs.append(_halt)
        try:
            while True:
                if not self._heap:
                    raise RuntimeError("ran out of events before `until`")
                self.step()
        except _Stop:
            return until.val


# --------------------------------------------------------------------------
# Observer with process admission (simtools.py:232-432)
# --------------------------------------------------------------------------
class Notifier:
    def __init__(self, sim):
        self.sim = sim
        self._cbs = []          # (priority, seq, fn, extra_args)
        self._seq = 0
        self._execs = []        # executors in subscription order
        self._event = None

    def subscribe_callback(self, fn, priority=0, extra=None):
        assert all(c[2] is not fn for c in self._cbs)
        self._cbs.append((priority, self._seq, fn, extra))
        self._seq += 1
        # higher priority first; equal priority keeps subscription order
        # (the reference keeps equal priorities in a set: order unspecified,
        #  simtools.py:255,315-320)
        self._cbs.sort(key=lambda c: (-c[0], c[1]))

    def unsubscribe_callback(self, fn):
        n = len(self._cbs)
        self._cbs = [c for c in self._cbs if c[2] is not fn]
        assert len(self._cbs) == n - 1

    def subscribe_process(self, genfn, blocking=True, queued=False):
        sim = self.sim
        state = {"running": False, "queue": deque()}

        def run_next(_ev):
            if state["queue"]:
                nxt = state["queue"].popleft()
                sim.process(genfn(nxt)).cbs.append(run_next)
            else:
                state["running"] = False

        def clear(_ev):
            state["running"] = False

        def executor(value):
            if not blocking:
                sim.process(genfn(value))
            elif state["running"]:
                if queued:
                    state["queue"].append(value)
                # else: dropped (simtools.py:352-357)
            else:
                state["running"] = True
                p = sim.process(genfn(value))
                p.cbs.append(run_next if queued else clear)

        self._execs.append(executor)

    def trigger(self, value=None):
        for _prio, _seq, fn, extra in list(self._cbs):
            if extra is not None:
                fn(value, *extra)
            else:
                fn(value)
        for ex in list(self._execs):
            ex(value)
        if self._event is not None:
            ev, self._event = self._event, None
            ev.succeed(value)

    @property
    def event(self):
        if self._event is None:
            self._event = self.sim.event()
        return self._event


# --------------------------------------------------------------------------
# Geometry (devices/core.py:15-123)
# --------------------------------------------------------------------------
class Position:
    def __init__(self, sim, x, y):
        self._x, self._y = float(x), float(y)
        self.n_change = Notifier(sim)

    x = property(lambda s: s._x)
    y = property(lambda s: s._y)

    def set_x(self, x):
        if x != self._x:
            self._x = x
            self.n_change.trigger(self)

    def set(self, x, y):
        if x != self._x or y != self._y:
            self._x, self._y = x, y
            self.n_change.trigger(self)

    def same_as(self, p):
        return p._x == self._x and p._y == self._y

    def distance_to(self, p):               # devices/core.py:88-95
        return sqrt((self._x - p._x) ** 2 + (self._y - p._y) ** 2)


class Device:
    def __init__(self, sim, name, x, y):
        self.name = name
        self.position = Position(sim, x, y)


# --------------------------------------------------------------------------
# PHY arithmetic (physical.py:25-98,160-212)
# --------------------------------------------------------------------------
_SQRT_2PI = sqrt(2 * pi)


def mw_to_dbm(mw):                          # physical.py:82-89
    return 10 * log10(mw)


def dbm_to_mw(dbm):                         # physical.py:91-98
    return 10 ** (dbm / 10)


def q_approx(x):                            # physical.py:46-58
    assert x >= 0
    return (1 - _E ** (-1.4 * x)) * _E ** (-(x ** 2 / 2)) / (1.135 * _SQRT_2PI * x)


def noise_power_density(celsius):           # physical.py:60-71
    return 1.38e-23 * (celsius + 273.15)


class BpskMcs:
    """physical.py:187-212; code rate 3/4 unless stated."""
    _vg_cache = {}

    def __init__(self, code_rate=Fraction(3, 4)):
        self.code_rate = code_rate
        self.bit_rate = 133.33333e3
        self.data_rate = float(code_rate) * self.bit_rate

    def max_correctable_ber(self):          # physical.py:160-185
        cr = self.code_rate
        if cr not in BpskMcs._vg_cache:
            k, n = cr.numerator, cr.denominator
            bound = 2 ** (n - k)
            total, t = 0, 0
            while total <= bound:
                total += math.comb(n, t)
                t += 1
            t -= 1
            BpskMcs._vg_cache[cr] = float(t) / n
        return BpskMcs._vg_cache[cr]

    def ber(self, sig_dbm, noise_dbm):      # physical.py:208-212, :25-42
        if sig_dbm <= noise_dbm:
            return 0.5
        ratio_db = sig_dbm - noise_dbm - 10 * log10(self.bit_rate)
        ratio = 10 ** (ratio_db / 10)
        return q_approx(sqrt(2 * ratio))


# --------------------------------------------------------------------------
# Messages: only sizes, addresses and flags survive (messages.py)
# --------------------------------------------------------------------------
class Blob:
    """Transmittable (messages.py:42-75): a value with a byte size."""
    kind = "blob"

    def __init__(self, value, byte_size=None):
        self.value = value
        self.byte_size = (len(str(value).encode("utf-8"))
                          if byte_size is None else byte_size)

    @property
    def bit_size(self):
        return self.byte_size * 8


class MacHeader(Blob):                      # messages.py:143-154 (13 B)
    kind = "machdr"

    def __init__(self, src, dst, flag):
        Blob.__init__(self, (src, dst, flag), 13)
        self.src, self.dst, self.flag = src, dst, flag


class NetHeader(Blob):                      # messages.py:172-180 (12 B)
    kind = "nethdr"

    def __init__(self, src, dst):
        Blob.__init__(self, (src, dst), 12)
        self.src, self.dst = src, dst


class Pkt(Blob):                            # messages.py:113-124
    kind = "pkt"

    def __init__(self, header, payload):
        Blob.__init__(self, (header, payload), header.byte_size + payload.byte_size)
        self.header, self.payload = header, payload


class Cmd:
    """Inter-layer message with a completion event (messages.py:201-225)."""

    def __init__(self, sim, kind, **args):
        self.kind, self.args = kind, args
        self.done = sim.event()

    def set_done(self, value=None):
        self.done.succeed(value)


# --------------------------------------------------------------------------
# Medium: attenuation, transmissions, band (physical.py:214-290,308-397,500-623)
# --------------------------------------------------------------------------
class FsplLink:
    STANDBY = 3000.0                        # physical.py:371

    def __init__(self, sim, freq, dev_a, dev_b, extra=None):
        assert dev_a is not dev_b
        self.freq, self.a, self.b = freq, dev_a, dev_b
        self.extra = extra                  # dB of the pair's custom models (JoinedAttenuationModel), or None
        self.attenuation = 0
        self.fspl = 0                       # the FsplAttenuation member's own value (AttenuationModel starts at 0, physical.py:322)
        self.n_changes = Notifier(sim)
        for dev in (dev_a, dev_b):          # physical.py:380-386
            dev.position.n_change.subscribe_callback(self._moved, extra=[dev])
        self._update()

    def _moved(self, _pos, _dev):
        if self.a.position.distance_to(self.b.position) < self.STANDBY:
            self._update()

    def _update(self):                      # attenuation_models.py:28-36
        pa, pb = self.a.position, self.b.position
        if not pa.same_as(pb):              # co-located: the FSPL member returns without setting anything (:31-33) and keeps
            self.fspl = 20 * log10(pa.distance_to(pb)) + 20 * log10(self.freq) - 147.55   # its previous value (0 at first)
        att = self.fspl
        if self.extra:                      # physical.py:457: sum() over the models' values, FSPL first
            att = sum([att, self.extra])
        if att != self.attenuation:         # physical.py:354-362
            self.attenuation = att
            self.n_changes.trigger(att)


class Tx:
    def __init__(self, sim, sender, power, packet, mcs_h, mcs_p):
        self.sim, self.sender, self.power, self.packet = sim, sender, power, packet
        self.mcs_h, self.mcs_p = mcs_h, mcs_p
        self.start = sim.now
        self.hdr_dur = packet.header.bit_size / mcs_h.data_rate
        self.pay_dur = packet.payload.bit_size / mcs_p.data_rate
        self.dur = self.hdr_dur + self.pay_dur
        self.stop = self.start + self.dur
        self.hdr_bits = packet.header.bit_size * float(2 - mcs_h.code_rate)
        self.pay_bits = packet.payload.bit_size * float(2 - mcs_p.code_rate)
        self.e_hdr = sim.timeout_until(self.start + self.hdr_dur, self)
        self.e_done = sim.timeout_until(self.stop, self)

    @property
    def completed(self):
        return self.sim.now >= self.stop


class Band:
    def __init__(self, sim, freq=2.4e9, bandwidth=22e6):
        self.sim, self.freq, self.bandwidth = sim, freq, bandwidth
        self._links = {}
        self.custom = {}                    # frozenset of two device ids -> extra dB (setCustomModels, physical.py:477-498)
        self._txs = deque()
        self.n_new_tx = Notifier(sim)
        self.log = []                       # every transmission, for tests

    def link(self, dev_a, dev_b):           # physical.py:500-528
        key = frozenset((id(dev_a), id(dev_b)))
        if key not in self._links:
            self._links[key] = FsplLink(self.sim, self.freq, dev_a, dev_b, self.custom.get(key))
        return self._links[key]

    def transmit(self, sender, power, packet, mcs_h, mcs_p):
        if self._txs and self._txs[0].completed:     # physical.py:614-616
            self._txs.popleft()
        tx = Tx(self.sim, sender, power, packet, mcs_h, mcs_p)
        self._txs.append(tx)
        self.log.append(tx)
        # notification deferred by a zero-delay event (physical.py:601-607)
        self.sim.timeout(0).cbs.append(lambda _ev: self.n_new_tx.trigger(tx))
        return tx

    def active(self):                       # physical.py:610-623
        while self._txs and self._txs[0].completed:
            self._txs.popleft()
        return list(self._txs)


# --------------------------------------------------------------------------
# SimplePhy (simple_stack.py:32-286)
# --------------------------------------------------------------------------
SLOT = 1e-6                                 # simple_stack.py:27


class Phy:
    NOISE_DENSITY = noise_power_density(20.0)        # simple_stack.py:57

    def __init__(self, sim, device, band):
        self.sim, self.device, self.band = sim, device, band
        self.mac_in = Notifier(sim)         # gate "macIn"
        self.mac_out = Notifier(sim)        # gate "macOut"
        self.transmitting = False
        self.cur_tx = None
        self.receiving = False
        self.n_rx_done = Notifier(sim)
        self.rx_mcs = None
        self._reset_errors()
        self.thermal = self.NOISE_DENSITY * band.bandwidth * 1000
        self._tx_pow = {}
        self._tx_attcb = {}
        self.rx_power = self.thermal
        self.n_pow = Notifier(sim)
        self.n_pow.subscribe_callback(self._add_power, priority=1)
        band.n_new_tx.subscribe_callback(self._on_new_tx)
        band.n_new_tx.subscribe_process(self._receive)           # blocking, not queued
        self.mac_in.subscribe_process(self._mac_in, queued=True)  # :192
        self.decisions = []                 # (what, ok, err_sum, bits) for tests

    def _add_power(self, delta):            # simple_stack.py:81-86
        self.rx_power += delta

    def _rx_from(self, tx, att=None):       # simple_stack.py:99-111
        if att is None:
            att = self.band.link(self.device, tx.sender).attenuation
        return dbm_to_mw(tx.power - att)

    def _on_att_change(self, tx, att):      # simple_stack.py:119-128
        new = self._rx_from(tx, att)
        delta = new - self._tx_pow[tx]
        self._tx_pow[tx] = new
        self.n_pow.trigger(delta)

    def _on_new_tx(self, tx):               # simple_stack.py:130-144
        if tx is not self.cur_tx:
            p = self._rx_from(tx)
            self._tx_pow[tx] = p
            self.n_pow.trigger(p)
            tx.e_done.cbs.append(self._on_tx_done)
            cb = lambda att, tx=tx: self._on_att_change(tx, att)
            self._tx_attcb[tx] = cb
            self.band.link(self.device, tx.sender).n_changes.subscribe_callback(cb)

    def _on_tx_done(self, ev):              # simple_stack.py:146-157
        tx = ev.val
        p = self._tx_pow.pop(tx)
        self.n_pow.trigger(-p)
        cb = self._tx_attcb.pop(tx)
        self.band.link(self.device, tx.sender).n_changes.unsubscribe_callback(cb)

    def _update_ber(self, tx):              # simple_stack.py:161-173
        sig = self._tx_pow[tx]
        noise = self.rx_power - sig
        assert sig >= 0 and noise >= 0
        self.ber = self.rx_mcs.ber(mw_to_dbm(sig), mw_to_dbm(noise))

    def _reset_errors(self):                # simple_stack.py:175-178
        self.err_sum = 0
        self.ber = 0.0
        self.t_seg = self.sim.now

    def _count_errors(self):                # simple_stack.py:180-188 (t_seg NOT advanced)
        dur = self.sim.now - self.t_seg
        self.err_sum += self.ber * dur * self.rx_mcs.bit_rate

    def _decide(self, what, err_sum, total_bits, mcs):          # :269-286
        ok = (round(err_sum) / total_bits) <= mcs.max_correctable_ber()
        self.decisions.append((what, ok, err_sum, total_bits))
        return ok

    def _mac_in(self, cmd):                 # simple_stack.py:192-212
        if cmd.kind == "SEND":
            if self.receiving:
                yield self.n_rx_done.event
            self.transmitting = True
            yield self.sim.next_slot(SLOT)
            a = cmd.args
            tx = self.band.transmit(self.device, a["power"], a["packet"], a["mcs"], a["mcs"])
            self.cur_tx = tx
            yield tx.e_done
            self.transmitting = False
            cmd.set_done()

    def _receive(self, tx):                 # simple_stack.py:214-267
        if not self.transmitting:
            self.receiving = True
            self.rx_mcs = tx.mcs_h
            self._reset_errors()

            def on_power_change(delta):
                if delta != 0:
                    self._count_errors()
                    if not tx.completed:
                        self._update_ber(tx)
            self.n_pow.subscribe_callback(on_power_change)
            self._update_ber(tx)
            yield tx.e_hdr
            self._count_errors()
            if self._decide("hdr", self.err_sum, tx.hdr_bits, tx.mcs_h):
                self.rx_mcs = tx.mcs_p
                self._reset_errors()
                self._update_ber(tx)
                yield tx.e_done
                self._count_errors()
                if self._decide("pay", self.err_sum, tx.pay_bits, tx.mcs_p):
                    self.mac_out.trigger(tx.packet)
            self.n_pow.unsubscribe_callback(on_power_change)
            self._reset_errors()
            self.receiving = False
            self.n_rx_done.trigger()


# --------------------------------------------------------------------------
# SimpleMac / SimpleRrmMac (simple_stack.py:289-561)
# --------------------------------------------------------------------------
RRM_ADDR = bytes(6)


class Mac:
    def __init__(self, sim, device, addr):
        self.sim, self.device, self.addr = sim, device, addr
        self.phy_in, self.phy_out = Notifier(sim), Notifier(sim)
        self.net_in, self.net_out = Notifier(sim), Notifier(sim)
        self.queue = deque(maxlen=100)      # simple_stack.py:361
        self._added = sim.event()
        self.mcs = BpskMcs()
        self.tx_power = 0.0
        self.receiving = False
        self._rx_cmd = None
        self._rx_timeout = None
        self.phy_in.subscribe_process(self._phy_in)          # blocking, NOT queued (:386)
        self.net_in.subscribe_callback(self._net_in)         # plain callback (:450)

    def _phy_in(self, packet):              # simple_stack.py:386-448
        hdr = packet.header
        if hdr.dst == self.addr:
            if hdr.src == RRM_ADDR:
                if hdr.flag == 1:
                    slots = packet.payload.value
                    total = slots * SLOT
                    stop = self.sim.now + total
                    t_out = self.sim.timeout(total)
                    have = True
                    while not t_out.processed:
                        if len(self.queue) == 0:
                            have = False
                            yield self._added | t_out
                            if not t_out.processed:
                                have = True
                        if have:
                            left = stop - self.sim.now
                            need = self.queue[0].bit_size / self.mcs.data_rate
                            if not left > need:
                                yield t_out
                            else:
                                pkt = self.queue.popleft()
                                cmd = Cmd(self.sim, "SEND", packet=pkt,
                                          power=self.tx_power, mcs=self.mcs)
                                self.phy_out.trigger(cmd)
                                yield cmd.done
            else:
                if self.receiving:
                    self._rx_cmd.set_done(packet.payload)
                    self._stop_rx()
        return
        yield  # pragma: no cover  (keeps this a generator on every path)

    def _net_in(self, cmd):                 # simple_stack.py:450-471
        if isinstance(cmd, Cmd):
            if cmd.kind == "RECEIVE":
                self._rx_cmd = cmd
                self.receiving = True
                self._rx_timeout = self.sim.timeout(cmd.args["duration"])
                self._rx_timeout.cbs.append(self._rx_timed_out)
        else:
            pkt = Pkt(MacHeader(self.addr, cmd.header.dst, 0), cmd)
            self.queue.append(pkt)
            self._added.succeed()
            self._added = self.sim.event()

    def _rx_timed_out(self, ev):            # simple_stack.py:473-478
        if ev is self._rx_timeout:
            self._rx_cmd.set_done()
            self._stop_rx()

    def _stop_rx(self):
        self._rx_cmd, self.receiving, self._rx_timeout = None, False, None


class RrmMac:
    def __init__(self, sim, device):
        self.sim, self.device, self.addr = sim, device, RRM_ADDR
        self.phy_in, self.phy_out = Notifier(sim), Notifier(sim)
        self.net_in, self.net_out = Notifier(sim), Notifier(sim)
        self.mcs = BpskMcs()
        self.tx_power = 0.0
        self._n_announce = Notifier(sim)
        self._n_announce.subscribe_process(self._announce, queued=True)   # :523
        self.phy_in.subscribe_callback(lambda p: self.net_out.trigger(p.payload))  # :527-529
        self.net_in.subscribe_callback(self._n_announce.trigger)               # :531-534

    def _announce(self, assign):            # simple_stack.py:536-561
        dest, slots = assign.args["dest"], assign.args["duration"]
        ann = Pkt(MacHeader(self.addr, dest, 1), Blob(slots))
        cmd = Cmd(self.sim, "SEND", packet=ann, power=self.tx_power, mcs=self.mcs)
        self.phy_out.trigger(cmd)
        yield cmd.done
        yield self.sim.timeout((slots + 1) * SLOT)
        assign.set_done()


def _wire(mac, phy):
    """Port.biConnectWith (construction.py:142-174), flattened."""
    mac.phy_out.subscribe_callback(phy.mac_in.trigger)
    phy.mac_out.subscribe_callback(mac.phy_in.trigger)


# --------------------------------------------------------------------------
# Devices (networking/devices.py)
# --------------------------------------------------------------------------
class World:
    """One simulation: clock, band and the MAC address counter
    (simple_stack.py:374-384 keeps the latter in a class global)."""

    def __init__(self, start_time=0):
        self.sim = Sim(start_time)
        self.band = Band(self.sim)
        self._mac_ctr = 0

    def new_mac(self):
        self._mac_ctr += 1
        addr = bytearray(6)
        addr[5] = self._mac_ctr
        return bytes(addr)


class NetDevice(Device):                    # networking/devices.py:40-111
    def __init__(self, world, name, x, y):
        Device.__init__(self, world.sim, name, x, y)
        self.world = world
        self.mac_addr = world.new_mac()
        self.phy = Phy(world.sim, self, world.band)
        self.mac = Mac(world.sim, self, self.mac_addr)
        _wire(self.mac, self.phy)

    def send(self, data, dst):              # :84-86
        self.mac.net_in.trigger(Pkt(NetHeader(self.mac_addr, dst), data))


class RrmDevice(Device):                    # networking/devices.py:113-203
    def __init__(self, world, name, x, y, index_to_mac, interpreter):
        Device.__init__(self, world.sim, name, x, y)
        self.world, self.interpreter = world, interpreter
        self.index_to_mac = index_to_mac
        self.mac_to_index = {m: i for i, m in index_to_mac.items()}
        self.phy = Phy(world.sim, self, world.band)
        self.mac = RrmMac(world.sim, self)
        _wire(self.mac, self.phy)
        self.mac.net_out.subscribe_callback(self._sniffed)

    def _sniffed(self, p):                  # :163-168
        self.interpreter.on_packet(self.mac_to_index[p.header.src],
                                   self.mac_to_index[p.header.dst], p.payload)

    def assign(self, index, slots):         # :178-203
        cmd = Cmd(self.world.sim, "ASSIGN", duration=slots, dest=self.index_to_mac[index])
        self.interpreter.on_assignment(slots, index)     # swapped on purpose (:200)
        self.mac.net_in.trigger(cmd)
        return cmd


# --------------------------------------------------------------------------
# CounterTraffic (envs/counter_traffic.py, envs/core.py)
# --------------------------------------------------------------------------
COUNTER_INTERVAL = 0.001
COUNTER_BYTE_LENGTH = 2
COUNTER_BOUND = 2 ** (8 * COUNTER_BYTE_LENGTH)
MAX_ASSIGN_DURATION = 20
ASSIGNMENT_DURATION_FACTOR = 1000


def circle_layout(num_devices, radius=2.0):
    """senders on a circle around the RRM; D=2 gives the
    reference's (0,2),(0,-2) (counter_traffic.py:124-127)."""
    if num_devices == 2:
        return [(0.0, 2.0), (0.0, -2.0)]
    out = []
    for i in range(num_devices):
        ang = math.pi / 2 - 2 * math.pi * i / num_devices
        out.append((radius * math.cos(ang), radius * math.sin(ang)))
    return out


def default_multiplicity(num_devices):
    return [1 if i % 2 == 0 else 3 for i in range(num_devices)]


class _Sender(NetDevice):                   # counter_traffic.py:37-61
    def __init__(self, world, name, x, y, mult, traffic=True, interval=None):
        NetDevice.__init__(self, world, name, x, y)
        self.mult = mult
        self.interval = COUNTER_INTERVAL if interval is None else interval
        self.counter = 1
        self.dest = None
        self.got = []                       # payloads handed up by a receive-mode MAC
        if traffic:
            world.sim.process(self._run())

    def keep_receiving(self, duration):
        """The receiver process of tests/networking/test_stack.py:176-186: a
        RECEIVE command re-issued as soon as the previous one completes."""
        def loop():
            while True:
                cmd = Cmd(self.world.sim, "RECEIVE", duration=duration)
                self.mac.net_in.trigger(cmd)
                res = yield cmd.done
                if res is not None:
                    self.got.append(res)
        self.world.sim.process(loop())

    def _run(self):
        assert self.dest is not None
        while True:
            for _ in range(self.mult):
                # swapped constructor arguments (:57): value=2, byteSize=counter
                self.send(Blob(COUNTER_BYTE_LENGTH, self.counter), self.dest)
            if self.counter < COUNTER_BOUND:
                self.counter += 1
            yield self.world.sim.timeout(self.interval)


class _Interp:                              # counter_traffic.py:63-112
    def __init__(self, n):
        self.n = n
        self.reset()

    def reset(self):
        self.latest_diff = 0
        self.last_abs = 0
        self.received = [0] * self.n
        self.done = False

    def on_packet(self, src, dst, payload):
        self.received[src] = payload.value
        self.latest_diff = self.received[0] - self.received[1]
        if payload.value == COUNTER_BOUND:
            self.done = True

    def on_assignment(self, device_index, duration):
        self.last_assign = device_index

    def feedback(self):                     # envs/core.py:142-153
        obs = self.latest_diff + COUNTER_BOUND
        abs_d = abs(self.latest_diff)
        reward = self.last_abs - abs_d
        self.last_abs = abs_d
        reward = 10 if reward > 10 else (-10 if reward < -10 else reward)
        return obs, float(reward), self.done, {"Latest received values": str(self.received)}


class CounterTrafficModel:
    """D-sender generalisation of CounterTrafficEnv; D=2 with the default
    layout IS the reference env (counter_traffic.py:114-158)."""

    def __init__(self, num_devices=2, positions=None, mult=None, dest=None,
                 rrm_pos=(0.0, 0.0), traffic=True, peer_receive=False,
                 rx_duration=10, float_duration=False, extra_att=None, counter_interval=None, start_time=0):
        """start_time: the clock's initial value (test hook; the reference's SimMan.init() starts at 0,
        simtools.py:90-95 -- everything else is unchanged, the first counter tick falls on it).
        traffic=False: no counter processes, packets come from enqueue();
        peer_receive: every sender MAC is kept in receive mode (SURVEY 8f rank
        2; the reference env never does this); float_duration: the assignment
        duration is passed as a float, as test_stack.py:197 does, which makes
        the announcement payload len(str(float)) bytes long."""
        D = num_devices
        positions = positions or circle_layout(D)
        mult = mult or default_multiplicity(D)
        dest = dest or [(i + 1) % D for i in range(D)]
        self.world = World(start_time)
        self.sim = self.world.sim
        self.senders = [_Sender(self.world, "Sender %d" % (i + 1),
                                positions[i][0], positions[i][1], mult[i], traffic, counter_interval)
                        for i in range(D)]
        idx2mac = {i: s.mac_addr for i, s in enumerate(self.senders)}
        for i, s in enumerate(self.senders):
            s.dest = self.senders[dest[i]].mac_addr
        self.interp = _Interp(D)
        self.rrm = RrmDevice(self.world, "RRM", rrm_pos[0], rrm_pos[1], idx2mac, self.interp)
        if extra_att:                       # {(a, b): dB}, radio index D = the RRM; set before any link exists
            radios = self.senders + [self.rrm]
            assert not self.world.band._links
            for (a, b), db in extra_att.items():
                self.world.band.custom[frozenset((id(radios[a]), id(radios[b])))] = float(db)
        self.num_devices = D
        self.float_duration = float_duration
        if peer_receive:
            for s in self.senders:
                s.keep_receiving(rx_duration)

    def enqueue(self, device, nbytes):      # networking/devices.py:84-86
        s = self.senders[device]
        s.send(Blob(COUNTER_BYTE_LENGTH, nbytes), s.dest)

    def reset(self):                        # counter_traffic.py:135-144
        for s in self.senders:
            s.counter = 0
        self.interp.reset()
        return self.interp.latest_diff + COUNTER_BOUND

    def step(self, device, duration):       # counter_traffic.py:146-158
        assert 0 <= device < self.num_devices and 0 <= duration < MAX_ASSIGN_DURATION
        slots = duration * ASSIGNMENT_DURATION_FACTOR
        sig = self.rrm.assign(device, float(slots) if self.float_duration else slots)
        self.sim.run(sig.done)
        return self.interp.feedback()

    # ---- state snapshot in the layout the C oracle / GPU path expose -------
    def snapshot(self):
        radios = self.senders + [self.rrm]
        return {
            "now": self.sim.now,
            "counters": [s.counter for s in self.senders],
            "qlen": [len(s.mac.queue) for s in self.senders],
            "queues": [[p.byte_size for p in s.mac.queue] for s in self.senders],
            "received": list(self.interp.received),
            "peer_received": [len(s.got) for s in self.senders],
            "rx_power": [r.phy.rx_power for r in radios],
            "n_tx": len(self.world.band.log),
        }


# --------------------------------------------------------------------------
# Networked control loop (second half) -- BUILDER-DEFINED
# --------------------------------------------------------------------------
class LinearPlant:
    """x <- A x + B u per substep, each row evaluated left to right without
    fused multiply-adds (the order the HIP kernel uses)."""

    def __init__(self, A, B, x0, u0):
        self.A, self.B = [list(r) for r in A], list(B)
        self.x, self.u = list(x0), float(u0)
        self.substeps = 0

    def step(self):
        x, u, nx = self.x, self.u, []
        for i in range(4):
            acc = self.A[i][0] * x[0]
            acc = acc + self.A[i][1] * x[1]
            acc = acc + self.A[i][2] * x[2]
            acc = acc + self.A[i][3] * x[3]
            acc = acc + self.B[i] * u
            nx.append(acc)
        self.x = nx
        self.substeps += 1


def default_plant_matrices(dt=1e-3):
    """The builder-defined pendulum of gw_plant_config_default (forward Euler)."""
    tau, g_l, damp = 0.05, 9.81, 0.2
    Ac = [[0, 1, 0, 0], [0, -1 / tau, 0, 0], [0, 0, 0, 1], [0, 1 / tau, -g_l, -damp]]
    Bc = [0, 1 / tau, 0, -1 / tau]
    A = [[(1.0 if i == j else 0.0) + dt * Ac[i][j] for j in range(4)] for i in range(4)]
    return A, [dt * b for b in Bc]


class ControlLoopModel:
    """The pendulum env with its loop CLOSED, as the reference intends it (envs/inverted_pendulum.py:60-113,
    plants/sliding_pendulum.py:116-155, control/inverted_pendulum.py:16-69) but never runs it (nobody sets
    `receiving`).  Builder-defined choices, all stated:
      * three network devices -- sensor (index 0), controller (1), actuator (2, not assignable) -- and the RRM;
        controller and actuator have `receiving = True` (networking/devices.py:71-111);
      * every SAMPLE_INTERVAL the sensor sends the plant's angle (value = angle, 2 bytes: the reference's
        Transmittable(2, angle) has the arguments swapped) and then advances the plant by one substep
        (OdePlant.updateState in whole substeps, plants/core.py:38-49); positions stay fixed;
      * the controller keeps the angle of the last sensor packet it received, in degrees (control/..:39-41), and
        every `period` ticks from tick `start` on sends -angle as the motor velocity when the angle is not 0
        (the PID of :52-69 with its shipped gains kp=1, ki=kd=0; aligned to the tick grid instead of its own
        1 s + k*10 ms timer);
      * the actuator applies the velocity of every command it receives (sliding_pendulum.py:154-155);
      * observation / reward as InvertedPendulumInterpreter computes them (envs/inverted_pendulum.py:27-57)."""
    SENSOR, CONTROLLER, ACTUATOR = 0, 1, 2

    def __init__(self, positions=((0.0, 0.0), (0.0, -1.0), (0.5, 0.0)), rrm_pos=(0.0, 1.0), start=20, period=10,
                 A=None, B=None, x0=(0.0, 0.0, 0.05, 0.0), u0=0.1):
        if A is None:                        # input sign chosen so that the reference's control law (u = -angle) damps
            A, B = default_plant_matrices()
            B = [-b for b in B]
        self.world = World()
        self.sim = sim = self.world.sim
        self.plant = LinearPlant(A, B, x0, u0)
        self.devs = [NetDevice(self.world, n, x, y) for n, (x, y) in zip(("Sensor", "Controller", "Actuator"), positions)]
        sensor, controller, actuator = self.devs
        self.angle_deg = 0.0                 # the controller's view
        self.start, self.period = start, period
        self.tick = 0
        self.n_cmd = 0

        def sensor_proc():
            while True:
                sensor.send(Blob(self.plant.x[2], 2), controller.mac_addr)
                self.plant.step()
                yield sim.timeout(COUNTER_INTERVAL)

        def controller_proc():
            k = 0
            while True:
                if k >= self.start and (k - self.start) % self.period == 0 and self.angle_deg != 0.0:
                    controller.send(Blob(-self.angle_deg, 1), actuator.mac_addr)
                    self.n_cmd += 1
                k += 1
                self.tick = k
                yield sim.timeout(COUNTER_INTERVAL)

        def receiver(dev, on_receive):
            def loop():
                while True:
                    cmd = Cmd(sim, "RECEIVE", duration=100)          # RECEIVE_TIMEOUT, devices.py:66
                    dev.mac.net_in.trigger(cmd)
                    res = yield cmd.done
                    if res is not None:
                        on_receive(res)
            sim.process(loop())

        sim.process(sensor_proc())
        sim.process(controller_proc())
        self.got = [0, 0, 0]

        def at_controller(packet):          # the network packet; control/inverted_pendulum.py:39-41
            self.angle_deg = math.degrees(packet.payload.value)
            self.got[1] += 1

        def at_actuator(packet):            # sliding_pendulum.py:154-155
            self.plant.u = packet.payload.value
            self.got[2] += 1
        receiver(controller, at_controller)
        receiver(actuator, at_actuator)
        idx2mac = {i: d.mac_addr for i, d in enumerate(self.devs)}
        self.rrm = RrmDevice(self.world, "RRM", rrm_pos[0], rrm_pos[1], idx2mac, _NullInterp())

    def feedback(self):
        deg = math.degrees(self.plant.x[2])
        return int(deg), float(abs(180 - deg)), False, {"Sensor angle": deg}

    def step(self, device, duration):
        assert 0 <= device < 2 and 0 <= duration < MAX_ASSIGN_DURATION
        sig = self.rrm.assign(device, duration * ASSIGNMENT_DURATION_FACTOR)
        self.sim.run(sig.done)
        return self.feedback()

    def snapshot(self):
        radios = self.devs + [self.rrm]
        return {"now": self.sim.now, "x": list(self.plant.x), "u": self.plant.u, "angle_deg": self.angle_deg,
                "qlen": [len(d.mac.queue) for d in self.devs], "received": list(self.got),
                "substeps": self.plant.substeps, "n_tx": len(self.world.band.log),
                "rx_power": [r.phy.rx_power for r in radios], "commands": self.n_cmd}


class _NullInterp:
    def on_packet(self, src, dst, payload):
        pass

    def on_assignment(self, device_index, duration):
        pass


# --------------------------------------------------------------------------
# The reference's own result-pinning test scenarios, restated
# --------------------------------------------------------------------------
def scenario_counter_traffic():
    """tests/envs/test_counter_traffic.py:17-34 -- returns the two
    (obs - 65536, reward) pairs the reference asserts on."""
    env = CounterTrafficModel(2)
    o1, r1, _, _ = env.step(0, 3)
    o2, r2, _, _ = env.step(1, 12)
    return [(o1 - COUNTER_BOUND, r1), (o2 - COUNTER_BOUND, r2)], env


def scenario_simple_phy():
    """tests/networking/test_stack.py:43-132.  Returns the observations the
    reference asserts on."""
    w = World()
    sim, band = w.sim, w.band
    d1, d2 = Device(sim, "1", 0, 0), Device(sim, "2", 1, 1)
    p1, p2 = Phy(sim, d1, band), Phy(sim, d2, band)
    delivered = []
    p2.mac_out.subscribe_callback(delivered.append)
    packet = Pkt(Blob(None, 8), Blob(None, 128))
    out = {}

    def sending():
        out["idle_before"] = len(band.active())
        mcs = BpskMcs()
        cmd = Cmd(sim, "SEND", packet=packet, power=0.0, mcs=mcs)
        p1.mac_in.trigger(cmd)
        yield sim.timeout(8 / mcs.data_rate)
        txs = band.active()
        out["active_during"] = len(txs)
        t = txs[0]
        out["tx_fields_ok"] = (t.packet is packet and t.power == 0.0
                               and t.mcs_h is mcs and t.mcs_p is mcs)
        power = p2.rx_power
        yield sim.timeout(64 / mcs.data_rate)
        d2.position.set_x(2)
        yield sim.timeout(16 / mcs.data_rate)
        out["power_dropped"] = p2.rx_power < power
        out["power_before"], out["power_after"] = power, p2.rx_power
        yield sim.timeout(1)
        out["active_after"] = len(band.active())

    def receiving():
        yield sim.timeout(4)
        out["delivered_last_is_packet"] = bool(delivered) and delivered[-1] is packet

    sim.process(sending())
    sim.process(receiving())
    sim.run(200)
    out["decisions"] = list(p2.decisions)
    return out


def scenario_simple_mac():
    """tests/networking/test_stack.py:134-235.  Returns the six delivery counts
    the reference asserts on: (rx2, rx1, rx2, rx1, rx1_final, rx2_final)."""
    w = World()
    sim, band = w.sim, w.band
    d1, d2 = Device(sim, "1", 0, 0), Device(sim, "2", 1, 1)
    p1, p2 = Phy(sim, d1, band), Phy(sim, d2, band)
    rrm = Device(sim, "RRM", 2, 2)
    rrm_phy = Phy(sim, rrm, band)
    rrm_mac = RrmMac(sim, rrm)
    m1 = Mac(sim, d1, w.new_mac())
    m2 = Mac(sim, d2, w.new_mac())
    _wire(m1, p1)          # collector proxies are pass-through (:147-156)
    _wire(m2, p2)
    _wire(rrm_mac, rrm_phy)

    def sender(src, dst, payloads):
        for p in payloads:
            src.net_in.trigger(Pkt(NetHeader(src.addr, dst.addr), p))
            yield sim.timeout(1e-4)

    def receiver(mac, got):
        while True:
            cmd = Cmd(sim, "RECEIVE", duration=10)
            mac.net_in.trigger(cmd)
            res = yield cmd.done
            if res is not None:
                got.append(res)

    ASSIGN_TIME = 0.01
    ANNOUNCE_TIME = (13 + log10(ASSIGN_TIME / SLOT)) * 8 / rrm_mac.mcs.data_rate

    def management():
        prev = None
        for i in range(10):
            dest = m1.addr if i % 2 == 0 else m2.addr
            cmd = Cmd(sim, "ASSIGN", duration=ASSIGN_TIME / SLOT, dest=dest)
            rrm_mac.net_in.trigger(cmd)
            if prev is not None:
                yield prev.done
            prev = cmd

    got1, got2 = [], []
    sim.process(sender(m1, m2, [Blob(i) for i in range(10)]))
    sim.process(sender(m2, m1, [Blob(i) for i in range(10, 20)]))
    sim.process(receiver(m1, got1))
    sim.process(receiver(m2, got2))
    sim.process(management())
    ROUND = ANNOUNCE_TIME + ASSIGN_TIME
    counts = []
    sim.run(ROUND); counts.append(len(got2))
    sim.run(ROUND); counts.append(len(got1))
    sim.run(ROUND); counts.append(len(got2))
    sim.run(ROUND); counts.append(len(got1))
    sim.run(6 * ROUND)
    counts += [len(got1), len(got2)]
    return counts


class Gate:
    """construction.py:77-111: send() triggers nReceives; a connection is a callback on it."""

    def __init__(self, sim):
        self.n_receives = Notifier(sim)

    def connect_to(self, gate):
        self.n_receives.subscribe_callback(gate.send)

    def send(self, obj):
        self.n_receives.trigger(obj)


class Port:
    """construction.py:114-174."""

    def __init__(self, sim):
        self.input, self.output = Gate(sim), Gate(sim)

    def bi_connect_with(self, port):
        self.output.connect_to(port.input)
        port.output.connect_to(self.input)


def scenario_module_ping_pong():
    """tests/networking/test_construction.py:73-135 -- two modules with ports a and b, connected in a
    bidirectional cycle, pass a counter around (one time unit per hop, direction reversed at every multiple of
    ten).  Returns what the reference asserts: (m1.msgVal, m2.msgVal) at t = 20 and the four receive counts of
    each module at t = 40."""
    sim = Sim()

    class Mod:
        def __init__(self):
            self.ports = {"a": Port(sim), "b": Port(sim)}
            self.count = {"a": 0, "b": 0}
            self.val = None
            sim.process(self.proc("a", "b"))
            sim.process(self.proc("b", "a"))

        def proc(self, frm, to):
            while True:
                msg = yield self.ports[frm].input.n_receives.event
                self.val = msg
                self.count[frm] += 1
                msg += 1
                yield sim.timeout(1)
                if msg % 10 == 0:
                    self.ports[frm].output.send(msg)
                else:
                    self.ports[to].output.send(msg)

    m1, m2 = Mod(), Mod()
    m1.ports["b"].bi_connect_with(m2.ports["b"])
    m2.ports["a"].bi_connect_with(m1.ports["a"])
    out = {}

    def simulation():
        m1.ports["a"].input.send(1)
        yield sim.timeout(20)
        out["vals_t20"] = [m1.val, m2.val]
        yield sim.timeout(20)
        out["counts_t40"] = [[m.count["a"], m.count["b"]] for m in (m1, m2)]
    sim.process(simulation())
    sim.run(50)
    return out


def scenario_gate_listeners():
    """tests/networking/test_construction.py:137-200 -- GateListener admission (construction.py:221-342): plain
    methods are callbacks; generator methods are blocking processes, queued or not.  Returns the logs the
    reference asserts on, for two identical modules."""
    sim = Sim()

    class Mod:
        def __init__(self):
            self.gates = {"aIn": Gate(sim), "bIn": Gate(sim)}
            self.logs = [[] for _ in range(4)]
            self.gates["aIn"].n_receives.subscribe_callback(lambda m: self.logs[0].append(m))
            self.gates["aIn"].n_receives.subscribe_callback(lambda m: self.logs[1].append(m))

            def b_plain(m):
                self.logs[2].append(m)
                yield sim.timeout(10)

            def b_queued(m):
                self.logs[3].append(m)
                yield sim.timeout(10)
            self.gates["bIn"].n_receives.subscribe_process(b_plain, blocking=True, queued=False)
            self.gates["bIn"].n_receives.subscribe_process(b_queued, blocking=True, queued=True)

    mods = (Mod(), Mod())
    a_ok = True
    for i in range(3):                       # test_gate_listener_method: callbacks see every message at once
        for m in mods:
            m.gates["aIn"].send("msg%d" % i)
            a_ok = a_ok and m.logs[0] == ["msg%d" % n for n in range(i + 1)]

    def main():                              # test_gate_listener_generator
        for i in range(3):
            for m in mods:
                m.gates["bIn"].send("msg%d" % i)
                yield sim.timeout(1)
    sim.process(main())
    sim.run(40)
    return {"callbacks_saw_every_message": a_ok,
            "non_queued": [m.logs[2] for m in mods], "queued": [m.logs[3] for m in mods]}


def scenario_notifier_admission():
    """tests/test_simtools.py:60-120 -- instance counts / last values of a
    non-blocking, a blocking and a blocking+queued subscriber."""
    sim = Sim()
    n = Notifier(sim)

    def make(length):
        st = {"count": 0, "value": None}

        def proc(value):
            st["count"] += 1
            st["value"] = value
            yield sim.timeout(length)
            st["count"] -= 1
        return proc, st

    (f1, s1), (f2, s2), (f3, s3) = make(10), make(10), make(10)
    n.subscribe_process(f1, blocking=False)
    n.subscribe_process(f2, blocking=True, queued=False)
    n.subscribe_process(f3, blocking=True, queued=True)

    def main():
        for i in range(1, 3):
            n.trigger("msg" + str(i))
            yield sim.timeout(1)
    sim.process(main())
    snap = lambda: [(s["count"], s["value"]) for s in (s1, s2, s3)]
    out = []
    sim.run(4); out.append(snap())
    sim.run(11); out.append(snap())
    n.trigger("msg3")
    sim.run(1); out.append(snap())
    sim.run(25); out.append(snap())
    return out


# --------------------------------------------------------------------------
# The reference's benchmark scenario (tests/test_benchmark.py:20-91), restated:
# a grid of PHY-only devices that each send one 39-byte packet every 10 ms at
# 40 dBm from a random phase, uncoordinated -- the one place the reference
# exercises concurrent transmissions (a non-trivial interference sum).
# --------------------------------------------------------------------------
GRID_SEND_INTERVAL = 1e-2            # tests/test_benchmark.py:17
GRID_MOVE_INTERVAL = 1e-3            # :18
GRID_TX_POWER_DBM = 40.0             # :47
GRID_MESSAGE = "A message to all my homies"     # :44 (26 bytes)


def grid_positions(n):
    """tests/test_benchmark.py:64-68: cols = int(sqrt(n)); device i at (i / cols, i % cols)."""
    cols = int(sqrt(n)) if n > 0 else 1
    return [(i / cols, float(i % cols)) for i in range(n)]


class GridDevice(Device):
    """SendingDevice (tests/test_benchmark.py:20-50): a PHY and a sender process, no MAC."""

    def __init__(self, world, index, x, y, send_interval, initial_delay):
        Device.__init__(self, world.sim, "Device%d" % index, x, y)
        self.index = index
        self.phy = Phy(world.sim, self, world.band)
        self.mcs = BpskMcs()
        self.n_sent = 0
        sim = world.sim

        def sender():
            yield sim.timeout(initial_delay)
            while True:
                yield sim.timeout(send_interval)
                packet = Pkt(MacHeader(bytes([0] * 5 + [index % 255]), bytes([255] * 6), 0), Blob(GRID_MESSAGE))
                cmd = Cmd(sim, "SEND", packet=packet, power=GRID_TX_POWER_DBM, mcs=self.mcs)
                self.n_sent += 1
                self.phy.mac_in.trigger(cmd)
        sim.process(sender())


def scenario_grid(n, initial_delays, sim_time, moves=None, positions=None):
    """Run the grid for `sim_time` seconds.  `initial_delays[i]` replaces random.uniform(0, SEND_INTERVAL)
    (tests/test_benchmark.py:67); `moves` (optional) is a list of (time, device, x, y) position updates
    standing in for the mover processes (:73-85).  Returns per-device counters and final state."""
    w = World()
    pos = positions or grid_positions(n)
    devs = [GridDevice(w, i, pos[i][0], pos[i][1], GRID_SEND_INTERVAL, initial_delays[i]) for i in range(n)]
    if moves:
        def mover():
            last = 0.0
            for (t, i, x, y) in moves:
                if t > last:
                    yield w.sim.timeout(t - last)
                    last = t
                devs[i].position.set(x, y)
        w.sim.process(mover())
    if n:
        w.sim.run(sim_time)
    out = {"n_sent": [d.n_sent for d in devs], "n_tx": len(w.band.log),
           "tx_start": [t.start for t in w.band.log], "tx_sender": [t.sender.index for t in w.band.log],
           "hdr_ok": [], "hdr_fail": [], "pay_ok": [], "pay_fail": [], "rx_power": [d.phy.rx_power for d in devs],
           "now": w.sim.now, "events": w.sim.n_popped}
    for d in devs:
        dec = d.phy.decisions
        out["hdr_ok"].append(sum(1 for k in dec if k[0] == "hdr" and k[1]))
        out["hdr_fail"].append(sum(1 for k in dec if k[0] == "hdr" and not k[1]))
        out["pay_ok"].append(sum(1 for k in dec if k[0] == "pay" and k[1]))
        out["pay_fail"].append(sum(1 for k in dec if k[0] == "pay" and not k[1]))
    return out


# ---- mobile variant (tests/test_benchmark.py:73-85) --------------------------------------------------
# The fixture draws from Python's global Mersenne Twister; to let a GPU replica take the same walk, the
# offsets come from a counter-based generator both sides can evaluate: splitmix64(seed, replica, device, k).
_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def grid_uniform(seed, replica, device, k, which):
    """U[0,1) with 53 random bits for draw `which` (0: first-move delay, 1: x offset, 2: y offset) of move k."""
    h = splitmix64((seed * 0x100000001B3 + replica) & _M64)
    h = splitmix64(h ^ ((device * 0x9E3779B1 + k) & _M64))
    h = splitmix64(h ^ which)
    return (h >> 11) * (1.0 / 9007199254740992.0)


def scenario_mobile_grid(n, initial_delays, sim_time, seed=0, replica=0, runs=None):
    """device_grid + mobile_device_grid: every device also runs a mover process that, from a random phase
    in [0, MOVE_INTERVAL), shifts its position by uniform(-.2, .2) in x and y every MOVE_INTERVAL
    (a random walk: `initialPos` aliases the live position object, tests/test_benchmark.py:76-82)."""
    w = World()
    pos = grid_positions(n)
    devs = [GridDevice(w, i, pos[i][0], pos[i][1], GRID_SEND_INTERVAL, initial_delays[i]) for i in range(n)]
    sim = w.sim

    def mover(dv):
        yield sim.timeout(grid_uniform(seed, replica, dv.index, 0, 0) * GRID_MOVE_INTERVAL)   # uniform(0, MOVE_INTERVAL)
        k = 0
        while True:
            xo = -.2 + (.2 - -.2) * grid_uniform(seed, replica, dv.index, k, 1)              # random.uniform(-.2, .2)
            yo = -.2 + (.2 - -.2) * grid_uniform(seed, replica, dv.index, k, 2)
            dv.position.set(dv.position.x + xo, dv.position.y + yo)
            k += 1
            yield sim.timeout(GRID_MOVE_INTERVAL)
    for dv in devs:
        sim.process(mover(dv))
    for t in (runs or [sim_time]):
        sim.run(t)
    out = {"n_sent": [d.n_sent for d in devs], "n_tx": len(w.band.log), "now": sim.now,
           "rx_power": [d.phy.rx_power for d in devs], "pos": [(d.position.x, d.position.y) for d in devs]}
    for key, what, ok in (("hdr_ok", "hdr", True), ("hdr_fail", "hdr", False), ("pay_ok", "pay", True), ("pay_fail", "pay", False)):
        out[key] = [sum(1 for k in d.phy.decisions if k[0] == what and k[1] is ok) for d in devs]
    return out
