"""
Host-side mirror of gymwipe/envs/counter_traffic.py on top of the HIP C-ABI.

``VecCounterTrafficEnv``  N independent CounterTraffic environments resident in HBM on one
                          GPU, advanced together by one kernel launch per ``step``.
``CounterTrafficEnv``     the N = 1 drop-in with the reference's exact Python surface:
                          ``step({"device": int, "duration": int}) -> (int, float, bool, dict)``.

Reference behaviour kept on purpose:
  * ``reset()`` zeroes the counters and the interpreter only; simulated time, MAC queues
    and radio state persist (counter_traffic.py:135-144)
  * a fresh env starts with counters at 1, a reset env at 0 (:48 vs :140)
  * data payloads carry ``value == 2`` and ``byteSize == counter`` (swapped constructor
    arguments, :57), so observations are 65534 / 65536 / 65538 and ``done`` never fires
"""
import ctypes as C
import weakref

import numpy as np

from .. import _native as nat
from .. import spaces
from .core import BaseEnv, Interpreter, VecInterpreter, VecPayload


def _torch():
    import torch
    return torch


class _DeviceInterpreter(Interpreter):
    """Facade over the interpreter state that lives on the GPU
    (CounterTrafficEnv.CounterTrafficInterpreter, counter_traffic.py:63-112)."""

    def __init__(self, env):
        self._env = env

    def reset(self):
        self._env.reset()

    def onPacketReceived(self, senderIndex, receiverIndex, payload):
        raise NotImplementedError("packets are interpreted inside the HIP step kernel")

    @property
    def receivedValues(self):
        rv = self._env.received()
        return rv[0].tolist() if self._env._scalar_api else rv

    def getObservation(self):
        return self._env._last[0]

    def getReward(self):
        return self._env._last[1]

    def getDone(self):
        return self._env._last[2]

    def getInfo(self):
        return self._env._info()


class StepOutputs:
    """Where one ``env.step(action, out=...)`` writes: ``obs`` int32[N], ``reward`` float32[N], ``done`` uint8[N] and, optionally,
    ``feedback_bytes`` uint8[N] (the step's feedback in the one-byte exchange format, ``gw_step_fb``) -- preallocated,
    contiguous tensors on the env's GPU.  Their device addresses are taken ONCE, here (a step is enqueued every few
    microseconds; four ``data_ptr()`` calls are 10 % of that), so the tensors must not be resized or re-pointed afterwards;
    the object keeps them alive."""
    __slots__ = ("obs", "reward", "done", "feedback_bytes", "_ptrs", "_as_tuple", "_dev")

    def __init__(self, obs, reward, done, feedback_bytes=None):
        torch = _torch()
        n = obs.shape[0]
        for t, dt in ((obs, torch.int32), (reward, torch.float32), (done, torch.uint8), (feedback_bytes, torch.uint8)):
            if t is None:
                continue
            assert (type(t) is torch.Tensor and t.dtype is dt and t.dim() == 1 and t.shape[0] == n and t.is_contiguous()
                    and t.device == obs.device and t.is_cuda), "StepOutputs: contiguous 1-D tensors of one length on one GPU"
        self.obs, self.reward, self.done, self.feedback_bytes = obs, reward, done, feedback_bytes
        self._ptrs = (obs.data_ptr(), reward.data_ptr(), done.data_ptr(),
                      feedback_bytes.data_ptr() if feedback_bytes is not None else 0)
        self._as_tuple = (obs, reward, done)
        self._dev = obs.device.index or 0                 # step() refuses outputs that live on another GPU than the env


class VecCounterTrafficEnv(BaseEnv):
    """N CounterTraffic environments on one MI355X.

    Args:
        num_envs: N.
        num_devices: D assignable senders (reference: 2).  For D > 2 the senders sit on a
            circle of radius 2 m around the RRM with multiplicities 1,3,1,3,....
        device: torch device string or index ('cuda:0').
        positions / multiplicity / dest / rrm_position: optional overrides of the layout.
        per_env_stats: explicit-queue mode only -- keep per-env event counters (the default mode always
            keeps them in its 32-byte counter record).
        explicit_queue: hold the MAC queues as explicit rings of packet sizes (generic, slower)
            instead of the default exact suffix encoding of counter traffic (gw_queue.h).
        reuse_outputs: return the same output tensors every step (fast path).
        extra_attenuation: custom attenuation models per device pair (the reference's
            AttenuationModelFactory.setCustomModels / JoinedAttenuationModel, physical.py:402-498), reduced to what they
            amount to with static geometry: ``{(a, b): dB}`` added to the free-space term of the pair (radio index
            ``num_devices`` is the RRM), or a callable ``(a, b, pos_a, pos_b) -> dB`` evaluated for every pair.
        per_env_geometry: positions per ENVIRONMENT (default queue mode): every env starts with the layout above and
            ``set_position`` / ``set_positions`` move radios between steps (the reference's ``Position.set``,
            devices/core.py:52-86); link powers are then kept per env and rebuilt on the GPU.
        counter_traffic / peer_receive / float_duration (explicit_queue only): switch the
            counter processes off so that packets come from enqueue() only; keep every sender MAC in receive
            mode (get_state("peer_received") counts what it hands up); pass assignment durations as floats
            like tests/networking/test_stack.py:197 does.
    """
    COUNTER_INTERVAL = 0.001                              # counter_traffic.py:31
    COUNTER_BYTE_LENGTH = 2                               # :33
    COUNTER_BOUND = 2 ** (8 * COUNTER_BYTE_LENGTH)        # :35

    _scalar_api = False

    def __init__(self, num_envs, num_devices=2, device="cuda:0", positions=None,
                 multiplicity=None, dest=None, rrm_position=None, per_env_stats=False,
                 reuse_outputs=True, explicit_queue=False, counter_bound=None, interpreter=None,
                 counter_traffic=True, peer_receive=False, float_duration=False, extra_attenuation=None,
                 start_time=None, per_env_geometry=False, counter_interval=None, duration_factor=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("gymwipe_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self._L = nat.lib()
        self.num_envs = int(num_envs)
        self.num_devices = int(num_devices)
        self.device = torch.device(device)
        BaseEnv.__init__(self, self.num_devices)
        self.observation_space = spaces.Discrete(2 * self.COUNTER_BOUND)   # :120

        cfg = nat.default_config(self.num_envs, self.num_devices)
        cfg.hip_device = self.device.index or 0
        D = self.num_devices
        if positions is not None:
            assert len(positions) == D
            for i, (x, y) in enumerate(positions):
                cfg.pos[i][0], cfg.pos[i][1] = float(x), float(y)
        if rrm_position is not None:
            cfg.pos[D][0], cfg.pos[D][1] = float(rrm_position[0]), float(rrm_position[1])
        if multiplicity is not None:
            assert len(multiplicity) == D
            for i, m in enumerate(multiplicity):
                cfg.mult[i] = int(m)
        if dest is not None:
            assert len(dest) == D
            for i, m in enumerate(dest):
                cfg.dest[i] = int(m)
        if extra_attenuation is not None:
            if callable(extra_attenuation):
                pos = [(cfg.pos[i][0], cfg.pos[i][1]) for i in range(D + 1)]
                extra_attenuation = {(a, b): extra_attenuation(a, b, pos[a], pos[b])
                                     for a in range(D + 1) for b in range(a + 1, D + 1)}
            for (a, b), db in extra_attenuation.items():
                cfg.extra_att_db[a][b] = cfg.extra_att_db[b][a] = float(db)
        if start_time is not None:             # test hook: simulated time at creation (the reference starts at 0)
            cfg.start_time = float(start_time)
        if per_env_stats:
            cfg.flags |= nat.CFG_PER_ENV_STATS
        if explicit_queue:
            cfg.flags |= nat.CFG_EXPLICIT_QUEUE
        if not counter_traffic:
            cfg.flags |= nat.CFG_NO_COUNTER_TRAFFIC
        if peer_receive:
            cfg.flags |= nat.CFG_PEER_RECEIVE
        if float_duration:
            cfg.flags |= nat.CFG_FLOAT_DURATION
        if per_env_geometry:
            cfg.flags |= nat.CFG_PER_ENV_GEOMETRY
        if counter_interval is not None:       # COUNTER_INTERVAL (counter_traffic.py:31)
            cfg.counter_interval = float(counter_interval)
        if duration_factor is not None:        # ASSIGNMENT_DURATION_FACTOR (envs/core.py:27)
            cfg.duration_factor = int(duration_factor)
            self.ASSIGNMENT_DURATION_FACTOR = int(duration_factor)
        if counter_bound is not None:          # tests: reach counter saturation quickly
            cfg.counter_bound = int(counter_bound)
            self.COUNTER_BOUND = int(counter_bound)
            self.observation_space = spaces.Discrete(2 * self.COUNTER_BOUND)
        self.config = cfg

        self._h = C.c_void_p()
        self._hv = 0
        self._fb = None
        self._fast = nat.fast()
        self._cuda_get_device = torch._C._cuda_getDevice
        self._cuda_raw_stream = torch._C._cuda_getCurrentRawStream
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_create(C.byref(cfg), C.byref(self._h)))
            self._hv = self._h.value or 0
            n = self.num_envs
            self._obs = torch.empty(n, dtype=torch.int32, device=self.device)
            self._rew = torch.empty(n, dtype=torch.float32, device=self.device)
            self._done = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._reuse = bool(reuse_outputs)
        self._seen = {}                                   # action tensor objects already validated (step())
        self._dev_index = self.device.index or 0
        self._last = (None, None, None)
        self._custom = interpreter
        if interpreter is not None:                       # a user-supplied Interpreter replaces the fused one
            if explicit_queue:
                raise ValueError("custom interpreters need the default queue mode")
            self._dest = torch.tensor([int(cfg.dest[i]) for i in range(D)], dtype=torch.int64, device=self.device)
            self._deliv = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            self._deliv_prev = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            self.interpreter = interpreter
        else:
            self.interpreter = _DeviceInterpreter(self)

    # -- helpers ------------------------------------------------------------------------------
    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def _as_i32(self, x, name):
        torch = _torch()
        if isinstance(x, torch.Tensor):
            t = x
        else:
            t = torch.as_tensor(np.asarray(x))
        if t.dim() == 0:
            t = t.reshape(1)
        if t.shape != (self.num_envs,):
            raise AssertionError("action[%r] must have shape (%d,), got %s"
                                 % (name, self.num_envs, tuple(t.shape)))
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    def _outputs(self):
        if self._reuse:
            return self._obs, self._rew, self._done
        torch = _torch()
        n = self.num_envs
        return (torch.empty(n, dtype=torch.int32, device=self.device),
                torch.empty(n, dtype=torch.float32, device=self.device),
                torch.empty(n, dtype=torch.uint8, device=self.device))

    def _info(self):
        return {}

    # -- gym surface ----------------------------------------------------------------------------
    def reset(self, mask=None):
        """Mirror of CounterTrafficEnv.reset (counter_traffic.py:135-144) for every env, or for
        the envs selected by ``mask`` (bool/uint8 [N])."""
        torch = _torch()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        obs = self._outputs()[0]
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_reset(self._h, m.data_ptr() if m is not None else None,
                                       obs.data_ptr(), self._stream()))
        if self._custom is not None:                      # counter_traffic.py:142-144
            self._custom.reset()
            return self._custom.getObservation()
        return obs

    def _ready(self, t):
        """int32, contiguous, right shape, on this env's GPU: usable in place."""
        torch = _torch()
        return (type(t) is torch.Tensor and t.dtype is torch.int32 and t.device == self.device
                and t.dim() == 1 and t.shape[0] == self.num_envs and t.is_contiguous())

    def _checked(self, t, name):
        """The action tensor, validated (int32, contiguous, right shape, on this env's GPU) or converted.  A tensor OBJECT
        that passed once is not re-validated (a caller steps with the same pre-staged tensors again and again; dtype, device
        and shape of a tensor object do not change under ordinary use): one dict probe instead of seven attribute tests."""
        hit = self._seen.get(id(t))
        if hit is not None and hit[0]() is t:
            return t
        if not self._ready(t):
            return self._as_i32(t, name)
        if len(self._seen) >= 8192:
            self._seen.clear()
        # weak: the env must not keep a caller's action buffers alive; the device address is taken once, with the validation
        self._seen[id(t)] = (weakref.ref(t), t.data_ptr())
        return t

    def step(self, action, out=None):
        """One env.step() for all N envs: ``action = {"device": int32[N], "duration": int32[N]}``
        (torch tensors on the env's GPU are used in place).  Returns
        ``(obs int32[N], reward float32[N], done uint8[N], info)``; an action outside the action
        space flags its env (``check()`` raises) and leaves that env untouched.
        ``out``: a ``StepOutputs`` to write this step's outputs into (instead of the env's own buffers).

        Aliasing contract of the fast path: an action tensor OBJECT that passed validation once, and the tensors inside a
        ``StepOutputs``, are taken at their word afterwards -- their device addresses, dtype, shape and device must not be
        changed behind the env's back (``set_()``, ``resize_()``, swapping ``.data``); writing new VALUES into them is what
        they are for.  A ``StepOutputs`` on another GPU than the env is refused."""
        # (this method is enqueued ~200 000 times a second: the common path -- pre-staged int32 tensors on this GPU, reused
        #  output buffers, the caller on this env's device -- is written out flat, without helper calls)
        dev = action["device"]
        dur = action["duration"]
        seen = self._seen
        hit = seen.get(id(dev))
        if hit is None or hit[0]() is not dev:
            dev = self._checked(dev, "device")
            dev_ptr = dev.data_ptr()
        else:
            dev_ptr = hit[1]
        hit = seen.get(id(dur))
        if hit is None or hit[0]() is not dur:
            dur = self._checked(dur, "duration")
            dur_ptr = dur.data_ptr()
        else:
            dur_ptr = hit[1]
        idx = self._dev_index
        if out is not None and out._dev != idx:
            raise ValueError("StepOutputs on cuda:%d passed to an env on cuda:%d" % (out._dev, idx))
        if out is not None and self._cuda_get_device() == idx and self._fast is not None and self._custom is None:
            p = out._ptrs                                       # preallocated outputs, addresses taken at construction
            if p[3]:
                rc = self._fast.step_fb(self._hv, dev_ptr, dur_ptr, p[0], p[1], p[2], p[3], self._cuda_raw_stream(idx))
            else:
                rc = self._fast.step(self._hv, dev_ptr, dur_ptr, p[0], p[1], p[2], self._cuda_raw_stream(idx))
            if rc:
                nat.check(rc)
            self._last = out._as_tuple
            return out.obs, out.reward, out.done, self._info()
        if out is not None:                                     # (no shim / another device current / custom interpreter: the general path)
            obs, rew, done, fb = out.obs, out.reward, out.done, out.feedback_bytes
        else:
            fb = self._fb                                       # feedback_bytes_into(): the step's one-byte feedback row
            if self._reuse:
                obs, rew, done = self._obs, self._rew, self._done
            else:
                obs, rew, done = self._outputs()
        if self._cuda_get_device() == idx:                     # the one-process-per-GPU case: no context switch
            fast = self._fast                                   # CPython fast-call shim (csrc/gw_pyfast.c) when built
            if fb is not None:
                rc = (fast.step_fb if fast is not None else self._L.gw_step_fb)(
                    self._hv, dev_ptr, dur_ptr, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), fb.data_ptr(),
                    self._cuda_raw_stream(idx))
            elif fast is not None:
                rc = fast.step(self._hv, dev_ptr, dur_ptr, obs.data_ptr(), rew.data_ptr(),
                               done.data_ptr(), self._cuda_raw_stream(idx))
            else:
                rc = self._L.gw_step(self._h, dev_ptr, dur_ptr, obs.data_ptr(), rew.data_ptr(),
                                     done.data_ptr(), self._cuda_raw_stream(idx))
        else:
            torch = _torch()
            with torch.cuda.device(self.device):
                rc = self._L.gw_step_fb(self._h, dev_ptr, dur_ptr, obs.data_ptr(), rew.data_ptr(),
                                        done.data_ptr(), fb.data_ptr() if fb is not None else None, self._stream())
        if rc:
            nat.check(rc)
        if self._custom is not None:
            return self._feed_custom(dev, dur)
        self._last = (obs, rew, done)
        return obs, rew, done, self._info()

    def feedback_bytes_into(self, row):
        """From now on every step() also writes its feedback in the one-byte exchange format of ``pack_feedback`` into
        ``row`` (uint8[N] on this env's GPU; ``None`` switches it off) -- the row a multi-GPU job gathers
        (``sharding.ChunkedFeedbackGather``).  In the default mode the step kernel stores the byte itself: no packing launch."""
        if row is not None:
            torch = _torch()
            assert (type(row) is torch.Tensor and row.dtype is torch.uint8 and row.device == self.device and row.dim() == 1
                    and row.shape[0] == self.num_envs and row.is_contiguous()), "feedback row: contiguous uint8[N] on the env's GPU"
        self._fb = row

    def _feed_custom(self, dev, dur):
        """Drive a user-supplied VecInterpreter from what the RRM sniffed in this step."""
        torch = _torch()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_delivered(self._h, self._deliv.data_ptr(), self._stream()))
        count = self._deliv - self._deliv_prev
        self._deliv_prev.copy_(self._deliv)
        it = self._custom
        it.onFrequencyBandAssignment(dur * self.ASSIGNMENT_DURATION_FACTOR, dev)    # swapped, networking/devices.py:200
        it.onPacketReceived(dev, self._dest[dev.long()].to(torch.int32), VecPayload(self.config.payload_value, count))
        self._last = it.getFeedback()
        return self._last

    def rollout(self, device, duration, out=None):
        """K consecutive steps from pre-staged actions ``int32[K][N]``; one launch per step, no
        Python in between.  Returns ``(obs[K][N], reward[K][N], done[K][N])``."""
        torch = _torch()
        dev = torch.as_tensor(device).to(device=self.device, dtype=torch.int32).contiguous()
        dur = torch.as_tensor(duration).to(device=self.device, dtype=torch.int32).contiguous()
        K = dev.shape[0]
        assert dev.shape == (K, self.num_envs) and dur.shape == dev.shape
        if out is None:
            out = (torch.empty((K, self.num_envs), dtype=torch.int32, device=self.device),
                   torch.empty((K, self.num_envs), dtype=torch.float32, device=self.device),
                   torch.empty((K, self.num_envs), dtype=torch.uint8, device=self.device))
        obs, rew, done = out
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_rollout(self._h, K, dev.data_ptr(), dur.data_ptr(), obs.data_ptr(),
                                         rew.data_ptr(), done.data_ptr(), self._stream()))
        if K:
            self._last = (obs[-1], rew[-1], done[-1])
        return obs, rew, done

    def render(self, mode='human', close=False):          # counter_traffic.py:160-162
        values = self.received()[0].tolist()
        print("Last Received: {}, difference: {:6d}".format(values, values[1] - values[0]), end='\r')

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.gw_destroy(self._h)
            self._h = C.c_void_p()
            self._hv = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state access ---------------------------------------------------------------------------
    def received(self):
        """interpreter.receivedValues of every env: int32[N][D] tensor on the GPU."""
        torch = _torch()
        out = torch.empty((self.num_envs, self.num_devices), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_received(self._h, out.data_ptr(), self._stream()))
        return out

    def pack_feedback(self, obs, reward, done, out=None, check=False):
        """(obs int32, reward float32, done uint8) tensors of any common shape -> one byte per element
        (bits 0-1 sign(obs - COUNTER_BOUND) + 1, bits 2-6 reward + 10, bit 7 done).  Lossless for the built-in
        interpreter; the unit the multi-GPU observation gather moves (sharding.ChunkedFeedbackGather)."""
        torch = _torch()
        assert obs.is_contiguous() and reward.is_contiguous() and done.is_contiguous()
        assert obs.dtype == torch.int32 and reward.dtype == torch.float32 and done.dtype == torch.uint8
        if out is None:
            out = torch.empty(obs.shape, dtype=torch.uint8, device=self.device)
        assert out.is_contiguous() and out.numel() == obs.numel() == reward.numel() == done.numel()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_pack_feedback(self._h, obs.numel(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                               out.data_ptr(), 1 if check else 0, self._stream()))
        return out

    def unpack_feedback(self, packed, obs=None, reward=None, done=None):
        torch = _torch()
        assert packed.is_contiguous() and packed.dtype == torch.uint8
        obs = torch.empty(packed.shape, dtype=torch.int32, device=self.device) if obs is None else obs
        reward = torch.empty(packed.shape, dtype=torch.float32, device=self.device) if reward is None else reward
        done = torch.empty(packed.shape, dtype=torch.uint8, device=self.device) if done is None else done
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_unpack_feedback(self._h, packed.numel(), packed.data_ptr(), obs.data_ptr(),
                                                 reward.data_ptr(), done.data_ptr(), self._stream()))
        return obs, reward, done

    def set_position(self, radio, x, y, mask=None):
        """``device.position.set(x, y)`` on radio ``radio`` (0..D-1 senders, D = the RRM) of every env (or of the envs
        selected by ``mask``), between two steps (devices/core.py:77-86): float64[N] tensors/arrays or scalars."""
        torch = _torch()
        xs = torch.as_tensor(x, dtype=torch.float64, device=self.device).expand(self.num_envs).contiguous()
        ys = torch.as_tensor(y, dtype=torch.float64, device=self.device).expand(self.num_envs).contiguous()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_set_position(self._h, int(radio), xs.data_ptr(), ys.data_ptr(),
                                              m.data_ptr() if m is not None else None, self._stream()))

    def set_positions(self, positions, mask=None):
        """Positions of every radio of every env at once: float64[N][D+1][2] (row D = the RRM)."""
        torch = _torch()
        pos = torch.as_tensor(positions, dtype=torch.float64, device=self.device).contiguous()
        assert pos.shape == (self.num_envs, self.num_devices + 1, 2)
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_set_positions(self._h, pos.data_ptr(), m.data_ptr() if m is not None else None, self._stream()))

    def enqueue(self, device, payload_bytes):
        """SimpleNetworkDevice.send(data, dest) on sender `device` of every env (networking/devices.py:84-86):
        payload_bytes is an int or an int32[N] tensor/array; negative entries enqueue nothing."""
        torch = _torch()
        pb = torch.as_tensor(payload_bytes, dtype=torch.int32, device=self.device).expand(self.num_envs).contiguous()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_enqueue(self._h, int(device), pb.data_ptr(), self._stream()))

    _FIELDS = {
        "peer_received": (np.uint32, lambda D, R: (D,)),
        "now": (np.float64, lambda D, R: ()), "wake": (np.float64, lambda D, R: (D,)),
        "counter": (np.uint32, lambda D, R: (D,)), "qlen": (np.int32, lambda D, R: (D,)),
        "queue": (np.uint32, lambda D, R: (D, nat.QUEUE_CAP)),
        "received": (np.int32, lambda D, R: (D,)), "latest_diff": (np.int32, lambda D, R: ()),
        "last_abs": (np.int32, lambda D, R: ()), "rx_power": (np.float64, lambda D, R: (R,)),
        "pos": (np.float64, lambda D, R: (R, 2)), "link_power": (np.float64, lambda D, R: (R, R)),
        "flags": (np.uint32, lambda D, R: ()), "n_tx": (np.uint64, lambda D, R: ()),
        "n_delivered": (np.uint64, lambda D, R: ()), "n_appended": (np.uint64, lambda D, R: ()),
        "n_popped": (np.uint64, lambda D, R: ()), "n_dropped": (np.uint64, lambda D, R: ()),
    }

    def get_state(self, field):
        """Synchronous host copy of one state field (numpy), in the oracle's logical layout."""
        dtype, shp = self._FIELDS[field]
        out = np.empty((self.num_envs,) + shp(self.num_devices, self.num_devices + 1), dtype)
        nat.check(self._L.gw_get_state(self._h, field.encode(), out.ctypes.data, out.nbytes))
        return out

    def stats(self):
        s = nat.Stats()
        nat.check(self._L.gw_stats_read(self._h, C.byref(s)))
        return s.as_dict()

    def state_bytes(self):
        b = C.c_uint64()
        nat.check(self._L.gw_state_bytes(self._h, C.byref(b)))
        return int(b.value)

    def snapshot(self):
        """Checkpoint: every byte of this env's device state as one uint8 numpy array (synchronises).  ``restore()`` puts it
        back -- into this env later on, or into a fresh env created with the same arguments -- and every later step continues
        bit for bit as this one would have (gw_get_snapshot / gw_set_state)."""
        n = C.c_uint64()
        nat.check(self._L.gw_snapshot_bytes(self._h, C.byref(n)))
        out = np.empty(int(n.value), np.uint8)
        nat.check(self._L.gw_get_snapshot(self._h, out.ctypes.data, out.nbytes))
        return out

    def restore(self, snap):
        snap = np.ascontiguousarray(snap, dtype=np.uint8)
        nat.check(self._L.gw_set_state(self._h, snap.ctypes.data, snap.nbytes))
        self._seen.clear()

    def link_info(self, frm, to):
        a, p = C.c_double(), C.c_double()
        nat.check(self._L.gw_link_info(self._h, frm, to, C.byref(a), C.byref(p)))
        return a.value, p.value

    def noise_states(self, radio):
        n = C.c_int32()
        vals = (C.c_double * nat.MAX_NSTATES)()
        nat.check(self._L.gw_noise_states(self._h, radio, C.byref(n), vals))
        return [vals[i] for i in range(n.value)]

    def check(self, strict=False):
        """Raise if any env hit a condition outside the modelled horizon or got a bad action.  The returned totals carry
        ``"ties"``: whether an exact f64 time tie was resolved by the insertion-order rule (GW_FLAG_TIE) -- the kernels and
        the oracle resolve it the same way, but the reference's order there rests on SimPy's event ids; ``strict=True``
        raises on it too.  Flags are sticky: ``clear_flags()`` resets them so that a later check tells when they arose."""
        st = self.stats()
        fl = st["flags_or"]
        st["ties"] = bool(fl & nat.FLAG_TIE)
        if fl & nat.FLAG_BADACT:
            raise AssertionError("%d env-step(s) had an action outside the action space" % st["bad_actions"])
        if fl & nat.FLAG_INTERNAL:
            raise RuntimeError("a kernel self-check failed in some env (GW_FLAG_INTERNAL): state invalid, please report")
        if fl & (nat.FLAG_CARRY | nat.FLAG_REFEXC):
            raise RuntimeError("step horizon not closed in some env (flags 0x%x)" % fl)
        if strict and st["ties"]:
            raise RuntimeError("an exact time tie was resolved by the insertion-order rule in some env (GW_FLAG_TIE)")
        return st

    def clear_flags(self):
        """Zero the sticky per-env GW_FLAG_* words (event counters are untouched)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_clear_flags(self._h, self._stream()))


class CounterTrafficEnv(VecCounterTrafficEnv):
    """Drop-in for gymwipe.envs.CounterTrafficEnv (2 senders + RRM, one env): Python scalars in,
    Python scalars out, ``AssertionError`` on an action outside the action space
    (counter_traffic.py:146-158)."""
    _scalar_api = True

    def __init__(self, device="cuda:0", num_devices=2, **kw):
        VecCounterTrafficEnv.__init__(self, 1, num_devices=num_devices, device=device,
                                      reuse_outputs=True, **kw)
        torch = _torch()
        self._act = torch.zeros((2, 1), dtype=torch.int32, device=self.device)

    def _info(self):
        return {"Latest received values": str(self.received()[0].tolist())}   # :109-112

    def reset(self):
        return int(VecCounterTrafficEnv.reset(self).item())

    def step(self, action):
        assert self.action_space.contains(action)                              # :147
        self._act[0, 0] = int(action["device"])
        self._act[1, 0] = int(action["duration"])
        obs, rew, done, _ = VecCounterTrafficEnv.step(self, {"device": self._act[0], "duration": self._act[1]})
        self._last = (int(obs.item()), float(rew.item()), bool(done.item()))
        return self._last + (self._info(),)
