"""
Host-side mirror of gymwipe/envs/inverted_pendulum.py for N environments on one MI355X.

The reference env assigns the frequency band to an angle sensor (device index 0) and a PID controller
(index 1) of a sliding pendulum and reads observation and reward straight off the plant
(``InvertedPendulumInterpreter``, envs/inverted_pendulum.py:27-57):

    observation = int(degrees(angle))        reward = float(abs(180 - degrees(angle)))       done = False

What this class keeps of it, and what it cannot:

* the gym surface (action space of two devices x 20 durations, ``observation_space = Discrete(180)``,
  ``step`` = assign, run to the end of the assignment, read the plant) and the interpreter's formulas, evaluated on
  the GPU (``gw_plant_feedback``);
* the network of the env AS SHIPPED: the sensor samples every millisecond (sliding_pendulum.py:116-135), nobody sets
  ``receiving``, so the controller's angle stays 0, its control loop never sends (control/inverted_pendulum.py:52-69)
  and the loop is open -- a sender with multiplicity 1 and a silent one (multiplicity 0) on the CounterTraffic step
  kernel, controller at (0, -1), RRM at (0, 1), sensor at the wagon's start (0, 0) (envs/inverted_pendulum.py:75-93);
* the plant is BUILDER-DEFINED: the reference integrates an ODE rigid-body world (py3ode, absent) inside an env that
  cannot even be constructed (simtools.py:39-42); here ``VecLinearPlant`` advances a linear model ``x <- A x + B u`` to the
  env clock on the f64 matrix cores.  Parity with the reference is therefore unpinned for this env; the sensor's packet
  sizes follow the counter-traffic rule (25 + counter bytes), not the reference's ``Transmittable(2, angle)``, whose
  byte size is the angle itself.
"""
import ctypes as C

from .. import _native as nat
from .. import spaces
from ..plants import VecLinearPlant
from .core import BaseEnv
from .counter_traffic import VecCounterTrafficEnv


class VecInvertedPendulumEnv(BaseEnv):
    SENSOR, CONTROLLER = 0, 1                              # deviceIndexToMacDict, envs/inverted_pendulum.py:86-89
    SAMPLE_INTERVAL = 0.001                                # :79

    _scalar_api = False

    def __init__(self, num_envs, device="cuda:0", plant=None):
        import torch
        self._torch = torch
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        BaseEnv.__init__(self, 2)                          # deviceCount=2, :70
        self.observation_space = spaces.Discrete(180)      # :73
        self.network = VecCounterTrafficEnv(self.num_envs, 2, device=self.device,
                                            positions=[(0.0, 0.0), (0.0, -1.0)], rrm_position=(0.0, 1.0),
                                            multiplicity=[1, 0], dest=[1, 0])
        self.plant = plant if plant is not None else VecLinearPlant(self.num_envs, device=self.device)
        base, stride = C.c_void_p(), C.c_int64()
        nat.check(self.network._L.gw_now_ptr(self.network._h, C.byref(base), C.byref(stride)))
        self._now = (base.value, stride.value)             # the env clock, read by the plant kernel in place
        n = self.num_envs
        self._obs = torch.empty(n, dtype=torch.int32, device=self.device)
        self._rew = torch.empty(n, dtype=torch.float32, device=self.device)
        self._angle = torch.empty(n, dtype=torch.float64, device=self.device)
        self._done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._out_ptrs = (self._obs.data_ptr(), self._rew.data_ptr(), self._angle.data_ptr())   # fixed output buffers
        self._fb = (self._obs, self._rew, self._done, {"Sensor angle": self._angle})

    def _feedback(self):
        torch = self._torch
        with torch.cuda.device(self.device):
            nat.check(self.plant._L.gw_plant_feedback(self.plant._h, self._obs.data_ptr(), self._rew.data_ptr(),
                                                      self._angle.data_ptr(),
                                                      torch.cuda.current_stream(self.device).cuda_stream))
        return self._obs, self._rew, self._done, {"Sensor angle": self._angle}

    def reset(self):
        """envs/inverted_pendulum.py:95-99: returns an observation, resets nothing."""
        return self._feedback()[0]

    def step(self, action, fused=True):
        """:101-113 for every env: assign the band, run to the end of the assignment, read the plant -- ONE kernel launch
        (``gw_pendulum_step``: band-assignment step + plant advance on the matrix cores + interpreter feedback).
        ``fused=False`` issues the same work as two launches (``gw_step`` + ``gw_plant_update_feedback``); results are identical."""
        torch = self._torch
        net = self.network
        if not fused:
            net.step(action)
            with torch.cuda.device(self.device):           # OdePlant.updateState to the new env clock + the interpreter's
                nat.check(self.plant._L.gw_plant_update_feedback(    # reading of the plant
                    self.plant._h, self._now[0], self._now[1], self._obs.data_ptr(), self._rew.data_ptr(),
                    self._angle.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
            return self._obs, self._rew, self._done, {"Sensor angle": self._angle}
        dev = net._checked(action["device"], "device")
        dur = net._checked(action["duration"], "duration")
        idx = net._dev_index
        if torch._C._cuda_getDevice() == idx:                  # the one-process-per-GPU case: no context switch
            fast = net._fast
            if fast is not None:
                rc = fast.pendulum_step(net._h.value or 0, self.plant._h.value or 0, dev.data_ptr(), dur.data_ptr(),
                                        self._out_ptrs[0], self._out_ptrs[1], self._out_ptrs[2],
                                        torch._C._cuda_getCurrentRawStream(idx))
            else:
                rc = net._L.gw_pendulum_step(net._h, self.plant._h, dev.data_ptr(), dur.data_ptr(), self._out_ptrs[0],
                                             self._out_ptrs[1], self._out_ptrs[2], torch._C._cuda_getCurrentRawStream(idx))
        else:
            with torch.cuda.device(self.device):
                rc = net._L.gw_pendulum_step(net._h, self.plant._h, dev.data_ptr(), dur.data_ptr(), self._out_ptrs[0],
                                             self._out_ptrs[1], self._out_ptrs[2],
                                             torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            nat.check(rc)
        return self._fb

    def render(self, mode="human", close=False):           # :115-116
        pass

    def close(self):
        self.network.close()
        self.plant.close()


class InvertedPendulumEnv(VecInvertedPendulumEnv):
    """N = 1 with the reference's scalar surface: ``step({"device": int, "duration": int}) ->
    (int, float, bool, {"Sensor angle": float})``, ``AssertionError`` on an action outside the action space."""
    _scalar_api = True

    def __init__(self, device="cuda:0"):
        VecInvertedPendulumEnv.__init__(self, 1, device=device)
        self._act = self._torch.zeros((2, 1), dtype=self._torch.int32, device=self.device)

    def _scalars(self, fb):
        obs, rew, done, info = fb
        return int(obs.item()), float(rew.item()), bool(done.item()), {"Sensor angle": float(info["Sensor angle"].item())}

    def reset(self):
        return self._scalars(self._feedback())[0]

    def step(self, action):
        assert self.action_space.contains(action)          # :102
        self._act[0, 0] = int(action["device"])
        self._act[1, 0] = int(action["duration"])
        return self._scalars(VecInvertedPendulumEnv.step(self, {"device": self._act[0], "duration": self._act[1]}))


class VecControlLoopEnv(BaseEnv):
    """The pendulum env with its control loop CLOSED over receive-mode MACs: what the reference
    intends (sensor -> controller -> actuator, plants/sliding_pendulum.py:116-155, control/inverted_pendulum.py:16-69)
    but never runs.  Builder-defined where the reference leaves it open -- see ``gw_ctrl_config`` in
    include/gymwipe_amd.h for the rules -- and checked bit for bit against an event-driven model of those rules.
    Same gym surface as ``VecInvertedPendulumEnv``: ``step({"device": int32[N] in {0 sensor, 1 controller},
    "duration": int32[N]}) -> (obs, reward, done, {"Sensor angle": deg})``."""
    SENSOR, CONTROLLER, ACTUATOR = 0, 1, 2

    def __init__(self, num_envs, device="cuda:0", ctrl_start_tick=None, ctrl_period_ticks=None, positions=None, x0=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("gymwipe_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self._torch = torch
        self._L = nat.lib()
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        BaseEnv.__init__(self, 2)
        self.observation_space = spaces.Discrete(180)
        cfg = nat.CtrlConfig()
        nat.check(self._L.gw_ctrl_config_default(C.byref(cfg), self.num_envs))
        cfg.net.hip_device = self.device.index or 0
        if ctrl_start_tick is not None:
            cfg.ctrl_start_tick = int(ctrl_start_tick)
        if ctrl_period_ticks is not None:
            cfg.ctrl_period_ticks = int(ctrl_period_ticks)
        if positions is not None:                          # sensor, controller, actuator, RRM
            for i, (x, y) in enumerate(positions):
                cfg.net.pos[i][0], cfg.net.pos[i][1] = float(x), float(y)
        if x0 is not None:
            for i, v in enumerate(x0):
                cfg.x0[i] = float(v)
        self.config = cfg
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_ctrl_create(C.byref(cfg), C.byref(self._h)))
            n = self.num_envs
            self._obs = torch.empty(n, dtype=torch.int32, device=self.device)
            self._rew = torch.empty(n, dtype=torch.float32, device=self.device)
            self._angle = torch.empty(n, dtype=torch.float64, device=self.device)
            self._done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._stepped = False

    def reset(self):
        """envs/inverted_pendulum.py:95-99: returns an observation, resets nothing."""
        import math
        if not self._stepped:                              # before the first step: the initial plant state
            self._obs.fill_(int(math.degrees(self.config.x0[2])))
        return self._obs

    def step(self, action):
        torch = self._torch
        self._stepped = True
        dev = action["device"].to(device=self.device, dtype=torch.int32).contiguous()
        dur = action["duration"].to(device=self.device, dtype=torch.int32).contiguous()
        assert dev.shape == (self.num_envs,) and dur.shape == (self.num_envs,)
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_ctrl_step(self._h, dev.data_ptr(), dur.data_ptr(), self._obs.data_ptr(), self._rew.data_ptr(),
                                           self._angle.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        return self._obs, self._rew, self._done, {"Sensor angle": self._angle}

    _FIELDS = {"now": ((), "f8"), "wake": ((), "f8"), "u": ((), "f8"), "angle_deg": ((), "f8"), "x": ((4,), "f8"),
               "rx_power": ((4,), "f8"), "qlen": ((2,), "i4"), "received": ((2,), "u4"), "n_tx": ((), "u4"),
               "commands": ((), "u4"), "substeps": ((), "u4"), "flags": ((), "u4")}

    def get_state(self, field):
        import numpy as np
        shape, dt = self._FIELDS[field]
        out = np.empty((self.num_envs,) + shape, np.dtype(dt))
        nat.check(self._L.gw_ctrl_get_state(self._h, field.encode(), out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.gw_ctrl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
