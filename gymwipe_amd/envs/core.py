"""
Host-side mirror of gymwipe/envs/core.py: the old-Gym ``Env`` contract
(``reset() -> obs``, ``step(action) -> (obs, reward, done, info)``, ``seed``,
``render``, ``action_space`` / ``observation_space``) and the ``Interpreter`` plug-in
interface.  Nothing here computes: the step itself runs in HIP behind the C-ABI.
"""
from abc import ABC, abstractmethod

import numpy as np

from .. import spaces


class BaseEnv:
    """Mirror of gymwipe.envs.core.BaseEnv (envs/core.py:14-57)."""
    metadata = {'render.modes': ['human']}

    MAX_ASSIGN_DURATION = 20            # * ASSIGNMENT_DURATION_FACTOR time slots (envs/core.py:25)
    ASSIGNMENT_DURATION_FACTOR = 1000   # envs/core.py:27

    reward_range = (-float('inf'), float('inf'))
    spec = None

    def __init__(self, deviceCount):
        self.deviceCount = deviceCount
        self.action_space = spaces.Dict({
            "device": spaces.Discrete(deviceCount),
            "duration": spaces.Discrete(self.MAX_ASSIGN_DURATION),
        })
        self.seed()

    def seed(self, seed=None):
        """envs/core.py:46-52 -- the env is deterministic; the generator is never consumed."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (2 ** 31))
        self.np_random = np.random.RandomState(seed % (2 ** 32))
        return [seed]

    def render(self, mode='human', close=False):
        pass

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Interpreter(ABC):
    """Mirror of gymwipe.envs.core.Interpreter (envs/core.py:59-159): observes the packets the
    RRM sniffs and turns them into observations and rewards.  In this build the
    CounterTraffic interpreter is fused into the step kernel; the class keeps the
    reference's method names so code written against it still reads the same."""

    @abstractmethod
    def onPacketReceived(self, senderIndex, receiverIndex, payload):
        ...

    def onFrequencyBandAssignment(self, deviceIndex, duration):
        pass

    @abstractmethod
    def getReward(self):
        ...

    @abstractmethod
    def getObservation(self):
        ...

    def getDone(self):
        return False

    def getInfo(self):
        return {}

    def getFeedback(self):
        return self.getObservation(), self.getReward(), self.getDone(), self.getInfo()

    def reset(self):
        pass


class VecPayload:
    """What a vectorised ``onPacketReceived`` gets in place of the reference's ``Transmittable`` payload:
    ``value`` (the payload value, 2 for every CounterTraffic packet -- swapped constructor arguments,
    counter_traffic.py:57) and ``count`` (int32[N]: how many packets of the assigned sender the RRM
    decoded in this step; 0 where none arrived)."""

    def __init__(self, value, count):
        self.value, self.count = value, count


class VecInterpreter(Interpreter):
    """The plug-in point of the reference (envs/core.py:59-159) for N envs at once: same method names,
    tensors instead of scalars.  Install with ``VecCounterTrafficEnv(..., interpreter=obj)``; per step the
    env calls ``onFrequencyBandAssignment(duration[N], deviceIndex[N])`` (argument order as the reference's
    RRM device calls it, networking/devices.py:200), then ``onPacketReceived(senderIndex[N],
    receiverIndex[N], VecPayload)`` and returns ``getFeedback()`` instead of the fused kernel's outputs."""
