"""
gymwipe_amd -- MI355X-native vectorised ``env.step()`` for the Gym-WiPE
frequency-band-assignment environments (reference: Gryph66/gymwipe).

Only one path of the reference is rebuilt here: the discrete-event advance behind
``CounterTrafficEnv.step()``.  The compute lives in hand-written HIP kernels for
gfx950 behind a plain C-ABI (``include/gymwipe_amd.h``,
``gymwipe_amd/lib/libgymwipe_amd.so``); this package is the thin Python host side that
keeps the reference's ``gym.Env`` / ``Interpreter`` surface.  There is no CPU
fallback: without the HIP library or a GPU the envs raise.
"""
from . import spaces                                   # noqa: F401
from .plants import VecLinearPlant                       # noqa: F401
from .grid import VecPhyGrid                            # noqa: F401
from .envs import (CounterTrafficEnv, StepOutputs, VecCounterTrafficEnv, InvertedPendulumEnv, VecInvertedPendulumEnv, VecControlLoopEnv,   # noqa: F401
                   Interpreter,
                   VecInterpreter, VecPayload,
                   make, register, registry)

__version__ = "0.1.0"
