"""
The two gym 0.12 space types the reference's action/observation spaces use
(gymwipe/envs/core.py:39-42, gymwipe/envs/counter_traffic.py:120), restated so
that the env surface works without the (absent) gym package: ``contains``,
``sample``, ``n`` / ``spaces`` and equality.
"""
from collections import OrderedDict

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape, self.dtype = shape, dtype
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def __contains__(self, x):
        return self.contains(x)


class Discrete(Space):
    """{0, 1, ..., n-1}"""

    def __init__(self, n):
        assert n >= 0
        self.n = int(n)
        Space.__init__(self, (), np.int64)

    def sample(self):
        return int(self.np_random.randint(self.n))

    def contains(self, x):
        if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
            as_int = int(x)
        elif isinstance(x, np.ndarray) and x.dtype.kind in "iu" and x.shape == ():
            as_int = int(x)
        else:
            return False
        return 0 <= as_int < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


class Dict(Space):
    """Dictionary of simpler spaces (keys kept sorted, as gym does for plain dicts)."""

    def __init__(self, spaces):
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(sorted(spaces.items()))
        self.spaces = OrderedDict(spaces)
        Space.__init__(self, None, None)

    def seed(self, seed=None):
        return [s.seed(seed) for s in self.spaces.values()]

    def sample(self):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def contains(self, x):
        if not isinstance(x, dict) or len(x) != len(self.spaces):
            return False
        for k, space in self.spaces.items():
            if k not in x or not space.contains(x[k]):
                return False
        return True

    def __repr__(self):
        return "Dict(" + ", ".join("%s:%r" % kv for kv in self.spaces.items()) + ")"

    def __eq__(self, other):
        return isinstance(other, Dict) and self.spaces == other.spaces
