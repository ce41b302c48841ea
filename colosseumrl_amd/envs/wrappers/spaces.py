"""Observation / action space descriptors for the RL-library wrappers.

The reference builds ``gym.spaces`` objects (envs/tron/TronRllibEnvironment.py:14-27, envs/tron/rllib.py:23-28).  gym is
optional here: when it is importable the real classes are used, otherwise small records with the same constructor
arguments and attributes (``n``; ``low / high / shape``; ``spaces``) stand in, so that code which only inspects the spaces
keeps working and nothing on the stepping path depends on gym."""
try:                                                      # pragma: no cover - depends on the installation
    from gym.spaces import Box, Dict, Discrete, Space
    HAVE_GYM = True
except ImportError:
    HAVE_GYM = False

    class Space:
        """Base of the stand-in descriptors."""

    class Discrete(Space):
        def __init__(self, n):
            self.n = int(n)

        def contains(self, x):
            return isinstance(x, (int,)) and 0 <= x < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape) if shape is not None else None, dtype

        def __repr__(self):
            return "Box(%r, %r, %r)" % (self.low, self.high, self.shape)

    class Dict(Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, key):
            return self.spaces[key]

        def __repr__(self):
            return "Dict(%s)" % ", ".join("%s: %r" % kv for kv in self.spaces.items())
