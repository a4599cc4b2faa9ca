"""Observation / action spaces.  gymnasium's own classes are used when gymnasium is importable; the image
this was built in has no gymnasium, so a minimal equivalent with the same attribute names is provided
(VectorEnv exposes single_*_space and batched *_space)."""
import numpy as np

try:  # pragma: no cover - gymnasium absent in the build image
    from gymnasium.spaces import Box, Discrete, MultiDiscrete  # type: ignore
    from gymnasium.vector.utils import batch_space  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:
    HAVE_GYMNASIUM = False

    class Space(object):
        shape = ()
        dtype = None

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        @property
        def rng(self):
            if not hasattr(self, "_rng"):
                self._rng = np.random.default_rng()
            return self._rng

    class Discrete(Space):
        def __init__(self, n, start=0):
            self.n, self.start = int(n), int(start)
            self.shape, self.dtype = (), np.int64

        def sample(self):
            return int(self.rng.integers(self.n)) + self.start

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class MultiDiscrete(Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape, self.dtype = self.nvec.shape, np.int64

        def sample(self):
            return (self.rng.random(self.nvec.shape) * self.nvec).astype(np.int64)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.nvec.shape and bool(np.all((x >= 0) & (x < self.nvec)))

        def __repr__(self):
            return "MultiDiscrete(%s)" % (self.nvec.tolist(),)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(self.dtype)
            self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(self.dtype)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return (lo + (hi - lo) * self.rng.random(self.shape)).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x >= self.low) & (x <= self.high)))

        def __repr__(self):
            return "Box(%s, %s)" % (self.shape, self.dtype)

    def batch_space(space, n):
        if isinstance(space, Discrete):
            return MultiDiscrete(np.full((n,), space.n, dtype=np.int64))
        if isinstance(space, MultiDiscrete):
            return MultiDiscrete(np.tile(space.nvec, (n,) + (1,) * space.nvec.ndim))
        if isinstance(space, Box):
            return Box(np.tile(space.low, (n,) + (1,) * len(space.shape)),
                       np.tile(space.high, (n,) + (1,) * len(space.shape)), dtype=space.dtype)
        raise TypeError(space)
