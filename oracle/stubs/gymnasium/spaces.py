import numpy


class Space(object):
    pass


class Discrete(Space):
    def __init__(self, n, start=0):
        self.n = int(n)
        self.start = int(start)

    def sample(self):
        return int(numpy.random.randint(self.n)) + self.start


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = numpy.asarray(nvec, dtype=numpy.int64)

    def sample(self):
        return numpy.array([numpy.random.randint(n) for n in self.nvec])


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=float):
        self.shape = tuple(shape) if shape is not None else numpy.shape(low)
        self.dtype = dtype
        self.low = numpy.broadcast_to(numpy.asarray(low, dtype=float), self.shape).astype(dtype)
        self.high = numpy.broadcast_to(numpy.asarray(high, dtype=float), self.shape).astype(dtype)

    def sample(self):
        lo = numpy.where(numpy.isfinite(self.low), self.low, -1.0)
        hi = numpy.where(numpy.isfinite(self.high), self.high, 1.0)
        return numpy.random.uniform(lo, hi).astype(self.dtype)
