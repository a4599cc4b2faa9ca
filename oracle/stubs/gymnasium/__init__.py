"""Test-tooling stand-in for `gymnasium` (absent from this image); just enough surface for the
reference's env classes to be *defined and stepped* by oracle/gen_golden.py.  No reference code."""
from . import spaces, error, utils  # noqa: F401
from .envs.registration import register, make  # noqa: F401


class Env(object):
    metadata = {}
    render_mode = None

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, a):
        raise NotImplementedError

    def close(self):
        pass
