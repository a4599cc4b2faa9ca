"""Test-tooling stand-in for `pygame` (absent): PIL-backed image loading with pygame's (W,H,3) axis
order, and inert display/font/key/time objects.  Only what importing + stepping the reference's
mazeworld needs in oracle/gen_golden.py.  No reference code."""
import numpy
from PIL import Image

K_LEFT, K_RIGHT, K_UP, K_DOWN, K_SPACE = range(5)
QUIT = 0


def init():
    return None


class Color(object):
    def __init__(self, *a):
        self.a = a


class Surface(object):
    def __init__(self, arr_or_size):
        if isinstance(arr_or_size, numpy.ndarray):
            self.arr = arr_or_size
        else:
            self.arr = numpy.zeros((int(arr_or_size[0]), int(arr_or_size[1]), 3), dtype=numpy.uint8)

    def get_width(self):
        return self.arr.shape[0]

    def get_height(self):
        return self.arr.shape[1]

    def fill(self, c):
        pass

    def blit(self, *a, **k):
        pass


class _Image(object):
    @staticmethod
    def load(path):
        im = numpy.asarray(Image.open(path).convert("RGB"))  # (H, W, 3)
        return Surface(numpy.ascontiguousarray(im.transpose(1, 0, 2)))  # pygame: (W, H, 3)

    @staticmethod
    def save(*a, **k):
        pass


class _Surfarray(object):
    @staticmethod
    def array3d(surf):
        return numpy.array(surf.arr)

    @staticmethod
    def make_surface(arr):
        return Surface(numpy.asarray(arr))


class _Font(object):
    @staticmethod
    def init():
        pass

    @staticmethod
    def SysFont(*a, **k):
        return None


class _Dummy(object):
    def __getattr__(self, name):
        return lambda *a, **k: None


image = _Image()
surfarray = _Surfarray()
font = _Font()
key = _Dummy()
time = _Dummy()
display = _Dummy()
draw = _Dummy()
event = _Dummy()
transform = _Dummy()
