"""Engine: one GPU + one HIP stream + one Philox key (C-ABI: xv_engine_*)."""
import ctypes as C

import torch

from . import _lib

AUTORESET = {"disabled": 0, "next_step": 1, "same_step": 2}


class Engine(object):
    """Owns an xv_engine.  By default it launches on torch's current stream of `device`, so kernels are
    ordered with the torch ops that produce actions / consume observations (torch is only plumbing:
    device memory and streams)."""

    def __init__(self, device="cuda:0", seed=0, env_id_base=0, stream=None):
        self.lib = _lib.load()   # raises if libxeno_hip.so is absent: there is no CPU path
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.XenoError("xenoverse_amd runs on a ROCm GPU only (got device %r)" % (device,))
        if not torch.cuda.is_available():
            raise _lib.XenoError("no ROCm GPU visible to torch; xenoverse_amd has no CPU fallback")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.torch_stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        _lib.check(self.lib.xv_engine_create(idx, int(seed) & (2**64 - 1), int(env_id_base),
                                             C.c_void_p(self.torch_stream.cuda_stream), C.byref(h)))
        self.handle = h
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)

    def sync(self):
        _lib.check(self.lib.xv_engine_sync(self.handle))

    def error_flags(self, clear=True):
        v = C.c_uint32(0)
        _lib.check(self.lib.xv_engine_error_flags(self.handle, 1 if clear else 0, C.byref(v)))
        return int(v.value)

    @property
    def tick(self):
        v = C.c_uint64(0)
        _lib.check(self.lib.xv_engine_get_tick(self.handle, C.byref(v)))
        return int(v.value)

    @tick.setter
    def tick(self, value):
        _lib.check(self.lib.xv_engine_set_tick(self.handle, int(value)))

    # ---- device tick (xv_engine_set_device_tick): what makes a step capturable in a torch.cuda.graph ------------
    @property
    def device_tick(self):
        return bool(self.lib.xv_engine_device_tick(self.handle))

    def set_device_tick(self, on=True):
        """on: the launch tick lives in device memory and every stochastic launch advances it there — a captured step
        draws fresh numbers at every replay, exactly those the same calls issued eagerly would draw.  `tick` then reads
        the device word back (synchronises)."""
        _lib.check(self.lib.xv_engine_set_device_tick(self.handle, 1 if on else 0))

    def tick_batch(self, on):
        """device tick mode: between tick_batch(True) and tick_batch(False) launches read tick + 0, + 1, ... and the word is
        advanced once at the end (an unrolled capture carries one tick node instead of one per step)"""
        _lib.check(self.lib.xv_engine_tick_batch(self.handle, 1 if on else 0))

    def set_stream(self, stream):
        """launch on another torch stream of the same device from now on (stream capture runs on a side stream)"""
        _lib.check(self.lib.xv_engine_set_stream(self.handle, C.c_void_p(stream.cuda_stream)))
        self.torch_stream = stream

    # ---- timing events on the engine's own stream (xv_engine_event_*) ----------------------------------------
    def event_record(self, slot):
        _lib.check(self.lib.xv_engine_event_record(self.handle, int(slot)))

    def event_done(self, slot):
        v = C.c_int(0)
        _lib.check(self.lib.xv_engine_event_done(self.handle, int(slot), C.byref(v)))
        return bool(v.value)

    def event_elapsed_ms(self):
        v = C.c_float(0.0)
        _lib.check(self.lib.xv_engine_event_elapsed_ms(self.handle, C.byref(v)))
        return float(v.value)

    def probe_side_streams(self):
        """diagnostic (xv_engine_probe_side_streams): the side-stream candidates the overlapped step_many paths would try
        beside this engine's stream -> list of dicts(priority, two_stream_us, one_stream_us, accepted)"""
        n = 8
        pr, ac, k = (C.c_int * n)(), (C.c_int * n)(), C.c_int(0)
        two, one = (C.c_float * n)(), (C.c_float * n)()
        _lib.check(self.lib.xv_engine_probe_side_streams(self.handle, n, pr, two, one, ac, C.byref(k)))
        return [dict(priority=int(pr[i]), two_stream_us=float(two[i]), one_stream_us=float(one[i]), accepted=bool(ac[i]))
                for i in range(k.value)]

    def philox(self, ctr, key):
        """Philox4x32-10 known-answer hook: ctr uint32[n,4], key uint32[2] -> uint32[n,4] (device)."""
        ctr = torch.as_tensor(ctr, dtype=torch.int64).to(torch.int32).to(self.device).contiguous() \
            if not torch.is_tensor(ctr) else ctr
        key = torch.as_tensor(key, dtype=torch.int64).to(torch.int32).to(self.device).contiguous() \
            if not torch.is_tensor(key) else key
        out = torch.empty_like(ctr)
        n = ctr.numel() // 4
        _lib.check(self.lib.xv_philox4x32_10(self.handle, _lib.ptr(ctr), _lib.ptr(key), _lib.ptr(out), n))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.xv_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
