"""ctypes binding of libxeno_hip.so (C-ABI: include/xeno.h).

There is NO fallback: if the HIP library is missing or does not load, importing the engine raises.  The
product never computes an environment step on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XV_LIB_PATH: A/B measurement of kernel variants built by scripts/devtools/build_variant.py (same ABI, same checks)
LIB_PATH = os.environ.get("XV_LIB_PATH") or os.path.join(_HERE, "libxeno_hip.so")

c_void_p, c_int, c_u64, c_i64, c_u32 = C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_uint32

# name -> argtypes (restype is int unless listed in _RESTYPE).  Keep in sync with include/xeno.h;
# tests/test_abi.py parses the header and checks that every declared symbol is exported and bound.
SIGNATURES = {
    "xv_abi_version": [],
    "xv_pack_rollout": [c_void_p, C.c_size_t] + [c_void_p] * 6,
    "xv_unpack_rollout": [c_void_p, C.c_size_t] + [c_void_p] * 6,
    "xv_pack_rollout_f32": [c_void_p, C.c_size_t, c_int] + [c_void_p] * 6,
    "xv_unpack_rollout_f32": [c_void_p, C.c_size_t, c_int] + [c_void_p] * 6,
    "xv_last_error": [],
    "xv_rccl_unique_id": [c_void_p],
    "xv_rccl_comm_create": [c_void_p, c_int, c_int, c_void_p, C.POINTER(c_void_p)],
    "xv_rccl_comm_destroy": [c_void_p],
    "xv_rccl_comm_count": [c_void_p, C.POINTER(c_int)],
    "xv_rollout_allgather": [c_void_p, c_void_p, c_void_p, c_void_p, C.c_size_t],
    "xv_engine_create": [c_int, c_u64, c_u64, c_void_p, C.POINTER(c_void_p)],
    "xv_engine_destroy": [c_void_p],
    "xv_engine_sync": [c_void_p],
    "xv_engine_stream": [c_void_p],
    "xv_engine_error_flags": [c_void_p, c_int, C.POINTER(c_u32)],
    "xv_engine_get_tick": [c_void_p, C.POINTER(c_u64)],
    "xv_engine_set_tick": [c_void_p, c_u64],
    "xv_engine_set_device_tick": [c_void_p, c_int],
    "xv_engine_device_tick": [c_void_p],
    "xv_engine_tick_batch": [c_void_p, c_int],
    "xv_engine_set_stream": [c_void_p, c_void_p],
    "xv_engine_event_record": [c_void_p, c_int],
    "xv_engine_event_done": [c_void_p, c_int, C.POINTER(c_int)],
    "xv_engine_event_elapsed_ms": [c_void_p, C.POINTER(C.c_float)],
    "xv_philox4x32_10": [c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "xv_anymdp_create": [c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 7 + [C.POINTER(c_void_p)],
    "xv_anymdp_destroy": [c_void_p],
    "xv_anymdp_reset": [c_void_p, c_void_p, c_void_p],
    "xv_anymdp_reset_injected": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_anymdp_step": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_anymdp_step_info": [c_void_p] + [c_void_p] * 9 + [c_int],
    "xv_anymdp_step_injected": [c_void_p] + [c_void_p] * 10 + [c_int],
    "xv_anymdp_step_many": [c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int],
    "xv_anymdp_set_step_many_graph": [c_void_p, c_int],
    "xv_anymdp_set_step_many_overlap": [c_void_p, c_int],
    "xv_anymdp_step_many_overlap_state": [c_void_p],
    "xv_anymdp_build_rows": [c_void_p, c_int, c_int, c_int] + [c_void_p] * 6 + [c_u64],
    "xv_anymdp_view": [c_void_p, c_void_p, c_int, c_int, C.POINTER(c_void_p)],
    "xv_anymdp_step_many_chains": [c_void_p, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 7 + [c_int],
    "xv_anymdp_solve": [c_void_p, C.c_double, C.c_double, c_int, c_void_p, c_void_p, c_void_p],
    "xv_anymdp_step_many_graph_state": [c_void_p],
    "xv_anymdp_sample_tasks": [c_void_p, c_u64, c_i64, c_int, c_int, c_int, c_int] + [c_void_p] * 11,
    "xv_anymdp_value_iteration_set_summation": [c_int],
    "xv_anymdp_sample_observation_model": [c_void_p, c_u64, c_i64, c_int, c_int, c_int, c_int, C.c_double, C.c_double, c_void_p],
    "xv_anymdp_value_iteration_gs": [c_void_p, c_void_p, c_int, c_int, C.c_double, c_int, c_void_p, c_void_p],
    "xv_anymdp_rollout": [c_void_p, c_int] + [c_void_p] * 7,
    "xv_anymdp_set_search": [c_void_p, c_int],
    "xv_anymdp_build_buckets": [c_void_p, c_int],
    "xv_anymdp_probe_buckets": [c_void_p, c_int, c_void_p],
    "xv_anymdp_bucket_census_get": [c_void_p, c_void_p],
    "xv_anymdp_effective_search": [c_void_p],
    "xv_anymdp_token_kernel": [c_void_p],
    "xv_anymdp_set_observation_model": [c_void_p, c_int, c_int, c_int, c_void_p],
    "xv_anymdp_reset_tokens": [c_void_p, c_void_p, c_void_p],
    "xv_anymdp_reset_tokens_injected": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_anymdp_step_tokens": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_anymdp_step_tokens_info": [c_void_p] + [c_void_p] * 9 + [c_int],
    "xv_anymdp_step_tokens_injected": [c_void_p] + [c_void_p] * 12 + [c_int],
    "xv_anymdp_step_tokens_many": [c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int],
    "xv_anymdp_rollout_teacher": [c_void_p, c_int, c_void_p, C.c_float] + [c_void_p] * 7,
    "xv_anymdp_get_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_anymdp_set_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_anymdp_transition_gt": [c_void_p, c_void_p, c_void_p],
    "xv_anymdp_synth_tasks": [c_void_p, c_u64, c_i64, c_int, c_int, c_int, c_int] + [c_void_p] * 6,
    "xv_mixed_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "xv_mixed_supported": [c_void_p, c_void_p, c_void_p],
    "xv_mixed_step_many": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int],
    "xv_mixed_step_many_overlap_state": [c_void_p],
    "xv_engine_probe_side_streams": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_linds_create": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, C.POINTER(c_void_p)],
    "xv_linds_destroy": [c_void_p],
    "xv_linds_step_many": [c_void_p, c_int, c_int] + [c_void_p] * 8 + [c_int],
    "xv_linds_set_path": [c_void_p, c_int],
    "xv_linds_set_command_table": [c_void_p, c_int],
    "xv_linds_reset": [c_void_p] + [c_void_p] * 4,
    "xv_linds_reset_injected": [c_void_p] + [c_void_p] * 5,
    "xv_linds_step": [c_void_p] + [c_void_p] * 8 + [c_int],
    "xv_linds_step_info": [c_void_p] + [c_void_p] * 10 + [c_int],
    "xv_linds_step_injected": [c_void_p] + [c_void_p] * 10 + [c_int],
    "xv_linds_rollout": [c_void_p, c_int] + [c_void_p] * 8,
    "xv_linds_get_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_linds_set_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_cartpole_create": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, C.POINTER(c_void_p)],
    "xv_cartpole_destroy": [c_void_p],
    "xv_cartpole_reset": [c_void_p, c_void_p, c_void_p],
    "xv_cartpole_reset_injected": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_cartpole_step": [c_void_p] + [c_void_p] * 6 + [c_int],
    "xv_cartpole_step_info": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_cartpole_rollout": [c_void_p, c_int] + [c_void_p] * 6 + [c_int],
    "xv_cartpole_step_injected": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_cartpole_get_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_cartpole_set_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_acrobot_create": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, C.POINTER(c_void_p)],
    "xv_acrobot_destroy": [c_void_p],
    "xv_acrobot_reset": [c_void_p, c_void_p, c_void_p],
    "xv_acrobot_reset_injected": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_acrobot_step": [c_void_p] + [c_void_p] * 6 + [c_int],
    "xv_acrobot_step_info": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_acrobot_rollout": [c_void_p, c_int] + [c_void_p] * 6 + [c_int],
    "xv_acrobot_step_injected": [c_void_p] + [c_void_p] * 7 + [c_int],
    "xv_acrobot_get_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_acrobot_set_state": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_maze_create": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, C.c_double, C.c_double,
                       c_void_p, c_void_p, C.POINTER(c_void_p)],
    "xv_maze_destroy": [c_void_p],
    "xv_maze_reset": [c_void_p, c_void_p, c_void_p, c_void_p],
    "xv_maze_step": [c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int],
    "xv_maze_get_state": [c_void_p] + [c_void_p] * 8,
    "xv_maze_set_state": [c_void_p] + [c_void_p] * 6,
    "xv_maze_render": [c_void_p, c_void_p, c_void_p],
    "xv_maze_set_precision": [c_void_p, c_int],
    "xv_maze_set_raycast_mapping": [c_void_p, c_int],
    "xv_maze_set_typing": [c_void_p, c_int],
    "xv_maze_set_move_kernel": [c_void_p, c_int],
    "xv_maze_agent_create": [c_void_p, c_int, C.c_double, c_int, c_int, c_int, c_void_p],
    "xv_maze_agent_destroy": [c_void_p],
    "xv_maze_agent_act": [c_void_p, c_void_p, c_void_p],
    "xv_maze_agent_get": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
}
_RESTYPE = {"xv_last_error": C.c_char_p, "xv_engine_stream": c_void_p}

_lib = None


class BucketCensus(C.Structure):
    """xv_anymdp_bucket_census (include/xeno.h)"""
    _fields_ = [("n_bucket", C.c_int32), ("format", C.c_int32), ("cuts_per_line", C.c_int32), ("built", C.c_int32),
                ("auto_uses_bucket", C.c_int32), ("reserved", C.c_int32), ("lines", C.c_uint64), ("lines_dirty", C.c_uint64),
                ("live_rows", C.c_uint64), ("p_fallback", C.c_double), ("fallbacks_per_launch", C.c_double),
                ("bytes", C.c_double), ("auto_limit", C.c_double), ("obs_lines", C.c_uint64), ("obs_lines_dirty", C.c_uint64),
                ("obs_p_fallback", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class XenoError(RuntimeError):
    code = 0      # the XV_ERR_* value check() saw (include/xeno.h)


XV_ERR_INVALID, XV_ERR_HIP, XV_ERR_UNSUPPORTED, XV_ERR_NOMEM = -1, -2, -3, -4


ABI_VERSION = 12     # include/xeno.h XV_ABI_VERSION


def load():
    """Load libxeno_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XenoError(
            "libxeno_hip.so is not built (%s). Run `python -m xenoverse_amd.build` (needs hipcc). "
            "xenoverse_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: loud by design
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, c_int)
    if lib.xv_abi_version() != ABI_VERSION:
        raise XenoError("libxeno_hip.so has ABI version %d, this package binds version %d: rebuild with "
                        "`python -m xenoverse_amd.build --force`" % (lib.xv_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_fast = [None, False]      # (module or None, tried)


def fast():
    """the CPython trampoline `_xvfast` (xenoverse_amd/_xvfast.so, built by xenoverse_amd.build.build_fast) or None: a shortcut
    for the BINDING of a few hot calls — ctypes serves the same C-ABI without it (XV_NO_FAST=1 forces that)"""
    if not _fast[1]:
        _fast[1] = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_xvfast.so")
        if os.path.exists(path) and not os.environ.get("XV_NO_FAST"):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("_xvfast", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _fast[0] = mod
            except Exception:
                _fast[0] = None
    return _fast[0]


def fn_address(name):
    """address of a C-ABI entry point (for _xvfast.icall)"""
    return C.cast(getattr(load(), name), C.c_void_p).value


def check(rc):
    if rc != 0:
        msg = load().xv_last_error()
        err = XenoError("libxeno_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
        err.code = int(rc)
        raise err


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    assert t.is_contiguous(), "tensors handed to the C-ABI must be contiguous"
    return t.data_ptr()
