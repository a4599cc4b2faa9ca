"""The C-ABI library loads on a GPU-less host and exports every symbol include/xeno.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from xenoverse_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "xeno.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xv_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_families():
    syms = declared_symbols()
    for must in ("xv_engine_create", "xv_anymdp_create", "xv_anymdp_step", "xv_anymdp_step_injected",
                 "xv_anymdp_reset", "xv_anymdp_rollout", "xv_anymdp_synth_tasks"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run python -m xenoverse_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libxeno_hip.so does not export %s" % s


def test_every_declared_symbol_is_bound_in_python():
    missing = [s for s in declared_symbols() if s not in _lib.SIGNATURES]
    assert not missing, missing
    extra = [s for s in _lib.SIGNATURES if s not in declared_symbols()]
    assert not extra, extra


def test_abi_version_and_loader():
    lib = _lib.load()
    assert lib.xv_abi_version() == _lib.ABI_VERSION == 12


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libxeno_hip.so")
    with pytest.raises(_lib.XenoError):
        _lib.load()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    with pytest.raises(_lib.XenoError):
        AnyMDPVecEnv(4)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under xenoverse_amd/ may reference it."""
    pkg = os.path.join(ROOT, "xenoverse_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "xeno_oracle.h" not in src.replace("oracle/xeno_oracle.c", ""), os.path.join(dp, f)


def test_graft_entry_build_runs():
    """the driver's build hook: compiles (incrementally) every HIP source, the CPU checker, and loads the library"""
    import __graft_entry__ as ge
    ge.build()


def test_header_is_plain_c(tmp_path):
    """include/xeno.h is the contract for non-Python callers: it must compile as C99 on its own, and its row-size
    macros must agree with the Python table builder"""
    import subprocess
    from xenoverse_amd.anymdp.tables import row_lines
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    checks = " && ".join("XV_ANYMDP_ROW_LINES(%d) == %d" % (S, row_lines(S)) for S in (2, 7, 8, 64, 112, 113, 128, 224, 225, 256))
    src.write_text('#include "xeno.h"\nint main(void) { return (%s) ? 0 : 1; }\n' % checks)
    exe = str(tmp_path / "h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           str(src), "-o", exe])
    assert subprocess.call([exe]) == 0


def test_build_tracks_sources_included_by_sources():
    """mixed.hip is compiled from the three family sources it #includes: a change to any of them must rebuild mixed.o
    (a stale object there reads the family handles with an old layout)"""
    import os
    from xenoverse_amd import build as xb
    deps = {os.path.basename(p) for p in xb._includes(os.path.join(xb.CSRC, "mixed.hip"), set())}
    assert {"anymdp.hip", "linds.hip", "cartpole.hip", "xv_common.h", "philox.h", "xeno.h"} <= deps
    newest = max(os.path.getmtime(os.path.join(xb.CSRC, f)) for f in ("anymdp.hip", "linds.hip", "cartpole.hip"))
    assert xb._deps_mtime(os.path.join(xb.CSRC, "mixed.hip")) >= newest
