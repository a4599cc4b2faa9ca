"""Host-side pieces of round 6 (no GPU): the roofline basis never prints a fraction above 1, the keys of the bench line's
`long_call` block, the committed floor probe is keyed by source hashes."""
import glob
import json
import os
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_basis_never_prints_a_fraction_above_one():
    algo = 562 * 65536
    # traffic below the algorithmic bytes (the bucket search reads one line of the row): the fraction prices the traffic
    r = bench.roofline_basis(algo, 14.14e6, 3.3, True)
    assert r["bound"] == "hbm" and abs(r["frac"] - 14.14e6 / 3.3e-6 / 1e9 / 8000.0) < 1e-12 and r["frac"] < 1.0
    assert "traffic" in r["basis"]
    # no profile of this kernel source, survey bytes above the peak: no fraction
    r = bench.roofline_basis(algo, None, 3.3, True)
    assert r["frac"] is None and r["achieved"] > 8000.0 and "frac null" in r["basis"]
    # no profile, below the peak: the survey's fraction
    r = bench.roofline_basis(algo, None, 5.0, True)
    assert 0.9 < r["frac"] < 0.93
    # traffic above the algorithmic bytes (wasted re-reads): the survey's fraction stays the claim
    r = bench.roofline_basis(algo, 2 * algo, 10.0, True)
    assert abs(r["frac"] - algo / 10e-6 / 1e9 / 8000.0) < 1e-12
    # shared tasks: cache resident, no HBM fraction whatever the time
    for us in (2.9, 5.0, 50.0):
        r = bench.roofline_basis(algo, 14e6, us, False)
        assert r["bound"] == "cache" and r["frac"] is None
    for us in (0.5, 1.0, 2.0, 3.0, 4.6, 5.0, 8.0):
        for tr in (None, 1e6, 14e6, 40e6):
            f = bench.roofline_basis(algo, tr, us, True)["frac"]
            assert f is None or f <= 1.0 or tr is not None      # (a PMC traffic above the peak would be a counter fault: shown as is)


def test_long_call_schema_constants():
    assert bench.LONG_CALL_MODES == ("one_stream", "overlapped", "fused_rollout")
    for k in ("steps", "warmup", "us_per_step", "wall_us_per_step", "env_steps_per_s", "overlap_state", "graph_state",
              "device_error_flags"):
        assert k in bench.LONG_CALL_ROW_KEYS
    for k in ("bound", "achieved", "peak", "frac", "traffic", "frac_survey_bytes"):
        assert k in bench.ROOFLINE_KEYS
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in bench.LONG_CALL_ROW_KEYS:          # the row builder writes exactly these keys
        assert '"%s":' % k in src, k


def test_committed_bench_lines_of_this_round_carry_the_block():
    """every r06 bench line kept under profiles/ has the long_call block with the three issue modes and both clocks"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*bench_2a*.json")))
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        lc = d["long_call"]
        for m in bench.LONG_CALL_MODES:
            row = lc[m]
            for k in bench.LONG_CALL_ROW_KEYS:
                assert k in row, (f, m, k)
            assert row["steps"] >= 2000 and row["warmup"] >= 100 and row["device_error_flags"] == 0
        fr = d["roofline"]["frac"]
        assert fr is None or fr <= 1.0
        for k in bench.ROOFLINE_KEYS:
            assert k in d["roofline"], (f, k)


def test_floor_probe_is_keyed_by_source_hashes():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*floor_probe*.json")))
    for f in files:
        d = json.load(open(f))
        assert d.get("probe_source_sha16") and "kernel_source_sha16" in d
    fl = bench.floor_probe()
    assert "kernel_source_sha16" in fl and fl["source"]


def test_the_cpython_trampoline_calls_the_c_abi_like_ctypes():
    """xenoverse_amd/_xvfast.so (csrc/xvfast.c, built by xenoverse_amd.build.build_fast with gcc): `icall(fn, *ints)` makes the
    same C-ABI call ctypes makes — checked on entry points that need no GPU (the version, and argument validation)"""
    import pytest
    from xenoverse_amd import _lib
    from xenoverse_amd import build as xb
    try:
        xb.build_fast()
    except RuntimeError as ex:
        pytest.skip("no Python.h / gcc here: %s" % (ex,))
    _lib._fast[0], _lib._fast[1] = None, False
    f = _lib.fast()
    assert f is not None
    lib = _lib.load()
    assert f.icall(_lib.fn_address("xv_abi_version")) == lib.xv_abi_version() == _lib.ABI_VERSION
    # null handle: XV_ERR_INVALID from the library's own argument check, through both bindings
    assert f.icall(_lib.fn_address("xv_anymdp_step_info"), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2) == _lib.XV_ERR_INVALID
    assert lib.xv_anymdp_step_info(None, None, None, None, None, None, None, None, None, None, 2) == _lib.XV_ERR_INVALID
    assert f.icall(_lib.fn_address("xv_linds_step_info"), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2) == _lib.XV_ERR_INVALID
    assert b"" != lib.xv_last_error()
    with pytest.raises(TypeError):
        f.icall(*([_lib.fn_address("xv_abi_version")] + [0] * 15))
    with pytest.raises(ValueError):
        f.icall(0)
    with pytest.raises(TypeError):
        f.icall(_lib.fn_address("xv_abi_version"), "not an int")


def test_the_trampoline_is_optional(monkeypatch):
    from xenoverse_amd import _lib
    monkeypatch.setenv("XV_NO_FAST", "1")
    _lib._fast[0], _lib._fast[1] = None, False
    assert _lib.fast() is None                      # the ctypes path serves
    monkeypatch.delenv("XV_NO_FAST")
    _lib._fast[0], _lib._fast[1] = None, False


def test_the_maze_roofline_prices_the_newest_committed_counters(tmp_path, monkeypatch):
    """bench_families.maze_valu_per_pixel: the newest counter profile of a filter and resolution by VISIT order (r06_m < r06_zz5 <
    r06_zz28 — a plain string sort puts zz5 behind zz28 and priced the 64 x 64 line with a kernel that no longer ran), and the
    committed ones belong to the kernel AUTO launches (the rows mapping, FILT 5)"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_families as bf
    prof = tmp_path / "profiles"
    prof.mkdir()
    (tmp_path / "scripts").mkdir()
    for name, valu in (("r06_m", 100.0), ("r06_zz5", 200.0), ("r06_zz28", 300.0)):
        (prof / ("%s_pmc_raycast_spec32_64.json" % name)).write_text(json.dumps(
            {"kernels": {"void maze_raycast_kernel<false, true, 5, false>": {"SQ_INSTS_VALU": valu}}}))
    monkeypatch.setattr(bf.os.path, "abspath", lambda p: str(tmp_path / "scripts" / "bench_families.py"))
    v, src = bf.maze_valu_per_pixel("spec32", 64, 16384)
    assert src == "r06_zz28_pmc_raycast_spec32_64.json" and abs(v - 300.0 * 64 / (64 * 64 * 16384)) < 1e-12
    monkeypatch.undo()
    for res in (64, 256):
        v, src = bf.maze_valu_per_pixel("spec32", res, 16384)
        d = json.load(open(os.path.join(ROOT, "profiles", src)))
        assert any("maze_raycast_kernel<false, true, 5, false>" in k for k in d["kernels"]), (res, src)
        assert 250.0 < v < 420.0
