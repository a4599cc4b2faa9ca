"""Host logic of the overlapped step_many paths (csrc/xv_pipe.h): how many ring cycles a cycle graph holds for a call of a
given length — whole steps per stream, never more cycles than the call has, sticky within 3 % of the best choice.  A small
host program built with hipcc (it calls nothing of the HIP runtime: no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cycle_graph_plan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "pipe_plan_check")
    src = os.path.join(ROOT, "tests", "native", "pipe_plan_check.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "xenoverse_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-500:]
    assert r.stdout.strip().endswith("ok (0 failures)")
