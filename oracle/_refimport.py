"""Import the Python reference (/root/reference) inside THIS build container only.

Test tooling: used by oracle/gen_golden.py and oracle/sample_ref_tasks.py to validate the
restatement in oracle/ and to emit the fixtures under tests/golden/.  Nothing here, and nothing
under /root/reference, travels to the GPU box or is imported by the product or by tests.

The reference needs numba / gymnasium / gym / pygame, none of which are installed here; the stand-ins in
oracle/stubs/ are put ahead of it on sys.path (identity `njit`, minimal spaces, PIL-backed image load).
`xenoverse.linds.__init__` is broken upstream (imports names that do not exist), so the
package object is pre-registered empty and its sub-modules imported directly.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("XENO_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "xenoverse"))


def setup():
    if not available():
        raise RuntimeError("reference tree not present at %s (only exists in the build container)" % REF_ROOT)
    for p in (REF_ROOT, _STUBS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, _STUBS)
    if "xenoverse.linds" not in sys.modules:
        import xenoverse  # noqa: F401
        m = types.ModuleType("xenoverse.linds")
        m.__path__ = [os.path.join(REF_ROOT, "xenoverse", "linds")]
        sys.modules["xenoverse.linds"] = m


def anymdp():
    setup()
    from xenoverse.anymdp.anymdp_env import AnyMDPEnv
    from xenoverse.anymdp import task_sampler
    return AnyMDPEnv, task_sampler


def linds():
    setup()
    from xenoverse.linds.linds_env import LinearDSEnv
    from xenoverse.linds import task_sampler
    return LinearDSEnv, task_sampler


def mazeworld():
    setup()
    from xenoverse.mazeworld.envs.maze_env import MazeWorldContinuous3D
    from xenoverse.mazeworld.envs import task_sampler, dynamics, ray_caster_utils
    return MazeWorldContinuous3D, task_sampler, dynamics, ray_caster_utils
