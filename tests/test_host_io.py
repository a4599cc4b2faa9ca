"""task (de)serialisation and registry (CPU)."""
import numpy as np

from xenoverse_amd import registration, tasks_io
from xenoverse_amd.linds import LinearDSSampler
from xenoverse_amd.mazeworld import MazeTaskSampler
from util import golden_files, load_anymdp_golden


def test_pickle_roundtrip_like_reference(tmp_path):
    t = LinearDSSampler(16, 8, 8, seed=1)
    tasks_io.dump_task(tmp_path / "t.pkl", t)
    t2 = tasks_io.load_task(tmp_path / "t.pkl")
    assert np.array_equal(t["ld_A"], t2["ld_A"]) and t2["target_type"] == t["target_type"]


def test_npz_batches_roundtrip(tmp_path):
    _, a = load_anymdp_golden(golden_files("anymdp_16x4")[0])
    tasks_io.save_task_batch(tmp_path / "a.npz", "anymdp", [a, a])
    fam, tab = tasks_io.load_task_batch(tmp_path / "a.npz")
    assert fam == "anymdp" and tab["rows"].shape == (2, 16, 4, 4, 16) and tab["S"] == 16 and tab["A"] == 4
    tasks_io.save_task_batch(tmp_path / "l.npz", "linds", [LinearDSSampler(16, 8, 8, seed=k) for k in range(3)])
    fam, tab = tasks_io.load_task_batch(tmp_path / "l.npz")
    assert fam == "linds" and tab["phiT"].shape == (3, 16, 16) and tab["NS"] == 16
    tasks_io.save_task_batch(tmp_path / "m.npz", "mazeworld", [MazeTaskSampler(n_range=(9, 10), seed=0)])
    fam, tab = tasks_io.load_task_batch(tmp_path / "m.npz")
    assert fam == "mazeworld" and tab["walls"].shape == (1, 9, 9)


def test_registry_ids_match_reference():
    assert set(registration.REGISTRY) == {"anymdp-v0", "linear-dynamics-v0", "mazeworld-v2", "random-cartpole-v0",
                                          "random-acrobot-v0"}
    assert registration.REGISTRY["random-acrobot-v0"][1] == {"frameskip": 1, "reset_bounds_scale": 0.10}
    assert registration.REGISTRY["mazeworld-v2"][1]["resolution"] == (256, 256)
    assert registration.REGISTRY["random-cartpole-v0"][1]["frameskip"] == 1
    registration.register_with_gymnasium()     # must not raise without gymnasium


def test_reference_ids_register_as_vector_entry_points(monkeypatch):
    """with a gymnasium present (a minimal stand-in here: registry + register), the reference's own ids get the batched
    engine as `vector_entry_point`; an id the reference package registered first keeps its scalar entry point"""
    import sys
    import types
    from xenoverse_amd import registration
    reg = {}

    class Spec(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def register(id, entry_point=None, vector_entry_point=None, kwargs=None, **other):
        reg[id] = Spec(id=id, entry_point=entry_point, vector_entry_point=vector_entry_point, kwargs=kwargs or {})
    gym = types.ModuleType("gymnasium")
    envs = types.ModuleType("gymnasium.envs")
    regmod = types.ModuleType("gymnasium.envs.registration")
    regmod.register, regmod.registry = register, reg
    gym.envs, envs.registration = envs, regmod
    for name, mod in (("gymnasium", gym), ("gymnasium.envs", envs), ("gymnasium.envs.registration", regmod)):
        monkeypatch.setitem(sys.modules, name, mod)
    reg["anymdp-v0"] = Spec(id="anymdp-v0", entry_point="xenoverse.anymdp:AnyMDPEnv", vector_entry_point=None, kwargs={"max_steps": 5000})
    assert registration.register_with_gymnasium() is True
    assert reg["anymdp-v0"].entry_point == "xenoverse.anymdp:AnyMDPEnv"
    assert reg["anymdp-v0"].vector_entry_point == "xenoverse_amd.anymdp:AnyMDPVecEnv"
    for env_id in ("linear-dynamics-v0", "mazeworld-v2", "random-cartpole-v0", "random-acrobot-v0"):
        assert reg[env_id].entry_point is None and reg[env_id].vector_entry_point.startswith("xenoverse_amd.")
        assert reg["xenoverse-amd/" + env_id].vector_entry_point == reg[env_id].vector_entry_point
    assert reg["mazeworld-v2"].kwargs["resolution"] == (256, 256)
    assert registration.register_with_gymnasium() is True and len(reg) == 10     # idempotent
