"""Environment ids.  The reference registers `anymdp-v0`, `linear-dynamics-v0`, `mazeworld-v2` and
`random-cartpole-v0` / `random-acrobot-v0` with gymnasium (anymdp/__init__.py:24-30, linds/__init__.py:21-35, mazeworld/__init__.py:19-33,
metacontrol/__init__.py:20-26).  `make_vec(id, num_envs, **kw)` builds the batched engine with the same registered
keyword defaults; when gymnasium is importable the ids are also registered as vector entry points."""
import importlib

REGISTRY = {
    "anymdp-v0": ("xenoverse_amd.anymdp:AnyMDPVecEnv", {"max_steps": 5000}),
    "linear-dynamics-v0": ("xenoverse_amd.linds:LinDSVecEnv", {"dt": 0.1, "max_steps": 1000, "pad_observation_dim": 16,
                                                               "pad_command_dim": 16, "pad_action_dim": 8}),
    "mazeworld-v2": ("xenoverse_amd.mazeworld:MazeWorldVecEnv",
                     {"enable_render": False, "render_scale": 480, "resolution": (256, 256), "max_steps": 5000,
                      "visibility_3D": 12.0, "command_in_observation": False, "action_space_type": "Discrete16"}),
    "random-cartpole-v0": ("xenoverse_amd.metacontrol:CartPoleVecEnv",
                           {"frameskip": 1, "reset_bounds_scale": [0.45, 0.90, 0.13, 1.0]}),
    "random-acrobot-v0": ("xenoverse_amd.metacontrol:AcrobotVecEnv", {"frameskip": 1, "reset_bounds_scale": 0.10}),
}


def make_vec(env_id, num_envs, **kwargs):
    entry, defaults = REGISTRY[env_id]
    mod, cls = entry.split(":")
    kw = dict(defaults)
    kw.update(kwargs)
    return getattr(importlib.import_module(mod), cls)(num_envs, **kw)


def register_with_gymnasium():
    """no-op when gymnasium is not installed"""
    try:
        from gymnasium.envs.registration import register
    except Exception:
        return False
    for env_id, (entry, defaults) in REGISTRY.items():
        try:
            register(id="xenoverse-amd/" + env_id, vector_entry_point=entry, kwargs=defaults)
        except Exception:
            pass
    return True
