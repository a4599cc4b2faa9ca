"""Environment ids.  The reference registers `anymdp-v0`, `linear-dynamics-v0`, `mazeworld-v2` and
`random-cartpole-v0` / `random-acrobot-v0` with gymnasium (anymdp/__init__.py:24-30, linds/__init__.py:21-35, mazeworld/__init__.py:19-33,
metacontrol/__init__.py:20-26).  `make_vec(id, num_envs, **kw)` builds the batched engine with the same registered
keyword defaults; when gymnasium is importable the ids are also registered as vector entry points."""
import importlib
import warnings

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
    """Register the reference's OWN ids (`anymdp-v0`, `linear-dynamics-v0`, `mazeworld-v2`, `random-cartpole-v0`,
    `random-acrobot-v0`) with gymnasium as VECTOR entry points, so that `gymnasium.make_vec("anymdp-v0", num_envs=N)`
    builds the batched engine.  An id that is already registered (the reference package imported first: it registers
    scalar `entry_point`s) keeps its scalar entry point — `gymnasium.make(id)` still builds the reference's env — and
    gains the vector one.  The same specs are also available as `xenoverse-amd/<id>`.  No-op (False) without gymnasium."""
    try:
        from gymnasium.envs.registration import register, registry
    except Exception:
        return False
    for env_id, (entry, defaults) in REGISTRY.items():
        scalar = None
        old = registry.get(env_id) if hasattr(registry, "get") else None
        if old is not None:
            if getattr(old, "vector_entry_point", None) == entry:
                continue
            scalar = getattr(old, "entry_point", None)
        for name, kw in ((env_id, dict(entry_point=scalar) if scalar is not None else {}), ("xenoverse-amd/" + env_id, {})):
            if name in registry and name != env_id:
                continue
            replaced = name == env_id and old is not None
            try:
                if replaced:
                    registry.pop(env_id, None)
                register(id=name, vector_entry_point=entry, kwargs=dict(defaults), order_enforce=False,
                         disable_env_checker=True, **kw)
            except Exception as exc:      # never lose the reference's own (scalar) registration over ours
                if replaced and env_id not in registry:
                    registry[env_id] = old
                warnings.warn("xenoverse_amd: could not register %r as a vector entry point: %r" % (name, exc))
    return True
