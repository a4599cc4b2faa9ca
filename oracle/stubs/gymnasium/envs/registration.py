import importlib

registry = {}


def register(id, entry_point=None, **kw):
    registry[id] = dict(entry_point=entry_point, **kw)


def make(id, **kw):
    spec = registry[id]
    mod, cls = spec["entry_point"].split(":")
    kwargs = dict(spec.get("kwargs", {}))
    kwargs.update(kw)
    return getattr(importlib.import_module(mod), cls)(**kwargs)
