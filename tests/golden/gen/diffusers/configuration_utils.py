import functools
import inspect


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class ConfigMixin:
    def register_to_config(self, **kw):
        if not hasattr(self, "config"):
            object.__setattr__(self, "config", _Cfg())
        self.config.update(kw)


def register_to_config(init):
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "config", _Cfg(cfg))
        init(self, *args, **kwargs)

    return wrapper
