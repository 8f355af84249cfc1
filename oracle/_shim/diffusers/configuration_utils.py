import functools
import inspect


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        if not hasattr(self, "_internal_dict"):
            object.__setattr__(self, "_internal_dict", _Cfg())
        self._internal_dict.update(kw)


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_internal_dict", _Cfg(cfg))
        init(self, *args, **kwargs)
    return wrapper
