"""ConfigMixin / register_to_config: constructor keywords (with defaults) become `self.config`, readable as attributes and items
and writable as items (`unet.config['in_channels'] = 8`, training/util/unet_prep.py:19)."""
import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        if not hasattr(self, "_internal_dict"):
            object.__setattr__(self, "_internal_dict", FrozenDict())
        self._internal_dict.update(kwargs)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for p in sig.parameters.values() if p.name != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update(kwargs)
        init(self, *args, **kwargs)
        merged = dict(cfg)
        merged.update(getattr(self, "_internal_dict", {}))   # explicit register_to_config calls inside __init__ win
        self.register_to_config(**merged)
    return inner
