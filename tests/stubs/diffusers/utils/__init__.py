"""diffusers.utils names the reference imports."""
import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch

USE_PEFT_BACKEND = True   # diffusers 0.30.2 with peft installed (requirements.txt): plain nn.Linear / nn.Conv2d, no `scale` plumbing


class BaseOutput(OrderedDict):
    """dataclass-style output that also supports tuple / dict access (diffusers.utils.outputs.BaseOutput)."""

    def __post_init__(self):
        if is_dataclass(self):
            for f in fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def __setitem__(self, key, value):
        # keys are mirrored as attributes, so that subclasses that are NOT dataclasses (the reference's MarigoldDepthOutput) still
        # give `out.depth_np` after `MarigoldDepthOutput(depth_np=...)` — as diffusers.utils.outputs.BaseOutput does
        super().__setitem__(key, value)
        super().__setattr__(key, value)

    def __setattr__(self, name, value):
        if name in self.keys() and value is not None:
            super().__setitem__(name, value)
        super().__setattr__(name, value)

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


def deprecate(*args, **kwargs):
    return None


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


logging = _Logging()


def scale_lora_layers(model, weight):
    return None


def unscale_lora_layers(model, weight=None):
    return None


def is_torch_version(op, version):
    import operator
    from packaging.version import parse
    ops = {">": operator.gt, ">=": operator.ge, "==": operator.eq, "<": operator.lt, "<=": operator.le, "!=": operator.ne}
    return ops[op](parse(parse(torch.__version__).base_version), parse(version))
