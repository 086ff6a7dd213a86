"""ModelMixin: nn.Module + the switches the reference's callers flip (train.py:317,343; run.py:284-289)."""
from functools import partial

import torch
from torch import nn


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    def enable_gradient_checkpointing(self):
        self.apply(partial(self._set_gradient_checkpointing, value=True))

    def disable_gradient_checkpointing(self):
        self.apply(partial(self._set_gradient_checkpointing, value=False))

    def _set_gradient_checkpointing(self, module, value=False):
        if hasattr(module, "gradient_checkpointing"):
            module.gradient_checkpointing = value

    def set_use_memory_efficient_attention_xformers(self, valid, attention_op=None):
        def rec(module):
            if hasattr(module, "set_use_memory_efficient_attention_xformers"):
                module.set_use_memory_efficient_attention_xformers(valid, attention_op)
            for child in module.children():
                rec(child)
        for m in self.children():
            if isinstance(m, nn.Module):
                rec(m)

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        self.set_use_memory_efficient_attention_xformers(True, attention_op)

    def disable_xformers_memory_efficient_attention(self):
        self.set_use_memory_efficient_attention_xformers(False)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype
