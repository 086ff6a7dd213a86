"""DiffusionPipeline surface used by the reference's pipelines (register_modules, to, device, dtype, progress_bar)."""
import torch


class DiffusionPipeline:
    def register_modules(self, **kwargs):
        self._module_names = list(kwargs)
        for name, module in kwargs.items():
            setattr(self, name, module)

    @property
    def components(self):
        return {k: getattr(self, k) for k in self._module_names}

    def to(self, *args, **kwargs):
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                m.to(*args, **kwargs)
        return self

    @property
    def device(self):
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    @property
    def dtype(self):
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                return next(m.parameters()).dtype
        return torch.float32

    def progress_bar(self, iterable=None, total=None):
        return iterable

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        for m in self.components.values():
            if hasattr(m, "enable_xformers_memory_efficient_attention"):
                m.enable_xformers_memory_efficient_attention(attention_op)
