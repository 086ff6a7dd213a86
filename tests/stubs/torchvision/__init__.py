"""Stub of torchvision (tests/stubs/README.md): the tensor resize / PIL conversion the reference's pipelines call."""
from . import transforms  # noqa: F401
