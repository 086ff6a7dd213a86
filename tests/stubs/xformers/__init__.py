"""Stub of xformers (tests/stubs/README.md): the one function the reference calls."""
from . import ops  # noqa: F401
