"""Test-only stand-in for torch_geometric (see ../README.md)."""
from . import nn, utils, data  # noqa: F401
