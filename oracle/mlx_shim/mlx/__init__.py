"""Stand-in for Apple's `mlx` package (TEST INFRASTRUCTURE, see oracle/mlx_shim/README.md)."""
from . import core  # noqa: F401

__version__ = "0.32.0+shim"
