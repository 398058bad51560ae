"""Import-name shim: the product package lives in the directory `mlx-vlm_amd/`
(the name the build contract asks for), which is not a valid Python identifier.
`import mlx_vlm_amd` executes that directory's __init__ under this name."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mlx-vlm_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
