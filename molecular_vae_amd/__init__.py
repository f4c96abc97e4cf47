"""Import alias: the product package lives in ``../molecular-vae_amd`` (directory name fixed by the layout contract,
not importable as written).  This shim points the package path there and executes its ``__init__``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "molecular-vae_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
