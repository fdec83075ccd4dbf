"""The code-object reader lives in the package (gaussian-object-modelling_amd/codeobj.py: __graft_entry__.build() runs its
guards on every build); the tests and scripts import it through this name."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_m = importlib.import_module("gaussian-object-modelling_amd.codeobj")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
