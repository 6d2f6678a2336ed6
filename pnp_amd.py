"""Import alias: `import pnp_amd` == the package in ./medical-cross-modality-domain-adaptation_amd/ (whose
mandated name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("medical-cross-modality-domain-adaptation_amd")
sys.modules[__name__] = _pkg


def _sub(name):
    return importlib.import_module("medical-cross-modality-domain-adaptation_amd." + name)
