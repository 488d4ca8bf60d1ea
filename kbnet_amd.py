"""Importable alias for the hyphen-named package directory.

`import kbnet_amd as kb; kb.modules.KBNetModel(...)`.  Sub-modules are exposed
as attributes of the one real package object (no second copy is imported).
"""

import importlib as _importlib
import sys as _sys

_pkg = _importlib.import_module("calibrated-backprojection-network_amd")


def __getattr__(name):
    try:
        return getattr(_pkg, name)
    except AttributeError:
        return _importlib.import_module(_pkg.__name__ + "." + name)


def __dir__():
    return dir(_pkg)
