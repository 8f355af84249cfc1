"""Importable alias for the package directory `ctrl-adapter_amd/` (a hyphen is not a valid identifier)."""
import importlib
import sys

_pkg = importlib.import_module("ctrl-adapter_amd")
sys.modules[__name__] = _pkg
