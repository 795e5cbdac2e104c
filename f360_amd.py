"""Importable alias of the package directory ``foveated-360-video_amd`` (hyphens are not
valid in an ``import`` statement): ``import f360_amd`` gives that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("foveated-360-video_amd")
sys.modules[__name__] = _pkg
