"""Importable alias of the package directory (whose mandated name contains hyphens)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd")
sys.modules[__name__] = _pkg
