"""MI355X-native batched SHEMS environment + DDPG hot path (gfx950), behind a C ABI.

Replaces the hot path of Lennart0HU/Master-Thesis-Deep-Reinforcement-Learning-DDPG-in-Home-Energy-Management
(RL-SHEMS/RL_environments/envs/shems_LU1.jl, algorithms/DDPG.jl, src/memory_plotting_saving.jl).
All compute goes through libshems_hip.so (include/shems_hip.h); there is no CPU fallback.
The directory name is not a Python identifier: import it via `import shems_amd` (repo-root alias)
or `importlib.import_module("master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd")`.
"""
from . import _capi, tables                     # noqa: F401
from ._capi import BoundsError, ShemsError      # noqa: F401
from .env import Shems, ShemsBatch, action, finished, make_config, mixed_profile_setup, reset_, step_   # noqa: F401

__all__ = ["Shems", "ShemsBatch", "reset_", "step_", "action", "finished", "make_config", "tables",
           "ShemsError", "BoundsError"]
