"""breedgym_amd: the vectorised BreedGym reset()/step() hot path on AMD MI355X (gfx950).

Public surface mirrors the reference package (``breedgym/__init__.py:4-34``):
``VecBreedGym``, ``BreedGym`` and -- when gymnasium is installed -- the same registered ids.
"""
from .breedgym import BreedGym
from .engine import DeviceArray, HipEngine, PackedPopulation
from .simulator import Simulator, TraitModel
from .vector.vec_env import VecBreedGym

__version__ = "0.1.0"
__all__ = ["BreedGym", "VecBreedGym", "Simulator", "TraitModel", "HipEngine", "DeviceArray", "PackedPopulation"]

try:  # same ids as the reference, under this package's entry points
    from gymnasium.envs.registration import register as _register

    _register(id="BreedGym", entry_point="breedgym_amd.breedgym:BreedGym")
    _register(id="VecBreedGym", entry_point="breedgym_amd.vector.vec_env:VecBreedGym")
except ImportError:
    pass
