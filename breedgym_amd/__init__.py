"""breedgym_amd: the vectorised BreedGym reset()/step() hot path on AMD MI355X (gfx950).

Public surface mirrors the reference package (``breedgym/__init__.py:4-34``):
``VecBreedGym``, ``BreedGym`` and -- when gymnasium is installed -- the same registered ids.
"""
from .breedgym import BreedGym
from .engine import DeviceArray, HipEngine, PackedPopulation
from .simulator import Simulator, TraitModel
from .vector.vec_env import VecBreedGym

__version__ = "0.5.1"  # = the library's (bh_version(); tests/test_abi.py compares them)
__all__ = ["BreedGym", "VecBreedGym", "Simulator", "TraitModel", "HipEngine", "DeviceArray", "PackedPopulation",
           "REGISTRY", "make"]

# The six ids the reference registers (breedgym/__init__.py:4-34), under this package's entry points.
REGISTRY = {
    "BreedGym": "breedgym_amd.breedgym:BreedGym",
    "SimplifiedBreedGym": "breedgym_amd.wrappers:SimplifiedBreedGym",
    "KBestBreedGym": "breedgym_amd.wrappers:KBestBreedGym",
    "VecBreedGym": "breedgym_amd.vector.vec_env:VecBreedGym",
    "SelectionScores": "breedgym_amd.vector.vec_wrappers:SelectionScores",
    "PairScores": "breedgym_amd.vector.vec_wrappers:PairScores",
}

try:  # with gymnasium installed they are genuine registrations: gym.make("SelectionScores", k=10, ...)
    from gymnasium.envs.registration import register as _register

    for _id, _entry in REGISTRY.items():
        _register(id=_id, entry_point=_entry)
except ImportError:
    pass


def make(id: str, **kwargs):
    """``gym.make`` for the ids above without gymnasium: ``"breedgym:KBestBreedGym"``-style module prefixes
    (the reference's tests use both spellings) are accepted."""
    import importlib

    name = id.split(":")[-1]
    if name not in REGISTRY:
        raise KeyError(f"unknown environment id {id!r}; known: {sorted(REGISTRY)}")
    module, attr = REGISTRY[name].split(":")
    return getattr(importlib.import_module(module), attr)(**kwargs)
