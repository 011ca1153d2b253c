from .vec_env import VecBreedGym
from .vec_wrappers import PairScores, RavelIndex, SelectionScores

from .breeding_programs_env import WheatBreedGym  # isort: skip

__all__ = ["VecBreedGym", "SelectionScores", "PairScores", "RavelIndex", "WheatBreedGym"]
