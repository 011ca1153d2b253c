from .breeding_programs_env import WheatBreedGym
from .vec_env import VecBreedGym
from .vec_wrappers import PairScores, RavelIndex, SelectionScores

__all__ = ["VecBreedGym", "SelectionScores", "PairScores", "RavelIndex", "WheatBreedGym"]
