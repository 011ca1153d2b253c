from .vec_env import VecBreedGym

__all__ = ["VecBreedGym"]
