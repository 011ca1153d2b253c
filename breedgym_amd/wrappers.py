"""Single-env policy wrappers with the behaviour of the reference's ``breedgym/wrappers.py``.

``SimplifiedBreedGym``: the agent sees two per-individual summaries (first-trait GEBV and the
correlation of each genotype with the population mean) and chooses how many of the best
individuals to keep and how many distinct pairs of them to cross (reference lines 16-103).
``KBestBreedGym``: one number, the size of the elite; all its pairs are crossed (lines 106-118).

Both are thin: truncation selection is the device top-k + row gather of ``Simulator.select``, the
crosses go through the same fused launch as the plain environment.
"""
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import spaces
from .breedgym import BreedGym
from .engine import DeviceArray


def yield_index(GEBV_model) -> Callable:
    """Selection index = breeding value of the first trait (reference utils/index_functions.py:6-10)."""
    return lambda population: GEBV_model(population)[..., 0]


def mean_genotype_correlation(population) -> DeviceArray:
    """Cosine between each individual's centred dosage vector (dosage - 1) and their mean.

    The quantity the reference calls ``_correlation`` (wrappers.py:92-98)."""
    dosage = population.as_subclass(torch.Tensor).sum(dim=-1, dtype=torch.float32) - 1.0
    centre = dosage.mean(dim=0)
    cos = (dosage @ centre) / (dosage.norm(dim=1) * centre.norm())
    return DeviceArray.wrap(cos)


class SimplifiedBreedGym(spaces.Wrapper):
    metadata = BreedGym.metadata

    def __init__(self, env: Optional[BreedGym] = None, individual_per_gen: int = 2250,
                 f_index: Optional[Callable] = None, **kwargs):
        super().__init__(env if env is not None else BreedGym(**kwargs))
        n = self.individual_per_gen = individual_per_gen
        self.f_index = f_index or yield_index(self.env.unwrapped.simulator.GEBV_model)
        unit = spaces.Box(-1, 1, shape=(n,))
        self.observation_space = spaces.Dict({"GEBV": unit, "corrcoef": spaces.Box(-1, 1, shape=(n,))})
        self.action_space = spaces.Dict(
            {"n_bests": spaces.Discrete(n - 1, start=2), "n_crosses": spaces.Discrete(n, start=1)}
        )

    # -- observation -------------------------------------------------------------------------------
    def _simplified_obs(self) -> dict:
        base = self.env.unwrapped
        return {
            "GEBV": base.GEBV["Yield"].to_numpy(),
            "corrcoef": mean_genotype_correlation(base.population),
        }

    # -- gym surface -------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        options = dict(options or {})
        if "index" in options:
            self.f_index = options["index"]
        options["n_individuals"] = self.individual_per_gen  # start from a random subset of this size
        _, info = self.env.reset(seed=seed, options=options)
        return self._simplified_obs(), info

    def _mating_plan(self, n_bests: int, n_crosses: int) -> np.ndarray:
        """n_crosses distinct pairs of the elite (which sits in rows 0..n_bests-1 after selection),
        each repeated enough times to refill the population."""
        sim = self.env.unwrapped.simulator
        candidates = sim._diallel_indices(np.arange(n_bests))
        chosen = self.np_random.choice(len(candidates), n_crosses, replace=False)  # raises if too many
        copies = math.ceil(self.individual_per_gen / n_crosses)
        return np.repeat(candidates[chosen], copies, axis=0)[: self.individual_per_gen]

    def step(self, action: dict):
        n_bests, n_crosses = action["n_bests"], action["n_crosses"]
        if n_bests < 2:
            raise ValueError("n_bests must be higher or equal to 2")
        if n_crosses > self.individual_per_gen:
            raise ValueError("n_crosses must be lower or equal to individual_per_gen")
        base = self.env.unwrapped
        base.population, _ = base.simulator.select(base.population, n_bests, self.f_index)
        _, reward, terminated, truncated, info = self.env.step(self._mating_plan(n_bests, n_crosses))
        return self._simplified_obs(), reward, terminated, truncated, info


class KBestBreedGym(SimplifiedBreedGym):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        # the largest elite whose k (k - 1) / 2 pairs still fit into one generation
        largest = int((1 + math.sqrt(1 + 8 * self.individual_per_gen)) // 2)
        self.action_space = spaces.Discrete(largest - 1, start=2)

    def step(self, action: int):
        k = int(action)
        return super().step({"n_bests": k, "n_crosses": k * (k - 1) // 2})
